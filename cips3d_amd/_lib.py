"""ctypes binding of libcips3d_hip.so — the C-ABI declared in include/cips3d_hip.h.

The product path has NO CPU fallback: if the shared object is missing or a symbol is absent,
import fails loudly (RuntimeError).  Build it with `python -m cips3d_amd.build` (hipcc,
--offload-arch=gfx950; cross-compiles without a GPU).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcips3d_hip.so")

vp = C.c_void_p
i32 = C.c_int
i64 = C.c_longlong
f32 = C.c_float


class SirenWeights(C.Structure):
    _fields_ = [(n, vp) for n in
                ("w0", "b0", "w1", "b1", "ws", "bs", "wc", "bc", "wf", "bf",
                 "g0", "p0", "g1", "p1", "gc", "pc")] + [("box_scale", f32), ("trig_mode", i32)]


class SirenGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("dg0", "dp0", "dg1", "dp1", "dgc", "dpc", "dw0", "db0", "dw1", "db1", "dws", "dbs", "dwc",
                                  "dbc", "dwf", "dbf")]


class RayParams(C.Structure):
    _fields_ = [("xg", vp), ("yg", vp), ("zg", vp), ("cam2world", vp), ("jitter", vp), ("zvals", vp), ("zc", f32), ("H", i32),
                ("W", i32), ("S", i32)]


class GlinJob(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("bias", vp), ("y", vp), ("dy", vp), ("dw", vp), ("db", vp), ("in_dim", i32),
                ("out_dim", i32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda", i32), ("ldb", i32), ("ldc", i32),
        ("strideA", i64), ("strideB", i64), ("strideC", i64),
        ("batch", i32), ("a_kmajor", i32), ("b_nmajor", i32),
        ("alpha", f32), ("bias", vp), ("bias_m", vp),
        ("act", i32), ("slope", f32), ("act_gain", f32),
        ("resid", vp), ("C2", vp), ("add", vp),
        ("rgb_g", vp), ("rgb_w", vp), ("C_unmasked", vp), ("mask", vp),
    ]


class ModfcPrepJob(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("s", C.c_void_p), ("wb_hi", C.c_void_p), ("wb_lo", C.c_void_p),
                ("wbt_hi", C.c_void_p), ("wbt_lo", C.c_void_p), ("demod", C.c_void_p), ("in_dim", C.c_int),
                ("out_dim", C.c_int)]


class ModfcBwdJob(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("s", C.c_void_p), ("demod", C.c_void_p), ("gwb", C.c_void_p),
                ("cbuf", C.c_void_p), ("dweight", C.c_void_p), ("ds", C.c_void_p), ("in_dim", C.c_int),
                ("out_dim", C.c_int)]


class GemmX3Desc(C.Structure):
    _fields_ = [
        ("A_hi", vp), ("A_lo", vp), ("B_hi", vp), ("B_lo", vp),
        ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32),
        ("strideA", i64), ("strideB", i64), ("batch", i32),
        ("C", vp), ("ldc", i32), ("strideC", i64),
        ("P_hi", vp), ("P_lo", vp), ("ldp", i32), ("strideP", i64),
        ("T_hi", vp), ("T_lo", vp), ("ldt", i32), ("strideT", i64),
        ("mask_out", vp), ("add", vp), ("rgb_g", vp), ("rgb_w", vp), ("C_unmasked", vp), ("mask", vp),
        ("act", i32), ("slope", f32), ("res_hi", vp), ("res_lo", vp), ("gate_bits", i32),
        ("torgb_w", vp), ("torgb_part", vp),
        ("addp_hi", vp), ("addp_lo", vp), ("addp_gate", vp), ("addp_gain", f32),
        ("kernel", i32),
    ]


class ConvX3Desc(C.Structure):
    _fields_ = [("w_hi", vp), ("w_lo", vp), ("x_hi", vp), ("x_lo", vp), ("y", vp),
                ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("O", i32), ("kh", i32), ("kw", i32),
                ("stride", i32), ("pad", i32), ("ksplit", i32), ("part", vp), ("bias", vp), ("act", i32), ("slope", f32),
                ("act_scale", f32)]


class ConvDgradS2Desc(C.Structure):
    _fields_ = [("w_hi", vp), ("w_lo", vp), ("dy_hi", vp), ("dy_lo", vp), ("dxp", vp),
                ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("O", i32), ("kh", i32), ("kw", i32),
                ("w_off", i64 * 4), ("out_off", i64 * 4)]


class WPrepJob(C.Structure):
    _fields_ = [("w", vp), ("fwd_hi", vp), ("fwd_lo", vp), ("alt_hi", vp), ("alt_lo", vp), ("scale", f32),
                ("O", i32), ("C", i32), ("kh", i32), ("kw", i32), ("alt_kind", i32), ("bank_off", i32 * 4)]


class ConvWgradDesc(C.Structure):
    _fields_ = [("dy_hi", vp), ("dy_lo", vp), ("x_hi", vp), ("x_lo", vp), ("part", vp),
                ("B", i32), ("C", i32), ("H", i32), ("W", i32), ("O", i32), ("kh", i32), ("kw", i32),
                ("stride", i32), ("pad", i32), ("nchunks", i32)]


# name -> (restype, argtypes); must list every symbol of include/cips3d_hip.h
SIGNATURES = {
    "cips_version": (i32, []),
    "cips_arch": (C.c_char_p, []),
    "cips_rays_fwd": (i32, [vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "cips_siren_fwd": (i32, [C.POINTER(SirenWeights), vp, vp, vp, i32, i32, vp]),
    "cips_siren_fwd_x3": (i32, [C.POINTER(SirenWeights), vp, vp, vp, i32, i32, vp]),
    "cips_siren_bwd_rows": (i32, [i32, i32]),
    "cips_siren_bwd_x3_chunks": (i32, [i32, i32]),
    "cips_siren_bwd_x3_gpart": (i32, []),
    "cips_siren_bwd_x3_sred": (i32, []),
    "cips_siren_bwd_x3": (i32, [C.POINTER(SirenWeights), vp, vp, vp, vp, vp, i32, i32, vp]),
    "cips_siren_fwd_x3_rays": (i32, [C.POINTER(SirenWeights), C.POINTER(RayParams), vp, vp, vp, i32, vp]),
    "cips_siren_bwd_x3_rays": (i32, [C.POINTER(SirenWeights), C.POINTER(RayParams), vp, vp, vp, vp, i32, vp]),
    "cips_siren_bwd_x3_finalize": (i32, [C.POINTER(SirenWeights), vp, vp, i32, i32, C.POINTER(SirenGrads), vp]),
    "cips_march_fwd_x3": (i32, [C.POINTER(SirenWeights), C.POINTER(RayParams), vp, f32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "cips_siren_bwd_data": (i32, [C.POINTER(SirenWeights)] + [vp] * 14 + [i32, i32, vp]),
    "cips_siren_bwd_data_f32": (i32, [C.POINTER(SirenWeights)] + [vp] * 9 + [i32, i32, vp]),
    "cips_resample_fwd": (i32, [vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, C.POINTER(RayParams), vp]),
    "cips_composite_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "cips_composite_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "cips_gemm_f32": (i32, [C.POINTER(GemmDesc), vp]),
    "cips_gemm_bf16x3": (i32, [C.POINTER(GemmX3Desc), vp]),
    "cips_gemm_bf16x3_fuses_torgb": (i32, [C.POINTER(GemmX3Desc)]),
    "cips_gemm_bf16x3_takes_addp": (i32, [C.POINTER(GemmX3Desc)]),
    "cips_torgb_finish": (i32, [vp, i32, vp, vp, i64, i32, vp]),
    "cips_equal_linear_scratch": (i64, [i32, i32, i32, i32]),
    "cips_equal_linear": (i32, [i32, vp, vp, vp, f32, f32, vp, vp, i32, i32, i32, vp]),
    "cips_gemm_bf16x3_km": (i32, [C.POINTER(GemmX3Desc), vp]),
    "cips_gemm_bf16x3_km_grouped": (i32, [C.POINTER(GemmX3Desc), i32, vp]),
    "cips_lrelu_bwd_bias_slices": (i32, [i32]),
    "cips_lrelu_bwd_bias": (i32, [vp, vp, vp, vp, i64, i32, f32, f32, vp]),
    "cips_lrelu_bwd_bias_finish": (i32, [vp, vp, i32, i32, i32, vp]),
    "cips_lrelu_bwd_bias_nhwc_tiles": (i32, [i32]),
    "cips_lrelu_bwd_bias_nhwc": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp]),
    "cips_conv1x1_smallk": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "cips_conv1x1_smallk_bwd_data": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "cips_conv1x1_smallk_bwd_weight_splits": (i32, [i32, i32]),
    "cips_conv1x1_smallk_bwd_weight": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "cips_conv2d_x3": (i32, [C.POINTER(ConvX3Desc), vp]),
    "cips_conv2d_x3_wgrad": (i32, [C.POINTER(ConvWgradDesc), vp]),
    "cips_split_planes": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i64, vp]),
    "cips_split_planes_nhwc": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "cips_conv2d_x3_ksplit": (i32, [i32, i32, i32, i32]),
    "cips_conv_weight_prep_max_jobs": (i32, []),
    "cips_conv_weight_prep_batch": (i32, [C.POINTER(WPrepJob), i32, vp]),
    "cips_conv2d_x3_dgrad_s2": (i32, [C.POINTER(ConvDgradS2Desc), vp]),
    "cips_conv_wgrad_finish": (i32, [vp, vp, i32, i32, i32, i32, f32, vp]),
    "cips_modfc_prep_x3": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "cips_torgb_fwd_x3": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "cips_torgb_bwd_w_x3": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "cips_modfc_prep": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "cips_modfc_prep_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "cips_modfc_max_jobs": (i32, []),
    "cips_modfc_prep_x3_batch": (i32, [C.POINTER(ModfcPrepJob), i32, i32, C.c_float, vp]),
    "cips_modfc_prep_bwd_batch": (i32, [C.POINTER(ModfcBwdJob), i32, i32, vp]),
    "cips_modfc_prep_bwd_batch_cores": (i32, [C.POINTER(ModfcBwdJob), i32, i32, vp]),
    "cips_cores_colsum_parts": (i32, []),
    "cips_opt_chunk": (i32, []),
    "cips_opt_step": (i32, [vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, f32, f32, i32, vp, vp]),
    "cips_camera_pose": (i32, [vp, vp, i32, f32, f32, f32, f32, vp, vp, vp, i32, vp]),
    "cips_grouped_linear_max_jobs": (i32, []),
    "cips_grouped_linear_fwd": (i32, [C.POINTER(GlinJob), i32, i32, vp]),
    "cips_grouped_linear_scratch": (i64, [C.POINTER(GlinJob), i32, i32]),
    "cips_grouped_linear_bwd": (i32, [C.POINTER(GlinJob), i32, i32, vp, vp, i64, vp]),
    "cips_rownorm_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "cips_rownorm_bwd": (i32, [vp] * 9 + [i32, i32, i32, f32, vp]),
    "cips_torgb_fwd": (i32, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "cips_torgb_bwd_partials": (i32, [i64]),
    "cips_torgb_bwd_w": (i32, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "cips_torgb_bwd_x": (i32, [vp, vp, vp, vp, f32, vp, vp, i64, i32, vp]),
    "cips_torgb_bwd_w_x3_batch": (i32, [vp, vp, i32, vp, vp, vp, vp, i64, i32, vp]),
    "cips_torgb_bwd_w_x3_batch_cores": (i32, [vp, vp, i32, vp, vp, vp, vp, i64, i32, vp]),
    "cips_torgb_bwd_x_x3": (i32, [vp, vp, vp, f32, vp, vp, vp, i64, i32, vp]),
    "cips_fused_bias_act": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, f32, vp]),
    "cips_diffaug": (i32, [vp] * 10 + [i32] * 8 + [vp]),
    "cips_avgpool2": (i32, [vp, vp, i64, i32, i32, i32, vp]),
    "cips_axpby": (i32, [vp, vp, vp, f32, f32, i64, vp]),
    "cips_image_to_u8": (i32, [vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "cips_upfirdn2d": (i32, [vp, vp, vp] + [i32] * 14 + [vp]),
    "cips_upfirdn2d_parity": (i32, [vp, C.POINTER(i64 * 4), vp, vp] + [i32] * 7 + [vp]),
    "cips_im2col": (i32, [vp, vp] + [i32] * 8 + [vp]),
    "cips_im2col_x3": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cips_col2im": (i32, [vp, vp] + [i32] * 8 + [vp]),
}

_lib = None


def load():
    """Load the shared object and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"cips3d_amd: HIP extension not built ({LIB_PATH} missing). "
            "Run `python -m cips3d_amd.build` (needs hipcc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"cips3d_amd: symbol {name} missing from {LIB_PATH}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(err, what):
    if err != 0:
        raise RuntimeError(f"cips3d_amd: {what} failed with hipError {err}")
