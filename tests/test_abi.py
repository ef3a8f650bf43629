"""CPU: the C-ABI shared library builds, loads and exports every symbol include/cips3d_hip.h
declares (no compute calls — there is no GPU here), and the product path refuses to run on CPU."""
import os
import re

import pytest
import torch

from conftest import ROOT, seeded_generator


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "cips3d_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cips_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from cips3d_amd import build, _lib
    build.build(verbose=False)
    assert os.path.exists(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(_lib.SIGNATURES.keys()) == syms, "ctypes table and header disagree"
    lib = _lib.load()
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.cips_version() == 7
    assert lib.cips_arch() == b"gfx950"


def test_library_is_stateless_reads_no_environment_and_exports_no_hooks():
    """the C-ABI keeps no mode between calls and no environment variable changes what it computes
    (round 4 shipped a process-global clamp hook, a process-global kernel selector and ~25 getenv sites, three of which
    produced wrong results by design).  Checked on the built binary: it does not import getenv, carries no CIPS_* variable
    name, and exports only what include/cips3d_hip.h declares — no cips_debug_* / cips_*_set_* entry points.  The tuning
    aids live behind -DCIPS_TUNING (csrc/common.h), which build.py never passes."""
    import subprocess
    from cips3d_amd import build, _lib
    build.build(verbose=False)
    dyn = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    undefined = {l.split()[-1].split("@")[0] for l in dyn.splitlines() if " U " in l}
    assert not ({"getenv", "secure_getenv", "setenv", "putenv"} & undefined), undefined
    exported = sorted(l.split()[-1] for l in dyn.splitlines() if " T cips_" in l)
    assert exported == header_symbols(), set(exported) ^ set(header_symbols())
    assert not [e for e in exported if "debug" in e or "_set_" in e or "prof" in e]
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"CIPS_" not in blob, "an environment variable name is compiled into the production library"
    assert build.HIPCC_EXTRA == [] and "CIPS_TUNING" not in open(build.__file__).read()
    # the C sources read the environment only through the tuning helper of common.h
    csrc = os.path.join(ROOT, "cips3d_amd", "csrc")
    sites = [(f, i + 1) for f in sorted(os.listdir(csrc)) for i, l in enumerate(open(os.path.join(csrc, f)))
             if "getenv" in l and not l.lstrip().startswith("//")]
    assert sites == [("common.h", next(i + 1 for i, l in enumerate(open(os.path.join(csrc, "common.h"))) if "getenv(name)" in l))], sites


def test_struct_layouts_match_header_field_order():
    from cips3d_amd import _lib
    txt = open(os.path.join(ROOT, "include", "cips3d_hip.h")).read()

    def fields(struct):
        body = txt[txt.index("typedef struct " + struct):]
        body = body[body.index("{") + 1:body.index("} " + struct)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            names = stmt.replace("*", " ").split()
            decl = stmt.split(None, 1)
            for part in stmt.split(","):
                out.append(part.replace("*", " ").split()[-1])
        return out
    assert fields("cips_siren_weights") == [f[0] for f in _lib.SirenWeights._fields_]
    assert fields("cips_gemm_desc") == [f[0] for f in _lib.GemmDesc._fields_]
    assert fields("cips_gemm_x3_desc") == [f[0] for f in _lib.GemmX3Desc._fields_]


def test_product_path_has_no_cpu_fallback():
    G = seeded_generator(0)
    zs = G.get_zs(1)
    with pytest.raises(RuntimeError, match="GPU"):
        G(zs, img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=4, h_stddev=0.3, v_stddev=0.155,
          hierarchical_sample=False, sample_dist="gaussian")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cips3d_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} references the oracle"


def test_product_holds_no_copy_of_the_reference_diffaugment():
    """Round-2 verdict: the op-by-op torch DiffAugment (a renamed copy of exp/cips3d/models/diffaug.py:30-85) left the
    product; the policy runs on the HIP operator (cips_diffaug) only.  The function bodies must stay gone."""
    pkg = os.path.join(ROOT, "cips3d_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                for needle in ("rand_brightness", "rand_saturation", "rand_contrast", "rand_translation", "rand_cutout",
                               "AUGMENT_FNS", "torch.meshgrid", "mask[gb, gx, gy]", "F.pad(x, [1, 1, 1, 1"):
                    assert needle not in src, (f, needle)


def test_entry_points_validate_arguments_before_touching_the_device():
    """Malformed descriptors are refused with hipErrorInvalidValue (1) / hipErrorNotSupported (801) before any HIP
    call — so this runs without a GPU.  (Contraction lengths must be multiples of 32 bf16, planes 16-byte aligned,
    bit-plane gates need 32-column rows, the implicit-GEMM conv 32-channel slices, ...)"""
    import ctypes
    from cips3d_amd import _lib
    lib = _lib.load()
    INVALID, UNSUPPORTED = 1, 801
    d = _lib.GemmX3Desc()
    d.M, d.N, d.K, d.lda, d.ldb, d.batch = 64, 64, 48, 48, 48, 1          # K not a multiple of 32
    assert lib.cips_gemm_bf16x3(ctypes.byref(d), None) == INVALID
    assert lib.cips_gemm_bf16x3_km(ctypes.byref(d), None) == INVALID
    d.K, d.lda, d.ldb = 64, 60, 64                                        # rows not 16-byte aligned
    assert lib.cips_gemm_bf16x3(ctypes.byref(d), None) == INVALID
    d.lda, d.batch = 64, 0
    assert lib.cips_gemm_bf16x3(ctypes.byref(d), None) == INVALID
    d.batch, d.N, d.ldp, d.gate_bits = 1, 72, 72, 1                       # bit-plane gate with 72-column rows
    assert lib.cips_gemm_bf16x3(ctypes.byref(d), None) == INVALID
    assert lib.cips_gemm_bf16x3(None, None) == INVALID
    assert lib.cips_gemm_bf16x3_km_grouped(ctypes.byref(d), 0, None) == INVALID
    assert lib.cips_gemm_bf16x3_km_grouped(ctypes.byref(d), 9, None) == UNSUPPORTED   # more than one launch holds
    c = _lib.ConvX3Desc()
    c.B, c.C, c.H, c.W, c.O, c.kh, c.kw, c.stride, c.pad = 2, 48, 16, 16, 64, 3, 3, 1, 1   # 48 channels: no 32-slices
    assert lib.cips_conv2d_x3(ctypes.byref(c), None) == UNSUPPORTED
    c.C, c.stride = 64, 0
    assert lib.cips_conv2d_x3(ctypes.byref(c), None) == INVALID
    w = _lib.ConvWgradDesc()
    w.B, w.C, w.H, w.W, w.O, w.kh, w.kw, w.stride, w.pad, w.nchunks = 2, 64, 15, 15, 64, 3, 3, 1, 1, 1
    assert lib.cips_conv2d_x3_wgrad(ctypes.byref(w), None) == INVALID    # no output buffer
    buf = (ctypes.c_float * 4)()
    w.part = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.cips_conv2d_x3_wgrad(ctypes.byref(w), None) == UNSUPPORTED   # 450 pixels: no 32-row k-tiles
    # round-2 entry points: host-side helpers and argument validation (no launches)
    w.H = w.W = 16                                                          # 2 * 256 pixels = 16 k-tiles
    w.nchunks = 17
    assert lib.cips_conv2d_x3_wgrad(ctypes.byref(w), None) == UNSUPPORTED   # more chunks than k-tiles
    c.stride, c.ksplit, c.part = 1, 3, None
    assert lib.cips_conv2d_x3(ctypes.byref(c), None) == INVALID             # split contraction without its scratch
    for (B, O, N, K) in [(32, 512, 256, 4608), (32, 256, 4096, 2304), (4, 64, 65536, 576), (1, 32, 256, 288), (32, 512, 1024, 512)]:
        ks = lib.cips_conv2d_x3_ksplit(B, O, N, K)
        tiles = -(-O // 256) * -(-N // 256) * B
        assert 1 <= ks <= 8 and (ks == 1 or (K // 32) // ks >= 8)
        assert ks == 1 or tiles < 256                                       # a full chip is never split
    assert lib.cips_conv2d_x3_ksplit(32, 512, 256, 4608) == 4               # 64 tiles -> 256
    assert lib.cips_conv1x1_smallk_bwd_weight(None, None, None, 2, 3, 8, 64, None) == INVALID
    assert lib.cips_conv1x1_smallk_bwd_weight_splits(64, 65536) == 32 and lib.cips_conv1x1_smallk_bwd_weight_splits(512, 16) == 1
    assert lib.cips_conv1x1_smallk_bwd_weight_splits(8, 30) == 0            # pixels not a multiple of 4
    assert lib.cips_lrelu_bwd_bias(None, None, None, None, 4, 64, 0.2, 1.0, None) == INVALID
    assert lib.cips_lrelu_bwd_bias_slices(64) == 1 and lib.cips_lrelu_bwd_bias_slices(65536) == 16 and lib.cips_lrelu_bwd_bias_slices(1 << 20) == 64
    assert lib.cips_split_planes_nhwc(None, None, None, 2, 64, 16, None) == INVALID
    assert lib.cips_split_planes_nhwc(buf, buf, buf, 2, 60, 16, None) == INVALID        # channels not a multiple of 8
    assert lib.cips_conv_wgrad_finish(None, None, 1, 9, 8, 8, 1.0, None) == INVALID
    assert lib.cips_col2im(None, None, 0, 3, 8, 8, 3, 3, 2, 0, None) == INVALID
    assert lib.cips_conv1x1_smallk(None, None, None, 2, 5, 8, 64, None) == INVALID      # more than 4 input channels
    assert lib.cips_conv1x1_smallk(None, None, None, 2, 3, 8, 30, None) == INVALID      # pixels not a multiple of 4
    assert lib.cips_upfirdn2d(None, None, None, 1, 8, 8, 1, 9, 9, 1, 1, 1, 1, 0, 0, 0, 0, None) == INVALID   # > 64 taps
    assert lib.cips_torgb_fwd(None, None, None, None, 16, 30, 0, None) == INVALID       # K not a multiple of 4
    assert lib.cips_fused_bias_act(None, buf, None, None, 16, 0, 1, 3, 0, 0.2, 1.0, None) == INVALID   # bias without its size
    assert lib.cips_fused_bias_act(None, None, None, None, 0, 0, 0, 3, 0, 0.2, 1.0, None) == 0          # empty tensor: no-op
    # round-3 entry points
    assert lib.cips_torgb_bwd_x_x3(None, None, None, 0.2, None, None, None, 16, 64, None) == INVALID    # no operands
    assert lib.cips_torgb_bwd_x_x3(buf, buf, None, 0.2, None, buf, buf, 16, 60, None) == INVALID        # K not a multiple of 8
    assert lib.cips_torgb_bwd_w_x3_batch(None, None, 2, None, None, None, None, 128, 512, None) == INVALID
    ptrs = (ctypes.c_void_p * 9)(*[ctypes.cast(buf, ctypes.c_void_p).value] * 9)
    assert lib.cips_torgb_bwd_w_x3_batch(ptrs, ptrs, 9, None, None, None, None, 128, 512, None) == UNSUPPORTED   # more than 8 taps
    assert lib.cips_torgb_bwd_w_x3_batch(ptrs, ptrs, 2, None, None, None, None, 128, 256, None) == UNSUPPORTED   # K != 512
    # the planes addend (ABI 4) is taken by the 256 x 256-tile kernel only: the query says no for anything else, and the
    # GEMM entry point refuses instead of dropping it
    e = _lib.GemmX3Desc()
    e.M, e.N, e.K, e.lda, e.ldb, e.batch, e.ldp, e.strideP, e.gate_bits = 160, 256, 64, 64, 64, 4, 256, 160 * 256, 1
    pv = ctypes.cast(buf, ctypes.c_void_p)
    e.A_hi = e.A_lo = e.B_hi = e.B_lo = e.P_hi = e.P_lo = e.mask = e.addp_hi = e.addp_lo = e.addp_gate = pv
    e.addp_gain = 5.0
    assert lib.cips_gemm_bf16x3_takes_addp(ctypes.byref(e)) == 0                                      # 160 rows: not a v3 shape
    assert lib.cips_gemm_bf16x3(ctypes.byref(e), None) == UNSUPPORTED
    e.addp_hi = None
    assert lib.cips_gemm_bf16x3_takes_addp(ctypes.byref(e)) == 0                                      # nothing to take
