"""Host-side operators over the C-ABI (include/cips3d_hip.h): thin launch wrappers plus the
torch.autograd.Functions that give the reference's Python API its backward.

PyTorch is used for device memory, streams and autograd bookkeeping only; every arithmetic
kernel on the hot path is a hand-written HIP kernel in libcips3d_hip.so.  There is no CPU
fallback: tensors must live on a ROCm device and the extension must be built.
"""
import ctypes as C
import ctypes as _ct
import math
import os as _os

import torch

from . import _lib
from ._lib import GemmDesc, GemmX3Desc, SirenWeights, check

LRELU_SLOPE = 0.2


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("cips3d_amd ops need tensors on the GPU (no CPU fallback)")
        if t.device.index != torch.cuda.current_device():
            # launches go to the CURRENT device's current stream (one process per GPU, torch.cuda.set_device(rank),
            # train.py:48): a tensor of another device would be read through the wrong context
            raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                               "call torch.cuda.set_device first")
        if t.dtype != torch.float32 and t.dtype not in (torch.int32, torch.int64, torch.bfloat16):
            raise RuntimeError(f"cips3d_amd ops are fp32 (got {t.dtype})")
        if not t.is_contiguous():
            raise RuntimeError("cips3d_amd ops need contiguous tensors")


def _c(t):
    return t.contiguous().float() if t is not None else None


# --------------------------------------------------------------------------------------
# H1 camera pose + rays
# --------------------------------------------------------------------------------------


def camera_pose(theta_raw, phi_raw, uniform, h_stddev, h_mean, v_stddev, v_mean):
    """raw draws (b, 1) x 2 -> pitch_yaw (b, 2), origin (b, 3), cam2world (b, 4, 4); see cips_camera_pose"""
    lib = _lib.load()
    dev = theta_raw.device
    B = theta_raw.shape[0]
    th, ph = _c(theta_raw), _c(phi_raw)
    py = torch.empty(B, 2, device=dev); origin = torch.empty(B, 3, device=dev); c2w = torch.empty(B, 4, 4, device=dev)
    check(lib.cips_camera_pose(_p(th), _p(ph), 1 if uniform else 0, float(h_stddev), float(h_mean), float(v_stddev),
                               float(v_mean), _p(py), _p(origin), _p(c2w), B, _stream()), "cips_camera_pose")
    return py, origin, c2w


_GRIDS = {}


def pixel_grids(W, H, S, ray_start, ray_end, device):
    """the three linspaces of the ray set-up (constants of the geometry), built once per geometry and device"""
    key = (W, H, S, float(ray_start), float(ray_end), str(device))
    g = _GRIDS.get(key)
    if g is None:
        g = (torch.linspace(-1, 1, W, device=device), torch.linspace(1, -1, H, device=device),
             torch.linspace(ray_start, ray_end, S, device=device))
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            return g        # built inside a hipGraph capture: the tensors live in the graph's pool and hold data only
                            # during replays — never cached for eager callers
        if len(_GRIDS) > 64:
            _GRIDS.clear()
        _GRIDS[key] = g
    return g

def rays_fwd(xg, yg, zg, zc, cam2world, jitter, B, H, W, S):
    """-> points (B,n,S,3), z (B,n,S), dirs (B,n,3); see cips_rays_fwd."""
    lib = _lib.load()
    dev = cam2world.device
    n = H * W
    points = torch.empty(B, n, S, 3, device=dev)
    z = torch.empty(B, n, S, device=dev)
    dirs = torch.empty(B, n, 3, device=dev)
    xg, yg, zg, cam2world, jitter = _c(xg), _c(yg), _c(zg), _c(cam2world), _c(jitter)
    _chk(xg, yg, zg, cam2world, jitter)
    check(lib.cips_rays_fwd(_p(xg), _p(yg), _p(zg), float(zc), _p(cam2world), _p(jitter),
                            _p(points), _p(z), _p(dirs), B, H, W, S, _stream()), "cips_rays_fwd")
    return points, z, dirs


# --------------------------------------------------------------------------------------
# generic GEMM
# --------------------------------------------------------------------------------------
def gemm(A, Bm, Cm, M, N, K, lda, ldb, ldc, batch=1, strideA=0, strideB=0, strideC=0,
         a_kmajor=False, b_nmajor=False, alpha=1.0, bias=None, bias_m=None, act=0, slope=LRELU_SLOPE,
         act_gain=1.0, resid=None, C2=None, add=None, rgb_g=None, rgb_w=None, C_unmasked=None, mask=None):
    lib = _lib.load()
    d = GemmDesc()
    d.A, d.B, d.C = _p(A), _p(Bm), _p(Cm)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.strideA, d.strideB, d.strideC = strideA, strideB, strideC
    d.batch = batch
    d.a_kmajor = 1 if a_kmajor else 0
    d.b_nmajor = 1 if b_nmajor else 0
    d.alpha = alpha
    d.bias, d.bias_m = _p(bias), _p(bias_m)
    d.act, d.slope, d.act_gain = act, slope, act_gain
    d.resid, d.C2, d.add = _p(resid), _p(C2), _p(add)
    d.rgb_g, d.rgb_w = _p(rgb_g), _p(rgb_w)
    d.C_unmasked, d.mask = _p(C_unmasked), _p(mask)
    check(lib.cips_gemm_f32(C.byref(d), _stream()), "cips_gemm_f32")
    return Cm


def bmm_nn(x, w, out=None, **epi):
    """out[b] (M,N) = x[b] (M,K) @ w[b] (K,N); x (B,M,K), w (B,K,N) contiguous."""
    B, M, K = x.shape
    N = w.shape[-1]
    if out is None:
        out = torch.empty(B, M, N, device=x.device)
    gemm(x, w, out, M, N, K, K, N, N, batch=B, strideA=M * K, strideB=K * N, strideC=M * N, **epi)
    return out


def bmm_tn(a, b, out=None):
    """out[b] (Ma,N) = a[b]^T @ b[b]; a (B,K,Ma), b (B,K,N)."""
    B, K, Ma = a.shape
    N = b.shape[-1]
    if out is None:
        out = torch.empty(B, Ma, N, device=a.device)
    gemm(a, b, out, Ma, N, K, Ma, N, N, batch=B, strideA=K * Ma, strideB=K * N, strideC=Ma * N, a_kmajor=True)
    return out


# --------------------------------------------------------------------------------------
# H2 SIREN
# --------------------------------------------------------------------------------------
_SIREN_NAMES = ("w0", "b0", "w1", "b1", "ws", "bs", "wc", "bc", "wf", "bf", "g0", "p0", "g1", "p1", "gc", "pc")
BOX_SCALE = 2.0 / 0.24   # UniformBoxWarp(0.24), generator.py:249
TRIG_MODE = 1            # 1 hardware v_sin_f32/v_cos_f32 after Cody-Waite reduction (default; measured as accurate
                         # as the polynomial on MI355X: 1.2e-6 vs 1.3e-6), 0 minimax polynomial


def _siren_struct(t):
    s = SirenWeights()
    for n in _SIREN_NAMES:
        setattr(s, n, _p(t[n]))
    s.box_scale = BOX_SCALE
    s.trig_mode = TRIG_MODE
    return s


def _split_k(P, target=8):
    """Largest split count <= target such that P/split is a multiple of 32."""
    for s in range(target, 0, -1):
        if P % s == 0 and (P // s) % 32 == 0:
            return s
    return 1


# The exact-fp32 mode of the path (the only environment variables the package reads besides
# CIPS_INR_MODE and CIPS_D_CONV_MODE).  Forward "x3": split-operand matrix-core chain (default: fp16 planes, sigma to fp32
# class); "f32": exact fp32 MFMA.  Backward "x3": fused split-bf16 kernel; "staged": fp32 data pass + split-bf16 K-major GEMMs;
# "staged_f32" (round 6): fp32 data pass with fp32-staged activations + fp32-MFMA weight-gradient GEMMs (the all-fp32 leg).
SIREN_FWD_MODE = __import__("os").environ.get("CIPS_SIREN_FWD", "x3")
SIREN_BWD_MODE = __import__("os").environ.get("CIPS_SIREN_BWD", "x3")


class SirenFunction(torch.autograd.Function):
    """feat (B,P,32), sigma (B,P) = siren(points; weights, per-image FiLM vectors).

    Mirrors NeRFNetwork.forward_with_frequencies_phase_shifts (generator.py:260-317); gains g*
    are already 15*gain_fc(style)+30 and phases p* = bias_fc(style) (film_layer.py:88-93), both
    computed by tiny torch Linears on the host side so autograd carries them to the mapping net.
    No gradient w.r.t. points (the reference never needs it: points come from no_grad ray math).
    """

    @staticmethod
    def forward(ctx, points, g0, p0, g1, p1, gc, pc, w0, b0, w1, b1, ws, bs, wc, bc, wf, bf):
        lib = _lib.load()
        t = dict(w0=w0, b0=b0, w1=w1, b1=b1, ws=ws, bs=bs, wc=wc, bc=bc, wf=wf, bf=bf,
                 g0=g0, p0=p0, g1=g1, p1=p1, gc=gc, pc=pc)
        t = {k: _c(v.detach()) for k, v in t.items()}
        points = _c(points.detach())
        _chk(points, *t.values())
        B, P, _ = points.shape
        feat = torch.empty(B, P, 32, device=points.device)
        sigma = torch.empty(B, P, device=points.device)
        sw = _siren_struct(t)
        fwd = lib.cips_siren_fwd_x3 if SIREN_FWD_MODE == "x3" else lib.cips_siren_fwd
        check(fwd(C.byref(sw), _p(points), _p(feat), _p(sigma), B, P, _stream()), "cips_siren_fwd")
        ctx.save_for_backward(points, *[t[n] for n in _SIREN_NAMES])
        return feat, sigma

    @staticmethod
    def backward(ctx, dfeat, dsigma):
        points = ctx.saved_tensors[0]
        t = dict(zip(_SIREN_NAMES, ctx.saved_tensors[1:]))
        B, P, _ = points.shape
        dev = points.device
        dfeat = _c(dfeat) if dfeat is not None else torch.zeros(B, P, 32, device=dev)
        dsigma = _c(dsigma) if dsigma is not None else torch.zeros(B, P, device=dev)
        return (None,) + _siren_backward(t, dfeat, dsigma, B, P, points=points)


def _siren_backward(t, dfeat, dsigma, B, P, points=None, rays=None):
    """SIREN backward for upstream gradients dfeat (B,P,32), dsigma (B,P): -> (dg0, dp0, dg1, dp1, dgc, dpc, dw0, db0, dw1,
    db1, dws, dbs, dwc, dbc, dwf, dbf).  The sample points are either given (B,P,3) or regenerated in-kernel from
    `rays` (a RayParams struct; fused bf16x3 form only)."""
    lib = _lib.load()
    dev = dfeat.device
    sw = _siren_struct(t)
    if True:
        if SIREN_BWD_MODE == "x3" or points is None:
            # fused kernel: recompute + data gradients + weight-gradient contractions, nothing staged in HBM
            chunks = lib.cips_siren_bwd_x3_chunks(B, P)
            gw = lib.cips_siren_bwd_x3_gpart()
            sw_ = lib.cips_siren_bwd_x3_sred()
            sred = torch.empty(B * chunks, sw_, device=dev)
            gpart = torch.empty(B * chunks, gw, device=dev)
            if points is not None:
                check(lib.cips_siren_bwd_x3(C.byref(sw), _p(points), _p(dfeat), _p(dsigma), _p(sred), _p(gpart), B, P,
                                            _stream()), "cips_siren_bwd_x3")
            else:
                check(lib.cips_siren_bwd_x3_rays(C.byref(sw), C.byref(rays), _p(dfeat), _p(dsigma), _p(sred), _p(gpart), B,
                                                 _stream()), "cips_siren_bwd_x3_rays")
            # the 16 gradient tensors from the partials in one launch (was ~40 tiny torch reductions)
            from ._lib import SirenGrads
            shapes = dict(dg0=(B, 128), dp0=(B, 128), dg1=(B, 128), dp1=(B, 128), dgc=(B, 64), dpc=(B, 64), dw0=(128, 3),
                          db0=(128,), dw1=(128, 128), db1=(128,), dws=(1, 128), dbs=(1,), dwc=(64, 128), dbc=(64,),
                          dwf=(32, 64), dbf=(32,))
            outs = {k: torch.empty(*v, device=dev) for k, v in shapes.items()}
            sg = SirenGrads()
            for k, v in outs.items():
                setattr(sg, k, _p(v))
            # the chunk partials are summed by two streaming reductions first (88 MB at C2: bandwidth-bound, one
            # launch each); the finalisation then walks B rows instead of B * chunks
            SRr = sred.view(B, chunks, sw_).sum(1) if chunks > 1 else sred
            Gpr = gpart.view(B, chunks, gw).sum(1) if chunks > 1 else gpart
            check(lib.cips_siren_bwd_x3_finalize(C.byref(sw), _p(SRr), _p(Gpr), B, 1, C.byref(sg), _stream()),
                  "cips_siren_bwd_x3_finalize")
            return tuple(outs[k] for k in ("dg0", "dp0", "dg1", "dp1", "dgc", "dpc", "dw0", "db0", "dw1", "db1", "dws",
                                           "dbs", "dwc", "dbc", "dwf", "dbf"))
        elif SIREN_BWD_MODE == "staged_f32":
            # the all-fp32 leg (round 6): the data pass stages fp32 rows, every weight-gradient contraction runs on the exact-fp32
            # MFMA GEMM (k-major A): no split operand anywhere in the SIREN backward
            BP = B * P
            h1, h2, da2 = (torch.empty(BP, 128, device=dev) for _ in range(3))
            hc, dac = (torch.empty(BP, 64, device=dev) for _ in range(2))
            rows = lib.cips_siren_bwd_rows(B, P)
            red = torch.empty(rows, 868, device=dev)
            check(lib.cips_siren_bwd_data_f32(C.byref(sw), _p(points), _p(dfeat), _p(dsigma), _p(h1), _p(h2), _p(hc), _p(da2), _p(dac),
                                              _p(red), B, P, _stream()), "cips_siren_bwd_data_f32")
            R = red.view(B, rows // B, 868).sum(1)
            sp = _split_k(P, 16)
            Kc = P // sp
            G1 = torch.empty(B * sp, 128, 128, device=dev)   # da2^T @ h1 per image and point chunk
            gemm(da2, h1, G1, 128, 128, Kc, 128, 128, 128, batch=B * sp, strideA=Kc * 128, strideB=Kc * 128, strideC=128 * 128,
                 a_kmajor=True)
            G1 = G1.view(B, sp, 128, 128).sum(1)
            Gc = torch.empty(B * sp, 64, 128, device=dev)    # dac^T @ h2
            gemm(dac, h2, Gc, 64, 128, Kc, 64, 128, 128, batch=B * sp, strideA=Kc * 64, strideB=Kc * 128, strideC=64 * 128,
                 a_kmajor=True)
            Gc = Gc.view(B, sp, 64, 128).sum(1)
            spf = _split_k(BP, 1024)
            Kf = BP // spf
            Gf = torch.empty(spf, 32, 64, device=dev)        # dfeat^T @ hc (no per-image scale)
            gemm(dfeat, hc, Gf, 32, 64, Kf, 32, 64, 64, batch=spf, strideA=Kf * 32, strideB=Kf * 64, strideC=32 * 64, a_kmajor=True)
            dwf = Gf.sum(0)
        else:
            BP = B * P
            h1, h2, da2 = (Planes.empty(BP, 128, device=dev) for _ in range(3))
            hc, dac = (Planes.empty(BP, 64, device=dev) for _ in range(2))
            rows = lib.cips_siren_bwd_rows(B, P)
            red = torch.empty(rows, 868, device=dev)
            check(lib.cips_siren_bwd_data(C.byref(sw), _p(points), _p(dfeat), _p(dsigma), _p(h1.hi), _p(h1.lo),
                                          _p(h2.hi), _p(h2.lo), _p(hc.hi), _p(hc.lo), _p(da2.hi), _p(da2.lo),
                                          _p(dac.hi), _p(dac.lo), _p(red), B, P, _stream()), "cips_siren_bwd_data")
            R = red.view(B, rows // B, 868).sum(1)          # (B, 868) deterministic reduction of partial rows
            # weight-gradient contractions over the points (K = P per image, split-K) on the bf16x3 K-major GEMM
            sp = _split_k(P, 16)
            Kc = P // sp
            G1 = torch.empty(B * sp, 128, 128, device=dev)   # da2^T @ h1
            gemm_x3_km(da2, h1, 128, 128, Kc, 128, 128, B * sp, Kc * 128, Kc * 128, G1)
            G1 = G1.view(B, sp, 128, 128).sum(1)
            Gc = torch.empty(B * sp, 64, 128, device=dev)    # dac^T @ h2
            gemm_x3_km(dac, h2, 64, 128, Kc, 64, 128, B * sp, Kc * 64, Kc * 128, Gc)
            Gc = Gc.view(B, sp, 64, 128).sum(1)
            hcf = hc.float()                                 # the 32x64 colour-linear gradient stays on the fp32 GEMM
            spf = _split_k(BP, 1024)
            Kf = BP // spf
            Gf = torch.empty(spf, 32, 64, device=dev)        # dfeat^T @ hc (no per-image scale)
            gemm(dfeat, hcf, Gf, 32, 64, Kf, 32, 64, 64, batch=spf, strideA=Kf * 32, strideB=Kf * 64,
                 strideC=32 * 64, a_kmajor=True)
            dwf = Gf.sum(0)
        # ---- assemble parameter / FiLM gradients (tiny tensors) ----
        g0, g1, gc = t["g0"], t["g1"], t["gc"]
        w0, b0, w1, b1, wc, bc = t["w0"], t["b0"], t["w1"], t["b1"], t["wc"], t["bc"]
        dp0 = R[:, 0:128]
        S0 = R[:, 128:512].view(B, 3, 128).transpose(1, 2) * BOX_SCALE   # (B,128,3): sum_p da1 * x0_c
        dg0 = (S0 * w0.unsqueeze(0)).sum(-1) + b0.unsqueeze(0) * dp0
        dw0 = (g0.unsqueeze(-1) * S0).sum(0)
        db0 = (g0 * dp0).sum(0)
        dp1 = R[:, 512:640]
        dg1 = (G1 * w1.unsqueeze(0)).sum(-1) + b1.unsqueeze(0) * dp1
        dw1 = (g1.unsqueeze(-1) * G1).sum(0)
        db1 = (g1 * dp1).sum(0)
        dpc = R[:, 640:704]
        dgc = (Gc * wc.unsqueeze(0)).sum(-1) + bc.unsqueeze(0) * dpc
        dwc = (gc.unsqueeze(-1) * Gc).sum(0)
        dbc = (gc * dpc).sum(0)
        dws = R[:, 704:832].sum(0, keepdim=True)
        dbf = R[:, 832:864].sum(0)
        dbs = R[:, 864].sum().view(1)
        return (dg0, dp0, dg1, dp1, dgc, dpc, dw0, db0, dw1, db1, dws, dbs, dwc, dbc, dwf, dbf)


# --------------------------------------------------------------------------------------
# H3 resample / composite
# --------------------------------------------------------------------------------------
def resample_fwd(sigma, z, noise, noise_std, u, origins, dirs, B, n, S, clamp_mode=0, debug=False, cdf_in=None, rays=None):
    """sigma/z/noise/u (B*n,S) -> fine_z (B*n,S), fine_pts (B*n,S,3) [+ weights, cdf, inds].
    rays (RayParams): ray directions / origins recomputed in-kernel, no fine_pts (the fine pass regenerates its points
    from fine_z): origins / dirs may be None, fine_pts is returned as None."""
    lib = _lib.load()
    dev = sigma.device
    sigma, z, noise, u, origins, dirs = _c(sigma), _c(z), _c(noise), _c(u), _c(origins), _c(dirs)
    _chk(sigma, z, noise, u, origins, dirs)
    R = B * n
    fine_z = torch.empty(R, S, device=dev)
    fine_pts = torch.empty(R, S, 3, device=dev) if rays is None else None
    w = cdf = inds = None
    if debug:
        w = torch.empty(R, S, device=dev)
        cdf = torch.empty(R, S - 1, device=dev)
        inds = torch.empty(R, S, device=dev, dtype=torch.int64)
    check(lib.cips_resample_fwd(_p(sigma), _p(z), _p(noise), float(noise_std), _p(u), _p(origins), _p(dirs),
                                _p(fine_z), _p(fine_pts), _p(w), _p(cdf), _p(inds), B, n, S, clamp_mode,
                                _p(_c(cdf_in)) if cdf_in is not None else None,
                                C.byref(rays) if rays is not None else None, _stream()), "cips_resample_fwd")
    if debug:
        return fine_z, fine_pts, w, cdf, inds
    return fine_z, fine_pts


_CLAMP = {"relu": 0, "softplus": 1}


class CompositeFunction(torch.autograd.Function):
    """Merge (optional) + alpha-composite; see cips_composite_fwd / _bwd.
    Inputs are flattened over rays: feat (R,S,32), sig (R,S), z (R,S); fine set may be None."""

    @staticmethod
    def forward(ctx, feat_c, sig_c, z_c, feat_f, sig_f, z_f, noise, noise_std, clamp_mode, flags):
        lib = _lib.load()
        feat_c, sig_c, z_c = _c(feat_c.detach()), _c(sig_c.detach()), _c(z_c.detach())
        hier = feat_f is not None
        if hier:
            feat_f, sig_f, z_f = _c(feat_f.detach()), _c(sig_f.detach()), _c(z_f.detach())
        noise = _c(noise) if (noise is not None and noise_std != 0.0) else None
        _chk(feat_c, sig_c, z_c, feat_f, sig_f, z_f, noise)
        R, S, _ = feat_c.shape
        E = 2 * S if hier else S
        dev = feat_c.device
        fea = torch.empty(R, 32, device=dev)
        depth = torch.empty(R, device=dev)
        weights = torch.empty(R, E, device=dev)
        order = torch.empty(R, E, device=dev, dtype=torch.int32)
        zs = torch.empty(R, E, device=dev)
        pin, rec = _clamp_debug_forward(R, E, clamp_mode, dev)
        check(lib.cips_composite_fwd(_p(feat_c), _p(sig_c), _p(z_c), _p(feat_f), _p(sig_f), _p(z_f), _p(noise),
                                     float(noise_std), _p(fea), _p(depth), _p(weights), _p(order), _p(zs),
                                     R, S, clamp_mode, flags, _p(pin), _p(rec), _stream()), "cips_composite_fwd")
        if rec is not None and CLAMP_REC is not None:
            CLAMP_REC.append(rec.reshape(R, E))
        ctx.clamp_mask = pin                      # pinned branches: the backward takes the same ones (else: its own sign test)
        ctx.save_for_backward(feat_c, sig_c, z_c, feat_f, sig_f, z_f, noise, order)
        ctx.meta = (float(noise_std), clamp_mode, flags, hier)
        ctx.mark_non_differentiable(depth, weights, order, zs)
        return fea, depth, weights, order, zs

    @staticmethod
    def backward(ctx, dfea, *unused):
        lib = _lib.load()
        feat_c, sig_c, z_c, feat_f, sig_f, z_f, noise, order = ctx.saved_tensors
        noise_std, clamp_mode, flags, hier = ctx.meta
        R, S, _ = feat_c.shape
        dfea = _c(dfea)
        dfeat_c = torch.empty_like(feat_c)
        dsig_c = torch.empty_like(sig_c)
        dfeat_f = torch.empty_like(feat_f) if hier else None
        dsig_f = torch.empty_like(sig_f) if hier else None
        check(lib.cips_composite_bwd(_p(feat_c), _p(sig_c), _p(z_c), _p(feat_f), _p(sig_f), _p(z_f), _p(noise),
                                     noise_std, _p(order), _p(dfea), _p(dfeat_c), _p(dsig_c), _p(dfeat_f),
                                     _p(dsig_f), R, S, clamp_mode, flags, _p(ctx.clamp_mask), _stream()), "cips_composite_bwd")
        _tail_gate_publish(dfea.device)
        return dfeat_c, dsig_c, None, dfeat_f, dsig_f, None, None, None, None, None


# --------------------------------------------------------------------------------------
# Grouped small Linear layers (style -> per-image vectors), one launch for all of them
# --------------------------------------------------------------------------------------


class GroupedLinearFunction(torch.autograd.Function):
    """ys = [x @ W_j^T + b_j for j] for Linear layers that share their input x (B, in): the SinStyleMod.modulation
    layers of the CIPS head (mod_conv_fc.py:433-436, 474) or the gain_fc / bias_fc of the FiLM layers
    (film_layer.py:59-63, 88-93).  forward(x, W_0, b_0, W_1, b_1, ...) -> tuple of (B, out_j)."""

    @staticmethod
    def forward(ctx, x, *wb):
        lib = _lib.load()
        from ._lib import GlinJob
        x = _c(x.detach())
        ws = [_c(w.detach()) for w in wb[0::2]]
        bs = [_c(b.detach()) if b is not None else None for b in wb[1::2]]
        _chk(x, *ws, *[b for b in bs if b is not None])
        B, in_dim = x.shape
        n = len(ws)
        if n > lib.cips_grouped_linear_max_jobs() or in_dim % 4 or in_dim > 512:
            raise RuntimeError("grouped linear: at most 32 jobs, in_dim a multiple of 4 and <= 512")
        jobs = (GlinJob * n)()
        ys = []
        for j, (w, b) in zip(jobs, zip(ws, bs)):
            y = torch.empty(B, w.shape[0], device=x.device)
            j.x, j.w, j.bias, j.y = _p(x), _p(w), _p(b), _p(y)
            j.in_dim, j.out_dim = in_dim, w.shape[0]
            ys.append(y)
        check(lib.cips_grouped_linear_fwd(jobs, n, B, _stream()), "cips_grouped_linear_fwd")
        ctx.save_for_backward(x, *ws)
        ctx.has_bias = [b is not None for b in bs]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        from ._lib import GlinJob
        x, *ws = ctx.saved_tensors
        B, in_dim = x.shape
        n = len(ws)
        jobs = (GlinJob * n)()
        dws, dbs, keep = [], [], []
        for j, w, dy, hb in zip(jobs, ws, dys, ctx.has_bias):
            dy = _c(dy) if dy is not None else torch.zeros(B, w.shape[0], device=x.device)
            keep.append(dy)
            dw = torch.empty_like(w)
            db = torch.empty(w.shape[0], device=x.device) if hb else None
            j.x, j.w, j.dy, j.dw, j.db = _p(x), _p(w), _p(dy), _p(dw), _p(db)
            j.in_dim, j.out_dim = in_dim, w.shape[0]
            dws.append(dw); dbs.append(db)
        dx = scratch = None
        nscr = 0
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nscr = lib.cips_grouped_linear_scratch(jobs, n, B)
            scratch = torch.empty(nscr, device=x.device)
        check(lib.cips_grouped_linear_bwd(jobs, n, B, _p(dx), _p(scratch), nscr, _stream()), "cips_grouped_linear_bwd")
        out = [dx]
        for dw, db in zip(dws, dbs):
            out += [dw, db]
        return tuple(out)


def grouped_linear(pairs):
    """pairs: list of (x (B,in), nn.Linear) -> list of Linear(x).  Layers whose inputs are the SAME tensor go into one
    grouped launch (the reference's style dicts alias one tensor per mapping network, multi_head_mapping.py:147-153);
    anything else (truncated / mixed styles) falls back to one launch per distinct input."""
    out = [None] * len(pairs)
    groups = {}
    for idx, (x, lin) in enumerate(pairs):
        groups.setdefault(id(x), []).append(idx)
    mx = 32
    for idxs in groups.values():
        x = pairs[idxs[0]][0]
        if x.dim() != 2 or x.shape[1] % 4 or x.shape[1] > 512 or x.shape[0] > 256 or not x.is_cuda:
            for i in idxs:
                out[i] = pairs[i][1](pairs[i][0])
            continue
        for c0 in range(0, len(idxs), mx):
            chunk = idxs[c0:c0 + mx]
            args = []
            for i in chunk:
                lin = pairs[i][1]
                args += [lin.weight, lin.bias]
            ys = GroupedLinearFunction.apply(x, *args)
            for i, y in zip(chunk, ys):
                out[i] = y
    return out


# --------------------------------------------------------------------------------------
# Row-wise normalisation + activation of the mapping MLPs
# --------------------------------------------------------------------------------------
class RowNormFunction(torch.autograd.Function):
    """y = act(norm(x)) per batch row (cips_rownorm_fwd / _bwd): mode bit 0 LayerNorm(gamma, beta; eps 1e-5), bit 1
    LeakyReLU(0.2) after it, bit 2 PixelNorm (multi_head_mapping.py:13-19, :62-84)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mode):
        lib = _lib.load()
        x = _c(x.detach())
        gamma = _c(gamma.detach()) if gamma is not None else None
        beta = _c(beta.detach()) if beta is not None else None
        _chk(x, gamma, beta)
        rows, cols = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(rows, 2, device=x.device)
        check(lib.cips_rownorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), rows, cols, mode, LRELU_SLOPE, _stream()),
              "cips_rownorm_fwd")
        ctx.save_for_backward(x, y, gamma, stats)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, gamma, stats = ctx.saved_tensors
        rows, cols = x.shape
        dy = _c(dy)
        dx = torch.empty_like(x)
        ln = bool(ctx.mode & 1)
        dyhat = torch.empty_like(x) if ln else None
        dgamma = torch.empty(cols, device=x.device) if ln else None
        dbeta = torch.empty(cols, device=x.device) if ln else None
        check(lib.cips_rownorm_bwd(_p(x), _p(y), _p(gamma), _p(stats), _p(dy), _p(dx), _p(dyhat), _p(dgamma), _p(dbeta), rows,
                                   cols, ctx.mode, LRELU_SLOPE, _stream()), "cips_rownorm_bwd")
        return dx, dgamma, dbeta, None


# --------------------------------------------------------------------------------------
# Fused ray-march (non-hierarchical sampling): rays + SIREN + composite in one kernel
# --------------------------------------------------------------------------------------
MARCH_FUSED = True      # False: flat sampling as rays / SIREN / composite launches (what the hierarchical path uses); the
                        # parity test of the fused kernel against that path flips it


def march_available():
    """the fused march runs on the split-bf16 register chain (forward) and regenerates points in the fused backward"""
    return MARCH_FUSED and SIREN_FWD_MODE == "x3" and SIREN_BWD_MODE == "x3"


def _ray_params(xg, yg, zg, zc, cam2world, jitter, H, W, S, zvals=None):
    from ._lib import RayParams
    r = RayParams()
    r.xg, r.yg, r.zg, r.cam2world, r.jitter, r.zvals = _p(xg), _p(yg), _p(zg), _p(cam2world), _p(jitter), _p(zvals)
    r.zc, r.H, r.W, r.S = float(zc), H, W, S
    return r


class SirenRaysFunction(torch.autograd.Function):
    """feat (B,P,32), sigma (B,P), z (B,P) = siren(points generated in-kernel) — SirenFunction without the (B,P,3)
    points tensor, for the two passes of the hierarchical path: the coarse pass generates its stratified samples from
    (grids, cam2world, jitter) (comm_utils.py:365-438, 584-679), the fine pass from the resampled depths
    `zvals`: origin + direction * z (generator_nerf_inr.py:590-592).  Backward: cips_siren_bwd_x3_rays."""

    @staticmethod
    def forward(ctx, geom, xg, yg, zg, cam2world, jitter, zvals, g0, p0, g1, p1, gc, pc, w0, b0, w1, b1, ws, bs, wc, bc,
                wf, bf):
        lib = _lib.load()
        B, H, W, S, zc = geom
        t = dict(w0=w0, b0=b0, w1=w1, b1=b1, ws=ws, bs=bs, wc=wc, bc=bc, wf=wf, bf=bf,
                 g0=g0, p0=p0, g1=g1, p1=p1, gc=gc, pc=pc)
        t = {k: _c(v.detach()) for k, v in t.items()}
        xg, yg, zg, cam2world = _c(xg), _c(yg), _c(zg), _c(cam2world)
        jitter = _c(jitter) if jitter is not None else None
        zvals = _c(zvals.detach()) if zvals is not None else None
        _chk(xg, yg, zg, cam2world, jitter, zvals, *t.values())
        dev = cam2world.device
        P = H * W * S
        feat = torch.empty(B, P, 32, device=dev)
        sigma = torch.empty(B, P, device=dev)
        z = torch.empty(B, P, device=dev) if zvals is None else zvals
        sw = _siren_struct(t)
        rp = _ray_params(xg, yg, zg, zc, cam2world, jitter, H, W, S, zvals)
        check(lib.cips_siren_fwd_x3_rays(C.byref(sw), C.byref(rp), _p(feat), _p(sigma), _p(z) if zvals is None else None, B,
                                         _stream()), "cips_siren_fwd_x3_rays")
        ctx.save_for_backward(xg, yg, zg, cam2world, jitter, zvals, *[t[k] for k in _SIREN_NAMES])
        ctx.geom = geom
        ctx.mark_non_differentiable(z)
        return feat, sigma, z

    @staticmethod
    def backward(ctx, dfeat, dsigma, _dz):
        xg, yg, zg, cam2world, jitter, zvals = ctx.saved_tensors[:6]
        t = dict(zip(_SIREN_NAMES, ctx.saved_tensors[6:]))
        B, H, W, S, zc = ctx.geom
        P = H * W * S
        dev = cam2world.device
        dfeat = _c(dfeat) if dfeat is not None else torch.zeros(B, P, 32, device=dev)
        dsigma = _c(dsigma) if dsigma is not None else torch.zeros(B, P, device=dev)
        rp = _ray_params(xg, yg, zg, zc, cam2world, jitter, H, W, S, zvals)
        return (None,) * 7 + _siren_backward(t, dfeat, dsigma, B, P, rays=rp)


class RayMarchFunction(torch.autograd.Function):
    """pixels_fea (B,n,32), depth (B,n) = composite(SIREN(points(rays, jitter))) for hierarchical_sample=False: the
    chain get_initial_rays_trig / perturb_points / transform_sampled_points (comm_utils.py:365-438, 584-679) ->
    NeRFNetwork (generator.py:260-317) -> fancy_integration (pigan_utils.py:212-273) as ONE kernel that walks the
    samples along each ray (cips_march_fwd_x3).  Under no_grad nothing per-sample reaches HBM (4 S + 132 B per ray);
    a training forward also writes feat / sigma / z for the backward, which is cips_composite_bwd followed by the fused
    SIREN backward with the points regenerated in-kernel (cips_siren_bwd_x3_rays)."""

    @staticmethod
    def forward(ctx, geom, xg, yg, zg, cam2world, jitter, noise, g0, p0, g1, p1, gc, pc, w0, b0, w1, b1, ws, bs, wc, bc,
                wf, bf):
        lib = _lib.load()
        B, H, W, S, zc, noise_std, clamp_mode, flags, grad_mode = geom
        t = dict(w0=w0, b0=b0, w1=w1, b1=b1, ws=ws, bs=bs, wc=wc, bc=bc, wf=wf, bf=bf,
                 g0=g0, p0=p0, g1=g1, p1=p1, gc=gc, pc=pc)
        t = {k: _c(v.detach()) for k, v in t.items()}
        xg, yg, zg, cam2world = _c(xg), _c(yg), _c(zg), _c(cam2world)
        jitter = _c(jitter) if jitter is not None else None
        noise = _c(noise) if (noise is not None and noise_std != 0.0) else None
        _chk(xg, yg, zg, cam2world, jitter, noise, *t.values())
        dev = cam2world.device
        n = H * W
        # per-sample outputs only when a backward can follow: the caller's grad mode counts (inside forward() it is
        # always off, and under torch.no_grad() needs_input_grad still reports the parameters' requires_grad)
        train = grad_mode and any(ctx.needs_input_grad)
        fea = torch.empty(B, n, 32, device=dev)
        depth = torch.empty(B, n, device=dev)
        feat = torch.empty(B, n * S, 32, device=dev) if train else None
        sigma = torch.empty(B, n * S, device=dev) if train else None
        z = torch.empty(B, n * S, device=dev) if train else None
        sw = _siren_struct(t)
        rp = _ray_params(xg, yg, zg, zc, cam2world, jitter, H, W, S)
        pin, rec = _clamp_debug_forward(B * n, S, clamp_mode, dev)
        check(lib.cips_march_fwd_x3(C.byref(sw), C.byref(rp), _p(noise), float(noise_std), clamp_mode, flags, _p(fea),
                                    _p(depth), None, _p(feat), _p(sigma), _p(z), B, _p(pin), _p(rec), _stream()),
              "cips_march_fwd_x3")
        if rec is not None and CLAMP_REC is not None:
            CLAMP_REC.append(rec.reshape(B * n, S))
        ctx.clamp_mask = pin
        if train:
            ctx.save_for_backward(xg, yg, zg, cam2world, jitter, noise, feat, sigma, z, *[t[k] for k in _SIREN_NAMES])
        ctx.geom = geom
        ctx.mark_non_differentiable(depth)
        return fea, depth

    @staticmethod
    def backward(ctx, dfea, _ddepth):
        lib = _lib.load()
        xg, yg, zg, cam2world, jitter, noise, feat, sigma, z = ctx.saved_tensors[:9]
        t = dict(zip(_SIREN_NAMES, ctx.saved_tensors[9:]))
        B, H, W, S, zc, noise_std, clamp_mode, flags, _ = ctx.geom
        n = H * W
        R = B * n
        dfea = _c(dfea)
        dfeat = torch.empty_like(feat)
        dsig = torch.empty_like(sigma)
        check(lib.cips_composite_bwd(_p(feat), _p(sigma), _p(z), None, None, None, _p(noise), float(noise_std), None,
                                     _p(dfea), _p(dfeat), _p(dsig), None, None, R, S, clamp_mode, flags, _p(ctx.clamp_mask),
                                     _stream()), "cips_composite_bwd")
        _tail_gate_publish(dfea.device)
        rp = _ray_params(xg, yg, zg, zc, cam2world, jitter, H, W, S)
        grads = _siren_backward(t, dfeat, dsig, B, n * S, rays=rp)
        return (None,) * 7 + grads


# --------------------------------------------------------------------------------------
# H4 CIPS INR head (all blocks of CIPSNet.forward in one autograd node)
# --------------------------------------------------------------------------------------
def modfc_prep(W, s, eps=1e-8):
    lib = _lib.load()
    in_dim, out_dim = W.shape
    B = s.shape[0]
    dev = W.device
    wb = torch.empty(B, in_dim, out_dim, device=dev)
    wbt = torch.empty(B, out_dim, in_dim, device=dev)
    demod = torch.empty(B, out_dim, device=dev)
    check(lib.cips_modfc_prep(_p(W), _p(s), _p(wb), _p(wbt), _p(demod), B, in_dim, out_dim, eps, _stream()),
          "cips_modfc_prep")
    return wb, wbt, demod


def modfc_prep_bwd(W, s, demod, gwb):
    lib = _lib.load()
    in_dim, out_dim = W.shape
    B = s.shape[0]
    dev = W.device
    cbuf = torch.empty(B, out_dim, device=dev)
    dW = torch.empty(in_dim, out_dim, device=dev)
    ds = torch.empty(B, in_dim, device=dev)
    check(lib.cips_modfc_prep_bwd(_p(W), _p(s), _p(demod), _p(gwb), _p(cbuf), _p(dW), _p(ds), B, in_dim, out_dim,
                                  _stream()), "cips_modfc_prep_bwd")
    return dW, ds


def torgb_fwd(x2d, w, b, rgb2d, accumulate):
    lib = _lib.load()
    M, K = x2d.shape
    check(lib.cips_torgb_fwd(_p(x2d), _p(w), _p(b), _p(rgb2d), M, K, 1 if accumulate else 0, _stream()),
          "cips_torgb_fwd")


def torgb_bwd_w(x2d, drgb2d):
    lib = _lib.load()
    M, K = x2d.shape
    chunks = lib.cips_torgb_bwd_partials(M)
    part = torch.empty(chunks, 4, K, device=x2d.device)
    dw = torch.empty(3, K, device=x2d.device)
    db = torch.empty(3, device=x2d.device)
    check(lib.cips_torgb_bwd_w(_p(x2d), _p(drgb2d), _p(part), _p(dw), _p(db), M, K, _stream()), "cips_torgb_bwd_w")
    return dw, db


def torgb_bwd_x_x3(drgb2d, w, gate_bits, out_unmasked, P):
    """P (Planes (.., K)) = (drgb2d (M,3) @ w (3,K)) * (gate bit ? 1 : slope); out_unmasked: optional fp32 copy before gating"""
    lib = _lib.load()
    M, K = drgb2d.shape[0], w.shape[1]
    check(lib.cips_torgb_bwd_x_x3(_p(drgb2d), _p(_c(w)), _p(gate_bits), LRELU_SLOPE, _p(out_unmasked), _p(P.hi), _p(P.lo), M, K,
                                  _stream()), "cips_torgb_bwd_x_x3")


def torgb_bwd_x(drgb2d, w, add, mask, out_unmasked, out):
    lib = _lib.load()
    M = drgb2d.shape[0]
    K = w.shape[1]
    check(lib.cips_torgb_bwd_x(_p(drgb2d), _p(w), _p(add), _p(mask), LRELU_SLOPE, _p(out_unmasked), _p(out), M, K,
                               _stream()), "cips_torgb_bwd_x")


# ---- LeakyReLU gate instrumentation of the head (parity tests; never set in production) --------------------------
# A LeakyReLU gate (pre-activation > 0) is the one discontinuity of the head: two fp32 evaluations of the same layer
# may disagree on it for pre-activations within rounding of zero, and each disagreement moves the gradients by a
# finite amount.  Gradient parity is therefore stated for a GIVEN set of gates:
#   GATE_PIN: iterator of uint8 bit planes (B, n, C/8) (bit c&7 of byte c>>3 = gate of column c), one per modulated-FC
#             layer in call order; the forward epilogues then apply `gate ? 1 : slope` from the plane instead of
#             deciding by the sign they computed, and the backward uses the same plane.
#   GATE_REC: list that receives the bit plane every layer actually used, in call order.
GATE_PIN = None
GATE_REC = None
# The other discontinuity of the path: importance resampling places the fine samples by a searchsorted over the coarse
# weights' cdf (pigan_utils.py:164-273); a cdf value within rounding of the uniform draw puts a sample in the
# neighbouring bin.  FINE_Z_REC collects the depths every hierarchical forward resampled (b*n, S); FINE_Z_PIN (an
# iterator of such tensors) replaces them — parity tests compare gradients for the SAME sample placement.
FINE_Z_PIN = None
FINE_Z_REC = None


# The third discontinuity: relu(sigma + nerf_noise * eps) in fancy_integration (pigan_utils.py:246-252).  The reference trains
# its first 5 000 steps with nerf_noise 1 -> 0 (train.py:325-327); a pre-activation within rounding of 0 takes either
# branch, and the two gradients of the sigma head (sums of d sigma with heavy cancellation) move by a finite amount per
# flipped sample.  CLAMP_PIN: iterator of uint8 tensors (R, E) (R rays, E sorted positions; 0 = clamped, else the linear
# branch), one per composite / fused-march forward in call order — forward AND backward then take the branch from it.
# CLAMP_REC: list receiving the branches each forward actually took.
CLAMP_PIN = None
CLAMP_REC = None


class clamp_debug:
    """with clamp_debug(pin=[mask, ...] or None, rec=list or None): ...   (tests only)"""

    def __init__(self, pin=None, rec=None):
        self.pin, self.rec = pin, rec

    def __enter__(self):
        global CLAMP_PIN, CLAMP_REC
        self.old = (CLAMP_PIN, CLAMP_REC)
        CLAMP_PIN = iter(self.pin) if self.pin is not None else None
        CLAMP_REC = self.rec
        return self

    def __exit__(self, *exc):
        global CLAMP_PIN, CLAMP_REC
        CLAMP_PIN, CLAMP_REC = self.old


def _clamp_debug_forward(R, E, clamp_mode, dev):
    """-> (pin, rec) for one composite forward of R rays x E positions: both None outside clamp_debug / for softplus"""
    if (CLAMP_PIN is None and CLAMP_REC is None) or clamp_mode != 0:
        return None, None
    pin = None
    if CLAMP_PIN is not None:
        pin = next(CLAMP_PIN).to(device=dev, dtype=torch.uint8).reshape(-1).contiguous()
        if pin.numel() != R * E:
            raise ValueError(f"clamp_debug: pinned mask of {pin.numel()} entries for {R} rays x {E} positions")
    return pin, (torch.empty(R * E, dtype=torch.uint8, device=dev) if CLAMP_REC is not None else None)


class resample_debug:
    """with resample_debug(pin=[fine_z, ...] or None, rec=list or None): ...   (tests only)"""

    def __init__(self, pin=None, rec=None):
        self.pin, self.rec = pin, rec

    def __enter__(self):
        global FINE_Z_PIN, FINE_Z_REC
        self.old = (FINE_Z_PIN, FINE_Z_REC)
        FINE_Z_PIN = iter(self.pin) if self.pin is not None else None
        FINE_Z_REC = self.rec
        return self

    def __exit__(self, *exc):
        global FINE_Z_PIN, FINE_Z_REC
        FINE_Z_PIN, FINE_Z_REC = self.old


class gate_debug:
    """with gate_debug(pin=planes or None, rec=list or None): ..."""

    def __init__(self, pin=None, rec=None):
        self.pin, self.rec = pin, rec

    def __enter__(self):
        global GATE_PIN, GATE_REC
        self.old = (GATE_PIN, GATE_REC)
        GATE_PIN = iter(self.pin) if self.pin is not None else None
        GATE_REC = self.rec
        return self

    def __exit__(self, *exc):
        global GATE_PIN, GATE_REC
        if exc[0] is None and GATE_PIN is not None:
            assert next(GATE_PIN, None) is None, "pinned gate planes left over"
        GATE_PIN, GATE_REC = self.old


def _next_pin(B, n, C, dev):
    if GATE_PIN is None:
        return None
    p = next(GATE_PIN)
    assert p.dtype == torch.uint8 and tuple(p.shape) == (B, n, C // 8), (tuple(p.shape), (B, n, C // 8))
    return p.to(dev).contiguous()


def _bits_to_pm1(p):
    """uint8 bit plane (..., C/8) -> fp32 (..., C) of +1 / -1 (the fp32 GEMM's `mask` operand reads the sign)"""
    sh = torch.arange(8, device=p.device)
    g = ((p.unsqueeze(-1).to(torch.int32) >> sh) & 1).reshape(*p.shape[:-1], p.shape[-1] * 8)
    return (g * 2 - 1).float()


def _sign_to_bits(a):
    """fp32 (..., C) -> uint8 bit plane (..., C/8) of a > 0"""
    w = (1 << torch.arange(8, device=a.device)).to(torch.int32)
    return ((a > 0).reshape(*a.shape[:-1], a.shape[-1] // 8, 8).to(torch.int32) * w).sum(-1).to(torch.uint8)


class InrHeadFunction(torch.autograd.Function):
    """rgb_pre (B,n,3) = CIPSNet body (generator.py:1107-1153, before the final tanh).

    args: x0 (B,n,in0); then per block k (nblocks of them): W1 (in,out), s1 (B,in), W2 (out,out),
    s2 (B,out); then per block with a ToRGB (k >= 3): T (3,out), tau (3).
    s* = SinStyleMod.modulation(style) (mod_conv_fc.py:474) computed on the host side.
    Blocks k >= 4 use the skip connection (generator.py:1128, 971-973)."""

    @staticmethod
    def forward(ctx, nblocks, x0, *params):
        nblocks, _grad_mode = nblocks if isinstance(nblocks, tuple) else (nblocks, True)
        x0 = _c(x0.detach())
        B, n, _ = x0.shape
        dev = x0.device
        blocks = []
        for k in range(nblocks):
            W1, s1, W2, s2 = [_c(p.detach()) for p in params[4 * k:4 * k + 4]]
            blocks.append((W1, s1, W2, s2))
        rgbp = [_c(p.detach()) for p in params[4 * nblocks:]]
        _chk(x0, *[t for blk in blocks for t in blk], *rgbp)
        saved = []
        x = x0
        rgb = torch.empty(B, n, 3, device=dev)
        first_rgb = True
        for k, (W1, s1, W2, s2) in enumerate(blocks):
            skip = k >= 4
            cout = W1.shape[1]
            wb1, wbt1, d1 = modfc_prep(W1, s1)
            pin1 = _next_pin(B, n, cout, dev)
            # pinned (tests): gate from the supplied plane (`mask` operand: C = acc * (mask > 0 ? 1 : slope))
            m1 = _bits_to_pm1(pin1) if pin1 is not None else None
            a1 = bmm_nn(x, wb1, mask=m1) if pin1 is not None else bmm_nn(x, wb1, act=1)
            wb2, wbt2, d2 = modfc_prep(W2, s2)
            pin2 = _next_pin(B, n, cout, dev)
            m2 = _bits_to_pm1(pin2) if pin2 is not None else None
            e2 = dict(mask=m2) if pin2 is not None else dict(act=1)
            if skip and a1.shape[-1] == x.shape[-1]:
                out = torch.empty_like(a1)
                a2 = bmm_nn(a1, wb2, resid=x, C2=out, **e2)
            else:
                a2 = bmm_nn(a1, wb2, **e2)
                out = a2
            if GATE_REC is not None:
                GATE_REC.append(pin1 if pin1 is not None else _sign_to_bits(a1))
                GATE_REC.append(pin2 if pin2 is not None else _sign_to_bits(a2))
            if k >= 3:
                T, tau = rgbp[2 * (k - 3)], rgbp[2 * (k - 3) + 1]
                torgb_fwd(out.view(B * n, -1), T, tau, rgb.view(B * n, 3), accumulate=not first_rgb)
                first_rgb = False
            # the backward's gate operands: the sign of the stored activations, or the pinned planes
            saved.append((x, a1 if m1 is None else m1, a2 if m2 is None else m2, out, wbt1, d1, wbt2, d2, a1))
            x = out
        if first_rgb:
            rgb.zero_()
        ctx.nblocks = nblocks
        ctx.blocks = blocks
        ctx.rgbp = rgbp
        ctx.saved = saved
        return rgb

    @staticmethod
    def backward(ctx, drgb):
        nblocks, blocks, rgbp, saved = ctx.nblocks, ctx.blocks, ctx.rgbp, ctx.saved
        drgb = _c(drgb)
        B, n, _ = drgb.shape
        dev = drgb.device
        drgb2 = drgb.view(B * n, 3)
        grads_blocks = [None] * nblocks
        grads_rgb = [None] * len(rgbp)
        width = saved[-1][3].shape[-1]
        g2 = torch.empty(B, n, width, device=dev)
        Dout = None     # unmasked grad wrt out_k (kept while block k has a skip)
        dx0 = None
        for k in range(nblocks - 1, -1, -1):
            xin, m1, a2, out, wbt1, d1, wbt2, d2, a1 = saved[k]      # m1, a2: gate operands (sign); a1: activation
            W1, s1, W2, s2 = blocks[k]
            skip = (k >= 4) and (a1.shape[-1] == xin.shape[-1])
            if k == nblocks - 1:
                if k >= 3:
                    T = rgbp[2 * (k - 3)]
                    Dout = torch.empty(B, n, width, device=dev) if skip else None
                    torgb_bwd_x(drgb2, T, None, a2, Dout, g2)
                else:  # degenerate configuration (fewer than 4 blocks): no gradient reaches the head
                    g2.zero_()
                    Dout = torch.zeros(B, n, width, device=dev) if skip else None
            if k >= 3:
                dT, dtau = torgb_bwd_w(out.view(B * n, -1), drgb2)
                grads_rgb[2 * (k - 3)], grads_rgb[2 * (k - 3) + 1] = dT, dtau
            # ---- mod2: y2 = a1 @ wb2 ----
            gwb2 = bmm_tn(a1, g2)                                  # (B, out, out) = a1^T @ g2
            dW2, ds2 = modfc_prep_bwd(W2, s2, d2, gwb2)
            g1 = bmm_nn(g2, wbt2, mask=m1)                         # dL/dy1 = (g2 @ wb2^T) * lrelu'(a1)
            # ---- mod1: y1 = xin @ wb1 ----
            gwb1 = bmm_tn(xin, g1)
            dW1, ds1 = modfc_prep_bwd(W1, s1, d1, gwb1)
            grads_blocks[k] = (dW1, ds1, dW2, ds2)
            if k == 0:
                dx0 = bmm_nn(g1, wbt1)
            else:
                # grad wrt out_{k-1} = dx1_k [+ Dout_k via skip] [+ drgb @ T_{k-1}], then the gate of a2_{k-1}
                a2_prev = saved[k - 1][2]
                prev_skip = (k - 1 >= 4) and (saved[k - 1][1].shape[-1] == saved[k - 1][0].shape[-1])
                newD = torch.empty(B, n, width, device=dev) if prev_skip else None
                epi = dict(mask=a2_prev, C_unmasked=newD)
                if skip:
                    epi["add"] = Dout
                if k - 1 >= 3:
                    epi["rgb_g"] = drgb2
                    epi["rgb_w"] = rgbp[2 * (k - 1 - 3)]
                g2 = bmm_nn(g1, wbt1, out=torch.empty(B, n, width, device=dev), **epi)
                Dout = newD
        flat = [None, dx0]
        for gb in grads_blocks:
            flat.extend(gb)
        flat.extend(grads_rgb)
        return tuple(flat)


# --------------------------------------------------------------------------------------
# H4 on the bf16 matrix cores: 3-pass split GEMM ("bf16x3"), see csrc/gemm_bf16x3.hip
# --------------------------------------------------------------------------------------
import os as _os
INR_MODE = _os.environ.get("CIPS_INR_MODE", "bf16x3")   # "bf16x3" (default, ~1e-5 rel. per layer) or "f32" (exact fp32 MFMA)
BF = torch.bfloat16
INR_W_KMAJOR = True   # dW GEMMs read the row-major planes through LDS transpose reads (no transposed copies in HBM)
# The head's weight-gradient tail (style / modulated-weight gradients of all layers, ToRGB weight gradients: ~0.75 ms of streaming
# kernels at C2) depends on nothing that follows the head in the backward pass.  "side": it hangs off the head through two
# gradient ports whose autograd nodes live on the generator's side stream, so it runs next to the compositing / SIREN backward
# instead of in front of it (CIPSNet.forward; profiles/r6_cores_tail_probe.txt).  "main": everything inside InrHeadX3Function.
INR_TAIL = _os.environ.get("CIPS_INR_TAIL", "side")
                                                                 # transpose reads: no transposed planes in HBM


class Planes:
    """An fp32 matrix carried as two bf16 planes (x = hi + lo)."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo

    @staticmethod
    def empty(*shape, device):
        return Planes(torch.empty(*shape, device=device, dtype=BF), torch.empty(*shape, device=device, dtype=BF))

    def float(self):
        return self.hi.float() + self.lo.float()


X3_KERNEL = 0        # descriptor field `kernel` of every split-bf16 GEMM issued from here: 0 = the library's automatic choice
                     # (production); 1 / 2 / 3 force a kernel (cips3d_hip.h) — set by the kernel parity tests and microbenchmarks


def _x3_desc(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, C=None, P=None, T=None, ldt=0, strideT=0,
             mask_out=None, add=None, rgb_g=None, rgb_w=None, C_unmasked=None, mask=None, act=0, res=None, gate_bits=0,
             torgb=None, addp=None):
    d = GemmX3Desc()
    d.A_hi, d.A_lo, d.B_hi, d.B_lo = _p(A.hi), _p(A.lo), _p(Bm.hi), _p(Bm.lo)
    d.M, d.N, d.K, d.lda, d.ldb = M, N, K, lda, ldb
    d.strideA, d.strideB, d.batch = strideA, strideB, batch
    d.C, d.ldc, d.strideC = _p(C), N, M * N
    d.P_hi, d.P_lo = (_p(P.hi), _p(P.lo)) if P is not None else (None, None)
    d.ldp, d.strideP = N, M * N
    d.T_hi, d.T_lo = (_p(T.hi), _p(T.lo)) if T is not None else (None, None)
    d.ldt, d.strideT = ldt, strideT
    d.mask_out, d.add, d.rgb_g, d.rgb_w = _p(mask_out), _p(add), _p(rgb_g), _p(rgb_w)
    d.C_unmasked, d.mask = _p(C_unmasked), _p(mask)
    d.act, d.slope = act, LRELU_SLOPE
    d.res_hi, d.res_lo = (_p(res.hi), _p(res.lo)) if res is not None else (None, None)
    d.gate_bits = gate_bits        # bit 0: `mask` is a uint8 bit plane (M, N/8); bit 1: `mask_out` is written as one
    d.kernel = X3_KERNEL
    if torgb is not None:          # (T (3, N), partials (N/128, batch*M, 4)): ToRGB forward folded into the epilogue
        d.torgb_w, d.torgb_part = _p(torgb[0]), _p(torgb[1])
    if addp is not None:           # (Planes of a gated tensor, bit plane of that gate): the addend, un-gated on the fly
        d.addp_hi, d.addp_lo, d.addp_gate, d.addp_gain = _p(addp[0].hi), _p(addp[0].lo), _p(addp[1]), 1.0 / LRELU_SLOPE
    return d


_ADDP_OK = {}


def _addp_shape_ok(n, cin, cout, nb, dev):
    """does the library take the planes addend at this shape?  (a property of the shape and of the kernel selector: asked
    once per shape with stand-in pointers — the query reads only shapes, flags and pointer alignment)"""
    key = (n, cin, cout, nb, dev.index, X3_KERNEL)
    ok = _ADDP_OK.get(key)
    if ok is None:
        gq = Planes.empty(1, 8, 8, device=dev)
        ok = _ADDP_OK[key] = gemm_x3_takes_addp(gq, gq, n, cin, cout, cout, cout, nb, n * cout, cin * cout, P=gq,
                                                addp=(gq, gq.hi), mask=gq.hi, gate_bits=1)
    return ok


def gemm_x3_takes_addp(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, **epi):
    """True when the library runs this descriptor with the planes addend (256x256-tile kernel, interior shapes)"""
    lib = _lib.load()
    d = _x3_desc(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, **epi)
    return bool(lib.cips_gemm_bf16x3_takes_addp(_ct.byref(d)))


def gemm_x3(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, **epi):
    """C[b][m][n] = epi(sum_k A[b][m][k] * B[b][n][k]); A, B, P, T, res: Planes; row-major aux use ld = N."""
    lib = _lib.load()
    d = _x3_desc(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, **epi)
    check(lib.cips_gemm_bf16x3(_ct.byref(d), _stream()), "cips_gemm_bf16x3")


def gemm_x3_torgb(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, P, rgb_w, rgb_b, rgb2d, accumulate, **epi):
    """gemm_x3 with planes output P followed by ToRGB forward on it (rgb2d (batch*M, 3) (+)= P . rgb_w^T + rgb_b):
    folded into the GEMM epilogue when the library's 256x256-tile kernel takes the shape (partials per 128-column
    block + one finishing launch), else the ToRGB kernel on the written planes."""
    lib = _lib.load()
    part = torch.empty(max(N // 128, 1), batch * M, 4, device=P.hi.device)
    d = _x3_desc(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, P=P, torgb=(rgb_w, part), **epi)
    if N % 128 == 0 and rgb_w.is_contiguous() and lib.cips_gemm_bf16x3_fuses_torgb(_ct.byref(d)):
        check(lib.cips_gemm_bf16x3(_ct.byref(d), _stream()), "cips_gemm_bf16x3")
        check(lib.cips_torgb_finish(_p(part), N // 128, _p(rgb_b), _p(rgb2d), batch * M, 1 if accumulate else 0, _stream()),
              "cips_torgb_finish")
        return
    d.torgb_w, d.torgb_part = None, None
    check(lib.cips_gemm_bf16x3(_ct.byref(d), _stream()), "cips_gemm_bf16x3")
    torgb_fwd_x3(P, rgb_w, rgb_b, rgb2d, accumulate)


def gemm_x3_km(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, C):
    """C[b][m][n] = sum_k A[b][k][m] * B[b][k][n]; A, B: Planes stored [K][ld] (k-major), C fp32 (M,N)."""
    lib = _lib.load()
    d = GemmX3Desc()
    d.A_hi, d.A_lo, d.B_hi, d.B_lo = _p(A.hi), _p(A.lo), _p(Bm.hi), _p(Bm.lo)
    d.M, d.N, d.K, d.lda, d.ldb = M, N, K, lda, ldb
    d.strideA, d.strideB, d.batch = strideA, strideB, batch
    d.C, d.ldc, d.strideC = _p(C), N, M * N
    d.slope = LRELU_SLOPE
    d.kernel = X3_KERNEL
    check(lib.cips_gemm_bf16x3_km(_ct.byref(d), _stream()), "cips_gemm_bf16x3_km")


def gemm_x3_km_grouped(problems, M, N, K, lda, ldb, batch, strideA, strideB):
    """problems: list of (A Planes, B Planes, C fp32) of one shape -> one launch when the library supports the shape
    (256x256 tiles), else one K-major GEMM per problem."""
    lib = _lib.load()
    descs = (GemmX3Desc * len(problems))()
    for d, (A, Bm, C) in zip(descs, problems):
        d.A_hi, d.A_lo, d.B_hi, d.B_lo = _p(A.hi), _p(A.lo), _p(Bm.hi), _p(Bm.lo)
        d.M, d.N, d.K, d.lda, d.ldb = M, N, K, lda, ldb
        d.strideA, d.strideB, d.batch = strideA, strideB, batch
        d.C, d.ldc, d.strideC = _p(C), N, M * N
        d.slope = LRELU_SLOPE
        d.kernel = X3_KERNEL
    rc = lib.cips_gemm_bf16x3_km_grouped(descs, len(problems), _stream())
    if rc == 801:      # hipErrorNotSupported
        for (A, Bm, C) in problems:
            gemm_x3_km(A, Bm, M, N, K, lda, ldb, batch, strideA, strideB, C)
        return
    check(rc, "cips_gemm_bf16x3_km_grouped")


def split_planes(x, want_p=True, want_t=True):
    """x (B, rows, cols) fp32 -> Planes row-major (B,rows,cols) and transposed (B,cols,rows)."""
    lib = _lib.load()
    B, rows, cols = x.shape
    dev = x.device
    P = Planes.empty(B, rows, cols, device=dev) if want_p else None
    T = Planes.empty(B, cols, rows, device=dev) if want_t else None
    check(lib.cips_split_planes(_p(x), _p(P.hi) if P else None, _p(P.lo) if P else None,
                                _p(T.hi) if T else None, _p(T.lo) if T else None, rows, cols, cols, cols, rows, B,
                                rows * cols, rows * cols, rows * cols, _stream()), "cips_split_planes")
    return P, T


def conv1x1_smallk(x, w):
    """x (B, C<=4, H, W), w (O, C) -> y (B, O, H, W): the RGB input convs as a streaming kernel"""
    lib = _lib.load()
    B, C, H, W = x.shape
    O = w.shape[0]
    y = torch.empty(B, O, H, W, device=x.device)
    check(lib.cips_conv1x1_smallk(_p(x), _p(w), _p(y), B, C, O, H * W, _stream()), "cips_conv1x1_smallk")
    return y


def conv1x1_smallk_bwd_data(dy, w, C):
    """dy (B, O, H, W), w (O, C<=4) -> dx (B, C, H, W)"""
    lib = _lib.load()
    B, O, H, W = dy.shape
    dx = torch.empty(B, C, H, W, device=dy.device)
    check(lib.cips_conv1x1_smallk_bwd_data(_p(dy), _p(w), _p(dx), B, C, O, H * W, _stream()), "cips_conv1x1_smallk_bwd_data")
    return dx


def conv1x1_smallk_bwd_weight(dy, x):
    """dy (B, O, H, W), x (B, C<=4, H, W) -> dw (O, C) = sum over images and pixels of dy x^T: the weight gradient of the
    RGB input convs (autograd of F.conv2d in EqualConv2d.forward, exp/cips3d/models/discriminator.py:40-48, for the 1x1
    layers of :457-459)"""
    lib = _lib.load()
    B, O, H, W = dy.shape
    C = x.shape[1]
    S = lib.cips_conv1x1_smallk_bwd_weight_splits(O, H * W)
    part = torch.empty(S, O, C, device=dy.device)
    check(lib.cips_conv1x1_smallk_bwd_weight(_p(dy), _p(x), _p(part), B, C, O, H * W, _stream()), "cips_conv1x1_smallk_bwd_weight")
    return part.sum(0) if S > 1 else part[0]


def split_planes_nhwc(x):
    """x (B, C, H, W) fp32 NCHW -> NHWC split planes (B*H*W + 1, C): the transposing form of cips_split_planes; the
    extra last row is zero (the implicit-GEMM convolution reads it wherever a tap falls into the padding)."""
    lib = _lib.load()
    B, C, H, W = x.shape
    n = H * W
    hi = torch.empty(B * n + 1, C, device=x.device, dtype=BF)
    lo = torch.empty(B * n + 1, C, device=x.device, dtype=BF)
    if C % 8 == 0 and x.dtype == torch.float32 and x.is_contiguous():
        check(lib.cips_split_planes_nhwc(_p(x), _p(hi), _p(lo), B, C, n, _stream()), "cips_split_planes_nhwc")
        return Planes(hi, lo)
    hi[B * n:].zero_(); lo[B * n:].zero_()
    check(lib.cips_split_planes(_p(x), None, None, _p(hi), _p(lo), C, n, n, n, C, B, C * n, C * n, C * n, _stream()),
          "cips_split_planes")
    return Planes(hi, lo)


def conv2d_x3(wP, xP, B, C, H, W, O, kh, kw, stride, pad, ksplit=None, bias=None, act=False, slope=0.2, act_scale=1.0):
    """Implicit-GEMM convolution: wP Planes (O, kh*kw*C) with contraction index (tap, channel), xP NHWC Planes from
    split_planes_nhwc -> y (B, O, Ho, Wo) fp32."""
    lib = _lib.load()
    from ._lib import ConvX3Desc
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = torch.empty(B, O, Ho, Wo, device=xP.hi.device)
    d = ConvX3Desc()
    d.w_hi, d.w_lo, d.x_hi, d.x_lo, d.y = _p(wP.hi), _p(wP.lo), _p(xP.hi), _p(xP.lo), _p(y)
    d.B, d.C, d.H, d.W, d.O, d.kh, d.kw, d.stride, d.pad = B, C, H, W, O, kh, kw, stride, pad
    ks = lib.cips_conv2d_x3_ksplit(B, O, Ho * Wo, kh * kw * C) if ksplit is None else ksplit
    part = None
    if ks > 1:                                   # few output tiles: split the contraction over the idle CUs
        part = torch.empty(ks, B, O, Ho, Wo, device=xP.hi.device)
        d.ksplit, d.part = ks, _p(part)
    else:
        d.ksplit, d.part = 1, None
    d.bias = _p(bias) if bias is not None else None            # EqualConv2d + FusedLeakyReLU in the epilogue
    d.act, d.slope, d.act_scale = (1 if act else 0), float(slope), float(act_scale)
    check(lib.cips_conv2d_x3(_ct.byref(d), _stream()), "cips_conv2d_x3")
    return y


def dgrad_s2_banks(w):
    """w (O, C, kh, kw) fp32 -> (Planes of the four parity filter banks back to back, element offsets [4]) for
    conv2d_x3_dgrad_s2: bank (a, b) = [C][(ty, tx, o)] with w[o][c][a + 2(Ta-1-ty)][b + 2(Tb-1-tx)] (include/cips3d_hip.h)."""
    O, C, kh, kw = w.shape
    parts, offs, tot = [], [], 0
    for a in range(2):
        for b in range(2):
            offs.append(tot)
            sub = w[:, :, a::2, b::2]
            if sub.numel() == 0:
                continue
            bank = sub.flip(2, 3).permute(1, 2, 3, 0).reshape(-1)          # (C, Ta, Tb, O)
            parts.append(bank)
            tot += bank.numel()
    flat = torch.cat(parts).contiguous()
    P, _ = split_planes(flat.view(1, tot // 32, 32), want_p=True, want_t=False)
    return P, offs


def dgrad_s2_layout(H, W):
    """-> (element offsets [4] of the parity blocks per (B*C) planes, plane sizes Np[4]) for an (H, W) input"""
    nps = []
    for a in range(2):
        for b in range(2):
            hs, ws = (H - a + 1) // 2, (W - b + 1) // 2
            nps.append((hs * ws + 7) // 8 * 8)
    return nps


def conv2d_x3_dgrad_s2(banks, w_off, dyP, B, C, H, W, O, kh, kw):
    """Data gradient of a stride-2 unpadded convolution as four parity sub-convolutions in one launch (cips_conv2d_x3_dgrad_s2).
    banks / w_off: dgrad_s2_banks(w); dyP: NHWC Planes of dy (B, O, Ho, Wo).  -> (dxp fp32 flat, out_off [4]): the gradient w.r.t.
    the (B, C, H, W) input in parity-block layout, consumed by upfirdn2d_parity (or parity_to_nchw in tests)."""
    lib = _lib.load()
    from ._lib import ConvDgradS2Desc
    nps = dgrad_s2_layout(H, W)
    # the four blocks are read TOGETHER by upfirdn2d_parity (one output needs all four parities); with B*C a power of two
    # their natural offsets are congruent modulo 64 KiB and the four streams queue on the same memory channels — skew them
    out_off, tot = [], 0
    for i, n_ in enumerate(nps):
        tot += i * 1184                     # floats: 4 736 B, 16-byte granular
        out_off.append(tot)
        tot += B * C * n_
    dxp = torch.empty(tot, device=dyP.hi.device)
    d = ConvDgradS2Desc()
    d.w_hi, d.w_lo, d.dy_hi, d.dy_lo, d.dxp = _p(banks.hi), _p(banks.lo), _p(dyP.hi), _p(dyP.lo), _p(dxp)
    d.B, d.C, d.H, d.W, d.O, d.kh, d.kw = B, C, H, W, O, kh, kw
    for i in range(4):
        d.w_off[i] = w_off[i]
        d.out_off[i] = out_off[i]
    check(lib.cips_conv2d_x3_dgrad_s2(_ct.byref(d), _stream()), "cips_conv2d_x3_dgrad_s2")
    return dxp, out_off


def parity_to_nchw(dxp, out_off, B, C, H, W):
    """the parity-block layout of conv2d_x3_dgrad_s2 as an ordinary (B, C, H, W) tensor (tests / fallbacks: torch indexing)"""
    out = torch.empty(B, C, H, W, device=dxp.device)
    nps = dgrad_s2_layout(H, W)
    for a in range(2):
        for b in range(2):
            hs, ws = (H - a + 1) // 2, (W - b + 1) // 2
            i = 2 * a + b
            blk = dxp[out_off[i]:out_off[i] + B * C * nps[i]].view(B, C, nps[i])[:, :, :hs * ws].reshape(B, C, hs, ws)
            out[:, :, a::2, b::2] = blk
    return out


def upfirdn2d_parity(dxp, out_off, kernel, major, in_h, in_w, pad_x0, pad_x1, pad_y0, pad_y1):
    """cips_upfirdn2d (4 x 4 kernel, up = down = 1) on planes stored as conv2d_x3_dgrad_s2's parity blocks -> (major, out_h, out_w)"""
    lib = _lib.load()
    k = _native_in(kernel, "kernel")
    if tuple(k.shape) != (4, 4):
        raise RuntimeError("upfirdn2d_parity: 4 x 4 kernels only")
    out_h, out_w = in_h + pad_y0 + pad_y1 - 3, in_w + pad_x0 + pad_x1 - 3
    out = torch.empty(major, out_h, out_w, device=dxp.device)
    offs = (_ct.c_longlong * 4)(*out_off)
    check(lib.cips_upfirdn2d_parity(_p(dxp), _ct.byref(offs), _p(k), _p(out), major, in_h, in_w, pad_x0, pad_x1, pad_y0, pad_y1,
                                    _stream()), "cips_upfirdn2d_parity")
    return out


def conv2d_x3_wgrad(dyP, xP, B, C, H, W, O, kh, kw, stride, pad, scale=1.0, nch=None):
    """Weight gradient of conv2d_x3: dyP, xP NHWC Planes from split_planes_nhwc -> dW (O, C, kh, kw) fp32, or None when
    the pixel count does not split into 32-row k-tiles.  The pixel range is cut into chunks so that a 512x512 filter bank
    still fills the chip (4 tiles per tap and chunk); cips_conv_wgrad_finish adds the partial sums, applies `scale` and
    lays the result out as (O, C, kh, kw)."""
    lib = _lib.load()
    from ._lib import ConvWgradDesc
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    K = B * Ho * Wo
    if K % 32:
        return None
    # chunks of the pixel range: the persistent grid runs ceil(tiles*nch / 256) rounds of ~K/nch rows each (+ an
    # epilogue worth ~512 rows); 36 tiles x 7 chunks = 252 of 256 CUs in one round (powers of two: 144, or 288 in two)
    tiles = ((O + 255) // 256) * ((C + 255) // 256) * kh * kw
    T = K // 32
    best = None
    forced = nch
    nch = 1
    for cand in range(1, 33):                   # any count: the kernel cuts the range at k-tile granularity
        if cand > 1 and T // cand < 8:
            break
        cost = -(-tiles * cand // 256) * (-(-T // cand) * 32 + 512)
        if best is None or cost < best:
            nch, best = cand, cost
    if forced is not None:
        nch = forced
    part = torch.empty(nch, kh * kw, O, C, device=xP.hi.device)
    d = ConvWgradDesc()
    d.dy_hi, d.dy_lo, d.x_hi, d.x_lo, d.part = _p(dyP.hi), _p(dyP.lo), _p(xP.hi), _p(xP.lo), _p(part)
    d.B, d.C, d.H, d.W, d.O, d.kh, d.kw, d.stride, d.pad, d.nchunks = B, C, H, W, O, kh, kw, stride, pad, nch
    check(lib.cips_conv2d_x3_wgrad(_ct.byref(d), _stream()), "cips_conv2d_x3_wgrad")
    dw = torch.empty(O, C, kh, kw, device=xP.hi.device)
    check(lib.cips_conv_wgrad_finish(_p(part), _p(dw), nch, kh * kw, O, C, float(scale), _stream()), "cips_conv_wgrad_finish")
    return dw


def modfc_prep_x3(W, s, eps=1e-8):
    lib = _lib.load()
    in_dim, out_dim = W.shape
    B = s.shape[0]
    dev = W.device
    wb = Planes.empty(B, in_dim, out_dim, device=dev)
    wbt = Planes.empty(B, out_dim, in_dim, device=dev)
    demod = torch.empty(B, out_dim, device=dev)
    check(lib.cips_modfc_prep_x3(_p(W), _p(s), _p(wb.hi), _p(wb.lo), _p(wbt.hi), _p(wbt.lo), _p(demod), B, in_dim,
                                 out_dim, eps, _stream()), "cips_modfc_prep_x3")
    return wb, wbt, demod


def modfc_prep_x3_batch(layers, eps=1e-8):
    """layers: list of (W (in,out), s (B,in)) -> list of (wb Planes (B,in,out), wbt Planes (B,out,in), demod (B,out)),
    all layers in two launches."""
    lib = _lib.load()
    from ._lib import ModfcPrepJob
    mx = lib.cips_modfc_max_jobs()
    out = []
    for c0 in range(0, len(layers), mx):
        chunk = layers[c0:c0 + mx]
        jobs = (ModfcPrepJob * len(chunk))()
        B = chunk[0][1].shape[0]
        for j, (W, s) in zip(jobs, chunk):
            in_dim, out_dim = W.shape
            dev = W.device
            wb = Planes.empty(B, in_dim, out_dim, device=dev)
            wbt = Planes.empty(B, out_dim, in_dim, device=dev)
            demod = torch.empty(B, out_dim, device=dev)
            j.weight, j.s, j.demod = _p(W), _p(s), _p(demod)
            j.wb_hi, j.wb_lo, j.wbt_hi, j.wbt_lo = _p(wb.hi), _p(wb.lo), _p(wbt.hi), _p(wbt.lo)
            j.in_dim, j.out_dim = in_dim, out_dim
            out.append((wb, wbt, demod))
        check(lib.cips_modfc_prep_x3_batch(jobs, len(chunk), B, eps, _stream()), "cips_modfc_prep_x3_batch")
    return out


def modfc_prep_bwd_batch(layers, cores=False):
    """layers: list of (W, s, demod, gwb) -> list of (dW, ds), all layers in three launches.  cores=True: the co-resident
    form (no LDS, <= 40 VGPRs: runs beside the fused SIREN backward when issued on a side stream; B <= 64, out % 4 == 0)."""
    lib = _lib.load()
    from ._lib import ModfcBwdJob
    mx = lib.cips_modfc_max_jobs()
    out = []
    cores = bool(cores) and layers[0][1].shape[0] <= 64 and all(W.shape[1] % 4 == 0 for W, _, _, _ in layers)
    parts = lib.cips_cores_colsum_parts() if cores else 1
    for c0 in range(0, len(layers), mx):
        chunk = layers[c0:c0 + mx]
        jobs = (ModfcBwdJob * len(chunk))()
        B = chunk[0][1].shape[0]
        keep = []
        for j, (W, s, demod, gwb) in zip(jobs, chunk):
            in_dim, out_dim = W.shape
            dev = W.device
            cbuf = torch.empty(parts, B, out_dim, device=dev)
            dW = torch.empty(in_dim, out_dim, device=dev)
            ds = torch.empty(B, in_dim, device=dev)
            j.weight, j.s, j.demod, j.gwb = _p(W), _p(s), _p(demod), _p(gwb)
            j.cbuf, j.dweight, j.ds = _p(cbuf), _p(dW), _p(ds)
            j.in_dim, j.out_dim = in_dim, out_dim
            keep.append(cbuf)
            out.append((dW, ds))
        if cores:
            check(lib.cips_modfc_prep_bwd_batch_cores(jobs, len(chunk), B, _stream()), "cips_modfc_prep_bwd_batch_cores")
        else:
            check(lib.cips_modfc_prep_bwd_batch(jobs, len(chunk), B, _stream()), "cips_modfc_prep_bwd_batch")
    return out


def torgb_fwd_x3(xp, w, b, rgb2d, accumulate):
    lib = _lib.load()
    K = xp.hi.shape[-1]
    M = xp.hi.numel() // K
    check(lib.cips_torgb_fwd_x3(_p(xp.hi), _p(xp.lo), _p(w), _p(b), _p(rgb2d), M, K, 1 if accumulate else 0,
                                _stream()), "cips_torgb_fwd_x3")


def torgb_bwd_w_x3(xp, drgb2d):
    lib = _lib.load()
    K = xp.hi.shape[-1]
    M = xp.hi.numel() // K
    chunks = lib.cips_torgb_bwd_partials(M)
    part = torch.empty(chunks, 4, K, device=drgb2d.device)
    dw = torch.empty(3, K, device=drgb2d.device)
    db = torch.empty(3, device=drgb2d.device)
    check(lib.cips_torgb_bwd_w_x3(_p(xp.hi), _p(xp.lo), _p(drgb2d), _p(part), _p(dw), _p(db), M, K, _stream()),
          "cips_torgb_bwd_w_x3")
    return dw, db


def torgb_bwd_w_x3_batch(xps, drgb2d, cores=False):
    """the ToRGB weight / bias gradients of several taps (Planes of one shape) against one drgb: list of (dw (3, K), db (3,)).
    Two launches for all of them when K = 512 (<= 8 taps), else one pair per tap.  cores=True: the co-resident form of the
    batched launch (see modfc_prep_bwd_batch)."""
    lib = _lib.load()
    K = xps[0].hi.shape[-1]
    M = xps[0].hi.numel() // K
    n = len(xps)
    if K != 512 or n > 8 or n < 2:
        return [torgb_bwd_w_x3(xp, drgb2d) for xp in xps]
    chunks = lib.cips_torgb_bwd_partials(M)
    dev = drgb2d.device
    part = torch.empty(n, chunks, 4, K, device=dev)
    dw = torch.empty(n, 3, K, device=dev)
    db = torch.empty(n, 3, device=dev)
    hi = (_ct.c_void_p * n)(*[_p(xp.hi) for xp in xps])
    lo = (_ct.c_void_p * n)(*[_p(xp.lo) for xp in xps])
    fn = lib.cips_torgb_bwd_w_x3_batch_cores if cores else lib.cips_torgb_bwd_w_x3_batch
    check(fn(hi, lo, n, _p(drgb2d), _p(part), _p(dw), _p(db), M, K, _stream()), "cips_torgb_bwd_w_x3_batch")
    return [(dw[i], db[i]) for i in range(n)]


# LeakyReLU gates of the head kept as bit planes (1 bit per activation, written by the forward GEMMs' epilogues) instead
# of bf16 planes: 2.5 GB less HBM traffic per C2 step (layer widths that are not multiples of 32 keep the bf16 form).
INR_GATE_BITS = True


# ToRGB forward folded into the epilogue of the block's second GEMM wherever the 256x256-tile kernel takes the shape
# (else the separate ToRGB kernel).  Backward: the skip gradient is re-read from the previous layer's GATED planes (un-gated
# on the fly with that gate's bit plane) instead of from a separate fp32 copy the previous GEMM would have to write, and the
# ToRGB tap's gradient into the last block's output is one streaming kernel — both where the shapes allow (bit-plane gates,
# interior tiles); the fp32 copy / the K = 32 zero-padded GEMM remain as the ragged-shape forms.


def _bsl(t, b0, b1):
    """images b0..b1 of a (B, ...) tensor / Planes (a contiguous view)"""
    if t is None:
        return None
    if isinstance(t, Planes):
        return Planes(t.hi[b0:b1], t.lo[b0:b1])
    return t[b0:b1]


def _chunk_ranges(B):
    """image ranges the head's launch chain is cut into: one (two chains on two streams measured slower, profiles/HISTORY.md section 3)"""
    return [(0, B)]


def _run_chunks(ranges, fn, dev):
    for r in ranges:
        fn(*r)


class InrHeadX3Function(torch.autograd.Function):
    """Same contract as InrHeadFunction, on the bf16x3 GEMM.  Every activation / gradient lives in HBM
    as bf16 hi/lo planes in both orientations (row-major for the forward / dX operand, transposed for
    the dW operand), written by the producing GEMM's epilogue."""

    @staticmethod
    def forward(ctx, nblocks, x0, *params):
        ports = None
        if isinstance(nblocks, tuple) and len(nblocks) == 3:
            # (nblocks, grad mode, state of the two gradient ports): params carries, after the usual tensors (passed detached),
            # one handle per modulated layer and the ToRGB handle; their "gradients" are dL/dWb and drgb (see _ModPrepGradPort)
            nblocks, grad_mode, ports = nblocks
            params = params[:len(params) - 2 * nblocks - 1]
        else:
            nblocks, grad_mode = nblocks if isinstance(nblocks, tuple) else (nblocks, True)
        x0 = _c(x0.detach())
        B, n, in0 = x0.shape
        if n % 32 or in0 % 32:
            raise RuntimeError("bf16x3 INR path needs pixels per image and feature width to be multiples of 32")
        dev = x0.device
        blocks = []
        for k in range(nblocks):
            blocks.append(tuple(_c(p.detach()) for p in params[4 * k:4 * k + 4]))
        rgbp = [_c(p.detach()) for p in params[4 * nblocks:]]
        _chk(x0, *[t for blk in blocks for t in blk], *rgbp)
        train = grad_mode and any(ctx.needs_input_grad)       # no-grad / inference: no transposed planes, nothing kept
        want_t = train and not INR_W_KMAJOR     # K-major dW form reads the row-major planes: no transposed copies
        x0P, x0T = split_planes(x0, want_t=want_t)
        rgb = torch.empty(B, n, 3, device=dev)
        # modulate / demodulate / split every layer's weights up front, in one batch
        prepped = modfc_prep_x3_batch([(W, s_) for (W1, s1, W2, s2) in blocks for (W, s_) in ((W1, s1), (W2, s2))])
        dbg = GATE_PIN is not None or GATE_REC is not None          # gate instrumentation (tests): bit planes always
        # every full-batch buffer is allocated here, on the caller's stream, before the chains fork
        plan = []
        for k, (W1, s1, W2, s2) in enumerate(blocks):
            cin, cout = W1.shape
            bits = (train or dbg) and (INR_GATE_BITS or dbg) and cout % 32 == 0
            pin1 = _next_pin(B, n, cout, dev)
            pin2 = _next_pin(B, n, cout, dev)
            if (pin1 is not None or GATE_REC is not None) and not bits:
                raise RuntimeError("gate instrumentation needs layer widths that are multiples of 32")
            skip = (k >= 4) and (cin == cout)
            e = dict(cin=cin, cout=cout, bits=bits, pin1=pin1, pin2=pin2, skip=skip)
            e["a1P"] = Planes.empty(B, n, cout, device=dev)
            e["a1T"] = Planes.empty(B, cout, n, device=dev) if want_t else None
            e["a1g"] = pin1 if pin1 is not None else (torch.empty(B, n, cout // 8, device=dev, dtype=torch.uint8) if bits else None)
            e["oP"] = Planes.empty(B, n, cout, device=dev)
            e["oT"] = Planes.empty(B, cout, n, device=dev) if want_t else None
            if pin2 is not None:
                e["m2"] = pin2
            elif bits:
                e["m2"] = torch.empty(B, n, cout // 8, device=dev, dtype=torch.uint8)
            elif skip:
                e["m2"] = torch.empty(B, n, cout, device=dev, dtype=BF) if train else None
            else:
                e["m2"] = e["oP"].hi
            plan.append(e)
        any_rgb = nblocks > 3

        def run(b0, b1):
            nb = b1 - b0
            xP = _bsl(x0P, b0, b1)
            first_rgb = True
            for k, e in enumerate(plan):
                cin, cout, bits, skip = e["cin"], e["cout"], e["bits"], e["skip"]
                wb1, wbt1, d1 = prepped[2 * k]
                wb2, wbt2, d2 = prepped[2 * k + 1]
                wbt1, wbt2 = _bsl(wbt1, b0, b1), _bsl(wbt2, b0, b1)
                a1P, a1T, a1g = _bsl(e["a1P"], b0, b1), _bsl(e["a1T"], b0, b1), _bsl(e["a1g"], b0, b1)
                oP, oT, m2 = _bsl(e["oP"], b0, b1), _bsl(e["oT"], b0, b1), _bsl(e["m2"], b0, b1)
                if e["pin1"] is not None:
                    # pinned: `gate ? 1 : slope` from the supplied plane in place of the LeakyReLU on the computed sign
                    gemm_x3(xP, wbt1, n, cout, cin, cin, cin, nb, n * cin, cout * cin, P=a1P, T=a1T, ldt=n, strideT=cout * n,
                            mask=a1g, gate_bits=1)
                else:
                    gemm_x3(xP, wbt1, n, cout, cin, cin, cin, nb, n * cin, cout * cin, P=a1P, T=a1T, ldt=n, strideT=cout * n,
                            act=1, mask_out=a1g, gate_bits=2 if bits else 0)
                if e["pin2"] is not None:
                    gemm_x3(a1P, wbt2, n, cout, cout, cout, cout, nb, n * cout, cout * cout, P=oP, T=oT, ldt=n,
                            strideT=cout * n, res=xP if skip else None, mask=m2, gate_bits=1)
                elif bits and k >= 3 and oT is None:
                    gemm_x3_torgb(a1P, wbt2, n, cout, cout, cout, cout, nb, n * cout, cout * cout, oP, rgbp[2 * (k - 3)],
                                  rgbp[2 * (k - 3) + 1], rgb[b0:b1].view(nb * n, 3), not first_rgb,
                                  act=1, res=xP if skip else None, mask_out=m2, gate_bits=2)
                    first_rgb = False
                    xP = oP
                    continue
                elif bits:
                    gemm_x3(a1P, wbt2, n, cout, cout, cout, cout, nb, n * cout, cout * cout, P=oP, T=oT, ldt=n,
                            strideT=cout * n, act=1, res=xP if skip else None, mask_out=m2, gate_bits=2)
                elif skip:
                    gemm_x3(a1P, wbt2, n, cout, cout, cout, cout, nb, n * cout, cout * cout, P=oP, T=oT, ldt=n,
                            strideT=cout * n, act=1, res=xP, mask_out=m2)
                else:
                    gemm_x3(a1P, wbt2, n, cout, cout, cout, cout, nb, n * cout, cout * cout, P=oP, T=oT, ldt=n,
                            strideT=cout * n, act=1)
                if k >= 3:
                    torgb_fwd_x3(oP, rgbp[2 * (k - 3)], rgbp[2 * (k - 3) + 1], rgb[b0:b1].view(nb * n, 3), accumulate=not first_rgb)
                    first_rgb = False
                xP = oP

        _run_chunks(_chunk_ranges(B), run, dev)
        saved = []
        xP, xT = x0P, x0T
        for k, e in enumerate(plan):
            if GATE_REC is not None:
                GATE_REC.append(e["a1g"])
                GATE_REC.append(e["m2"])
            if train:
                # keep for backward: xT (dW1), a1 gate + a1T (dW2), out planes (ToRGB grad), m2 gate, weights
                wb1, _, d1 = prepped[2 * k]
                wb2, _, d2 = prepped[2 * k + 1]
                saved.append(dict(xT=xT, xP=xP, a1P=e["a1P"], a1m=e["a1g"] if e["bits"] else e["a1P"].hi, a1T=e["a1T"], oP=e["oP"],
                                  m2=e["m2"], wb1=wb1, d1=d1, wb2=wb2, d2=d2, skip=e["skip"], bits=e["bits"]))
            xP, xT = e["oP"], e["oT"]
        if not any_rgb:
            rgb.zero_()
        ctx.nblocks, ctx.blocks, ctx.rgbp, ctx.saved = nblocks, blocks, rgbp, saved
        ctx.dims = (B, n)
        ctx.kmajor = INR_W_KMAJOR
        ctx.ports = ports if train else None
        if ports is not None and train:
            ports["demod"] = [d for k in range(nblocks) for d in (saved[k]["d1"], saved[k]["d2"])]
            ports["taps"] = [saved[k]["oP"] for k in range(3, nblocks)]
        return rgb

    @staticmethod
    def backward(ctx, drgb):
        nblocks, blocks, rgbp, saved = ctx.nblocks, ctx.blocks, ctx.rgbp, ctx.saved
        B, n = ctx.dims
        drgb = _c(drgb)
        dev = drgb.device
        width = blocks[-1][2].shape[1]
        km = ctx.kmajor
        # full-batch outputs of the chains: per-image dL/dWb of every layer, dx0, and the ToRGB weight-gradient partials
        gwb = []
        for k in range(nblocks):
            cin, cout = blocks[k][0].shape
            gwb.append((torch.empty(B, cin, cout, device=dev), torch.empty(B, cout, cout, device=dev)))       # (gwb1, gwb2)
        cin0 = blocks[0][0].shape[0]
        dx0 = torch.empty(B, n, cin0, device=dev)
        ranges = _chunk_ranges(B)
        rgb_parts = [[None] * len(ranges) for _ in range(nblocks)]
        def run(b0, b1):
            nb = b1 - b0
            ci = [r[0] for r in ranges].index(b0)
            drgb_c = drgb[b0:b1]
            drgb2 = drgb_c.reshape(nb * n, 3)
            PT = (lambda c: None) if km else (lambda c: Planes.empty(nb, c, n, device=dev))
            k = nblocks - 1
            gP, gT = Planes.empty(nb, n, width, device=dev), PT(width)
            Dout = None
            # the skip gradient D (un-gated) is either kept as an fp32 copy next to the gated planes every GEMM writes
            # for its successors, or — when every layer has bit-plane gates and the 256x256-tile kernel takes the shapes —
            # recovered from those planes by the consumer (INR_ADDP): no copy written, same bytes read
            addp = km and all(saved[j]["bits"] for j in range(nblocks))
            if addp:
                for j in range(1, nblocks):
                    if saved[j]["skip"]:
                        cj_in, cj_out = blocks[j][0].shape
                        addp = addp and _addp_shape_ok(n, cj_in, cj_out, nb, dev)
            if k >= 3 and km and saved[k]["bits"] and width % 8 == 0:
                # grad wrt out_k = drgb @ T_k (rank 3), gate of a2_k fused: one streaming kernel writing the planes
                Dout = torch.empty(nb, n, width, device=dev) if saved[k]["skip"] and not addp else None
                torgb_bwd_x_x3(drgb2, rgbp[2 * (k - 3)], _bsl(saved[k]["m2"], b0, b1), Dout, gP)
            elif k >= 3:
                # ... as a K=32 zero-padded bf16x3 GEMM where the transposed planes are wanted too
                tpad = torch.zeros(1, width, 32, device=dev); tpad[0, :, :3] = rgbp[2 * (k - 3)].t()
                tP, _ = split_planes(tpad, want_t=False)
                dpad = torch.zeros(nb, n, 32, device=dev); dpad[..., :3] = drgb_c
                dP, _ = split_planes(dpad, want_t=False)
                Dout = torch.empty(nb, n, width, device=dev) if saved[k]["skip"] and not addp else None
                gemm_x3(dP, tP, n, width, 32, 32, 32, nb, n * 32, 0, P=gP, T=gT, ldt=n, strideT=width * n,
                        C_unmasked=Dout, mask=_bsl(saved[k]["m2"], b0, b1), gate_bits=1 if saved[k]["bits"] else 0)
            else:
                gP.hi.zero_(); gP.lo.zero_()
                if gT is not None:
                    gT.hi.zero_(); gT.lo.zero_()
                Dout = torch.zeros(nb, n, width, device=dev) if saved[k]["skip"] and not addp else None
            # ToRGB weight / bias gradients of all taps: they need only the saved block outputs and drgb
            taps = list(range(3, nblocks)) if ctx.ports is None else []
            if taps:
                for k_, res_ in zip(taps, torgb_bwd_w_x3_batch([_bsl(saved[k_]["oP"], b0, b1) for k_ in taps], drgb2)):
                    rgb_parts[k_][ci] = res_
            for k in range(nblocks - 1, -1, -1):
                sv = saved[k]
                W1, s1, W2, s2 = blocks[k]
                cin, cout = W1.shape
                gP_in = gP              # D_{k+1} gated by a2_k's gate: the planes form of the skip gradient
                # ---- mod2: gradient through the gate of a1 ----
                g1P, g1T = Planes.empty(nb, n, cout, device=dev), PT(cout)
                gemm_x3(gP, _bsl(sv["wb2"], b0, b1), n, cout, cout, cout, cout, nb, n * cout, cout * cout, P=g1P, T=g1T, ldt=n,
                        strideT=cout * n, mask=_bsl(sv["a1m"], b0, b1), gate_bits=1 if sv["bits"] else 0)
                # ---- weight gradients of both layers: dWb2 = a1^T g, dWb1 = x^T g1 ----
                gwb1, gwb2 = gwb[k][0][b0:b1], gwb[k][1][b0:b1]
                a1P, xP = _bsl(sv["a1P"], b0, b1), _bsl(sv["xP"], b0, b1)
                if km:
                    # dWb[b] = X[b]^T G[b] contracts over the n pixels of image b.  With few images per GPU a 512x512
                    # output is too few tiles for the chip, so the pixel range is split in `ksp` parts — a pure view of
                    # the row-major planes, (B, n, C) -> (B*ksp, n/ksp, C) — and the partial products are summed.
                    ksp = 1
                    while nb * ksp < 32 and n % (2 * ksp * 32) == 0 and n // (2 * ksp) >= 512:
                        ksp *= 2
                    nk_ = n // ksp
                    part2 = gwb2 if ksp == 1 else torch.empty(nb * ksp, cout, cout, device=dev)
                    part1 = gwb1 if ksp == 1 else torch.empty(nb * ksp, cin, cout, device=dev)
                    if cin == cout:
                        gemm_x3_km_grouped([(a1P, gP, part2), (xP, g1P, part1)], cout, cout, nk_, cout, cout,
                                           nb * ksp, nk_ * cout, nk_ * cout)
                    else:
                        # the block's square problem alone is half a chip of 256 x 256 tiles: two pixel halves fill it
                        ksp2 = ksp
                        while nb * ksp2 * ((cout + 255) // 256) ** 2 < 192 and n % (2 * ksp2 * 32) == 0 and n // (2 * ksp2) >= 512:
                            ksp2 *= 2
                        if ksp2 != ksp:
                            part2 = torch.empty(nb * ksp2, cout, cout, device=dev)
                        nk2 = n // ksp2
                        gemm_x3_km(a1P, gP, cout, cout, nk2, cout, cout, nb * ksp2, nk2 * cout, nk2 * cout, part2)
                        if ksp2 != ksp:
                            torch.sum(part2.view(nb, ksp2, cout, cout), dim=1, out=gwb2)
                            part2 = gwb2
                        # a narrow first layer (cin = 32: one row tile, 2 column tiles per image = 64 workgroups) streams
                        # the whole gradient plane through a quarter of the chip: split its pixel range further
                        ksp1 = ksp
                        while cin <= 128 and nb * ksp1 * ((cout + 255) // 256) < 192 and n % (2 * ksp1 * 32) == 0 and n // (2 * ksp1) >= 512:
                            ksp1 *= 2
                        if ksp1 != ksp:
                            part1 = torch.empty(nb * ksp1, cin, cout, device=dev)
                        nk1 = n // ksp1
                        gemm_x3_km(xP, g1P, cin, cout, nk1, cin, cout, nb * ksp1, nk1 * cin, nk1 * cout, part1)
                        if ksp1 != ksp:
                            torch.sum(part1.view(nb, ksp1, cin, cout), dim=1, out=gwb1)
                            part1 = gwb1
                    if ksp > 1:
                        if part2 is not gwb2:
                            torch.sum(part2.view(nb, ksp, cout, cout), dim=1, out=gwb2)
                        if part1 is not gwb1:
                            torch.sum(part1.view(nb, ksp, cin, cout), dim=1, out=gwb1)
                else:
                    gemm_x3(_bsl(sv["a1T"], b0, b1), gT, cout, cout, n, n, n, nb, cout * n, cout * n, C=gwb2)
                    gemm_x3(_bsl(sv["xT"], b0, b1), g1T, cin, cout, n, n, n, nb, cin * n, cout * n, C=gwb1)
                if k == 0:
                    gemm_x3(g1P, _bsl(sv["wb1"], b0, b1), n, cin, cout, cout, cout, nb, n * cout, cin * cout, C=dx0[b0:b1])
                else:
                    pv = saved[k - 1]
                    newD = torch.empty(nb, n, cin, device=dev) if pv["skip"] and not addp else None
                    gP, gT = Planes.empty(nb, n, cin, device=dev), PT(cin)
                    gemm_x3(g1P, _bsl(sv["wb1"], b0, b1), n, cin, cout, cout, cout, nb, n * cout, cin * cout, P=gP, T=gT, ldt=n,
                            strideT=cin * n, add=Dout if sv["skip"] and not addp else None,
                            addp=(gP_in, _bsl(sv["m2"], b0, b1)) if sv["skip"] and addp else None,
                            rgb_g=drgb2 if k - 1 >= 3 else None, rgb_w=rgbp[2 * (k - 1 - 3)] if k - 1 >= 3 else None,
                            C_unmasked=newD, mask=_bsl(pv["m2"], b0, b1), gate_bits=1 if pv["bits"] else 0)
                    Dout = newD

        _run_chunks(ranges, run, dev)
        if ctx.ports is not None:
            _tail_gate_clear(dev)
            # the tail runs in the ports' backward nodes (side stream): hand them dL/dWb of every layer and drgb
            return (None, dx0) + (None,) * (4 * nblocks + len(rgbp)) + tuple(g for k in range(nblocks) for g in gwb[k]) + (drgb,)
        grads_blocks = [None] * nblocks
        grads_rgb = [None] * len(rgbp)
        for k in range(3, nblocks):
            parts = rgb_parts[k]
            dT, dtau = parts[0]
            for q in parts[1:]:
                dT = dT + q[0]; dtau = dtau + q[1]
            grads_rgb[2 * (k - 3)], grads_rgb[2 * (k - 3) + 1] = dT, dtau
        pending = []      # (W, s, demod, dL/dWb) of every layer: their prep backward runs as one batch
        for k in range(nblocks - 1, -1, -1):
            sv = saved[k]
            W1, s1, W2, s2 = blocks[k]
            pending.append((W2, s2, sv["d2"], gwb[k][1]))
            pending.append((W1, s1, sv["d1"], gwb[k][0]))
        res = modfc_prep_bwd_batch(pending)          # pending order: block nblocks-1 (mod2, mod1), ..., block 0
        for i, k in enumerate(range(nblocks - 1, -1, -1)):
            (dW2, ds2), (dW1, ds1) = res[2 * i], res[2 * i + 1]
            grads_blocks[k] = (dW1, ds1, dW2, ds2)
        flat = [None, dx0]
        for gb in grads_blocks:
            flat.extend(gb)
        flat.extend(grads_rgb)
        return tuple(flat)


class _ModPrepGradPort(torch.autograd.Function):
    """Gradient port of the head's modulated weights.  forward: no launch — one handle per layer, a stride-0 zero shaped like
    the per-image weights (B, in, out), which InrHeadX3Function takes as an input and answers with dL/dWb; backward: the
    modulation / demodulation backward of all layers (mod_conv_fc.py:392-496) -> (dW, ds) per layer.  The node lives on the
    stream that is current when apply() is called: CIPSNet.forward calls it under the generator's side stream."""

    @staticmethod
    def forward(ctx, state, B, *ws):
        ctx.state = state
        ctx.ws = tuple(_c(t.detach()) for t in ws)
        z = _zero_handle(ws[0].device)
        return tuple(z.expand(B, ws[2 * i].shape[0], ws[2 * i].shape[1]) for i in range(len(ws) // 2))

    @staticmethod
    def backward(ctx, *gwb):
        _tail_gate_wait(gwb[0].device)
        demod = ctx.state["demod"]
        layers = [(ctx.ws[2 * i], ctx.ws[2 * i + 1], demod[i], _c(gwb[i])) for i in range(len(gwb))]
        res = modfc_prep_bwd_batch(layers, cores=True)
        return (None, None) + tuple(t for pair in res for t in pair)


class _ToRGBGradPort(torch.autograd.Function):
    """Gradient port of the ToRGB taps: forward returns a stride-0 zero shaped like the head's output (its gradient IS drgb),
    backward computes every tap's weight / bias gradient from the block outputs the head kept (state["taps"])."""

    @staticmethod
    def forward(ctx, state, shape, *rgbp):
        ctx.state, ctx.n = state, len(rgbp)
        return _zero_handle(rgbp[0].device).expand(shape)

    @staticmethod
    def backward(ctx, drgb):
        _tail_gate_wait(drgb.device)
        taps = ctx.state["taps"]
        drgb2 = _c(drgb).reshape(-1, 3)
        out = []
        for dT, dtau in torgb_bwd_w_x3_batch(taps, drgb2, cores=True):
            out += [dT, dtau]
        return (None, None) + tuple(out)


_ZERO_HANDLE = {}


def _zero_handle(dev):
    """one cached zero element per device (never written): the storage behind every port handle.  Not cached under a graph
    capture (a tensor made inside a capture lives in the graph's pool)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(1, device=dev)
    z = _ZERO_HANDLE.get(dev)
    if z is None:
        z = _ZERO_HANDLE[dev] = torch.zeros(1, device=dev)
    return z


# The ports' tail must not start while the compositing backward (HBM-bound, on the critical path) is running: it would share the
# bandwidth and delay it by as much as the tail gains.  The NeRF path's backward publishes an event right after that launch; the
# ports' backward nodes — which the engine runs AFTER it when the ports were opened before the ray march in the forward pass
# (lower sequence numbers; GeneratorNerfINR._render does that) — make their stream wait for it.  No event (frozen NeRF, ports
# opened late): the tail starts as soon as the head's backward is done.
_TAIL_GATE = {}


def _tail_gate_clear(dev):
    _TAIL_GATE.pop(torch.device(dev).index, None)


def _tail_gate_publish(dev):
    if INR_TAIL == "side" and torch.device(dev).type == "cuda":
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        _TAIL_GATE[torch.device(dev).index] = ev


def _tail_gate_wait(dev):
    ev = _TAIL_GATE.get(torch.device(dev).index)
    if ev is not None:
        torch.cuda.current_stream(dev).wait_event(ev)


def inr_head_ports_ok(nblocks, B, n, in0, params, dev):
    """the gradient-port form applies: split-bf16 head in training, batch within the co-resident kernels' limit, ToRGB taps present"""
    return (INR_TAIL == "side" and INR_MODE == "bf16x3" and torch.device(dev).type == "cuda" and torch.is_grad_enabled()
            and nblocks > 3 and n % 32 == 0 and in0 % 32 == 0 and B <= 64 and any(t.requires_grad for t in params))


def inr_head_open_ports(nblocks, B, n, params, side):
    """the two gradient ports of one head evaluation (see INR_TAIL), their autograd nodes on `side` -> (state, handles, H)"""
    state = {}
    ws, rgbp = params[:4 * nblocks], params[4 * nblocks:]
    dev = params[0].device
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        handles = _ModPrepGradPort.apply(state, B, *ws)
        H = _ToRGBGradPort.apply(state, (B, n, 3), *rgbp)
    return state, handles, H


def inr_head_with_ports(nblocks, x0, params, ports):
    """inr_head with the weight-gradient tail behind the gradient ports `ports` = inr_head_open_ports(...)"""
    state, handles, H = ports
    det = [t.detach() for t in params]
    return InrHeadX3Function.apply((nblocks, True, state), x0, *det, *handles, H)


def inr_head(nblocks, x0, *params):
    # the split-bf16 kernels tile pixels and features by 32; anything else (part_grad_forward with an arbitrary
    # grad_points, generator.py:1591-1593) runs on the exact fp32 MFMA path, which has no such granule
    x3 = INR_MODE == "bf16x3" and x0.shape[1] % 32 == 0 and x0.shape[2] % 32 == 0
    fn = InrHeadX3Function if x3 else InrHeadFunction
    # (nblocks, caller's grad mode): under torch.no_grad() nothing is kept for a backward
    return fn.apply((nblocks, torch.is_grad_enabled()), x0, *params)


# --------------------------------------------------------------------------------------
# H5 discriminator native ops (same contracts as the reference's pybind ops)
# --------------------------------------------------------------------------------------
_FLOATS = (torch.float32, torch.float16, torch.bfloat16, torch.float64)


def _native_in(t, what):
    """The reference's ops dispatch fp16 / fp32 / fp64 (fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:177-211); the
    HIP kernels compute in fp32: other float dtypes make an fp32 round trip (the result is cast back to the INPUT's
    dtype by the caller), anything else is an error."""
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be a CUDA tensor")      # CHECK_CUDA, fused_bias_act.cpp:13
    if t.dtype not in _FLOATS:
        raise RuntimeError(f"{what}: unsupported dtype {t.dtype} (floating point expected)")
    return t.contiguous().float()


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale) -> Tensor of input's dtype and shape
    (exp/comm/op/fused_bias_act.cpp:11-21).  Empty tensor = absent."""
    lib = _lib.load()
    x = _native_in(input, "input")
    b = _native_in(bias, "bias") if (bias is not None and bias.numel()) else None
    r = _native_in(refer, "refer") if (refer is not None and refer.numel()) else None
    if r is not None and r.numel() != x.numel():
        raise RuntimeError("refer must have input's number of elements")
    y = torch.empty_like(x)
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    size_b = b.numel() if b is not None else 0
    with torch.cuda.device(x.device):
        check(lib.cips_fused_bias_act(_p(x), _p(b), _p(r), _p(y), x.numel(), size_b, step_b, act, grad, float(alpha),
                                      float(scale), _stream()), "cips_fused_bias_act")
    return y if input.dtype == torch.float32 else y.to(input.dtype)


def lrelu_bwd_bias(grad, refer, alpha, scale):
    """grad, refer (B, C, H, W) fp32 contiguous -> (grad_input, grad_bias (C)): FusedLeakyReLUFunctionBackward.forward
    (exp/comm/op/fused_act.py:26-44: fused_bias_act(grad, empty, out, 3, 1, slope, scale) then grad_input.sum(0, 2, 3)) with
    the bias gradient accumulated in the same pass"""
    lib = _lib.load()
    B, C = grad.shape[0], grad.shape[1]
    hw = grad.numel() // (B * C)
    S = lib.cips_lrelu_bwd_bias_slices(hw)
    gin = torch.empty_like(grad)
    part = torch.empty(B, C, S, device=grad.device)
    with torch.cuda.device(grad.device):
        check(lib.cips_lrelu_bwd_bias(_p(grad), _p(refer), _p(gin), _p(part), B * C, hw, float(alpha), float(scale), _stream()),
              "cips_lrelu_bwd_bias")
        gb = torch.empty(C, device=grad.device)
        check(lib.cips_lrelu_bwd_bias_finish(_p(part), _p(gb), B, C, S, _stream()), "cips_lrelu_bwd_bias_finish")
    return gin, gb


def lrelu_bwd_bias_nhwc(grad, refer, alpha, scale):
    """lrelu_bwd_bias with the gated gradient written as NHWC split planes only (cips_lrelu_bwd_bias_nhwc) -> (Planes, grad_bias)"""
    lib = _lib.load()
    B, C, H, W = grad.shape
    hw = H * W
    S = lib.cips_lrelu_bwd_bias_nhwc_tiles(hw)
    hi = torch.empty(B * hw + 1, C, device=grad.device, dtype=BF)
    lo = torch.empty(B * hw + 1, C, device=grad.device, dtype=BF)
    part = torch.empty(B, C, S, device=grad.device)
    gb = torch.empty(C, device=grad.device)
    with torch.cuda.device(grad.device):
        check(lib.cips_lrelu_bwd_bias_nhwc(_p(grad), _p(refer), _p(hi), _p(lo), _p(part), B, C, hw, float(alpha), float(scale), _stream()),
              "cips_lrelu_bwd_bias_nhwc")
        check(lib.cips_lrelu_bwd_bias_finish(_p(part), _p(gb), B, C, S, _stream()), "cips_lrelu_bwd_bias_finish")
    return Planes(hi, lo), gb


def upfirdn2d_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """upfirdn2d_op.upfirdn2d(input[N,H,W,minor], kernel, ...) -> Tensor of input's dtype
    (exp/comm/op/upfirdn2d.cpp:12-23)."""
    lib = _lib.load()
    x = _native_in(input, "input")
    k = _native_in(kernel, "kernel")
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    out = torch.empty(major, out_h, out_w, minor, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.cips_upfirdn2d(_p(x), _p(k), _p(out), major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y,
                                 pad_x0, pad_x1, pad_y0, pad_y1, _stream()), "cips_upfirdn2d")
    return out if input.dtype == torch.float32 else out.to(input.dtype)


def im2col(x, kh, kw, stride, pad):
    lib = _lib.load()
    B, Cc, H, W = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    col = torch.empty(B, Cc * kh * kw, Ho * Wo, device=x.device)
    check(lib.cips_im2col(_p(x), _p(col), B, Cc, H, W, kh, kw, stride, pad, _stream()), "cips_im2col")
    return col, Ho, Wo


def im2col_x3(x, kh, kw, stride, pad):
    """im2col as split-bf16 planes (B, C*kh*kw, Ho*Wo)"""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    col = Planes.empty(B, Cc * kh * kw, Ho * Wo, device=x.device)
    check(lib.cips_im2col_x3(_p(x), _p(col.hi), _p(col.lo), B, Cc, H, W, kh, kw, stride, pad, _stream()), "cips_im2col_x3")
    return col, Ho, Wo


def col2im(col, B, Cc, H, W, kh, kw, stride, pad):
    lib = _lib.load()
    dx = torch.empty(B, Cc, H, W, device=col.device)
    check(lib.cips_col2im(_p(col), _p(dx), B, Cc, H, W, kh, kw, stride, pad, _stream()), "cips_col2im")
    return dx
