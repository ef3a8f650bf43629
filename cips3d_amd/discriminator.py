"""MI355X-native drop-in for the reference discriminator API
(exp/cips3d/models/discriminator.py: Discriminator_MultiScale :406-585, Discriminator_MultiScale_Aux
:589-664) with identical constructor / forward signatures and state_dict layout.

Native ops, same contracts as the reference's CUDA extensions (exp/comm/op/fused_act.py,
exp/comm/op/upfirdn2d.py) but backed by libcips3d_hip.so:
  * fused_leaky_relu / FusedLeakyReLU   -> cips_fused_bias_act (fwd, bwd, double-bwd)
  * upfirdn2d (Blur)                     -> cips_upfirdn2d      (fwd, bwd, double-bwd)
  * EqualConv2d                          -> cips_im2col + cips_gemm_f32 (fp32 MFMA) + cips_col2im,
                                            as three mutually-recursive autograd Functions so the R1
                                            double-backward of train.py:387-394 works.
"""
import collections
import math

import torch
from torch import nn
from torch.autograd import Function
import torch.nn.functional as F

from . import ops


# ------------------------------------------------------------------------------------------
# fused bias + leaky relu  (mirrors exp/comm/op/fused_act.py:19-86)
# ------------------------------------------------------------------------------------------
# LeakyReLU gate instrumentation (parity tests; never set in production) — the discriminator's counterpart of
# ops.GATE_PIN / ops.GATE_REC: GATE_PIN is an iterator of bool tensors shaped like the activations, one per
# fused_leaky_relu call in call order; the op then applies `gate ? 1 : slope` from the tensor (cips_fused_bias_act
# in its gradient form: y = (refer > 0 ? x + b : (x + b) * slope) * scale) and every backward order uses the same
# gate.  GATE_REC receives the bool gates actually used.
GATE_PIN = None
GATE_REC = None


class gate_debug:
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
            assert next(GATE_PIN, None) is None, "pinned gates left over"
        GATE_PIN, GATE_REC = self.old




class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale, planes_only=False):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        if planes_only:
            # the gated gradient as NHWC split planes ONLY (cips_lrelu_bwd_bias_nhwc): the tensor handed on is a stride-0 zero
            # placeholder of the right shape that carries the planes (attribute `_cips_nhwc`); its consumers — the implicit-GEMM
            # data and weight gradient of ONE convolution, chosen by ConvBiasActFunction.backward — read nothing else (_nhwc
            # refuses a placeholder that lost its planes; _dense refuses to materialise one)
            P, gb = ops.lrelu_bwd_bias_nhwc(grad_output, out, negative_slope, scale)
            ph = _zero1(grad_output).expand(grad_output.shape)
            ph._cips_nhwc = (ph._version, ph.data_ptr(), {(): P})
            ph._cips_planes_only = True
            return ph, gb
        if (grad_output.dim() == 4 and grad_output.dtype == torch.float32 and out.dtype == torch.float32
                and grad_output.is_cuda and grad_output.is_contiguous() and out.is_contiguous()):
            return ops.lrelu_bwd_bias(grad_output, out, negative_slope, scale)     # one pass: gated gradient + bias sums
        empty = grad_output.new_empty(0)
        grad_input = ops.fused_bias_act(grad_output, empty, out, 3, 1, negative_slope, scale)
        dim = [0] + list(range(2, grad_input.ndim))
        grad_bias = grad_input.sum(dim).detach()
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        gradgrad_out = ops.fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        empty = input.new_empty(0)
        if GATE_PIN is not None:
            gate = next(GATE_PIN).to(input.device).reshape(input.shape)
            refer = torch.where(gate, 1.0, -1.0).to(input.dtype)       # only its sign is read
            out = ops.fused_bias_act(input, bias, refer, 3, 1, negative_slope, scale)
        else:
            out = ops.fused_bias_act(input, bias, empty, 3, 0, negative_slope, scale)
            refer = out                                                    # sign(out) = sign(input + bias)
        if GATE_REC is not None:
            GATE_REC.append((refer > 0).clone())
        ctx.save_for_backward(refer)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


# ------------------------------------------------------------------------------------------
# upfirdn2d  (mirrors exp/comm/op/upfirdn2d.py:18-149)
# ------------------------------------------------------------------------------------------
class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1 = g_pad
        grad_output = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        grad_input = ops.upfirdn2d_op(grad_output, grad_kernel, down_x, down_y, up_x, up_y,
                                      g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.in_size, ctx.out_size = in_size, out_size
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        gradgrad_input = gradgrad_input.reshape(-1, ctx.in_size[2], ctx.in_size[3], 1)
        gradgrad_out = ops.upfirdn2d_op(gradgrad_input, kernel, ctx.up[0], ctx.up[1], ctx.down[0], ctx.down[1], *ctx.pad)
        gradgrad_out = gradgrad_out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0], ctx.out_size[1])
        return gradgrad_out, None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        batch, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        input = input.reshape(-1, in_h, in_w, 1)
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) // down_y + 1
        out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) // down_x + 1
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
        g_pad_x0 = kernel_w - pad_x0 - 1
        g_pad_y0 = kernel_h - pad_y0 - 1
        g_pad_x1 = in_w * up_x - out_w * down_x + pad_x0 - up_x + 1
        g_pad_y1 = in_h * up_y - out_h * down_y + pad_y0 - up_y + 1
        ctx.g_pad = (g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        out = ops.upfirdn2d_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad,
                                             ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


# ------------------------------------------------------------------------------------------
# conv2d on the fp32 MFMA GEMM (replaces F.conv2d in EqualConv2d.forward, discriminator.py:40-48)
# ------------------------------------------------------------------------------------------
def _pad4(n):
    return (n + 3) // 4 * 4


# Conv GEMMs of eligible layers run as 3-pass split-bf16 MFMA (cips_gemm_bf16x3*, ~5e-6 relative, ~3x the fp32 MFMA
# rate); CIPS_D_CONV_MODE=f32 keeps every layer on the exact fp32 MFMA GEMM.  Eligible: all three contraction lengths
# (C*kh*kw, Ho*Wo, O) multiples of 32 — i.e. everything but the RGB input convs and the 4x4 tail.
import os as _os
CONV_MODE = _os.environ.get("CIPS_D_CONV_MODE", "bf16x3")


def _x3_ok(K, N, O):
    return CONV_MODE == "bf16x3" and K % 32 == 0 and N % 32 == 0 and O % 32 == 0


# Implicit-GEMM form of the x3 convs (CIPS_D_CONV_IMPLICIT=0 keeps the materialised im2col everywhere): the activation is
# transposed once into NHWC split planes and the NT GEMM's loader gathers the taps; used from 16x16 output planes up
# (smaller planes leave most of a 256-pixel tile empty).


# Small output planes (8x8, 4x4: fewer than 256 pixels per image): the batch is folded into the pixel dimension, one
# GEMM over B*Ho*Wo columns with the shared weights instead of B GEMMs whose 16- or 64-column tiles are mostly padding
# (the 4x4 layers spent 14 ms per GAN step in the fp32 GEMM for 5 GFLOP).  CIPS_D_CONV_FOLD=0 restores that.


def _fold_ok(K, N, B, O):
    return CONV_MODE == "bf16x3" and N < 256 and (B * N) % 32 == 0 and K % 32 == 0 and O % 32 == 0


def _split_count(tiles, length):
    """Folded GEMMs are single problems of few 256x128 tiles (16 for a 4x4 plane): cut the contraction (`length` rows,
    in 32-row k-tiles) into chunks that become the GEMM's batch, partial sums added by the caller."""
    n = 1
    while tiles * n < 192 and length % (64 * n) == 0 and length // (2 * n) >= 128:
        n *= 2
    return n


def _folded_col_planes(x, kh, kw, stride, pad):
    """im2col with the batch folded into the columns: Planes (1, K, B*N), k-major"""
    col, Ho, Wo = ops.im2col(x, kh, kw, stride, pad)                  # (B, K, N) fp32
    B, K, N = col.shape
    colP, _ = ops.split_planes(col.permute(1, 0, 2).reshape(1, K, B * N).contiguous(), want_p=True, want_t=False)
    return colP, Ho, Wo


def _folded_rows_planes(t):
    """(B, O, Ho, Wo) -> Planes (1, O, B*N)"""
    B, O = t.shape[0], t.shape[1]
    P, _ = ops.split_planes(t.reshape(B, O, -1).permute(1, 0, 2).reshape(1, O, -1).contiguous(), want_p=True, want_t=False)
    return P


def _implicit_ok(C, N, O):
    # any plane size: output planes of fewer than 256 pixels (8 x 8, 4 x 4) run with the batch folded into the pixel
    # dimension of the same kernel (round 6; before, im2col + a K-major GEMM + col2im and their permuting copies)
    return CONV_MODE == "bf16x3" and C % 32 == 0 and O % 32 == 0 and N % 8 == 0


# NHWC planes of an activation / gradient are shared between the convolution ops that read the same tensor inside ONE
# autograd node (dy feeds both the data and the weight gradient; the forward's planes of x feed its weight gradient):
# a memo that only lives while that node's forward / backward runs.  It is per thread (autograd runs backward nodes
# on its own threads) and every entry holds the SOURCE tensor as well as its planes: while the memo is alive the
# source's storage cannot be freed and handed to another tensor of the same shape, so the (address, shape, version) key
# cannot alias stale planes.
import threading as _threading
_tls = _threading.local()


def _memo():
    if not hasattr(_tls, "shared"):
        _tls.shared, _tls.depth = {}, 0
    return _tls


def _key(t):
    return (t.data_ptr(), tuple(t.shape), t._version)


class _Shared:
    """dict-like view used by the Functions below: key -> planes"""

    @staticmethod
    def get(key_or_tensor):
        m = _memo()
        ent = m.shared.get(key_or_tensor)
        return ent[1] if ent is not None else None


_shared = _Shared()


class _share_planes:
    def __init__(self, *pairs):                 # (tensor, Planes or None) whose planes are already known
        self.pairs = pairs

    def __enter__(self):
        m = _memo()
        m.depth += 1
        for ent in self.pairs:
            t, p = ent[0], ent[1]
            if p is not None:
                m.shared[_key(t) + _pre_sig(ent[2] if len(ent) > 2 else None)] = (t, p)

    def __exit__(self, *exc):
        m = _memo()
        m.depth -= 1
        if m.depth == 0:
            m.shared.clear()


def _is_planes_only(t):
    """a FusedLeakyReLU-backward placeholder: flagged, or — should the flag have been lost with the Python object — recognisable
    by its storage: an expansion of THIS module's cached zero element (an ordinary stride-0 tensor, e.g. autograd's gradient of
    a .sum(), is not one)"""
    if getattr(t, "_cips_planes_only", False):
        return True
    z = _ZERO1.get((t.device, t.dtype))
    return z is not None and t.dim() == 4 and t.numel() > 1 and t.data_ptr() == z.data_ptr()


def _dense(t):
    """t.contiguous() for a tensor whose VALUES are about to be read: a planes-only gradient has none"""
    if _is_planes_only(t):
        raise RuntimeError("cips3d_amd: a planes-only gradient (FusedLeakyReLU backward written as NHWC planes) reached a path that "
                           "reads fp32 values — ConvBiasActFunction.backward chose it for a convolution that is not on the implicit-GEMM path")
    return t.contiguous()


def _nhwc(t, pre=None):
    """NHWC split planes of t — of Blur(t) when `pre` is given (upfirdn2d + split: a fused kernel was built, was bit-identical
    and did not beat the two, profiles/r6_blur_nhwc_planes_experiment.txt).

    The planes travel WITH the tensor object (attribute `_cips_nhwc`, checked against the version counter): a convolution's
    input keeps them from its forward to every later backward that needs them for the weight gradient (the tensor is saved by
    the node anyway), and a gradient tensor keeps them from the data-gradient node to the weight-gradient node, which are
    separate autograd nodes (_WeightGradPort).  They die with the tensor."""
    sig = _pre_sig(pre)
    att = getattr(t, "_cips_nhwc", None)
    if att is not None and att[0] == t._version and att[1] == t.data_ptr():
        hit = att[2].get(sig)
        if hit is not None:
            return hit
    if _is_planes_only(t):
        raise RuntimeError("cips3d_amd: a planes-only gradient lost its planes")
    m = _memo()
    key = _key(t) + sig
    ent = m.shared.get(key)
    if ent is not None:
        return ent[1]
    if pre is None:
        p = ops.split_planes_nhwc(t)
    else:
        p = ops.split_planes_nhwc(_pre_fp32(t, pre))
    if m.depth:
        m.shared[key] = (t, p)
    if att is not None and att[0] == t._version and att[1] == t.data_ptr():
        att[2][sig] = p
    else:
        try:
            t._cips_nhwc = (t._version, t.data_ptr(), {sig: p})
        except Exception:                            # noqa: BLE001 — a tensor subclass without a __dict__: no caching
            pass
    return p


# ------------------------------------------------------------------------------------------
# The Blur of a down-sampling ConvLayer folded into its convolution: `pre` = (kernel (4 x 4 buffer),
# pad0, pad1, down) describes y = conv(upfirdn2d(x, kernel, down=down, pad=(pad0, pad1)), w).  The three convolution
# Functions below carry it through every order of differentiation: forward and weight gradient read the blurred planes
# (made once per tensor and shared between them; the blurred tensor is a temporary of that step, not a node of the graph), the data gradient ends in the
# Blur's transpose — for the 3 x 3 stride-2 layers applied directly to the parity blocks of cips_conv2d_x3_dgrad_s2.
# ------------------------------------------------------------------------------------------
def _pre_sig(pre):
    return () if pre is None else ("pre", int(pre[1]), int(pre[2]), int(pre[3]))


def _pre_shape(H, W, pre):
    if pre is None:
        return H, W
    _, p0, p1, down = pre
    return (H + p0 + p1 - 4) // down + 1, (W + p0 + p1 - 4) // down + 1


_FLIPPED = {}


def _flipped(k):
    """torch.flip(k, [0, 1]) of a Blur kernel buffer, cached per (storage, version)"""
    key = (k.data_ptr(), k._version, str(k.device))
    f = _FLIPPED.get(key)
    if f is None:
        f = torch.flip(k, [0, 1]).contiguous()
        if not (k.is_cuda and torch.cuda.is_current_stream_capturing()):      # a tensor made inside a capture lives in the graph's pool
            if len(_FLIPPED) > 64:
                _FLIPPED.clear()
            _FLIPPED[key] = f
    return f


def _pre_fp32(x, pre):
    """upfirdn2d(x, kernel, down, pad) as an fp32 NCHW tensor (the paths without an implicit-GEMM form)"""
    k, p0, p1, down = pre
    B, C, H, W = x.shape
    out = ops.upfirdn2d_op(x.contiguous().reshape(-1, H, W, 1), k, 1, 1, down, down, p0, p1, p0, p1)
    Hb, Wb = _pre_shape(H, W, pre)
    return out.view(B, C, Hb, Wb)


def _pre_adjoint_pads(H, W, pre):
    _, p0, p1, down = pre
    Hb, Wb = _pre_shape(H, W, pre)
    return (3 - p0, W - Wb * down + p0, 3 - p0, H - Hb * down + p0)          # (x0, x1, y0, y1): upfirdn2d.py:104-110


def _pre_adjoint(dxb, pre, in_shape):
    """transpose of the Blur: gradient w.r.t. its input from the gradient w.r.t. its output (upfirdn2d.py:18-52)"""
    k, p0, p1, down = pre
    B, C, H, W = in_shape
    Hb, Wb = _pre_shape(H, W, pre)
    gx0, gx1, gy0, gy1 = _pre_adjoint_pads(H, W, pre)
    out = ops.upfirdn2d_op(dxb.contiguous().reshape(-1, Hb, Wb, 1), _flipped(k), down, down, 1, 1, gx0, gx1, gy0, gy1)
    return out.view(B, C, H, W)


# Operand planes of a convolution WEIGHT are a function of the parameter alone, but every conv call of a step needs
# them: three discriminator forwards, the backward and the R1 double-backward re-derive the same planes (412
# split_planes launches per GAN step).  They are cached per nn.Parameter and invalidated by the tensor's version
# counter (bumped by every in-place update: torch.optim, load_state_dict, and cips3d_amd.optim.FusedClipAdamEMA, which
# writes through raw pointers and therefore bumps it explicitly).  Transient weights (the double-backward's `ggw`) are
# never cached.  CIPS_D_WCACHE=0 disables the cache.
# A write that bypasses the version counter (`p.data.copy_()`, `p.data.mul_()` — the reference's own EMA helper writes
# through .data —, an optimiser step replayed from a hipGraph, a raw-pointer kernel) leaves stale planes in use: such
# writers call invalidate_weight_cache(module_or_parameters) afterwards.  The cache lives in a weakly keyed table beside
# the parameters, not in Parameter.__dict__: pickling / deepcopying a module does not carry GPU planes along.
import weakref as _weakref
_WCACHE = {}        # id(Parameter) -> (weakref to it, {(kind, scale): (version, data_ptr, operand)}); the weakref's callback
                    # drops the entry with the Parameter (a WeakKeyDictionary would compare tensor keys with ==)


def invalidate_weight_cache(what=None):
    """Drop the cached operand planes of a module's parameters (or of an iterable of parameters; None: of everything).
    Needed after any weight write that does not bump Tensor._version (see above); cheap to call when nothing is cached."""
    if what is None:
        _WCACHE.clear()
        return
    params = what.parameters() if isinstance(what, nn.Module) else what
    for p in params:
        _WCACHE.pop(id(p), None)


def _cached(w, scale, kind, build):
    """build(w_eff) -> operand for `kind`, memoised on (parameter, version, scale)"""
    if not isinstance(w, nn.Parameter):
        return build(w if scale == 1.0 else w * scale)
    if w.is_cuda and torch.cuda.is_current_stream_capturing():
        # hipGraph capture: a replay runs no Python and never bumps `_version`, so cached planes would go
        # stale against the captured optimizer's updates, and planes built here live in the graph's private pool —
        # build them inside the capture (every replay rebuilds them from the current weights) and keep nothing
        with torch.no_grad():
            return build(w.detach() if scale == 1.0 else w.detach() * scale)
    slot = _WCACHE.get(id(w))                              # entries die with the Parameter object
    if slot is None or slot[0]() is not w:
        key_id = id(w)
        slot = _WCACHE[key_id] = (_weakref.ref(w, lambda _r, k=key_id: _WCACHE.pop(k, None)), {})
    ent = slot[1]
    key = (kind, float(scale))
    hit = ent.get(key)
    if hit is not None and hit[0] == w._version and hit[1] == w.data_ptr():
        return hit[2]
    with torch.no_grad():
        val = build(w.detach() if scale == 1.0 else w.detach() * scale)
    ent[key] = (w._version, w.data_ptr(), val)
    return val


def _bank_offsets(O, C, kh, kw):
    """element offsets of the four parity banks (ops.dgrad_s2_banks' layout) and their total size"""
    offs, tot = [], 0
    for a in range(2):
        for b in range(2):
            offs.append(tot)
            tot += C * O * len(range(a, kh, 2)) * len(range(b, kw, 2))
    return offs, tot


def prepare_weight_planes(layers):
    """layers: iterable of (EqualConv2d, alt) with alt "flipT" (stride-1 data gradient) or "s2banks" (stride-2 data gradient behind a
    Blur).  Builds every stale operand form of every listed weight in ONE launch per 24 layers (cips_conv_weight_prep_batch) and
    files them in the weight-plane cache under the keys the convolution Functions look up — after an optimizer step the whole
    network's planes used to be rebuilt layer by layer, form by form (multiply, permuting copy, flip, split: ~12 launches per
    layer, ~480 per GAN step).  Same values bit for bit.  Not under hipGraph capture (planes built there live in the graph's pool and
    are rebuilt by every replay: _cached handles that case per layer)."""
    if CONV_MODE != "bf16x3" or torch.cuda.is_current_stream_capturing():
        return
    import ctypes as C
    from . import _lib
    from ._lib import WPrepJob
    todo = []
    want_alt = torch.is_grad_enabled()
    for conv, alt in layers:
        w = conv.weight
        if not (isinstance(w, nn.Parameter) and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            continue
        O, Cc, kh, kw = w.shape
        if Cc % 32 or O % 32 or (alt == "s2banks" and O < 64):
            continue
        slot = _WCACHE.get(id(w))
        if slot is None or slot[0]() is not w:
            key_id = id(w)
            slot = _WCACHE[key_id] = (_weakref.ref(w, lambda _r, k=key_id: _WCACHE.pop(k, None)), {})
        ent = slot[1]
        sc = float(conv.scale)

        def stale(kind):
            hit = ent.get((kind, sc))
            return hit is None or hit[0] != w._version or hit[1] != w.data_ptr()
        # the alternate form serves the DATA gradient: needed whenever a backward may run, frozen weights included (the G step)
        need_fwd, need_alt = stale("fwd"), bool(alt) and want_alt and stale(alt)
        if need_fwd or need_alt:
            todo.append((w, ent, sc, alt if need_alt else None, need_fwd))
    if not todo:
        return
    lib = _lib.load()
    mx = lib.cips_conv_weight_prep_max_jobs()
    dev = todo[0][0].device
    with torch.no_grad(), torch.cuda.device(dev):
        for c0 in range(0, len(todo), mx):
            chunk = todo[c0:c0 + mx]
            jobs = (WPrepJob * len(chunk))()
            made = []
            for j, (w, ent, sc, alt, need_fwd) in zip(jobs, chunk):
                O, Cc, kh, kw = w.shape
                K = kh * kw * Cc
                j.w, j.scale, j.O, j.C, j.kh, j.kw = w.data_ptr(), sc, O, Cc, kh, kw
                fwd = altv = None
                if need_fwd:
                    fwd = ops.Planes.empty(1, O, K, device=dev)
                    j.fwd_hi, j.fwd_lo = fwd.hi.data_ptr(), fwd.lo.data_ptr()
                if alt == "flipT":
                    P = ops.Planes.empty(1, Cc, kh * kw * O, device=dev)
                    j.alt_kind, j.alt_hi, j.alt_lo = 1, P.hi.data_ptr(), P.lo.data_ptr()
                    altv = P
                elif alt == "s2banks":
                    offs, tot = _bank_offsets(O, Cc, kh, kw)
                    P = ops.Planes.empty(1, tot // 32, 32, device=dev)
                    j.alt_kind, j.alt_hi, j.alt_lo = 2, P.hi.data_ptr(), P.lo.data_ptr()
                    for i in range(4):
                        j.bank_off[i] = offs[i]
                    altv = (P, offs)
                made.append((w, ent, sc, alt, fwd, altv))
            _lib.check(lib.cips_conv_weight_prep_batch(jobs, len(chunk), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "cips_conv_weight_prep_batch")
            for w, ent, sc, alt, fwd, altv in made:
                if fwd is not None:
                    ent[("fwd", sc)] = (w._version, w.data_ptr(), fwd)
                if altv is not None:
                    ent[(alt, sc)] = (w._version, w.data_ptr(), altv)


def _w_planes_raw(w):
    O, C, kh, kw = w.shape
    P, _ = ops.split_planes(w.permute(0, 2, 3, 1).reshape(1, O, kh * kw * C).contiguous(), want_p=True, want_t=False)
    return P


def _w_planes(w, scale=1.0):
    """(O, C, kh, kw) -> Planes (O, kh*kw*C) of w * scale: contraction index (tap, channel)"""
    return _cached(w, scale, "fwd", _w_planes_raw)


def _w_planes_flipT(w, scale=1.0):
    """planes of the data-gradient filter bank: flipped taps, channel roles swapped, (C, kh*kw*O)"""
    return _cached(w, scale, "flipT", lambda we: _w_planes_raw(we.flip(2, 3).transpose(0, 1)))


def _w_banks_s2(w, scale=1.0):
    """the four parity filter banks of the stride-2 data gradient (ops.dgrad_s2_banks): (Planes, offsets)"""
    return _cached(w, scale, "s2banks", ops.dgrad_s2_banks)


def _w_rows(w, scale, want_t):
    """(O, K) matrix of w * scale as planes (1, O, K) (want_t False) or transposed (1, K, O)"""
    def build(we):
        O = we.shape[0]
        P, T = ops.split_planes(we.reshape(1, O, -1).contiguous(), want_p=not want_t, want_t=want_t)
        return T if want_t else P
    return _cached(w, scale, "rowsT" if want_t else "rows", build)


def _scaled(w, scale):
    return w if scale == 1.0 else w * scale


def _conv_fwd(x, w, stride, pad, scale=1.0, pre=None):
    """y = conv(Blur(x), w * scale); `w` may be the nn.Parameter itself (its operand planes are cached, see _cached)"""
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    x = x.contiguous()
    if pre is not None:
        Hb, Wb = _pre_shape(H, W, pre)
        if _implicit_ok(C, ((Hb + 2 * pad - kh) // stride + 1) * ((Wb + 2 * pad - kw) // stride + 1), O):
            return ops.conv2d_x3(_w_planes(w, scale), _nhwc(x, pre), B, C, Hb, Wb, O, kh, kw, stride, pad)
        x = _pre_fp32(x, pre)
        H, W = Hb, Wb
    Ho_, Wo_ = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and C <= 4 and (H * W) % 4 == 0:
        return ops.conv1x1_smallk(x, _scaled(w, scale).reshape(O, C).contiguous())      # RGB input convs: streaming, no GEMM
    if _implicit_ok(C, Ho_ * Wo_, O):
        return ops.conv2d_x3(_w_planes(w, scale), _nhwc(x), B, C, H, W, O, kh, kw, stride, pad)
    if _fold_ok(C * kh * kw, Ho_ * Wo_, B, O):
        K, N = C * kh * kw, Ho_ * Wo_
        colP, _, _ = _folded_col_planes(x, kh, kw, stride, pad)
        wT = _w_rows(w, scale, True)                                                           # (1, K, O)
        nch = _split_count(((O + 255) // 256) * ((B * N + 127) // 128), K)
        kc = K // nch
        y2 = torch.empty(nch, O, B * N, device=x.device)
        ops.gemm_x3_km(wT, colP, O, B * N, kc, O, B * N, nch, kc * O, kc * B * N, y2)   # y (O, B*N) = W (O,K) @ col (K, B*N)
        y2 = y2.sum(0) if nch > 1 else y2[0]
        return y2.view(O, B, Ho_, Wo_).permute(1, 0, 2, 3).contiguous()
    if _x3_ok(C * kh * kw, Ho_ * Wo_, O):
        K, N = C * kh * kw, Ho_ * Wo_
        colP, _, _ = ops.im2col_x3(x, kh, kw, stride, pad)               # planes (B, K, N): k-major B operand
        wT = _w_rows(w, scale, True)                                     # (1, K, O): k-major A operand
        y = torch.empty(B, O, Ho_, Wo_, device=x.device)
        ops.gemm_x3_km(wT, colP, O, N, K, O, N, B, 0, K * N, y)          # y[b] (O,N) = W (O,K) @ col[b] (K,N)
        return y
    col, Ho, Wo = ops.im2col(x, kh, kw, stride, pad)                  # (B, K, Ho*Wo)
    K, N = C * kh * kw, Ho * Wo
    if N % 4:
        raise NotImplementedError("conv output plane must have a multiple of 4 pixels")
    wm = _scaled(w, scale).reshape(O, K)
    Kp = _pad4(K)
    if Kp != K:   # RGB input (C_in = 3, 1x1): pad the contraction dim to the GEMM's 16-byte vector granule
        wm = F.pad(wm, (0, Kp - K))
        col = F.pad(col, (0, 0, 0, Kp - K))
    wm = wm.contiguous()
    y = torch.empty(B, O, Ho, Wo, device=x.device)
    ops.gemm(wm, col, y, O, N, Kp, Kp, N, N, batch=B, strideA=0, strideB=Kp * N, strideC=O * N)
    return y


def _s2_parity_ok(C, O, N):
    """stride-2 data gradient as parity sub-convolutions: O a multiple of 32 with two k-tiles in the single-tap class (small
    planes: the batch folded into the pixel dimension)"""
    return CONV_MODE == "bf16x3" and O % 32 == 0 and O >= 64 and C % 8 == 0 and N % 8 == 0


def _conv_bwd_data(dy, w, in_shape, stride, pad, scale=1.0, pre=None):
    if pre is not None:
        B, C, H, W = in_shape
        O, _, kh, kw = w.shape
        Hb, Wb = _pre_shape(H, W, pre)
        if (stride == 2 and pad == 0 and pre[3] == 1 and dy.is_cuda and _s2_parity_ok(C, O, dy.shape[2] * dy.shape[3])):
            banks, w_off = _w_banks_s2(w, scale)
            dxp, out_off = ops.conv2d_x3_dgrad_s2(banks, w_off, _nhwc(dy if _is_planes_only(dy) else dy.contiguous()), B, C, Hb, Wb, O, kh, kw)
            gx0, gx1, gy0, gy1 = _pre_adjoint_pads(H, W, pre)
            return ops.upfirdn2d_parity(dxp, out_off, _flipped(pre[0]), B * C, Hb, Wb, gx0, gx1, gy0, gy1).view(B, C, H, W)
        return _pre_adjoint(_conv_bwd_data(dy, w, (B, C, Hb, Wb), stride, pad, scale), pre, in_shape)
    B, C, H, W = in_shape
    O, _, kh, kw = w.shape
    po = _is_planes_only(dy)
    if not po:
        dy = dy.contiguous()
    Ho, Wo = dy.shape[2], dy.shape[3]
    K, N = C * kh * kw, Ho * Wo
    if stride == 1 and _implicit_ok(O, H * W, C) and Ho + kh - 1 - 2 * pad == H and kh - 1 - pad >= 0 and not (kh == 1 and kw == 1 and C <= 4):
        # dx = conv(dy, flipped weights with the channel roles swapped), padding kh-1-pad
        return ops.conv2d_x3(_w_planes_flipT(w, scale), _nhwc(dy), B, O, Ho, Wo, C, kh, kw, 1, kh - 1 - pad)
    if po:
        dy = _dense(dy)                   # raises: every path below reads values
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and C <= 4 and (H * W) % 4 == 0:
        return ops.conv1x1_smallk_bwd_data(dy, _scaled(w, scale).reshape(O, C).contiguous(), C)
    if _fold_ok(K, N, B, O):
        wP = _w_rows(w, scale, False)                                                            # (1, O, K)
        dcol2 = torch.empty(K, B * N, device=dy.device)
        ops.gemm_x3_km(wP, _folded_rows_planes(dy), K, B * N, O, K, B * N, 1, 0, 0, dcol2)       # dcol (K, B*N) = W^T dy
        return ops.col2im(dcol2.view(K, B, N).permute(1, 0, 2).contiguous(), B, C, H, W, kh, kw, stride, pad)
    if _x3_ok(K, N, O):
        wP = _w_rows(w, scale, False)                                    # (1, O, K): contraction index o = rows
        dyP, _ = ops.split_planes(dy.view(B, O, N), want_p=True, want_t=False)
        dcol = torch.empty(B, K, N, device=dy.device)
        ops.gemm_x3_km(wP, dyP, K, N, O, K, N, B, 0, O * N, dcol)        # dcol[b] (K,N) = W^T (K,O) @ dy[b] (O,N)
        return ops.col2im(dcol, B, C, H, W, kh, kw, stride, pad)
    wm = _scaled(w, scale).reshape(O, K)
    Kp = _pad4(K)
    if Kp != K:
        wm = F.pad(wm, (0, Kp - K))
    wm = wm.contiguous()
    dcol = torch.empty(B, Kp, N, device=dy.device)
    ops.gemm(wm, dy, dcol, Kp, N, O, Kp, N, N, batch=B, strideA=0, strideB=O * N, strideC=Kp * N, a_kmajor=True)
    if Kp != K:
        dcol = dcol[:, :K].contiguous()
    return ops.col2im(dcol, B, C, H, W, kh, kw, stride, pad)


def _conv_bwd_weight(dy, x, w_shape, stride, pad, scale=1.0, pre=None):
    """d/dw of conv(Blur(x), w * scale): scale * (dy correlated with Blur(x))"""
    if pre is not None:
        O, C, kh, kw = w_shape
        Hb, Wb = _pre_shape(x.shape[2], x.shape[3], pre)
        if _implicit_ok(C, dy.shape[2] * dy.shape[3], O) and x.is_cuda:
            dw = ops.conv2d_x3_wgrad(_nhwc(dy if _is_planes_only(dy) else dy.contiguous()), _nhwc(x.contiguous(), pre), x.shape[0], C, Hb, Wb, O, kh, kw, stride, pad, scale)
            if dw is not None:
                return dw
        x = _pre_fp32(x, pre)
    dw = _conv_bwd_weight_raw(dy, x, w_shape, stride, pad, scale)
    if isinstance(dw, tuple):            # (tensor,): the path applied the scale itself
        return dw[0]
    return dw if scale == 1.0 else dw * scale


def _conv_bwd_weight_raw(dy, x, w_shape, stride, pad, scale):
    O, C, kh, kw = w_shape
    B = x.shape[0]
    x = x.contiguous()
    po = _is_planes_only(dy)
    if not po:
        dy = dy.contiguous()
    K, N = C * kh * kw, dy.shape[2] * dy.shape[3]
    if _implicit_ok(C, N, O):
        dw = ops.conv2d_x3_wgrad(_nhwc(dy), _nhwc(x), B, C, x.shape[2], x.shape[3], O, kh, kw, stride, pad, scale)
        if dw is not None:
            return (dw,)
    if po:
        dy = _dense(dy)                   # raises: every path below reads values
    if kh == 1 and kw == 1 and stride == 1 and pad == 0 and C <= 4 and N % 4 == 0:
        return ops.conv1x1_smallk_bwd_weight(dy, x).view(O, C, 1, 1)       # RGB input convs: streaming reduction
    if _fold_ok(K, N, B, O):
        colP, _, _ = _folded_col_planes(x, kh, kw, stride, pad)
        nch = _split_count(((O + 255) // 256) * ((K + 127) // 128), B * N)
        nc = B * N // nch
        part = torch.empty(nch, O, K, device=x.device)
        ops.gemm_x3(_folded_rows_planes(dy), colP, O, K, nc, B * N, B * N, nch, nc, nc, C=part)   # dW (O,K) = dy (O,B*N) col^T
        return (part.sum(0) if nch > 1 else part[0]).view(O, C, kh, kw)
    if _x3_ok(K, N, O):
        colP, _, _ = ops.im2col_x3(x, kh, kw, stride, pad)
        dyP, _ = ops.split_planes(dy.view(B, O, N), want_p=True, want_t=False)
        part = torch.empty(B, O, K, device=x.device)
        ops.gemm_x3(dyP, colP, O, K, N, N, N, B, O * N, K * N, C=part)   # part[b] (O,K) = dy[b] (O,N) @ col[b]^T (N,K)
        return part.sum(0).view(O, C, kh, kw)
    col, Ho, Wo = ops.im2col(x, kh, kw, stride, pad)
    part = torch.empty(B, O, K, device=x.device)
    ops.gemm(dy, col, part, O, K, N, N, N, K, batch=B, strideA=O * N, strideB=K * N, strideC=O * K, b_nmajor=True)
    return part.sum(0).view(O, C, kh, kw)


class _WeightGradPort(Function):
    """(w, x) -> a stride-0 zero tensor shaped like the convolution's output, handed to the convolution node as a third input.
    Its only purpose is to be a SEPARATE NODE in front of the weight: the convolution node sends dy back through it, and the
    engine runs this backward — the weight gradient — only in graph tasks that ask for d/dw.  torch.autograd.grad(r_preds.sum(),
    real_imgs, create_graph=True) of the R1 penalty (train.py:387-394) does not; torch's own conv2d learns that from
    task_should_compute_output, a Python Function's ctx.needs_input_grad is fixed at forward time and said "yes": before round
    6 every convolution computed (and dropped) its weight gradient in that pass — a quarter of all weight-gradient GEMMs of
    the D step.  The backward is Conv2dBwdWeightFunction, so every higher order is unchanged."""

    @staticmethod
    def forward(ctx, w, xbox, out_shape, stride, pad, scale, pre):
        # x arrives boxed, NOT as a graph input: with an edge to x this node would sit on every path to the images and run in
        # the R1 pass after all.  The reference keeps x's own graph (the weight gradient's dependence on x is differentiated
        # through Conv2dBwdWeightFunction when this backward runs under create_graph).
        ctx.x = xbox[0]
        ctx.cfg = (tuple(w.shape), stride, pad, scale, pre)
        return _zero1(w).expand(out_shape)

    @staticmethod
    def backward(ctx, g):
        w_shape, stride, pad, scale, pre = ctx.cfg
        with _share_planes():
            dw = Conv2dBwdWeightFunction.apply(g, ctx.x, w_shape, stride, pad, scale, pre)
        return dw, None, None, None, None, None, None


_ZERO1 = {}


def _zero1(like):
    """one zero element per (device, dtype), made once (a fill launch per convolution call otherwise); never written"""
    key = (like.device, like.dtype)
    z = _ZERO1.get(key)
    if z is None:
        z = torch.zeros(1, device=like.device, dtype=like.dtype)
        if like.is_cuda and torch.cuda.is_current_stream_capturing():
            return z                                   # not cached: it lives in the capturing graph's pool
        _ZERO1[key] = z
    return z


def _weight_port(x, w, out_shape, stride, pad, scale, pre):
    if torch.is_grad_enabled() and w.requires_grad:
        return _WeightGradPort.apply(w, (x,), out_shape, stride, pad, scale, pre)
    return None


def _conv_out_shape(x, w, stride, pad, pre):
    Hb, Wb = _pre_shape(x.shape[2], x.shape[3], pre)
    return (x.shape[0], w.shape[0], (Hb + 2 * pad - w.shape[2]) // stride + 1, (Wb + 2 * pad - w.shape[3]) // stride + 1)


class Conv2dFunction(Function):
    """y = conv(Blur(x), w * scale).  `scale` is EqualConv2d's constant 1/sqrt(C k^2) (discriminator.py:33, 44): keeping it
    out of the tensor lets the parameter itself arrive here, so that its operand planes can be cached (_cached).  `port`: the
    weight's _WeightGradPort (or None): dy goes back through it, the weight gradient is that node's."""

    @staticmethod
    def forward(ctx, x, w, port, stride, pad, scale=1.0, pre=None):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.pad, ctx.scale, ctx.pre = stride, pad, scale, pre
        ctx.w_obj = w if isinstance(w, nn.Parameter) else None       # the Parameter object: the cache key
        with _share_planes():
            y = _conv_fwd(x, w, stride, pad, scale, pre)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        w = ctx.w_obj if ctx.w_obj is not None else w
        dx = None
        dy = dy.contiguous()
        with _share_planes():
            if ctx.needs_input_grad[0]:
                dx = Conv2dBwdDataFunction.apply(dy, w, x.shape, ctx.stride, ctx.pad, ctx.scale, ctx.pre)
        return dx, None, (dy if ctx.needs_input_grad[2] else None), None, None, None, None


def _conv_apply(x, w, stride, pad, scale=1.0, pre=None):
    x = x.contiguous()
    return Conv2dFunction.apply(x, w, _weight_port(x, w, _conv_out_shape(x, w, stride, pad, pre), stride, pad, scale, pre),
                                stride, pad, scale, pre)


class ConvBiasActFunction(Function):
    """out = leaky_relu(conv(x, w * scale) + bias, slope) * act_scale in ONE kernel (bias and activation in the epilogue
    of the implicit-GEMM convolution): EqualConv2d followed by FusedLeakyReLU (discriminator.py:205-215).  The backward is
    the composition of the two layers' own backward Functions, so every higher-order path (R1) is theirs; the weight
    gradient is the port's (see _WeightGradPort)."""

    @staticmethod
    def forward(ctx, x, w, bias, port, stride, pad, scale, slope, act_scale, pre=None):
        B, C, H, W = x.shape
        O, _, kh, kw = w.shape
        x = x.contiguous()
        Hb, Wb = _pre_shape(H, W, pre)
        with _share_planes():
            out = ops.conv2d_x3(_w_planes(w, scale), _nhwc(x, pre), B, C, Hb, Wb, O, kh, kw, stride, pad, bias=bias.detach().contiguous(),
                                act=True, slope=slope, act_scale=act_scale)
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (stride, pad, scale, slope, act_scale, pre)
        ctx.w_obj = w if isinstance(w, nn.Parameter) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        w = ctx.w_obj if ctx.w_obj is not None else w
        stride, pad, scale, slope, act_scale, pre = ctx.cfg
        po = PLANES_ONLY_GRADIENT and _planes_only_ok(x.shape, w.shape, stride, pad, pre, ctx.needs_input_grad[0], dout)
        dpre, dbias = FusedLeakyReLUFunctionBackward.apply(dout.contiguous(), out, slope, act_scale, po)
        dx = None
        with _share_planes():
            if ctx.needs_input_grad[0]:
                dx = Conv2dBwdDataFunction.apply(dpre, w, x.shape, stride, pad, scale, pre)
        return (dx, None, (dbias if ctx.needs_input_grad[2] else None), (dpre if ctx.needs_input_grad[3] else None),
                None, None, None, None, None, None)


PLANES_ONLY_GRADIENT = True      # False: FusedLeakyReLU's backward writes the fp32 gradient and the convolution splits it (the parity test flips it)


def _planes_only_ok(x_shape, w_shape, stride, pad, pre, need_dx, dout):
    """may the gated gradient of this ConvBiasAct layer exist as NHWC planes only?  Yes when BOTH of its consumers — the
    convolution's data gradient (if asked for) and weight gradient — take their implicit-GEMM forms, which read planes and
    nothing else (the conditions of _conv_bwd_data / _conv_bwd_weight, restated)"""
    if not (CONV_MODE == "bf16x3" and dout.is_cuda and dout.dtype == torch.float32 and dout.dim() == 4):
        return False
    B, C, H, W = x_shape
    O, _, kh, kw = w_shape
    Hb, Wb = _pre_shape(H, W, pre)
    Ho, Wo = dout.shape[2], dout.shape[3]
    N = Ho * Wo
    if not _implicit_ok(C, N, O) or (B * N) % 32 or O % 8:
        return False                                                     # weight gradient: cips_conv2d_x3_wgrad's conditions
    if need_dx:
        if stride == 1:
            if pre is not None or not (_implicit_ok(O, Hb * Wb, C) and Ho + kh - 1 - 2 * pad == Hb and kh - 1 - pad >= 0):
                return False
        elif not (stride == 2 and pad == 0 and pre is not None and pre[3] == 1 and _s2_parity_ok(C, O, N)):
            return False
    return True


def _conv_bias_act_apply(x, w, bias, stride, pad, scale, slope, act_scale, pre=None):
    x = x.contiguous()
    return ConvBiasActFunction.apply(x, w, bias, _weight_port(x, w, _conv_out_shape(x, w, stride, pad, pre), stride, pad, scale, pre),
                                     stride, pad, scale, slope, act_scale, pre)


FOLD_BLUR = True            # False: Blur as its own upfirdn2d node in front of the convolution (the folding's parity test flips it)
_CONV_ACT_FUSED = True      # False: bias + LeakyReLU as a separate pass after the convolution (the fusion's parity test flips it)


def _conv_act_fusable(x, conv, act, pre=None):
    """EqualConv2d (no bias of its own) + FusedLeakyReLU on a GPU batch whose conv takes the implicit-GEMM path"""
    if not (_CONV_ACT_FUSED and x.is_cuda and x.dtype == torch.float32 and conv.bias is None and isinstance(act, FusedLeakyReLU)):
        return False
    if GATE_PIN is not None or GATE_REC is not None:        # gate instrumentation works on the separate activation op
        return False
    B, C, H, W = x.shape
    H, W = _pre_shape(H, W, pre)
    O, _, kh, kw = conv.weight.shape
    Ho, Wo = (H + 2 * conv.padding - kh) // conv.stride + 1, (W + 2 * conv.padding - kw) // conv.stride + 1
    rgb = kh == 1 and kw == 1 and C <= 4
    return (not rgb) and Ho > 0 and Wo > 0 and _implicit_ok(C, Ho * Wo, O)


class Conv2dBwdDataFunction(Function):
    @staticmethod
    def forward(ctx, dy, w, in_shape, stride, pad, scale=1.0, pre=None):
        ctx.save_for_backward(dy, w)
        ctx.in_shape, ctx.stride, ctx.pad, ctx.scale, ctx.pre = in_shape, stride, pad, scale, pre
        ctx.w_obj = w if isinstance(w, nn.Parameter) else None
        return _conv_bwd_data(dy, w, in_shape, stride, pad, scale, pre)

    @staticmethod
    def backward(ctx, ggx):
        dy, w = ctx.saved_tensors
        w = ctx.w_obj if ctx.w_obj is not None else w
        g_dy = g_w = None
        ggx = ggx.contiguous()
        with _share_planes():
            if ctx.needs_input_grad[0]:
                g_dy = _conv_apply(ggx, w, ctx.stride, ctx.pad, ctx.scale, ctx.pre)
            if ctx.needs_input_grad[1]:
                g_w = Conv2dBwdWeightFunction.apply(dy, ggx, w.shape, ctx.stride, ctx.pad, ctx.scale, ctx.pre)
        return g_dy, g_w, None, None, None, None, None


class Conv2dBwdWeightFunction(Function):
    """dw = scale * wgrad(dy, x): the gradient of conv(x, w * scale) w.r.t. w"""

    @staticmethod
    def forward(ctx, dy, x, w_shape, stride, pad, scale=1.0, pre=None):
        ctx.save_for_backward(dy, x)
        ctx.w_shape, ctx.stride, ctx.pad, ctx.scale, ctx.pre = w_shape, stride, pad, scale, pre
        return _conv_bwd_weight(dy, x, w_shape, stride, pad, scale, pre)

    @staticmethod
    def backward(ctx, ggw):
        dy, x = ctx.saved_tensors
        g_dy = g_x = None
        if ctx.needs_input_grad[0]:
            g_dy = _conv_apply(x, ggw, ctx.stride, ctx.pad, ctx.scale, ctx.pre)
        if ctx.needs_input_grad[1]:
            g_x = Conv2dBwdDataFunction.apply(dy, ggw, x.shape, ctx.stride, ctx.pad, ctx.scale, ctx.pre)
        return g_dy, g_x, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, padding=0, scale=1.0, pre=None):
    """conv2d(Blur(x), w * scale) + bias; pass the raw nn.Parameter and its constant scale to get the plane cache.
    pre = (4 x 4 kernel, pad0, pad1, down): the Blur of a down-sampling ConvLayer folded in (no blurred tensor exists)"""
    y = _conv_apply(x, w, stride, padding, scale, pre)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y


# ------------------------------------------------------------------------------------------
# modules (same names / shapes / init as the reference)
# ------------------------------------------------------------------------------------------
class EqualConv2d(nn.Module):
    """discriminator.py:20-54"""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return conv2d(input, self.weight, bias=self.bias, stride=self.stride, padding=self.padding, scale=self.scale)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Blur(nn.Module):
    """discriminator.py:67-82"""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer('kernel', kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class ConvLayer(nn.Sequential):
    """discriminator.py:134-222 (down path only: the discriminator never upsamples)"""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True, upsample=False, padding="zero"):
        layers = collections.OrderedDict()
        self.padding = 0
        stride = 1
        if upsample or padding != "zero":
            raise NotImplementedError("only the discriminator's ConvLayer modes are built (zero padding, no upsample)")
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers['down_blur'] = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
            stride = 2
        else:
            self.padding = (kernel_size - 1) // 2
        layers['equal_conv'] = EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                           bias=bias and not activate)
        if activate:
            if bias:
                layers['flrelu'] = FusedLeakyReLU(out_channel)
            else:
                layers['slrelu'] = ScaledLeakyReLU(0.2)
        super().__init__(layers)

    def forward(self, input):
        # Blur followed by a 1x1 stride-2 conv (the ResBlock skip branch) only ever reads blur[2i][2j]: ask upfirdn2d for
        # exactly those samples (down = 2: a quarter of the blur's output bytes, the same taps in the same order) and run
        # the 1x1 conv at stride 1 on the quarter-size map — implicit GEMM forward and backward, no im2col / col2im.
        blur = getattr(self, "down_blur", None)
        conv = self.equal_conv
        fold = (FOLD_BLUR and blur is not None and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
                and tuple(blur.kernel.shape) == (4, 4) and min(blur.pad) >= 0)
        if (blur is not None and input.is_cuda and conv.weight.shape[2] == 1 and conv.weight.shape[3] == 1
                and conv.stride == 2 and conv.padding == 0):
            if fold:      # Blur (sampled at stride 2) + 1 x 1 convolution as one op: blurred planes straight from the fused kernel
                x = conv2d(input, conv.weight, bias=conv.bias, stride=1, padding=0, scale=conv.scale,
                           pre=(blur.kernel, blur.pad[0], blur.pad[1], 2))
            else:
                x = upfirdn2d(input, blur.kernel, down=2, pad=blur.pad)
                x = conv2d(x, conv.weight, bias=conv.bias, stride=1, padding=0, scale=conv.scale)
            for name, m in self.named_children():
                if name not in ("down_blur", "equal_conv"):
                    x = m(x)
            return x
        act = getattr(self, "flrelu", None)
        if act is not None and input.is_cuda:
            if fold:
                pre = (blur.kernel, blur.pad[0], blur.pad[1], 1)
                if _conv_act_fusable(input, conv, act, pre):
                    return _conv_bias_act_apply(input, conv.weight, act.bias, conv.stride, conv.padding, conv.scale,
                                                act.negative_slope, act.scale, pre)
                return act(conv2d(input, conv.weight, bias=conv.bias, stride=conv.stride, padding=conv.padding, scale=conv.scale, pre=pre))
            x = blur(input) if blur is not None else input
            if _conv_act_fusable(x, conv, act):
                return _conv_bias_act_apply(x, conv.weight, act.bias, conv.stride, conv.padding, conv.scale,
                                            act.negative_slope, act.scale)
            return act(conv(x))
        return super().forward(input)


class ResBlock(nn.Module):
    """discriminator.py:224-252"""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1], kernel_size=3, downsample=True,
                 first_downsample=False):
        super().__init__()
        if first_downsample:
            self.conv1 = ConvLayer(in_channel, in_channel, kernel_size, downsample=downsample)
            self.conv2 = ConvLayer(in_channel, out_channel, kernel_size)
        else:
            self.conv1 = ConvLayer(in_channel, in_channel, kernel_size)
            self.conv2 = ConvLayer(in_channel, out_channel, kernel_size, downsample=downsample)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=downsample, activate=False, bias=False)

    def forward(self, input):
        out = self.conv2(self.conv1(input))
        skip = self.skip(input)
        if out.is_cuda:
            c = 1.0 / math.sqrt(2)
            return _BlendFunction.apply(out, skip, c, c)         # (out + skip) / sqrt(2) in one pass (cips_axpby)
        return (out + skip) / math.sqrt(2)


def _eql(mode, a, b, s, B, K, O, bias=None, bias_scale=1.0):
    """cips_equal_linear: mode 0  s a b^T (+ bias * bias_scale) -> (B, O);  1  s a b -> (B, K);  2  s a^T b -> (O, K)"""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    a, b = a.contiguous().float(), b.contiguous().float()
    if not a.is_cuda:
        raise RuntimeError("EqualLinear runs on the HIP library only: GPU tensors required")
    out = torch.empty({0: (B, O), 1: (B, K), 2: (O, K)}[mode], device=a.device)
    n = int(lib.cips_equal_linear_scratch(mode, B, K, O))
    scratch = torch.empty(n, device=a.device) if n else None
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(a.device):
        _lib.check(lib.cips_equal_linear(mode, P(a), P(b), P(bias.contiguous().float() if bias is not None else None),
                                         float(bias_scale), float(s), P(out), P(scratch), B, K, O,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cips_equal_linear")
    return out


class _EqLinFwd(Function):
    """y = s x w^T (+ bias * bias_scale).  Together with _EqLinDx (s g w) and _EqLinDw (s g^T x) the three forms are
    closed under differentiation: every backward below is again one of them, so the R1 double-backward
    (train.py:387-394) needs nothing else."""

    @staticmethod
    def forward(ctx, x, w, bias, s, bias_scale):
        ctx.save_for_backward(x, w)
        ctx.s, ctx.bs, ctx.has_bias = s, bias_scale, bias is not None
        return _eql(0, x, w, s, x.shape[0], x.shape[1], w.shape[0], bias, bias_scale)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = _EqLinDx.apply(g, w, ctx.s) if ctx.needs_input_grad[0] else None
        dw = _EqLinDw.apply(g, x, ctx.s) if ctx.needs_input_grad[1] else None
        db = g.sum(0) * ctx.bs if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None


class _EqLinDx(Function):
    @staticmethod
    def forward(ctx, g, w, s):
        ctx.save_for_backward(g, w)
        ctx.s = s
        return _eql(1, g, w, s, g.shape[0], w.shape[1], w.shape[0])

    @staticmethod
    def backward(ctx, u):
        g, w = ctx.saved_tensors
        dg = _EqLinFwd.apply(u, w, None, ctx.s, 1.0) if ctx.needs_input_grad[0] else None
        dw = _EqLinDw.apply(g, u, ctx.s) if ctx.needs_input_grad[1] else None
        return dg, dw, None


class _EqLinDw(Function):
    @staticmethod
    def forward(ctx, g, x, s):
        ctx.save_for_backward(g, x)
        ctx.s = s
        return _eql(2, g, x, s, g.shape[0], x.shape[1], g.shape[1])

    @staticmethod
    def backward(ctx, u):
        g, x = ctx.saved_tensors
        dg = _EqLinFwd.apply(x, u, None, ctx.s, 1.0) if ctx.needs_input_grad[0] else None
        dx = _EqLinDx.apply(g, u, ctx.s) if ctx.needs_input_grad[1] else None
        return dg, dx, None




class EqualLinear(nn.Module):
    """discriminator.py:254-288.  (b x 8192) @ (8192 x 512) and (b x 512) @ (512 x 1) on the HIP library
    (cips_equal_linear: the exact-fp32 MFMA GEMM with the long contraction cut into chunks, streaming kernels for the
    one-column output layer), double-differentiable through _EqLinFwd / _EqLinDx / _EqLinDw; the activation is the fused
    bias + LeakyReLU op.  Inputs of more than two dimensions (not what the discriminator feeds it) take torch's linear."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        if input.dim() == 2:
            if self.activation:
                out = _EqLinFwd.apply(input, self.weight, None, self.scale, 1.0)
                return fused_leaky_relu(out, self.bias * self.lr_mul)
            return _EqLinFwd.apply(input, self.weight, self.bias, self.scale, self.lr_mul)
        if self.activation:
            out = F.linear(input, self.weight * self.scale)
            return fused_leaky_relu(out, self.bias * self.lr_mul)
        return F.linear(input, self.weight * self.scale, bias=self.bias * self.lr_mul)


class _DiffAugFunction(Function):
    """y = A x (+ c): the fused DiffAugment operator (cips_diffaug) or, with adjoint=True, its transpose.  The two are
    each other's backward, which gives the R1 double-backward for free (the operator is affine)."""

    @staticmethod
    def forward(ctx, x, draws, adjoint, affine):
        import ctypes as C
        from . import _lib
        rb, rs, rc, tx, ty, ox, oy, (cut_h, cut_w, color) = draws
        x = x.contiguous().float()
        B, Cc, H, W = x.shape
        y = torch.empty_like(x)
        sums = torch.empty(33 * B, device=x.device)          # per-image sums + 32 slices of partials (cips_diffaug)
        lib = _lib.load()
        P = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(x.device):
            _lib.check(lib.cips_diffaug(P(x), P(y), P(rb), P(rs), P(rc), P(tx), P(ty), P(ox), P(oy), P(sums), B, Cc, H, W,
                                        cut_h, cut_w, 1 if adjoint else 0, (1 if affine else 0) | (0 if color else 2),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cips_diffaug")
        ctx.draws, ctx.adjoint = draws, adjoint
        return y

    @staticmethod
    def backward(ctx, g):
        return _DiffAugFunction.apply(g, ctx.draws, not ctx.adjoint, False), None, None, None


_POLICY_ORDER = ("color", "translation", "cutout")


def _diffaug_hip(x, parts):
    """`parts`: the policy's stages, a subsequence of (color, translation, cutout).  The reference's draws for the stages
    that are present, in its order and with its calls (diffaug.py:32, 38, 44, 50-51, 66-67), then one fused operator;
    an absent stage is the operator's identity setting (no colour arithmetic, zero shift, empty hole)."""
    b, _, h, w = x.shape
    dev = x.device
    if "color" in parts:
        rb = torch.rand(b, 1, 1, 1, dtype=x.dtype, device=dev)
        rs = torch.rand(b, 1, 1, 1, dtype=x.dtype, device=dev)
        rc = torch.rand(b, 1, 1, 1, dtype=x.dtype, device=dev)
    else:
        rb = rs = rc = torch.full((b,), 0.5, device=dev)
    if "translation" in parts:
        sx, sy = int(h * 0.125 + 0.5), int(w * 0.125 + 0.5)
        tx = torch.randint(-sx, sx + 1, size=[b, 1, 1], device=dev)
        ty = torch.randint(-sy, sy + 1, size=[b, 1, 1], device=dev)
    else:
        tx = ty = torch.zeros(b, dtype=torch.long, device=dev)
    ch, cw = (int(h * 0.2 + 0.5), int(w * 0.2 + 0.5)) if "cutout" in parts else (0, 0)
    if "cutout" in parts:
        ox = torch.randint(0, h + (1 - ch % 2), size=[b, 1, 1], device=dev)
        oy = torch.randint(0, w + (1 - cw % 2), size=[b, 1, 1], device=dev)
    else:
        ox = oy = torch.zeros(b, dtype=torch.long, device=dev)
    draws = tuple(t.reshape(b).contiguous() for t in (rb.float(), rs.float(), rc.float(), tx.long(), ty.long(), ox.long(), oy.long()))
    return _DiffAugFunction.apply(x, draws + ((ch, cw, "color" in parts),), False, True)


def DiffAugment(x, policy='', channels_first=True):
    """exp/cips3d/models/diffaug.py:9-85 (color, translation, cutout) as one fused HIP operator with its adjoint
    (cips_diffaug).  Policies are the reference's stage names, applied in the order listed like the reference does
    (color,translation,cutout and its subsequences — what the configs use — as one launch); no CPU path."""
    if not policy:
        return x
    parts = tuple(policy.split(','))
    if any(p not in _POLICY_ORDER for p in parts):
        raise ValueError(f"DiffAugment policy {policy!r}: unknown stage (known: {','.join(_POLICY_ORDER)})")
    # the canonical order (what the reference's configs use) is ONE fused launch; any other order or a repeated stage
    # runs the reference's way — stage by stage in the order listed (diffaug.py:13-16), each stage one launch of the
    # same operator, draws in the reference's order
    runs = [parts] if list(parts) == [p for p in _POLICY_ORDER if p in parts] else [(p,) for p in parts]
    if not x.is_cuda or x.dim() != 4:
        raise RuntimeError("DiffAugment runs on the HIP operator only: a 4-d GPU batch is required")
    if not channels_first:
        x = x.permute(0, 3, 1, 2)
    if x.shape[1] > 4:
        raise RuntimeError("DiffAugment: at most 4 channels (images)")
    y = x
    for run in runs:
        y = _diffaug_hip(y, run)
    if not channels_first:
        y = y.permute(0, 2, 3, 1)
    return y.contiguous()


def _hip_unary(name, *args):
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args, C.c_void_p(torch.cuda.current_stream().cuda_stream)), name)


class _AvgPool2Function(Function):
    """the 2x2 mean of the fade-in path (discriminator.py:525) and its transpose, each the other's backward"""

    @staticmethod
    def forward(ctx, x, adjoint):
        import ctypes as C
        x = x.contiguous().float()
        B, Cc, H, W = x.shape
        Ho, Wo = (H * 2, W * 2) if adjoint else (H // 2, W // 2)
        y = torch.empty(B, Cc, Ho, Wo, device=x.device)
        with torch.cuda.device(x.device):
            _hip_unary("cips_avgpool2", C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B * Cc, max(H, Ho), max(W, Wo),
                       1 if adjoint else 0)
        ctx.adjoint = adjoint
        return y

    @staticmethod
    def backward(ctx, g):
        return _AvgPool2Function.apply(g, not ctx.adjoint), None


class _BlendFunction(Function):
    """out = a x + b y (alpha * cur + (1 - alpha) * down, discriminator.py:534); y None: out = a x"""

    @staticmethod
    def forward(ctx, x, y, a, b):
        import ctypes as C
        x = x.contiguous().float()
        y = y.contiguous().float() if y is not None else None
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _hip_unary("cips_axpby", C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()) if y is not None else None,
                       C.c_void_p(out.data_ptr()), a, b, x.numel())
        ctx.ab, ctx.has_y = (a, b), y is not None
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        gx = _BlendFunction.apply(g, None, a, 0.0) if ctx.needs_input_grad[0] else None
        if ctx.has_y and ctx.needs_input_grad[1]:
            gy = gx if (gx is not None and a == b) else _BlendFunction.apply(g, None, b, 0.0)    # same scale: one pass
        else:
            gy = None
        return gx, gy, None, None


class Discriminator_MultiScale(nn.Module):
    """discriminator.py:406-585"""

    def __init__(self, diffaug, max_size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], input_size=3,
                 first_downsample=False, channels=None, stddev_group=4, **kwargs):
        super().__init__()
        self.epoch = 0
        self.step = 0
        self.diffaug, self.max_size, self.input_size, self.stddev_group = diffaug, max_size, input_size, stddev_group
        self.module_name_list = []
        if channels is None:
            channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                        256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.conv_in = nn.ModuleDict()
        self.module_name_list.append('conv_in')
        for name, channel_ in channels.items():
            self.conv_in[f"{name}"] = ConvLayer(input_size, channel_, 1)
        self.convs = nn.ModuleDict()
        self.module_name_list.append('convs')
        log_size = int(math.log(max_size, 2))
        in_channel = channels[max_size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            self.convs[f"{2 ** i}"] = ResBlock(in_channel, out_channel, blur_kernel, first_downsample=first_downsample)
            in_channel = out_channel
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + (1 if self.stddev_group > 1 else 0), channels[4], 3)
        self.module_name_list.append('final_conv')
        self.space_linear = EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu')
        self.module_name_list.append('space_linear')
        self.out_linear = EqualLinear(channels[4], 1)
        self.module_name_list.append('out_linear')

    def diff_aug_img(self, img):
        return DiffAugment(img, policy='color,translation,cutout')

    def _conv_layers(self, log_size):
        """(EqualConv2d, alternate operand form of its weight) for every convolution a forward at 2^log_size runs"""
        out = []
        for i in range(log_size, 2, -1):
            blk = self.convs[f"{2 ** i}"]
            for layer in (blk.conv1, blk.conv2, blk.skip):
                conv = layer.equal_conv
                s2 = hasattr(layer, "down_blur") and conv.weight.shape[2] > 1 and FOLD_BLUR
                out.append((conv, "s2banks" if s2 else "flipT"))
        out.append((self.final_conv.equal_conv, "flipT"))
        return out

    def forward(self, input, alpha, summary_ddict=None):
        if self.diffaug:
            input = self.diff_aug_img(input)
        size = input.shape[-1]
        log_size = int(math.log(size, 2))
        if input.is_cuda:
            prepare_weight_planes(self._conv_layers(log_size))
        cur = self.conv_in[f"{2 ** log_size}"](input)
        cur = self.convs[f"{2 ** log_size}"](cur)
        if alpha < 1:
            if input.is_cuda and input.shape[-1] % 2 == 0 and input.shape[-2] % 2 == 0:
                down_input = _AvgPool2Function.apply(input, False)       # = F.interpolate(input, 0.5, 'bilinear') on even sizes
                down = self.conv_in[f"{2 ** (log_size - 1)}"](down_input)
                out = _BlendFunction.apply(cur, down, float(alpha), float(1 - alpha))
            else:
                down_input = F.interpolate(input, scale_factor=0.5, mode='bilinear')
                down = self.conv_in[f"{2 ** (log_size - 1)}"](down_input)
                out = alpha * cur + (1 - alpha) * down
        else:
            out = cur
        for i in range(log_size - 1, 2, -1):
            out = self.convs[f"{2 ** i}"](out)
        batch, channel, height, width = out.shape
        if self.stddev_group > 0:
            group = min(batch, self.stddev_group)
            stddev = out.view(group, -1, self.stddev_feat, channel // self.stddev_feat, height, width)
            stddev = torch.sqrt(stddev.var(0, unbiased=False) + 1e-8)
            stddev = stddev.mean([2, 3, 4], keepdims=True).squeeze(2)
            stddev = stddev.repeat(group, 1, height, width)
            out = torch.cat([out, stddev], 1)
        out = self.final_conv(out)
        out = out.view(batch, -1)
        out = self.space_linear(out)
        if summary_ddict is not None:
            with torch.no_grad():
                summary_ddict['logits_norm']['logits_norm'] = out.norm(dim=1).mean().item()
                summary_ddict['w_norm']['w_norm'] = self.out_linear.weight.norm(dim=1).mean().item()
        out = self.out_linear(out)
        return out, None, None


AUX_SIDE_STREAM = True
_AUX_STREAMS = {}


def _aux_stream(device):
    key = torch.device(device).index
    st = _AUX_STREAMS.get(key)
    if st is None:
        st = _AUX_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class Discriminator_MultiScale_Aux(nn.Module):
    """discriminator.py:589-664"""

    def __init__(self, diffaug, max_size, channel_multiplier=2, first_downsample=False, stddev_group=0, **kwargs):
        super().__init__()
        self.epoch = 0
        self.step = 0
        self.main_disc = Discriminator_MultiScale(diffaug=diffaug, max_size=max_size,
                                                  channel_multiplier=channel_multiplier,
                                                  first_downsample=first_downsample, stddev_group=stddev_group)
        channel_multiplier = 2
        channels = {4: 128 * channel_multiplier, 8: 128 * channel_multiplier, 16: 128 * channel_multiplier,
                    32: 128 * channel_multiplier, 64: 128 * channel_multiplier, 128: 128 * channel_multiplier,
                    256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.aux_disc = Discriminator_MultiScale(diffaug=diffaug, max_size=max_size,
                                                 channel_multiplier=channel_multiplier, first_downsample=True,
                                                 channels=channels, stddev_group=stddev_group)

    def forward(self, input, use_aux_disc=False, summary_ddict=None, alpha=1., **kwargs):
        if use_aux_disc:
            b = input.shape[0] // 2
            if AUX_SIDE_STREAM and input.is_cuda:
                # EXPERIMENT (round 5): the two discriminators are independent networks on different halves of the batch; the
                # auxiliary one (256 channels at half resolution: grids that do not fill 256 CUs) runs on a side stream next to
                # the main one.  autograd keeps every node on its forward's stream, so both backward passes overlap as well.
                cur = torch.cuda.current_stream(input.device)
                side = _aux_stream(input.device)
                side.wait_stream(cur)
                input.record_stream(side)
                # main first in program order (the gate tapes of the parity tests record in call order; the host runs ahead of
                # the GPU either way, so the auxiliary launches are queued long before the main network's kernels have run)
                main_out, latent, position = self.main_disc(input[:b], alpha, summary_ddict=summary_ddict)
                with torch.cuda.stream(side):
                    aux_out, _, _ = self.aux_disc(input[b:], alpha)
                cur.wait_stream(side)
                aux_out.record_stream(cur)
            else:
                main_out, latent, position = self.main_disc(input[:b], alpha, summary_ddict=summary_ddict)
                aux_out, _, _ = self.aux_disc(input[b:], alpha)
            out = torch.cat([main_out, aux_out], dim=0)
        else:
            out, latent, position = self.main_disc(input, alpha, summary_ddict=summary_ddict)
        return out, latent, position
