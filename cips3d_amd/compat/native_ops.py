"""Stand-ins for the reference's two JIT-compiled pybind11 modules, on the HIP library.

The reference builds them at import time (exp/comm/op/fused_act.py:10-16, exp/comm/op/upfirdn2d.py:9-15):

    fused = load('fused', sources=[fused_bias_act.cpp, fused_bias_act_kernel.cu])
    upfirdn2d_op = load('upfirdn2d', sources=[upfirdn2d.cpp, upfirdn2d_kernel.cu])

and calls `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)` and
`upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)` from its
autograd Functions.  A maintainer either replaces those two `load(...)` calls by

    from cips3d_amd.compat import fused            # fused_act.py
    from cips3d_amd.compat import upfirdn2d_op     # upfirdn2d.py

or leaves the files untouched and calls `cips3d_amd.compat.patch_cpp_extension_load()` before importing them
(tests/test_compat_cpu.py does exactly that with the unmodified reference files).

Conventions kept from the pybind modules: positional arguments in the same order and with the same names; a non-GPU
input raises RuntimeError (CHECK_CUDA, fused_bias_act.cpp:13-14); inputs are made contiguous internally
(fused_bias_act_kernel.cu:58-60); the output is allocated by the callee with the INPUT's dtype; an empty tensor means
"absent" for bias / refer (:62-63); the launch goes to the current stream of the input's device.
dtype: the reference dispatches fp16 / fp32 / fp64 (fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:177-211); the HIP
kernels compute in fp32, so fp16 / bf16 / fp64 inputs make an fp32 round trip and come back in their own dtype (fp64
therefore carries fp32 precision — stated, not silent); other dtypes raise."""
import torch

from .. import ops


class _FusedModule:
    """pybind module `fused` (exp/comm/op/fused_bias_act.cpp:24-26: m.def("fused_bias_act", ...))"""
    __name__ = "fused"

    @staticmethod
    def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
        return ops.fused_bias_act(input, bias, refer, int(act), int(grad), float(alpha), float(scale))


class _UpFirDn2dModule:
    """pybind module `upfirdn2d` (exp/comm/op/upfirdn2d.cpp:26-28: m.def("upfirdn2d", ...))"""
    __name__ = "upfirdn2d"

    @staticmethod
    def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
        return ops.upfirdn2d_op(input, kernel, int(up_x), int(up_y), int(down_x), int(down_y), int(pad_x0), int(pad_x1),
                                int(pad_y0), int(pad_y1))


fused = _FusedModule()
upfirdn2d_op = _UpFirDn2dModule()
_MODULES = {"fused": fused, "upfirdn2d": upfirdn2d_op}


def load(name, sources=None, **kwargs):
    """Stand-in for torch.utils.cpp_extension.load for the reference's two extensions: returns the HIP-backed module
    object instead of JIT-compiling CUDA sources.  Any other extension name is an error (nothing else is built here)."""
    try:
        return _MODULES[name]
    except KeyError:
        raise RuntimeError(f"cips3d_amd.compat.load: no HIP stand-in for extension {name!r} (have: {sorted(_MODULES)})")


def patch_cpp_extension_load():
    """Make `from torch.utils.cpp_extension import load` hand out the stand-ins for 'fused' / 'upfirdn2d' and defer to
    the real loader for anything else.  Call before importing the reference's exp.comm.op package."""
    import torch.utils.cpp_extension as ce
    real = getattr(ce.load, "_cips3d_real", ce.load)

    def patched(name, sources=None, **kwargs):
        if name in _MODULES:
            return _MODULES[name]
        return real(name, sources, **kwargs)
    patched._cips3d_real = real
    ce.load = patched
    return patched
