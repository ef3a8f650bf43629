"""Reference-side bindings, shipped:

  native_ops   `fused` and `upfirdn2d_op` — objects with the exact call signatures of the reference's two pybind11
               extension modules (exp/comm/op/fused_bias_act.cpp:11-21, exp/comm/op/upfirdn2d.cpp:12-23), backed by
               libcips3d_hip.so, plus `load()`, a stand-in for torch.utils.cpp_extension.load so that the reference's
               own exp/comm/op/fused_act.py / upfirdn2d.py run unmodified on top of them;
  registry     registers the drop-in generator / discriminator classes in tl2's MODEL_REGISTRY under this module's
               name, so that `register_modules: [cips3d_amd.compat.registry]` in the reference's YAML selects them.
"""
from .native_ops import fused, upfirdn2d_op, load, patch_cpp_extension_load  # noqa: F401
