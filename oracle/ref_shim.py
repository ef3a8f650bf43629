"""TEST INFRASTRUCTURE ONLY — import shim that lets the *unmodified* reference
modules under /root/reference import in this container (no tl2 / easydict /
streamlit / torchvision / CUDA ops here).  Used exclusively by
`oracle/make_golden.py` to mint the golden fixtures under tests/golden/.
Nothing in the product package (cips3d_amd/) imports this file, and it is
never used on the GPU box (/root/reference does not exist there).

What is stubbed (call sites: SURVEY.md §8c):
  tl2.proj.fvcore.{MODEL_REGISTRY, build_model}     generator.py:17, discriminator.py:10
  tl2.launch.launch_utils.global_cfg (.tl_debug)    generator.py:19
  tl2.proj.pytorch.pytorch_hook.VerboseModel        generator.py:20
  tl2.proj.pytorch.{torch_utils, init_func}         generator.py:21-22
  tl2.tl2_utils.{get_class_repr, dict2string}       generator.py:23
  tl2.proj.stylegan2_ada.persistence                generator_nerf_inr.py:15
  easydict / streamlit / torchvision                comm_utils.py:8-9,15
  exp.comm.op (CUDA JIT ext) -> pure-torch restatement of
      fused_bias_act_kernel.cu:36-47 and upfirdn2d.py:152-186 (upfirdn2d_native)
"""
import sys
import types
import math
import torch
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


def _mod(name):
  m = types.ModuleType(name)
  m.__path__ = []
  sys.modules[name] = m
  return m


def upfirdn2d_native(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
  # follows exp/comm/op/upfirdn2d.py:152-186 (the reference's own unused pure-torch restatement)
  _, in_h, in_w, minor = input.shape
  kernel_h, kernel_w = kernel.shape
  out = input.view(-1, in_h, 1, in_w, 1, minor)
  out = F.pad(out, [0, 0, 0, up_x - 1, 0, 0, 0, up_y - 1])
  out = out.view(-1, in_h * up_y, in_w * up_x, minor)
  out = F.pad(out, [0, 0, max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
  out = out[:, max(-pad_y0, 0): out.shape[1] - max(-pad_y1, 0),
            max(-pad_x0, 0): out.shape[2] - max(-pad_x1, 0), :]
  out = out.permute(0, 3, 1, 2)
  out = out.reshape([-1, 1, in_h * up_y + pad_y0 + pad_y1, in_w * up_x + pad_x0 + pad_x1])
  w = torch.flip(kernel, [0, 1]).view(1, 1, kernel_h, kernel_w)
  out = F.conv2d(out, w)
  out = out.reshape(-1, minor, in_h * up_y + pad_y0 + pad_y1 - kernel_h + 1,
                    in_w * up_x + pad_x0 + pad_x1 - kernel_w + 1)
  out = out.permute(0, 2, 3, 1)
  return out[:, ::down_y, ::down_x, :]


def install():
  if "tl2" in sys.modules and getattr(sys.modules["tl2"], "_cips3d_shim", False):
    return
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)

  tl2 = _mod("tl2"); tl2._cips3d_shim = True
  proj = _mod("tl2.proj"); tl2.proj = proj

  class _Registry:
    def __init__(self): self._d = {}
    def register(self, name=None, name_prefix=None):
      def deco(cls):
        key = (name_prefix + "." if name_prefix else "") + (name or cls.__name__)
        self._d[key] = cls
        return cls
      return deco
    def get(self, name): return self._d[name]
  fv = _mod("tl2.proj.fvcore"); proj.fvcore = fv
  fv.MODEL_REGISTRY = _Registry()
  def build_model(cfg, kwargs_priority=False, cfg_to_kwargs=True, **kwargs):
    import importlib
    cfg = dict(cfg)
    for m in cfg.pop("register_modules", []): importlib.import_module(m)
    name = cfg.pop("name")
    merged = {**kwargs, **cfg} if not kwargs_priority else {**cfg, **kwargs}
    return fv.MODEL_REGISTRY.get(name)(**merged)
  fv.build_model = build_model

  launch = _mod("tl2.launch"); tl2.launch = launch
  lu = _mod("tl2.launch.launch_utils"); launch.launch_utils = lu
  class _Cfg(dict):
    def __getattr__(self, k):
      try: return self[k]
      except KeyError: raise AttributeError(k)
  lu.global_cfg = _Cfg(tl_debug=False)
  lu.TLCfgNode = _Cfg

  pt = _mod("tl2.proj.pytorch"); proj.pytorch = pt
  hook = _mod("tl2.proj.pytorch.pytorch_hook"); pt.pytorch_hook = hook
  class VerboseModel:
    @staticmethod
    def forward_verbose(*a, **k): return None
  hook.VerboseModel = VerboseModel
  tu = _mod("tl2.proj.pytorch.torch_utils"); pt.torch_utils = tu
  tu.print_number_params = lambda *a, **k: None
  def requires_grad(model, flag=True):
    for p in model.parameters(): p.requires_grad_(flag)
  tu.requires_grad = requires_grad
  initf = _mod("tl2.proj.pytorch.init_func"); pt.init_func = initf
  def kaiming_leaky_init(m):
    # identical in-tree copy: exp/cips3d/models/multi_head_mapping.py:22-25
    if m.__class__.__name__.find("Linear") != -1:
      torch.nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
  initf.kaiming_leaky_init = kaiming_leaky_init
  for sub in ("ddp", "examples"):
    _mod("tl2.proj.pytorch." + sub)

  tlu = _mod("tl2.tl2_utils"); tl2.tl2_utils = tlu
  tlu.get_class_repr = lambda self: f"{self.__class__.__name__}({getattr(self, 'repr_str', '')})"
  tlu.dict2string = lambda dict_obj=None, **k: str(dict_obj)
  sg = _mod("tl2.proj.stylegan2_ada"); proj.stylegan2_ada = sg
  per = _mod("tl2.proj.stylegan2_ada.persistence"); sg.persistence = per
  per.persistent_class = lambda c: c
  _mod("tl2.proj.fvcore.checkpoint")
  _mod("tl2.proj.logger")
  modelarts = _mod("tl2.modelarts")

  ed = _mod("easydict"); ed.EasyDict = _Cfg
  _mod("streamlit")
  tv = _mod("torchvision"); tvt = _mod("torchvision.transforms"); tv.transforms = tvt
  tvf = _mod("torchvision.transforms.functional"); tvt.functional = tvf

  # exp.comm.op replacement (pure torch, double-differentiable by construction)
  import importlib
  importlib.import_module("exp.comm")
  op = _mod("exp.comm.op")
  class FusedLeakyReLU(torch.nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
      super().__init__()
      self.bias = torch.nn.Parameter(torch.zeros(channel))
      self.negative_slope = negative_slope
      self.scale = scale
    def forward(self, input):
      return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
  def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    # fused_bias_act_kernel.cu:36-47, act=3 (lrelu), grad=0
    rest = [1] * (input.ndim - bias.ndim - 1)
    return F.leaky_relu(input + bias.view(1, bias.shape[0], *rest), negative_slope) * scale
  def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    b, c, h, w = input.shape
    out = upfirdn2d_native(input.reshape(-1, h, w, 1), kernel, up, up, down, down,
                           pad[0], pad[1], pad[0], pad[1])
    return out.view(-1, c, out.shape[1], out.shape[2])
  op.FusedLeakyReLU = FusedLeakyReLU
  op.fused_leaky_relu = fused_leaky_relu
  op.upfirdn2d = upfirdn2d
  sys.modules["exp.comm"].op = op
