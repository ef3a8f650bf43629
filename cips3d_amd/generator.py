"""MI355X-native drop-in for the reference generator API (exp/cips3d/models/generator.py).

Same constructor / forward signatures, attribute names and state_dict layout as
`GeneratorNerfINR` (generator.py:1159-1951) and `GeneratorNerfINR_freeze_NeRF` (:1955-2083), so
checkpoints and `exp/cips3d` scripts work unchanged; the arithmetic runs in the hand-written
HIP kernels of libcips3d_hip.so (see ops.py / include/cips3d_hip.h).  Modules here only HOLD
parameters (same names, shapes and initialisers, created in the reference's order so that the
same torch seed yields the same initial weights) and orchestrate kernel launches.
"""
import math
import os
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# ------------------------------------------------------------------------------------------
# initialisers (film_layer.py:11-18, inr_network.py:20-27, tl2 init_func.kaiming_leaky_init)
# ------------------------------------------------------------------------------------------
def frequency_init(freq):
    def init(m):
        with torch.no_grad():
            if isinstance(m, nn.Linear):
                num_input = m.weight.size(-1)
                m.weight.uniform_(-np.sqrt(6 / num_input) / freq, np.sqrt(6 / num_input) / freq)
    return init


def kaiming_leaky_init(m):
    if m.__class__.__name__.find('Linear') != -1:
        torch.nn.init.kaiming_normal_(m.weight, a=0.2, mode='fan_in', nonlinearity='leaky_relu')


# ------------------------------------------------------------------------------------------
# parameter holders
# ------------------------------------------------------------------------------------------
class PixelNorm(nn.Module):
    """multi_head_mapping.py:13-19"""

    def forward(self, input):
        assert input.dim() == 2
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


class MultiHeadMappingNetwork(nn.Module):
    """z -> style MLP (multi_head_mapping.py:28-153).  Tiny (b x 512) GEMMs: stays on
    torch/rocBLAS; must remain differentiable and state-dict compatible."""

    def __init__(self, z_dim, hidden_dim, base_layers, head_layers, head_dim_dict,
                 add_norm=False, norm_out=False, **kwargs):
        super().__init__()
        self.z_dim = z_dim
        self.head_dim_dict = head_dim_dict
        out_dim = z_dim
        self.module_name_list = []
        self.norm = PixelNorm()
        base_net = []
        for i in range(base_layers):
            in_dim = out_dim
            out_dim = hidden_dim
            layer = nn.Linear(in_features=in_dim, out_features=out_dim)
            layer.apply(kaiming_leaky_init)
            base_net.append(layer)
            if head_layers > 0 or i != base_layers - 1:
                if add_norm:
                    base_net.append(nn.LayerNorm(out_dim))
                base_net.append(nn.LeakyReLU(0.2, inplace=True))
        if len(base_net) > 0:
            if norm_out and head_layers <= 0:
                base_net.append(nn.LayerNorm(out_dim))
            self.base_net = nn.Sequential(*base_net)
            self.num_z = 1
            self.module_name_list.append('base_net')
        else:
            self.base_net = None
            self.num_z = len(head_dim_dict)
        head_in_dim = out_dim
        for name, head_dim in head_dim_dict.items():
            if head_layers > 0:
                head_net = []
                out_dim = head_in_dim
                for i in range(head_layers):
                    in_dim = out_dim
                    out_dim = head_dim if i == head_layers - 1 else hidden_dim
                    hl = nn.Linear(in_features=in_dim, out_features=out_dim)
                    hl.apply(kaiming_leaky_init)
                    head_net.append(hl)
                    if i != head_layers - 1:
                        head_net.append(nn.LeakyReLU(0.2, inplace=True))
                    elif norm_out:
                        head_net.append(nn.LayerNorm(out_dim))
                head_net = nn.Sequential(*head_net)
                self.module_name_list.append(name)
            else:
                head_net = nn.Identity()
            self.add_module(name, head_net)

    def _base_hip(self, z):
        """base_net on the HIP kernels: PixelNorm, then per layer the Linear (grouped-linear kernel, one job) and one
        fused LayerNorm / LeakyReLU launch (cips_rownorm_*) instead of three torch modules"""
        x = ops.RowNormFunction.apply(z, None, None, 4)
        mods = list(self.base_net)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                x = ops.grouped_linear([(x, m)])[0]
                i += 1
                ln = mods[i] if i < len(mods) and isinstance(mods[i], nn.LayerNorm) else None
                if ln is not None:
                    i += 1
                act = i < len(mods) and isinstance(mods[i], nn.LeakyReLU)
                if act:
                    i += 1
                if ln is not None or act:
                    x = ops.RowNormFunction.apply(x, ln.weight if ln is not None else None, ln.bias if ln is not None else None,
                                                  (1 if ln is not None else 0) | (2 if act else 0))
            else:               # anything else the constructor could have put here
                x = m(x)
                i += 1
        return x

    def _hip_ok(self, z):
        # (large batches — the 10 000 latents of generate_avg_frequencies — stay on hipBLASLt: the row kernels are built for
        # the few rows of a training batch)
        return (z.is_cuda and z.dim() == 2 and z.dtype == torch.float32 and z.shape[0] <= 256 and z.shape[1] <= 512 and z.shape[1] % 4 == 0
                and all(not isinstance(m, nn.Linear) or (m.in_features <= 512 and m.in_features % 4 == 0 and m.out_features <= 1024)
                        for m in self.base_net))

    def forward(self, z):
        if self.base_net is not None:
            if self._hip_ok(z):
                base_fea = self._base_hip(z)
            else:
                z = self.norm(z)
                base_fea = self.base_net(z)
            head_inputs = {name: base_fea for name in self.head_dim_dict.keys()}
        else:
            head_inputs = {name: self.norm(z[idx]) for idx, name in enumerate(self.head_dim_dict.keys())}
        return {name: getattr(self, name)(head_inputs[name]) for name in self.head_dim_dict.keys()}


class FiLMLayer(nn.Module):
    """Parameters of film_layer.FiLMLayer (film_layer.py:41-107); the sine layer itself is
    evaluated inside the fused SIREN kernel."""

    def __init__(self, in_dim, out_dim, style_dim, use_style_fc=True, **kwargs):
        super().__init__()
        assert use_style_fc
        self.in_dim, self.out_dim, self.style_dim, self.use_style_fc = in_dim, out_dim, style_dim, use_style_fc
        self.linear = nn.Linear(in_dim, out_dim)
        self.linear.apply(frequency_init(25))
        self.gain_fc = nn.Linear(style_dim, out_dim)
        self.bias_fc = nn.Linear(style_dim, out_dim)
        self.gain_fc.weight.data.mul_(0.25)
        self.bias_fc.weight.data.mul_(0.25)

    def film(self, style):
        """gain = 15 * gain_fc(style) + 30 (LinearScale, film_layer.py:21-32, :59), bias = bias_fc(style)."""
        return self.gain_fc(style) * 15 + 30, self.bias_fc(style)


def _film_all(layers, styles):
    """FiLM vectors of several layers in one grouped launch: -> [(gain, bias), ...]"""
    ys = ops.grouped_linear([(st, m) for lay, st in zip(layers, styles) for m in (lay.gain_fc, lay.bias_fc)])
    return [(ys[2 * i] * 15 + 30, ys[2 * i + 1]) for i in range(len(layers))]


class NeRFNetwork(nn.Module):
    """generator.py:151-340.  forward() keeps the reference op boundary ((b,P,3) -> (b,P,33));
    the generator uses `evaluate()` which returns feat / sigma separately (no 33-wide cat)."""

    def __init__(self, in_dim=3, hidden_dim=256, hidden_layers=2, style_dim=512, rgb_dim=3, device=None,
                 name_prefix='nerf', **kwargs):
        super().__init__()
        if not (in_dim == 3 and rgb_dim == 32 and hidden_layers >= 1):
            raise NotImplementedError(
                "the ray set-up produces 3-vectors and the composite / CIPS head take 32 colour features "
                "(in_dim 3, rgb_dim 32: every shipped config, ffhq_exp.yaml:51-58)")
        # the fused SIREN kernels hold exactly the shipped matrices in LDS (hidden 128, two FiLM layers); any other width or depth
        # runs the same arithmetic as plain GPU tensor operations (hipBLASLt linears + elementwise sine, torch autograd) through
        # the unfused rays -> SIREN -> composite path: functional parity with the reference module, not its speed
        self.fused = hidden_dim == 128 and hidden_layers == 2
        self.device = device
        self.in_dim, self.hidden_dim, self.rgb_dim = in_dim, hidden_dim, rgb_dim
        self.style_dim, self.hidden_layers, self.name_prefix = style_dim, hidden_layers, name_prefix
        self.module_name_list = []
        self.style_dim_dict = {}
        self.network = nn.ModuleList()
        self.module_name_list.append('network')
        _out = in_dim
        for idx in range(hidden_layers):
            _in, _out = _out, hidden_dim
            layer = FiLMLayer(in_dim=_in, out_dim=_out, style_dim=style_dim, use_style_fc=True)
            self.network.append(layer)
            self.style_dim_dict[f'{name_prefix}_w{idx}'] = layer.style_dim
        self.final_layer = nn.Linear(hidden_dim, 1)
        self.module_name_list.append('final_layer')
        self.color_layer_sine = FiLMLayer(in_dim=hidden_dim, out_dim=hidden_dim // 2, style_dim=style_dim,
                                          use_style_fc=True)
        self.style_dim_dict[f'{name_prefix}_rgb'] = self.color_layer_sine.style_dim
        self.module_name_list.append('color_layer_sine')
        self.color_layer_linear = nn.Sequential(nn.Linear(hidden_dim // 2, rgb_dim))
        self.color_layer_linear.apply(kaiming_leaky_init)
        self.module_name_list.append('color_layer_linear')
        self.dim_styles = sum(self.style_dim_dict.values())

    def _evaluate_unfused(self, points, style_dict):
        """generator.py:260-317 for any hidden width / depth: UniformBoxWarp (x 2 / 0.24), FiLM sine layers
        (film_layer.py:78-107: sin(gain * linear(x) + bias)), sigma head, colour sine layer, colour linear."""
        if not points.is_cuda:
            raise RuntimeError("NeRFNetwork runs on the GPU only (there is no CPU path)")
        p = self.name_prefix
        x = points * (2.0 / 0.24)
        for idx, layer in enumerate(self.network):
            gain, bias = layer.film(style_dict[f'{p}_w{idx}'])
            x = torch.sin(gain.unsqueeze(1) * layer.linear(x) + bias.unsqueeze(1))
        sigma = self.final_layer(x).squeeze(-1)
        gain, bias = self.color_layer_sine.film(style_dict[f'{p}_rgb'])
        c = torch.sin(gain.unsqueeze(1) * self.color_layer_sine.linear(x) + bias.unsqueeze(1))
        return self.color_layer_linear(c), sigma

    def evaluate(self, points, style_dict):
        """points (b,P,3) -> feat (b,P,32), sigma (b,P) via the fused HIP kernel (shipped shape) or tensor operations."""
        if not self.fused:
            return self._evaluate_unfused(points, style_dict)
        p = self.name_prefix
        (g0, p0), (g1, p1), (gc, pc) = _film_all([self.network[0], self.network[1], self.color_layer_sine],
                                                 [style_dict[f'{p}_w0'], style_dict[f'{p}_w1'], style_dict[f'{p}_rgb']])
        return ops.SirenFunction.apply(
            points, g0, p0, g1, p1, gc, pc,
            self.network[0].linear.weight, self.network[0].linear.bias,
            self.network[1].linear.weight, self.network[1].linear.bias,
            self.final_layer.weight, self.final_layer.bias,
            self.color_layer_sine.linear.weight, self.color_layer_sine.linear.bias,
            self.color_layer_linear[0].weight, self.color_layer_linear[0].bias)

    def evaluate_rays(self, style_dict, geom, xg, yg, zg, cam2world, jitter=None, zvals=None):
        """evaluate() with the sample points generated in-kernel (coarse: from the jitter draw; fine: from the resampled
        depths `zvals`) -> feat (b,P,32), sigma (b,P), z (b,P)"""
        p = self.name_prefix
        (g0, p0), (g1, p1), (gc, pc) = _film_all([self.network[0], self.network[1], self.color_layer_sine],
                                                 [style_dict[f'{p}_w0'], style_dict[f'{p}_w1'], style_dict[f'{p}_rgb']])
        return ops.SirenRaysFunction.apply(
            geom, xg, yg, zg, cam2world, jitter, zvals, g0, p0, g1, p1, gc, pc,
            self.network[0].linear.weight, self.network[0].linear.bias,
            self.network[1].linear.weight, self.network[1].linear.bias,
            self.final_layer.weight, self.final_layer.bias,
            self.color_layer_sine.linear.weight, self.color_layer_sine.linear.bias,
            self.color_layer_linear[0].weight, self.color_layer_linear[0].bias)

    def march(self, style_dict, geom, xg, yg, zg, cam2world, jitter, noise):
        """fused rays + SIREN + composite for non-hierarchical sampling -> pixels_fea (b,n,32), depth (b,n)"""
        p = self.name_prefix
        (g0, p0), (g1, p1), (gc, pc) = _film_all([self.network[0], self.network[1], self.color_layer_sine],
                                                 [style_dict[f'{p}_w0'], style_dict[f'{p}_w1'], style_dict[f'{p}_rgb']])
        return ops.RayMarchFunction.apply(
            geom, xg, yg, zg, cam2world, jitter, noise, g0, p0, g1, p1, gc, pc,
            self.network[0].linear.weight, self.network[0].linear.bias,
            self.network[1].linear.weight, self.network[1].linear.bias,
            self.final_layer.weight, self.final_layer.bias,
            self.color_layer_sine.linear.weight, self.color_layer_sine.linear.bias,
            self.color_layer_linear[0].weight, self.color_layer_linear[0].bias)

    def forward(self, input, style_dict, ray_directions=None, **kwargs):
        feat, sigma = self.evaluate(input, style_dict)
        return torch.cat([feat, sigma.unsqueeze(-1)], dim=-1)

    forward_with_frequencies_phase_shifts = forward


class SinStyleMod(nn.Module):
    """Parameters of mod_conv_fc.SinStyleMod (mod_conv_fc.py:392-450).  `norm` exists in the
    reference state_dict but is never used in forward (:444-445)."""

    def __init__(self, in_channel, out_channel, kernel_size=1, style_dim=None, use_style_fc=False,
                 demodulate=True, eps=1e-8, **kwargs):
        super().__init__()
        assert kernel_size == 1 and use_style_fc and demodulate
        self.eps, self.in_channel, self.out_channel, self.style_dim = eps, in_channel, out_channel, style_dim
        self.weight = nn.Parameter(torch.randn(1, in_channel, out_channel))
        torch.nn.init.kaiming_normal_(self.weight[0], a=0.2, mode='fan_in', nonlinearity='leaky_relu')
        self.modulation = nn.Linear(style_dim, in_channel)
        self.modulation.apply(kaiming_leaky_init)
        self.norm = nn.LayerNorm(in_channel)


class SinBlock(nn.Module):
    """generator.py:893-980 (two modulated FCs + LeakyReLU(0.2), optional skip)."""

    def __init__(self, in_dim, out_dim, style_dim, name_prefix):
        super().__init__()
        self.in_dim, self.out_dim, self.style_dim, self.name_prefix = in_dim, out_dim, style_dim, name_prefix
        self.style_dim_dict = {}
        self.mod1 = SinStyleMod(in_channel=in_dim, out_channel=out_dim, style_dim=style_dim, use_style_fc=True)
        self.style_dim_dict[f'{name_prefix}_0'] = self.mod1.style_dim
        self.mod2 = SinStyleMod(in_channel=out_dim, out_channel=out_dim, style_dim=style_dim, use_style_fc=True)
        self.style_dim_dict[f'{name_prefix}_1'] = self.mod2.style_dim


class ToRGB(nn.Module):
    """generator.py:983-1006"""

    def __init__(self, in_dim, dim_rgb=3, use_equal_fc=False):
        super().__init__()
        assert not use_equal_fc
        self.in_dim, self.dim_rgb = in_dim, dim_rgb
        self.linear = nn.Linear(in_dim, dim_rgb)


class CIPSNet(nn.Module):
    """generator.py:1009-1154.  NOTE (reference behaviour): `points_forward` calls
    `self.inr_net(pixels_fea, style_dict)` WITHOUT img_size (generator.py:1754), so the default
    img_size=1024 applies and ALL nine blocks "4".."1024" run at every resolution."""

    def __init__(self, input_dim, style_dim, hidden_dim=256, pre_rgb_dim=32, device=None, name_prefix='inr',
                 **kwargs):
        super().__init__()
        if pre_rgb_dim != 3:
            raise NotImplementedError("shipped configs use pre_rgb_dim 3 (ffhq_exp.yaml:72)")
        self.device, self.pre_rgb_dim, self.name_prefix = device, pre_rgb_dim, name_prefix
        self.channels = {str(2 ** i): hidden_dim for i in range(2, 11)}
        self.module_name_list = []
        self.style_dim_dict = {}
        _out = input_dim
        network, to_rgbs = OrderedDict(), OrderedDict()
        for name, channel in self.channels.items():
            _in, _out = _out, channel
            blk = SinBlock(in_dim=_in, out_dim=_out, style_dim=style_dim, name_prefix=f'{name_prefix}_w{name}')
            self.style_dim_dict.update(blk.style_dim_dict)
            network[name] = blk
            to_rgbs[name] = ToRGB(in_dim=_out, dim_rgb=pre_rgb_dim, use_equal_fc=False)
        self.network = nn.ModuleDict(network)
        self.to_rgbs = nn.ModuleDict(to_rgbs)
        self.to_rgbs.apply(frequency_init(100))
        self.module_name_list += ['network', 'to_rgbs']
        self.tanh = nn.Sequential(nn.Tanh())
        self.module_name_list.append('tanh')

    def _names(self, img_size):
        img_size = str(2 ** int(np.log2(img_size)))
        names = []
        for name in self.network.keys():
            names.append(name)
            if name == img_size:
                break
        return names

    def _build_params(self, names, style_dict):
        # s = SinStyleMod.modulation(style) of all 18 layers (mod_conv_fc.py:474) in one grouped launch
        mods = ops.grouped_linear([(style_dict[f'{self.network[name].name_prefix}_{j}'], m.modulation)
                                   for name in names for j, m in enumerate((self.network[name].mod1, self.network[name].mod2))])
        params = []
        for k, name in enumerate(names):
            blk = self.network[name]
            # (1, in, out) -> (in, out) as a VIEW: indexing with [0] makes autograd materialise a zero-filled (1, in, out)
            # buffer and copy the gradient into it, 18 x (fill + 1 MiB copy) per step
            w1, w2 = blk.mod1.weight, blk.mod2.weight
            params += [w1.view(w1.shape[1], w1.shape[2]), mods[2 * k], w2.view(w2.shape[1], w2.shape[2]), mods[2 * k + 1]]
        for idx, name in enumerate(names):
            if idx >= 3:
                params += [self.to_rgbs[name].linear.weight, self.to_rgbs[name].linear.bias]
        return mods, params

    def open_tail_ports(self, style_dict, B, n, in0, img_size=1024, join=True):
        """The head's weight-gradient tail behind gradient ports on the side stream (ops.INR_TAIL), opened for the NEXT forward()
        with these styles and (B, n, in0) inputs.  Every autograd node between the ports and the parameters — the modulation
        Linears, the weight views — is created under that stream as well (a node on the caller's stream that consumed a port's
        output would make the caller's stream wait for the whole tail).  GeneratorNerfINR._render calls this BEFORE the ray march:
        nodes created earlier run later in the backward pass, so the NeRF backward is issued first and the tail starts behind
        its compositing kernel (ops._TAIL_GATE).  Returns True when ports are pending."""
        self._tail = None
        dev = next(self.parameters()).device
        names = self._names(img_size)
        if not (dev.type == "cuda" and B <= 64 and torch.is_grad_enabled() and ops.INR_TAIL == "side" and len(names) > 3):
            return False
        side = _side_stream(dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            mods, params = self._build_params(names, style_dict)
        for t in mods:
            t.record_stream(main)
        ok = ops.inr_head_ports_ok(len(names), B, n, in0, params, dev)
        ports = ops.inr_head_open_ports(len(names), B, n, params, side) if ok else None
        if join:
            main.wait_stream(side)
        # (not ok: the modulation Linears are kept all the same — forward() takes its parameters from here)
        self._tail = dict(key=(B, n, in0, len(names), id(style_dict)), params=params, ports=ports)
        return True

    def forward(self, input, style_dict, img_size=1024, **kwargs):
        names = self._names(img_size)
        tail, self._tail = getattr(self, "_tail", None), None
        key = (input.shape[0], input.shape[1], input.shape[2], len(names), id(style_dict))
        if (tail is None and torch.is_grad_enabled() and input.is_cuda and input.requires_grad
                and self.open_tail_ports(style_dict, *key[:3], img_size=img_size)):
            tail, self._tail = self._tail, None          # not opened ahead (a direct call): open them now
        if tail is not None and tail["key"] == key and torch.is_grad_enabled():
            if tail["ports"] is not None:
                rgb = ops.inr_head_with_ports(len(names), input, tail["params"], tail["ports"])
            else:
                rgb = ops.inr_head(len(names), input, *tail["params"])
        else:
            _, params = self._build_params(names, style_dict)
            rgb = ops.inr_head(len(names), input, *params)
        return self.tanh(rgb)


class _ToRGBFunction(torch.autograd.Function):
    """y (M,3) = x (M,K) @ w^T + b on the HIP ToRGB kernels (used for the aux 32->3 head)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.detach().contiguous().view(-1, x.shape[-1])
        w, b = w.detach().contiguous(), b.detach().contiguous()
        y = torch.empty(x2.shape[0], 3, device=x.device)
        ops.torgb_fwd(x2, w, b, y, accumulate=False)
        ctx.save_for_backward(x2, w)
        ctx.shape = x.shape
        return y.view(*x.shape[:-1], 3)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, 3)
        dw, db = ops.torgb_bwd_w(x2, dy2)
        dx = torch.empty_like(x2)
        ops.torgb_bwd_x(dy2, w, None, None, None, dx)
        return dx.view(ctx.shape), dw, db


# The INR mapping MLP runs on a side stream.  Measured on one box, C2: 17.15 -> 16.79 ms per step (graph replay), 17.22 ->
# 16.88 eager against everything on the caller's stream.
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


# ------------------------------------------------------------------------------------------
# camera helpers (O(batch) host-side math; comm_utils.py:451-581)
# ------------------------------------------------------------------------------------------
def _normalize(v):
    return v / torch.norm(v, dim=-1, keepdim=True)


def camera_origin_from_angles(theta, phi, r=1.0):
    """comm_utils.py:527-535 (phi already drawn; clamp + spherical -> cartesian)."""
    phi = torch.clamp(phi, 1e-5, math.pi - 1e-5)
    out = torch.zeros((theta.shape[0], 3), device=theta.device)
    out[:, 0:1] = r * torch.sin(phi) * torch.cos(theta)
    out[:, 2:3] = r * torch.sin(phi) * torch.sin(theta)
    out[:, 1:2] = r * torch.cos(phi)
    return out, phi


def _truncated_normal(shape, device):
    """comm_utils.py:441-448: per element the first of four standard-normal draws that lies in (-2, 2) (the first
    draw if none does)."""
    tmp = torch.randn(tuple(shape) + (4,), device=device)      # = empty().normal_(): the same draws
    ok = (tmp < 2) & (tmp > -2)
    first = ok.max(-1, keepdim=True)[1]
    return tmp.gather(-1, first).squeeze(-1)


def sample_camera_positions(device, bs=1, r=1, horizontal_stddev=1, vertical_stddev=1, horizontal_mean=math.pi * 0.5,
                            vertical_mean=math.pi * 0.5, mode='normal'):
    """comm_utils.py:451-535: camera origins on the sphere of radius r -> (origin (bs,3), phi = pitch (bs,1),
    theta = yaw (bs,1)), every distribution of the reference with its draws in the reference's order
    (yaw first; 'hybrid' flips Python's `random.random()` first; anything else is an assertion error)."""
    hs, vs, hm, vm = horizontal_stddev, vertical_stddev, horizontal_mean, vertical_mean
    u = lambda: torch.rand((bs, 1), device=device)
    g = lambda: torch.randn((bs, 1), device=device)
    if mode == 'uniform':
        theta = (u() - 0.5) * 2 * hs + hm
        phi = (u() - 0.5) * 2 * vs + vm
    elif mode in ('normal', 'gaussian'):
        theta = g() * hs + hm
        phi = g() * vs + vm
    elif mode == 'hybrid':
        if random.random() < 0.5:
            theta = (u() - 0.5) * 2 * hs * 2 + hm
            phi = (u() - 0.5) * 2 * vs * 2 + vm
        else:
            theta = g() * hs + hm
            phi = g() * vs + vm
    elif mode == 'truncated_gaussian':
        theta = _truncated_normal((bs, 1), device) * hs + hm
        phi = _truncated_normal((bs, 1), device) * vs + vm
    elif mode == 'spherical_uniform':
        theta = (u() - 0.5) * 2 * hs + hm
        v = (u() - 0.5) * 2 * (vs / math.pi) + vm / math.pi
        phi = torch.arccos(1 - 2 * torch.clamp(v, 1e-5, 1 - 1e-5))
    elif mode == 'mean':
        theta = torch.ones((bs, 1), device=device) * hm
        phi = torch.ones((bs, 1), device=device) * vm
    else:
        assert 0, f"camera distribution {mode!r}"
    origin, phi = camera_origin_from_angles(theta, phi, r)
    return origin, phi, theta


def create_cam2world_matrix(forward_vector, origin, up_vector=None):
    """comm_utils.py:538-581"""
    device = origin.device
    forward_vector = _normalize(forward_vector)
    if up_vector is None:
        # (0, 1, 0) built on the device — no host-to-device copy, so the step can be captured in a hipGraph
        up_vector = (torch.arange(3, device=device) == 1).to(torch.float).expand_as(forward_vector)
    left_vector = _normalize(torch.cross(up_vector, forward_vector, dim=-1))
    up_vector = _normalize(torch.cross(forward_vector, left_vector, dim=-1))
    rot = torch.eye(4, device=device).unsqueeze(0).repeat(forward_vector.shape[0], 1, 1)
    rot[:, :3, :3] = torch.stack((-left_vector, up_vector, -forward_vector), axis=-1)
    trans = torch.eye(4, device=device).unsqueeze(0).repeat(forward_vector.shape[0], 1, 1)
    trans[:, :3, 3] = origin
    return trans @ rot


# ------------------------------------------------------------------------------------------
# generator
# ------------------------------------------------------------------------------------------
class GeneratorNerfINR(nn.Module):
    """Drop-in for exp.cips3d.models.generator.GeneratorNerfINR."""

    def __init__(self, z_dim, nerf_cfg, mapping_nerf_cfg, inr_cfg, mapping_inr_cfg, device='cuda', **kwargs):
        super().__init__()
        self.epoch = 0
        self.step = 0
        self.z_dim = z_dim
        self.device = device
        self.module_name_list = []
        self.siren = NeRFNetwork(**nerf_cfg)
        self.module_name_list.append('siren')
        self.mapping_network_nerf = MultiHeadMappingNetwork(
            **{**mapping_nerf_cfg, 'head_dim_dict': self.siren.style_dim_dict})
        self.module_name_list.append('mapping_network_nerf')
        self.inr_net = CIPSNet(**{**inr_cfg, "input_dim": self.siren.rgb_dim})
        self.module_name_list.append('inr_net')
        self.mapping_network_inr = MultiHeadMappingNetwork(
            **{**mapping_inr_cfg, 'head_dim_dict': self.inr_net.style_dim_dict})
        self.module_name_list.append('mapping_network_inr')
        self.aux_to_rbg = nn.Sequential(nn.Linear(self.siren.rgb_dim, 3), nn.Tanh())
        self.aux_to_rbg.apply(frequency_init(25))
        self.module_name_list.append('aux_to_rbg')
        self.filters = nn.Identity()

    # ---- latent / style plumbing (generator.py:1764-1826) ----
    def z_sampler(self, shape, device, dist='gaussian'):
        if dist == 'gaussian':
            return torch.randn(shape, device=device)
        return torch.rand(shape, device=device) * 2 - 1

    def get_zs(self, b, batch_split=1):
        z_nerf = self.z_sampler(shape=(b, self.mapping_network_nerf.z_dim), device=self.device)
        z_inr = self.z_sampler(shape=(b, self.mapping_network_inr.z_dim), device=self.device)
        if batch_split > 1:
            return [{'z_nerf': a, 'z_inr': c} for a, c in
                    zip(z_nerf.split(b // batch_split), z_inr.split(b // batch_split))]
        return {'z_nerf': z_nerf, 'z_inr': z_inr}

    def mapping_network(self, z_nerf, z_inr, defer_join=False):
        style_dict = {}
        if z_inr.is_cuda and z_inr.shape[0] <= 256:
            # the two z -> style MLPs are independent chains of latency-bound launches: the INR one runs on a side stream —
            # its forward next to the NeRF mapping (and, with defer_join, next to the ray march: forward() joins right
            # before the INR head, the first consumer of its styles), its backward (autograd keeps a node on its forward's
            # stream) next to the NeRF path's backward.  Fork / join by stream waits, so a captured step records it as
            # parallel branches.
            main = torch.cuda.current_stream(z_inr.device)
            side = _side_stream(z_inr.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                inr = self.mapping_network_inr(z_inr)
            z_inr.record_stream(side)
            style_dict.update(self.mapping_network_nerf(z_nerf))
            for t in inr.values():
                t.record_stream(main)
            style_dict.update(inr)
            self._pending_side = side
            if not defer_join:
                self._join_side()
            return style_dict
        style_dict.update(self.mapping_network_nerf(z_nerf))
        style_dict.update(self.mapping_network_inr(z_inr))
        return style_dict

    def _join_side(self):
        """the caller's stream waits for the INR mapping MLP (no-op when nothing is pending)"""
        side = getattr(self, "_pending_side", None)
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)
            self._pending_side = None

    def generate_avg_frequencies(self, num_samples=10000, device='cuda'):
        zs = self.get_zs(num_samples)
        with torch.no_grad():
            style_dict = self.mapping_network(**zs)
        self.avg_styles = {name: style.mean(0, keepdim=True) for name, style in style_dict.items()}
        return self.avg_styles

    def get_truncated_freq_phase(self, raw_style_dict, avg_style_dict, raw_lambda):
        """generator_nerf_inr.py:770-782"""
        return {name: avg_style_dict[name].lerp(raw, raw_lambda) for name, raw in raw_style_dict.items()}

    def set_device(self, device):
        pass

    def staged_forward(self, *args, **kwargs):
        raise NotImplementedError        # as in the reference (generator.py:1819-1820: the same two lines)

    # ---- the hot path ----
    def _nerf_styles(self, style_dict):
        return style_dict

    def _render(self, style_dict, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean,
                v_mean, hierarchical_sample, sample_dist, clamp_mode, nerf_noise, white_back, last_back,
                return_aux_img, forward_points=None, camera_pos=None, camera_lookup=None, up_vector=None,
                nerf_grad=True, rand_override=None, grad_points=None):
        """whole_grad_forward / part_grad_forward + points_forward (generator.py:1378-1657, 1659-1762) on the HIP path.

        `grad_points` (a count of pixels < img_size^2): part_grad_forward — after the ray set-up a `randperm(n)`
        splits the pixels of every image into a subset rendered with gradients and a rest rendered under
        no_grad (each with its own noise draws, in that order), and the two are scattered back (int64
        bookkeeping of comm_utils.py:240-282).

        Random tensors are drawn with the reference's calls, shapes and order
        so that a same-device, same-seed run consumes the generator identically; `rand_override`
        (dict with any of jitter/theta/phi/noise_c/u/noise_f) injects fixed draws for parity tests.
        With `forward_points` the reference evaluates image by image in chunks under no_grad; the
        fused kernels need no chunking, so only the per-image/per-chunk draw order is reproduced."""
        ro = rand_override or {}
        device = next(self.parameters()).device
        b = list(style_dict.values())[0].shape[0]
        H = W = img_size
        n = H * W
        S = num_steps
        E = 2 * S if hierarchical_sample else S
        clamp = ops._CLAMP[clamp_mode]
        flags = (1 if last_back else 0) | (2 if white_back else 0)
        staged = forward_points is not None
        part = grad_points is not None and grad_points < n
        if part:
            staged = False          # generator.py:1325-1347: part_grad_forward is not handed forward_points
        if not part and not staged and nerf_grad and torch.is_grad_enabled():
            # the INR head's gradient ports, opened before the NeRF path so that its backward is issued first
            # (CIPSNet.open_tail_ports); they sit on the INR mapping network's side stream, joined right before the head.
            # Only where a NeRF backward follows the head's: with a frozen NeRF there is nothing to run beside, and the
            # side-stream form measured 0.4 ms slower than the plain one at the r256 stages (profiles/r6_tail_ab.txt)
            if self.inr_net.open_tail_ports(style_dict, b, n, 32, join=False):
                self._pending_side = _side_stream(device)

        # ---------------- random draws in reference order ----------------
        def draw(kind, fn, shape, override=True):
            t = fn(shape, device=device)
            return ro[kind].to(device).reshape(shape).float() if (override and kind in ro) else t

        need_cam = camera_pos is None or camera_lookup is None
        mode = sample_dist
        # the two distributions the training configs use keep their raw draws separate (rand_override can inject
        # them); the others go through sample_camera_positions as a whole
        simple_cam = mode in ('gaussian', 'normal', 'uniform')
        assert not need_cam or simple_cam or mode in ('hybrid', 'truncated_gaussian', 'spherical_uniform', 'mean'), \
            f"camera distribution {mode!r}"          # comm_utils.py:526 (`assert 0`), incl. the default None

        def cam_angles(th_raw, ph_raw):
            if mode == 'uniform':
                return (th_raw - 0.5) * 2 * h_stddev + h_mean, (ph_raw - 0.5) * 2 * v_stddev + v_mean
            return th_raw * h_stddev + h_mean, ph_raw * v_stddev + v_mean

        def draw_cam(bs, override=True):
            """-> the RAW draws (b,1) x 2, or for the other distributions the finished (theta, phi)"""
            if not simple_cam:
                _, ph, th = sample_camera_positions(device, bs, 1, h_stddev, v_stddev, h_mean, v_mean, mode)
                return th, ph
            fn = torch.rand if mode == 'uniform' else torch.randn
            return draw('theta', fn, (bs, 1), override), draw('phi', fn, (bs, 1), override)

        if part:
            jitter = draw('jitter', torch.rand, (b, n, S, 1))
            th_raw, ph_raw = draw_cam(b) if need_cam else (None, None)
            noise_c = u = noise_f = None          # drawn per pixel subset below, after the randperm
        elif not staged:
            jitter = draw('jitter', torch.rand, (b, n, S, 1))
            th_raw, ph_raw = draw_cam(b) if need_cam else (None, None)
            noise_c = draw('noise_c', torch.randn, (b, n, S, 1)) if hierarchical_sample else None
            u = draw('u', torch.rand, (b * n, S)) if hierarchical_sample else None
            noise_f = draw('noise_f', torch.randn, (b, n, E, 1))
        else:
            js, ths, phs, ncs, us, nfs = [], [], [], [], [], []
            for _ in range(b):
                js.append(torch.rand((1, n, S, 1), device=device))
                if need_cam:
                    th, ph = draw_cam(1, override=False)
                    ths.append(th); phs.append(ph)
                head = 0
                while head < n:
                    c = min(forward_points, n - head)
                    if hierarchical_sample:
                        ncs.append(torch.randn((1, c, S, 1), device=device))
                        us.append(torch.rand((c, S), device=device))
                    nfs.append(torch.randn((1, c, E, 1), device=device))
                    head += forward_points
            jitter = ro.get('jitter', torch.cat(js, 0))
            th_raw = ro.get('theta', torch.cat(ths, 0)) if need_cam else None
            ph_raw = ro.get('phi', torch.cat(phs, 0)) if need_cam else None
            noise_c = ro.get('noise_c', torch.cat(ncs, 1).view(b, n, S, 1)) if hierarchical_sample else None
            u = ro.get('u', torch.cat(us, 0)) if hierarchical_sample else None
            noise_f = ro.get('noise_f', torch.cat(nfs, 1).view(b, n, E, 1))

        # ---------------- camera (O(b) host math) ----------------
        with torch.no_grad():
            pitch_yaw_fused = None
            if need_cam and simple_cam and th_raw.is_cuda and not (staged and up_vector is not None):
                # draws -> pitch, yaw, origin, cam2world in one launch (the ~45 one-wave torch kernels of the op-by-op form
                # below are 0.2 ms of a captured step)
                pitch_yaw_fused, origin, cam2world = ops.camera_pose(th_raw, ph_raw, mode == 'uniform', h_stddev, h_mean,
                                                                     v_stddev, v_mean)
                pitch, yaw = pitch_yaw_fused[:, 0:1], pitch_yaw_fused[:, 1:2]
            else:
                if need_cam:
                    theta, phi = cam_angles(th_raw, ph_raw) if simple_cam else (th_raw, ph_raw)
                    origin, pitch = camera_origin_from_angles(theta, phi)
                    yaw = theta
                    forward_vector = _normalize(-origin)
                else:
                    origin = camera_pos
                    pitch = yaw = torch.zeros(b, 1, device=device)
                    forward_vector = _normalize(camera_lookup)
                # reference quirk kept: only the staged branch of whole_grad_forward hands `up_vector` on
                # (generator.py:1437 vs :1481-1497); the one-shot branch always uses (0, 1, 0)
                cam2world = create_cam2world_matrix(forward_vector, origin, up_vector=up_vector if staged else None)
            xg, yg, zg = ops.pixel_grids(W, H, S, ray_start, ray_end, device)
            zc = float((-torch.ones(1) / np.tan((2 * math.pi * fov / 360) / 2)).item())
            # non-hierarchical sampling of whole images: rays + SIREN + composite fused in one kernel that walks the
            # samples along each ray (ops.RayMarchFunction); no (b,n,S,3) points, no per-sample features in HBM
            fused = (not hierarchical_sample) and (not part) and ops.march_available() and self.siren.fused
            # hierarchical sampling of whole images: both SIREN passes and the resampler regenerate rays / points
            # in-kernel (no rays kernel, no (b,n,S,3) point tensors for either pass)
            gen_rays = hierarchical_sample and (not part) and ops.march_available() and self.siren.fused
            if not fused and not gen_rays:
                points, z_vals, dirs = ops.rays_fwd(xg, yg, zg, zc, cam2world, jitter.reshape(b, n, S), b, H, W, S)
            ray_origins = origin if pitch_yaw_fused is not None else cam2world[:, :3, 3].contiguous()       # every ray starts at the camera

        nerf_styles = self._nerf_styles(style_dict)
        def pipeline(points, z_vals, dirs, n, noise_c, u, noise_f, nerf_grad):
            """points_forward (generator.py:1659-1762) for n rays per image: -> inr rgb (b,n,3), aux rgb or None"""
            ctx_nerf = torch.enable_grad() if nerf_grad else torch.no_grad()
            with ctx_nerf:
                if gen_rays:
                    rgeom = (b, H, W, S, zc)
                    jit3 = jitter.reshape(b, n, S)
                    feat_c, sig_c, z_c = self.siren.evaluate_rays(nerf_styles, rgeom, xg, yg, zg, cam2world, jitter=jit3)
                else:
                    feat_c, sig_c = self.siren.evaluate(points.reshape(b, n * S, 3), nerf_styles)
                    z_c = z_vals
                feat_c = feat_c.view(b * n, S, 32)
                sig_c = sig_c.view(b * n, S)
                z_c = z_c.reshape(b * n, S)
                if hierarchical_sample:
                    with torch.no_grad():
                        rp = ops._ray_params(xg, yg, zg, zc, cam2world, None, H, W, S) if gen_rays else None
                        fine_z, fine_pts = ops.resample_fwd(
                            sig_c, z_c, noise_c.reshape(b * n, S) if nerf_noise != 0 else None, nerf_noise,
                            u, ray_origins, dirs.reshape(b * n, 3) if dirs is not None else None, b, n, S, clamp, rays=rp)
                        if ops.FINE_Z_REC is not None:
                            ops.FINE_Z_REC.append(fine_z.detach().clone())
                        if ops.FINE_Z_PIN is not None:       # parity tests: the reference's sample placement
                            fine_z = next(ops.FINE_Z_PIN).to(fine_z.device).reshape(fine_z.shape).contiguous()
                            if not gen_rays:             # materialised points: origin + direction * depth (generator_nerf_inr.py:537-598)
                                fine_pts = (ray_origins.view(b, 1, 1, 3) + dirs.reshape(b, n, 1, 3) * fine_z.view(b, n, S, 1)).reshape(b, n * S, 3).contiguous()
                    if gen_rays:
                        feat_f, sig_f, _ = self.siren.evaluate_rays(nerf_styles, rgeom, xg, yg, zg, cam2world,
                                                                    zvals=fine_z.view(b, n * S))
                    else:
                        feat_f, sig_f = self.siren.evaluate(fine_pts.view(b, n * S, 3), nerf_styles)
                    feat_f = feat_f.view(b * n, S, 32)
                    sig_f = sig_f.view(b * n, S)
                else:
                    feat_f = sig_f = fine_z = None
                pixels_fea, depth, weights, order, zsorted = ops.CompositeFunction.apply(
                    feat_c, sig_c, z_c, feat_f, sig_f, fine_z,
                    noise_f.reshape(b * n, E) if nerf_noise != 0 else None, nerf_noise, clamp, flags)
                pixels_fea = pixels_fea.view(b, n, 32)
                if return_aux_img:
                    aux = torch.tanh(_ToRGBFunction.apply(pixels_fea, self.aux_to_rbg[0].weight,
                                                          self.aux_to_rbg[0].bias))
                else:
                    aux = None
            if not nerf_grad:
                pixels_fea = pixels_fea.detach()
            self._join_side()
            return self.inr_net(pixels_fea, style_dict), aux

        if fused:
            ctx_nerf = torch.enable_grad() if nerf_grad else torch.no_grad()
            with ctx_nerf:
                geom = (b, H, W, S, zc, float(nerf_noise), clamp, flags, torch.is_grad_enabled())
                pixels_fea, _depth = self.siren.march(nerf_styles, geom, xg, yg, zg, cam2world, jitter.reshape(b, n, S),
                                                      noise_f.reshape(b, n, S) if nerf_noise != 0 else None)
                aux_img = torch.tanh(_ToRGBFunction.apply(pixels_fea, self.aux_to_rbg[0].weight,
                                                          self.aux_to_rbg[0].bias)) if return_aux_img else None
            if not nerf_grad:
                pixels_fea = pixels_fea.detach()
            self._join_side()
            inr_img = self.inr_net(pixels_fea, style_dict)
        elif gen_rays:
            inr_img, aux_img = pipeline(None, None, None, n, noise_c, u, noise_f, nerf_grad)
        elif not part:
            inr_img, aux_img = pipeline(points, z_vals, dirs, n, noise_c, u, noise_f, nerf_grad)
        else:
            # ---- part_grad_forward (generator.py:1536-1657) ----
            rand_idx = ro['rand_idx'].to(device) if 'rand_idx' in ro else torch.randperm(n, device=device)
            idx_grad, idx_rest = rand_idx[:grad_points], rand_idx[grad_points:]
            pts4, z3, dirs3 = points.view(b, n, S, 3), z_vals.view(b, n, S), dirs.view(b, n, 3)

            def subset(idx, tag, with_grad):
                m = idx.numel()
                nc = draw('noise_c' + tag, torch.randn, (b, m, S, 1)) if hierarchical_sample else None
                uu = draw('u' + tag, torch.rand, (b * m, S)) if hierarchical_sample else None
                nf = draw('noise_f' + tag, torch.randn, (b, m, E, 1))
                return pipeline(pts4.index_select(1, idx).contiguous(), z3.index_select(1, idx).contiguous(),
                                dirs3.index_select(1, idx).contiguous(), m, nc, uu, nf, with_grad)

            inr_g, aux_g = subset(idx_grad, '_grad', nerf_grad)
            with torch.no_grad():
                inr_r, aux_r = subset(idx_rest, '_rest', False)

            def scatter(pg, pr):       # comm_utils.scatter_points: rows idx_grad <- pg (with grad), idx_rest <- pr
                out = torch.zeros(b, n, pg.shape[-1], device=device, dtype=pg.dtype)
                out = out.index_copy(1, idx_grad, pg)
                return out.index_copy(1, idx_rest, pr)

            inr_img = scatter(inr_g, inr_r)
            aux_img = scatter(aux_g, aux_r) if return_aux_img else None

        inr_img = inr_img.view(b, H, W, 3).permute(0, 3, 1, 2)
        inr_img = self.filters(inr_img)
        pitch_yaw = pitch_yaw_fused if pitch_yaw_fused is not None else torch.cat([pitch, yaw], -1)
        if return_aux_img:
            aux_img = aux_img.view(b, H, W, 3).permute(0, 3, 1, 2)
            imgs = torch.cat([inr_img, aux_img])
            pitch_yaw = torch.cat([pitch_yaw, pitch_yaw])
        else:
            imgs = inr_img.contiguous()
        return imgs, pitch_yaw

    def forward(self, zs, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, hierarchical_sample,
                h_mean=math.pi * 0.5, v_mean=math.pi * 0.5, psi=1, sample_dist=None, lock_view_dependence=False,
                clamp_mode='relu', nerf_noise=0., white_back=False, last_back=False, return_aux_img=False,
                grad_points=None, forward_points=None, **kwargs):
        """generator.py:1256-1370.  Returns (imgs (b or 2b,3,H,W), pitch_yaw (b or 2b,2))."""
        style_dict = self.mapping_network(**zs, defer_join=not psi < 1)     # joined right before the INR head (_render)
        if psi < 1:
            avg_styles = self.generate_avg_frequencies(device=self.device)
            style_dict = self.get_truncated_freq_phase(raw_style_dict=style_dict, avg_style_dict=avg_styles,
                                                       raw_lambda=psi)
        part = grad_points if (grad_points is not None and grad_points < img_size ** 2) else None
        return self._forward_styles(style_dict, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                    h_mean, v_mean, hierarchical_sample, sample_dist, clamp_mode, nerf_noise,
                                    white_back, last_back, return_aux_img, forward_points,
                                    rand_override=kwargs.get('rand_override'), grad_points=part)

    def _forward_styles(self, style_dict, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                        h_mean, v_mean, hierarchical_sample, sample_dist, clamp_mode, nerf_noise, white_back,
                        last_back, return_aux_img, forward_points, rand_override=None, grad_points=None, **cam):
        try:
            part = grad_points is not None and grad_points < img_size ** 2       # generator.py:1325: part_grad_forward, forward_points unused
            if forward_points is not None and not part:
                with torch.no_grad():
                    return self._render(style_dict, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                        h_mean, v_mean, hierarchical_sample, sample_dist, clamp_mode, nerf_noise,
                                        white_back, last_back, return_aux_img, forward_points=forward_points,
                                        rand_override=rand_override, **cam)
            return self._render(style_dict, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev, h_mean,
                                v_mean, hierarchical_sample, sample_dist, clamp_mode, nerf_noise, white_back, last_back,
                                return_aux_img, rand_override=rand_override, grad_points=grad_points, **cam)
        finally:
            # a deferred INR-mapping side stream is joined on EVERY exit (no-op after _render's own join): an exception
            # before the INR head must not leave the fork open — the main stream would never wait for it, and a
            # hipGraph capture would end with an unjoined stream
            self._join_side()

    def forward_camera_pos_and_lookup(self, zs, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                      h_mean, v_mean, hierarchical_sample, camera_pos, camera_lookup, psi=1,
                                      sample_dist=None, lock_view_dependence=False, clamp_mode='relu',
                                      nerf_noise=0., white_back=False, last_back=False, return_aux_img=False,
                                      grad_points=None, forward_points=None, up_vector=None, **kwargs):
        """generator.py:1828-1951 (explicit camera; pitch/yaw are zeros)."""
        style_dict = self.mapping_network(**zs, defer_join=not psi < 1)     # joined right before the INR head (_render)
        if psi < 1:
            avg_styles = self.generate_avg_frequencies(device=self.device)
            style_dict = self.get_truncated_freq_phase(raw_style_dict=style_dict, avg_style_dict=avg_styles,
                                                       raw_lambda=psi)
        part = grad_points if (grad_points is not None and grad_points < img_size ** 2) else None
        return self._forward_styles(style_dict, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                                    h_mean, v_mean, hierarchical_sample, sample_dist, clamp_mode, nerf_noise,
                                    white_back, last_back, return_aux_img, forward_points,
                                    rand_override=kwargs.get('rand_override'), grad_points=part,
                                    camera_pos=camera_pos, camera_lookup=camera_lookup, up_vector=up_vector)


class GeneratorNerfINR_freeze_NeRF(GeneratorNerfINR):
    """generator.py:1955-2083: mapping_nerf, both SIREN passes, the composite and aux_to_rbg run
    under no_grad; gradients flow only through the CIPS INR head and mapping_inr."""

    def load_nerf_ema(self, G_ema):
        self.siren.load_state_dict(G_ema.siren.state_dict())
        self.mapping_network_nerf.load_state_dict(G_ema.mapping_network_nerf.state_dict())
        self.aux_to_rbg.load_state_dict(G_ema.aux_to_rbg.state_dict())

    def mapping_network(self, z_nerf, z_inr, defer_join=False):
        style_dict = {}
        if z_inr.is_cuda and z_inr.shape[0] <= 256:       # see GeneratorNerfINR.mapping_network
            main = torch.cuda.current_stream(z_inr.device)
            side = _side_stream(z_inr.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                inr = self.mapping_network_inr(z_inr)
            z_inr.record_stream(side)
            with torch.no_grad():
                style_dict.update(self.mapping_network_nerf(z_nerf))
            for t in inr.values():
                t.record_stream(main)
            style_dict.update(inr)
            self._pending_side = side
            if not defer_join:
                self._join_side()
            return style_dict
        with torch.no_grad():
            style_dict.update(self.mapping_network_nerf(z_nerf))
        style_dict.update(self.mapping_network_inr(z_inr))
        return style_dict

    def _render(self, *args, **kwargs):
        kwargs['nerf_grad'] = False
        return super()._render(*args, **kwargs)
