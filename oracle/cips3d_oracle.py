"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU (torch fp32) restatement of the CIPS-3D hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; it
is the checker, never the product (the product path is cips3d_amd/ over libcips3d_hip.so and
fails loudly without the HIP extension).

Every function restates — in a flat, functional form over a plain state_dict — the algorithm
of the reference code it cites (paths relative to the reference repo).  The restatement is
PINNED: oracle/make_golden.py runs the unmodified reference (imported through
oracle/ref_shim.py in the build container) on seeded inputs with every random draw captured,
and tests/test_oracle_golden.py checks this file against those committed fixtures
(tests/golden/*.pt) to ~1e-6.

All random tensors are explicit inputs (`rand` dict) in the reference's draw order
(SURVEY.md §8a): jitter rand(b,n,S,1) comm_utils.py:432 | theta randn(b,1) :485 | phi randn(b,1)
:488 | noise_c randn(b,n,S,1) pigan_utils.py:246 via generator_nerf_inr.py:564 | u rand(b*n,S)
pigan_utils.py:192 | noise_f randn(b,n,E,1) pigan_utils.py:246 via generator.py:1744.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU = 0.2


# ----------------------------------------------------------------------------------------
# LeakyReLU gate tape (test instrumentation).  LeakyReLU is the only discontinuity on the gradient
# path that sees ~1e6+ activations per call (nn.LeakyReLU in SinBlock, generator.py:922/936;
# FusedLeakyReLU in the discriminator, fused_act.py:51-86).  A pre-activation within rounding of 0 may
# take either branch in two fp32 evaluations, so gradient parity is only well-defined for a GIVEN set
# of gates.  The tape records the gates (pre-activation > 0) of every such LeakyReLU in call order, or
# replays ("pins") gates recorded elsewhere — from the reference run (tests/golden/gates_*.pt) or from
# the HIP path — making this restatement evaluate the same piecewise-linear branch.
# ----------------------------------------------------------------------------------------
class GateTape:
    def __init__(self, pin=None):
        """pin=None: record; else an iterable of bool tensors replayed in call order."""
        self.rec = []
        self.preact = []           # |pre-activation| kept alongside when recording (ambiguity of a gate)
        self._pin = iter(pin) if pin is not None else None
        self.keep_preact = False

    def lrelu(self, y, slope):
        if self._pin is None:
            self.rec.append((y > 0).detach())
            if self.keep_preact:
                self.preact.append(y.detach().abs())
            return F.leaky_relu(y, slope)
        g = next(self._pin).to(y.device).reshape(y.shape)
        self.rec.append(g)
        if self.keep_preact:
            self.preact.append(y.detach().abs())
        return y * torch.where(g, torch.ones((), dtype=y.dtype), torch.full((), slope, dtype=y.dtype))

    def done(self):
        assert self._pin is None or next(self._pin, None) is None, "pinned gates left over"


_TAPE = [None]


class gate_tape:
    """with gate_tape(tape): every head / discriminator LeakyReLU below goes through `tape`."""

    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.old = _TAPE[0]
        _TAPE[0] = self.tape
        return self.tape

    def __exit__(self, *exc):
        _TAPE[0] = self.old


def _gated_lrelu(y, slope=LRELU):
    return F.leaky_relu(y, slope) if _TAPE[0] is None else _TAPE[0].lrelu(y, slope)


# ----------------------------------------------------------------------------------------
# mapping networks — exp/cips3d/models/multi_head_mapping.py:13-19 (PixelNorm), :130-153
# ----------------------------------------------------------------------------------------
def pixel_norm(z):
    return z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)


def mapping_nerf(sd, z, prefix="mapping_network_nerf.base_net."):
    """4 x Linear(., 128) with LeakyReLU(0.2) between (ffhq_exp.yaml:59-64: head_layers 0)."""
    x = pixel_norm(z)
    idxs = [0, 2, 4, 6]
    for j, i in enumerate(idxs):
        x = F.linear(x, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"])
        if j != len(idxs) - 1:
            x = F.leaky_relu(x, LRELU)
    return x


def mapping_inr(sd, z, prefix="mapping_network_inr.base_net."):
    """8 x (Linear, LayerNorm, LeakyReLU); the last is Linear + LayerNorm (add_norm, norm_out;
    ffhq_exp.yaml:73-81, multi_head_mapping.py:62-84)."""
    x = pixel_norm(z)
    for j in range(8):
        i = 3 * j
        x = F.linear(x, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"])
        x = F.layer_norm(x, (x.shape[-1],), sd[f"{prefix}{i + 1}.weight"], sd[f"{prefix}{i + 1}.bias"])
        if j != 7:
            x = F.leaky_relu(x, LRELU)
    return x


# ----------------------------------------------------------------------------------------
# H1 rays — exp/comm/comm_utils.py:365-412, 416-438, 451-535, 538-581, 584-679
# ----------------------------------------------------------------------------------------
def _unit(v):
    return v / torch.norm(v, dim=-1, keepdim=True)


def rays(b, img_size, fov, ray_start, ray_end, S, jitter, theta_n, phi_n, h_stddev, v_stddev,
         h_mean=math.pi * 0.5, v_mean=math.pi * 0.5, camera_pos=None, camera_lookup=None, up_vector=None):
    """-> points (b,n,S,3) world, z (b,n,S,1), dirs (b,n,3), origins (b,n,3), pitch, yaw.
    jitter = rand(b,n,S,1); theta_n/phi_n = the raw randn(b,1) draws ('gaussian' camera).
    With camera_pos / camera_lookup (comm_utils.py:626-641) the camera is explicit and pitch = yaw = 0."""
    W = H = img_size
    gx, gy = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="ij")
    x = gx.T.flatten()
    y = gy.T.flatten()
    zc = -torch.ones_like(x) / np.tan((2 * math.pi * fov / 360) / 2)
    d_cam = _unit(torch.stack([x, y, zc], -1))                                   # (n,3)
    z = torch.linspace(ray_start, ray_end, S).reshape(1, S, 1).repeat(W * H, 1, 1)  # (n,S,1)
    pts = d_cam.unsqueeze(1).repeat(1, S, 1) * z
    pts = torch.stack(b * [pts]); z = torch.stack(b * [z]); d_cam = torch.stack(b * [d_cam])
    # stratified jitter (perturb_points)
    step = z[:, :, 1:2, :] - z[:, :, 0:1, :]
    off = (jitter - 0.5) * step
    z = z + off
    pts = pts + off * d_cam.unsqueeze(2)
    # camera on the unit sphere (sample_camera_positions, mode gaussian)
    if camera_pos is None or camera_lookup is None:
        theta = theta_n * h_stddev + h_mean
        phi = torch.clamp(phi_n * v_stddev + v_mean, 1e-5, math.pi - 1e-5)
        o = torch.zeros(b, 3)
        o[:, 0:1] = torch.sin(phi) * torch.cos(theta)
        o[:, 2:3] = torch.sin(phi) * torch.sin(theta)
        o[:, 1:2] = torch.cos(phi)
        fwd = _unit(-o)
    else:
        o = camera_pos
        theta = phi = torch.zeros(b, 1)
        fwd = _unit(camera_lookup)
    up0 = (torch.tensor([0., 1., 0.]) if up_vector is None else up_vector).expand_as(fwd)
    left = _unit(torch.cross(up0, fwd, dim=-1))
    up = _unit(torch.cross(fwd, left, dim=-1))
    rot = torch.eye(4).unsqueeze(0).repeat(b, 1, 1)
    rot[:, :3, :3] = torch.stack((-left, up, -fwd), axis=-1)
    tr = torch.eye(4).unsqueeze(0).repeat(b, 1, 1)
    tr[:, :3, 3] = o
    c2w = tr @ rot
    n = W * H
    ph = torch.ones(b, n, S, 4)
    ph[..., :3] = pts
    wpts = torch.bmm(c2w, ph.reshape(b, -1, 4).permute(0, 2, 1)).permute(0, 2, 1).reshape(b, n, S, 4)[..., :3]
    wdir = torch.bmm(c2w[..., :3, :3], d_cam.reshape(b, -1, 3).permute(0, 2, 1)).permute(0, 2, 1).reshape(b, n, 3)
    ho = torch.zeros(b, 4, n)
    ho[:, 3, :] = 1
    worig = torch.bmm(c2w, ho).permute(0, 2, 1).reshape(b, n, 4)[..., :3]
    return dict(points=wpts, z=z, dirs=wdir, origins=worig, pitch=phi, yaw=theta, cam2world=c2w)


# ----------------------------------------------------------------------------------------
# H2 SIREN — exp/cips3d/models/generator.py:260-317, exp/comm/models/film_layer.py:78-107,
#            exp/comm/models/nerf_network.py:39-45
# ----------------------------------------------------------------------------------------
def film(sd, prefix, x, style):
    gain = F.linear(style, sd[prefix + "gain_fc.weight"], sd[prefix + "gain_fc.bias"]) * 15 + 30
    bias = F.linear(style, sd[prefix + "bias_fc.weight"], sd[prefix + "bias_fc.bias"])
    y = F.linear(x, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    return torch.sin(gain.unsqueeze(1) * y + bias.unsqueeze(1))


def siren(sd, points, w_nerf, prefix="siren."):
    """points (b,P,3), w_nerf (b,128) -> (b,P,33) = [feat(32), sigma(1)]."""
    x = points * (2 / 0.24)
    idx = 0
    while prefix + f"network.{idx}.linear.weight" in sd:          # generator.py:287: `for index, layer in enumerate(self.network)`
        x = film(sd, prefix + f"network.{idx}.", x, w_nerf)
        idx += 1
    sigma = F.linear(x, sd[prefix + "final_layer.weight"], sd[prefix + "final_layer.bias"])
    c = film(sd, prefix + "color_layer_sine.", x, w_nerf)
    feat = F.linear(c, sd[prefix + "color_layer_linear.0.weight"], sd[prefix + "color_layer_linear.0.bias"])
    return torch.cat([feat, sigma], dim=-1)


# ----------------------------------------------------------------------------------------
# H3 — exp/pigan/pigan_utils.py:164-209 (sample_pdf), :212-273 (fancy_integration),
#       exp/dev/nerf_inr/models/generator_nerf_inr.py:537-598, generator.py:1733-1752
# ----------------------------------------------------------------------------------------
class ClampTape:
    """Test infrastructure, like GateTape: the branch relu(sigma + nerf_noise * eps) took per (image, ray, sorted position)
    in the differentiable composite (pigan_utils.py:246-252).  pin=None records (x > 0); else an iterable of bool / uint8
    tensors (b, n, E[, 1]) replayed in call order: relu(x) becomes x * pinned, the pinned branch's linear extension."""

    def __init__(self, pin=None):
        self.rec = []
        self.preact = []
        self._pin = iter(pin) if pin is not None else None

    def relu(self, x):
        self.preact.append(x.detach())
        if self._pin is None:
            self.rec.append((x > 0).detach())
            return F.relu(x)
        g = next(self._pin).to(x.device).reshape(x.shape) != 0
        self.rec.append(g)
        return x * g.to(x.dtype)


_CLAMP_TAPE = [None]


class clamp_tape:
    """with clamp_tape(tape): the FINAL composite of every generator forward below (not the no-grad coarse composite that
    only feeds the resampler — the fine-sample placement has its own pin, fine_z_pin) goes through `tape`."""

    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.old = _CLAMP_TAPE[0]
        _CLAMP_TAPE[0] = self.tape
        return self.tape

    def __exit__(self, *exc):
        _CLAMP_TAPE[0] = self.old


def integrate(rgb_sigma, z, noise, noise_std, dim_rgb=32, clamp_mode="relu", last_back=False, white_back=False, tape=None):
    """rgb_sigma (b,n,E,33), z (b,n,E,1), noise (b,n,E,1) raw randn -> rgb (b,n,32), depth, weights."""
    rgbs, sig = rgb_sigma[..., :dim_rgb], rgb_sigma[..., dim_rgb:]
    d = z[:, :, 1:] - z[:, :, :-1]
    d = torch.cat([d, 1e10 * torch.ones_like(d[:, :, :1])], -2)
    nz = noise * noise_std
    if clamp_mode == "softplus":
        a = 1 - torch.exp(-d * F.softplus(sig + nz))
    elif clamp_mode == "relu":
        a = 1 - torch.exp(-d * (tape.relu(sig + nz) if tape is not None else F.relu(sig + nz)))
    else:
        raise AssertionError("Need to choose clamp mode")
    shifted = torch.cat([torch.ones_like(a[:, :, :1]), 1 - a + 1e-10], -2)
    w = a * torch.cumprod(shifted, -2)[:, :, :-1]
    wsum = w.sum(2)
    if last_back:
        w[:, :, -1] += (1 - wsum)
    rgb = torch.sum(w * rgbs, -2)
    depth = torch.sum(w * z, -2)
    if white_back:
        rgb = rgb + 1 - wsum
    return rgb, depth, w


def sample_pdf(bins, weights, u, eps=1e-5):
    """bins (R,S-1), weights (R,S-2), u (R,S) -> samples (R,S) + (cdf, inds, below, above)."""
    n_w = weights.shape[1]
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_w)
    sel = torch.stack([below, above], -1).view(u.shape[0], -1)
    cg = torch.gather(cdf, 1, sel).view(u.shape[0], -1, 2)
    bg = torch.gather(bins, 1, sel).view(u.shape[0], -1, 2)
    den = cg[..., 1] - cg[..., 0]
    den[den < eps] = 1
    smp = bg[..., 0] + (u - cg[..., 0]) / den * (bg[..., 1] - bg[..., 0])
    return smp, dict(cdf=cdf, inds=inds, below=below, above=above)


def fine_points(coarse, z, noise_c, nerf_noise, u, origins, dirs, clamp_mode="relu"):
    """-> fine_points (b, n*S, 3), fine_z (b,n,S,1), plus the bookkeeping dict (all no-grad)."""
    b, n, S, _ = z.shape
    _, _, w = integrate(coarse, z, noise_c, nerf_noise, clamp_mode=clamp_mode)
    w2 = w.reshape(b * n, S) + 1e-5
    z2 = z.reshape(b * n, S)
    mid = 0.5 * (z2[:, :-1] + z2[:, 1:])
    fz, book = sample_pdf(mid, w2[:, 1:-1], u)
    fz = fz.detach().reshape(b, n, S, 1)
    fp = origins.unsqueeze(2).contiguous() + dirs.unsqueeze(2).contiguous() * fz.expand(-1, -1, -1, 3).contiguous()
    book["weights"] = w
    return fp.reshape(b, n * S, 3), fz, book


# ----------------------------------------------------------------------------------------
# H4 CIPS INR head — exp/cips3d/models/generator.py:949-974, 983-1006, 1107-1153,
#                     exp/comm/models/mod_conv_fc.py:470-489
# ----------------------------------------------------------------------------------------
def mod_fc(sd, prefix, x, style, eps=1e-8):
    s = F.linear(style, sd[prefix + "modulation.weight"], sd[prefix + "modulation.bias"])
    w = sd[prefix + "weight"] * (s.unsqueeze(-1) + 1)                    # (b,in,out)
    w = w * torch.rsqrt(w.pow(2).sum([1]) + eps).unsqueeze(1)
    return torch.bmm(x, w)


INR_NAMES = [str(2 ** i) for i in range(2, 11)]


def inr_head(sd, fea, w_inr, prefix="inr_net.", return_all=False):
    """fea (b,n,32), w_inr (b,512) -> rgb (b,n,3) in [-1,1].  All nine blocks run (the reference
    calls inr_net without img_size, generator.py:1754 -> default 1024)."""
    x = fea
    rgb = 0
    outs = []
    for idx, name in enumerate(INR_NAMES):
        x0 = x
        x = _gated_lrelu(mod_fc(sd, f"{prefix}network.{name}.mod1.", x, w_inr), LRELU)
        x = _gated_lrelu(mod_fc(sd, f"{prefix}network.{name}.mod2.", x, w_inr), LRELU)
        if idx >= 4 and x.shape[-1] == x0.shape[-1]:
            x = x + x0
        if idx >= 3:
            rgb = F.linear(x, sd[f"{prefix}to_rgbs.{name}.linear.weight"], sd[f"{prefix}to_rgbs.{name}.linear.bias"]) + rgb
        outs.append(x)
    out = torch.tanh(rgb)
    return (out, outs) if return_all else out


# ----------------------------------------------------------------------------------------
# generator.forward — exp/cips3d/models/generator.py:1256-1370, 1378-1534, 1659-1762
# ----------------------------------------------------------------------------------------
_FINE_Z_PIN = [None]


class fine_z_pin:
    """with fine_z_pin(fz): the hierarchical pass below takes its fine depths (b, n, S, 1) from `fz` instead of resampling
    them (test infrastructure: the searchsorted of sample_pdf is a discontinuity — an fp64 evaluation of the same network
    places ~0.2 % of the samples in a neighbouring bin; gradients are compared for ONE placement)."""

    def __init__(self, fz):
        self.fz = fz

    def __enter__(self):
        self.old = _FINE_Z_PIN[0]
        _FINE_Z_PIN[0] = self.fz
        return self

    def __exit__(self, *exc):
        _FINE_Z_PIN[0] = self.old


def _points_forward(sd, w_nerf, w_inr, pts, z, origins, dirs, b, n, S, hierarchical_sample, nerf_noise, clamp_mode,
                    noise_c, u, noise_f, return_aux_img, nerf_nograd, keep=None, last_back=False, white_back=False):
    """points_forward (exp/cips3d/models/generator.py:1659-1762) for n rays per image:
    -> inr rgb (b,n,3), aux rgb (b,n,3) or None"""
    ctx = torch.no_grad() if nerf_nograd else torch.enable_grad()
    with ctx:
        coarse = siren(sd, pts.reshape(b, n * S, 3), w_nerf).reshape(b, n, S, 33)
    if hierarchical_sample:
        with torch.no_grad():
            fp, fz, book = fine_points(coarse, z, noise_c, nerf_noise, u, origins, dirs, clamp_mode)
            if _FINE_Z_PIN[0] is not None:
                fz = _FINE_Z_PIN[0].to(z.dtype).reshape(b, n, S, 1)
                fp = (origins.unsqueeze(2).contiguous() + dirs.unsqueeze(2).contiguous() * fz.expand(-1, -1, -1, 3).contiguous()).reshape(b, n * S, 3)
        with ctx:
            fine = siren(sd, fp, w_nerf).reshape(b, n, S, 33)
        all_o = torch.cat([fine, coarse], dim=-2)
        all_z = torch.cat([fz, z], dim=-2)
        _, idx = torch.sort(all_z, dim=-2)
        all_z = torch.gather(all_z, -2, idx)
        all_o = torch.gather(all_o, -2, idx.expand(-1, -1, -1, all_o.shape[-1]))
    else:
        all_o, all_z, idx, fz, fp, book, fine = coarse, z, None, None, None, None, None
    fea, depth, weights = integrate(all_o, all_z, noise_f, nerf_noise, clamp_mode=clamp_mode, last_back=last_back,
                                    white_back=white_back, tape=_CLAMP_TAPE[0])
    inr = inr_head(sd, fea, w_inr)
    aux = None
    if return_aux_img:
        with ctx:
            aux = torch.tanh(F.linear(fea, sd["aux_to_rbg.0.weight"], sd["aux_to_rbg.0.bias"]))
    if keep is not None:
        keep.update(coarse=coarse, fine=fine, fine_z=fz, fine_points=fp, book=book, sort_idx=idx,
                    pixels_fea=fea, depth=depth, weights=weights)
    return inr, aux


def generator_forward(sd, zs, rand, img_size, fov, ray_start, ray_end, num_steps, h_stddev, v_stddev,
                      hierarchical_sample, nerf_noise=0., clamp_mode="relu", return_aux_img=False,
                      freeze_nerf=False, keep=False, grad_points=None, last_back=False, white_back=False, psi=1.,
                      avg_styles=None, camera=None, forward_points=None):
    """Returns dict(imgs, pitch_yaw, + intermediates when keep).  `sd` values may require grad.

    psi < 1: truncation towards `avg_styles` = (avg_w_nerf (1,128), avg_w_inr (1,512)), the batch mean of the
    mapping networks over 10 000 latents (generator.py:1320-1323, 1804-1817; generator_nerf_inr.py:770-782).
    camera = dict(camera_pos, camera_lookup, up_vector): forward_camera_pos_and_lookup (generator.py:1828-1951).
    The staged forward (`forward_points`, generator.py:1406-1473) computes the same image chunk by chunk; its
    observable differences are the order of the random draws, which `rand` already carries in assembled form, and
    that ONLY it hands `up_vector` to the camera matrix (generator.py:1437; the one-shot branch :1481-1497 drops it).

    grad_points (< img_size^2): part_grad_forward (generator.py:1536-1657) — `rand["rand_idx"]` (the randperm)
    splits the pixels into a subset rendered with gradients (draws rand["noise_c_grad"/"u_grad"/"noise_f_grad"])
    and the rest rendered under no_grad (rand["..._rest"]), scattered back like comm_utils.py:240-258."""
    b = zs["z_nerf"].shape[0]
    S = num_steps
    n = img_size * img_size
    if freeze_nerf:
        with torch.no_grad():
            w_nerf = mapping_nerf(sd, zs["z_nerf"])
    else:
        w_nerf = mapping_nerf(sd, zs["z_nerf"])
    w_inr = mapping_inr(sd, zs["z_inr"])
    if psi < 1:
        w_nerf = avg_styles[0] + psi * (w_nerf - avg_styles[0])
        w_inr = avg_styles[1] + psi * (w_inr - avg_styles[1])
    with torch.no_grad():
        r = rays(b, img_size, fov, ray_start, ray_end, S, rand["jitter"], rand.get("theta"), rand.get("phi"),
                 h_stddev, v_stddev, **{k: (v if (k != "up_vector" or forward_points is not None) else None)
                                        for k, v in (camera or {}).items()})
    out = {}
    kept = {} if keep else None
    if grad_points is None or grad_points >= n:
        inr, aux = _points_forward(sd, w_nerf, w_inr, r["points"], r["z"], r["origins"], r["dirs"], b, n, S,
                                   hierarchical_sample, nerf_noise, clamp_mode, rand.get("noise_c"), rand.get("u"),
                                   rand["noise_f"], return_aux_img, freeze_nerf, kept, last_back=last_back,
                                   white_back=white_back)
    else:
        ridx = rand["rand_idx"]
        parts = []
        for tag, idx, nograd in (("_grad", ridx[:grad_points], freeze_nerf), ("_rest", ridx[grad_points:], True)):
            m = idx.numel()
            sub = lambda t: t.index_select(1, idx)
            def run():
                return _points_forward(sd, w_nerf, w_inr, sub(r["points"].reshape(b, n, S, 3)), sub(r["z"]),
                                       sub(r["origins"]), sub(r["dirs"]), b, m, S, hierarchical_sample, nerf_noise,
                                       clamp_mode, rand.get("noise_c" + tag), rand.get("u" + tag),
                                       rand["noise_f" + tag], return_aux_img, nograd)
            if tag == "_rest":
                with torch.no_grad():
                    parts.append((idx, run()))
            else:
                parts.append((idx, run()))

        def scatter(k):
            o = torch.zeros(b, n, 3)
            for idx, res in parts:
                o = o.index_copy(1, idx, res[k])
            return o
        inr = scatter(0)
        aux = scatter(1) if return_aux_img else None
    imgs = inr.reshape(b, img_size, img_size, 3).permute(0, 3, 1, 2)
    pitch_yaw = torch.cat([r["pitch"], r["yaw"]], -1)
    if return_aux_img:
        imgs = torch.cat([imgs, aux.reshape(b, img_size, img_size, 3).permute(0, 3, 1, 2)])
        pitch_yaw = torch.cat([pitch_yaw, pitch_yaw])
    out.update(imgs=imgs, pitch_yaw=pitch_yaw)
    if keep:
        out.update(w_nerf=w_nerf, w_inr=w_inr, points=r["points"], z=r["z"], dirs=r["dirs"], origins=r["origins"])
        out.update(kept)
    return out


# ----------------------------------------------------------------------------------------
# H5 discriminator — exp/cips3d/models/discriminator.py:20-288, 502-585, 647-664;
#   native ops: exp/comm/op/fused_bias_act_kernel.cu:36-47, exp/comm/op/upfirdn2d.py:152-186
# ----------------------------------------------------------------------------------------
def fused_leaky_relu(x, bias, slope=0.2, scale=2 ** 0.5):
    rest = [1] * (x.ndim - bias.ndim - 1)
    return _gated_lrelu(x + bias.view(1, bias.shape[0], *rest), slope) * scale


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """Element-wise restatement of the CUDA kernel's act*10+grad switch (fused_bias_act_kernel.cu:36-47)."""
    v = x.clone()
    if bias is not None and bias.numel():
        step = 1
        for i in range(2, x.dim()):
            step *= x.size(i)
        idx = (torch.arange(x.numel()) // step) % bias.numel()
        v = (v.reshape(-1) + bias[idx]).reshape(x.shape)
    mode = act * 10 + grad
    if mode in (12, 32):
        y = torch.zeros_like(v)
    elif mode == 30:
        y = torch.where(v > 0, v, v * alpha)
    elif mode == 31:
        y = torch.where(ref > 0, v, v * alpha)
    else:
        y = v
    return y * scale


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """x (b,c,h,w); zero-stuff by `up`, pad, correlate with the FLIPPED kernel, decimate by `down`."""
    b, c, h, w = x.shape
    kh, kw = kernel.shape
    t = x.reshape(b * c, 1, h, 1, w, 1)
    t = F.pad(t, [0, up - 1, 0, 0, 0, up - 1])
    t = t.reshape(b * c, 1, h * up, w * up)
    p0, p1 = pad
    t = F.pad(t, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    t = t[:, :, max(-p0, 0): t.shape[2] - max(-p1, 0), max(-p0, 0): t.shape[3] - max(-p1, 0)]
    t = F.conv2d(t, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw))
    t = t[:, :, ::down, ::down]
    return t.reshape(b, c, t.shape[2], t.shape[3])


def _blur_kernel():
    k = torch.tensor([1., 3., 3., 1.])
    k = k[None, :] * k[:, None]
    return k / k.sum()


def conv_layer(sd, prefix, x, k, downsample=False, activate=True, bias=True):
    """ConvLayer (discriminator.py:134-222): [Blur] -> EqualConv2d -> [FusedLeakyReLU]."""
    w = sd[prefix + "equal_conv.weight"]
    cin = w.shape[1]
    scale = 1 / math.sqrt(cin * k * k)
    if downsample:
        p = (4 - 2) + (k - 1)
        x = upfirdn2d(x, sd.get(prefix + "down_blur.kernel", _blur_kernel()), pad=((p + 1) // 2, p // 2))
        stride, padding = 2, 0
    else:
        stride, padding = 1, (k - 1) // 2
    cb = sd.get(prefix + "equal_conv.bias") if (bias and not activate) else None
    x = F.conv2d(x, w * scale, bias=cb, stride=stride, padding=padding)
    if activate:
        x = fused_leaky_relu(x, sd[prefix + "flrelu.bias"]) if bias else F.leaky_relu(x, 0.2) * math.sqrt(2)
    return x


def res_block(sd, prefix, x, first_downsample=False):
    if first_downsample:
        o = conv_layer(sd, prefix + "conv1.", x, 3, downsample=True)
        o = conv_layer(sd, prefix + "conv2.", o, 3)
    else:
        o = conv_layer(sd, prefix + "conv1.", x, 3)
        o = conv_layer(sd, prefix + "conv2.", o, 3, downsample=True)
    s = conv_layer(sd, prefix + "skip.", x, 1, downsample=True, activate=False, bias=False)
    return (o + s) / math.sqrt(2)


def equal_linear(sd, prefix, x, activation=False):
    w = sd[prefix + "weight"]
    scale = 1 / math.sqrt(w.shape[1])
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), sd[prefix + "bias"])
    return F.linear(x, w * scale, bias=sd[prefix + "bias"])


def diff_augment(x, draws, policy="color,translation,cutout"):
    """DiffAugment (exp/cips3d/models/diffaug.py:9-85) with its random draws handed in: `draws` is an iterator over
    the tensors the reference drew, in its order — color: 3 x rand(b,1,1,1) (brightness, saturation, contrast);
    translation: 2 x randint(b,1,1) (rows, columns; shift up to 1/8 of the size); cutout: 2 x randint(b,1,1) (the
    centre of a square hole of 0.2 x the size)."""
    b, c, h, w = x.shape
    for p in policy.split(","):
        if p == "color":
            x = x + (next(draws) - 0.5)
            m = x.mean(dim=1, keepdim=True)
            x = (x - m) * (next(draws) * 2) + m
            m = x.mean(dim=[1, 2, 3], keepdim=True)
            x = (x - m) * (next(draws) + 0.5) + m
        elif p == "translation":
            tx, ty = next(draws), next(draws)
            gb, gx, gy = torch.meshgrid(torch.arange(b), torch.arange(h), torch.arange(w), indexing="ij")
            gx = torch.clamp(gx + tx + 1, 0, h + 1)
            gy = torch.clamp(gy + ty + 1, 0, w + 1)
            xp = F.pad(x, [1, 1, 1, 1, 0, 0, 0, 0])
            x = xp.permute(0, 2, 3, 1).contiguous()[gb, gx, gy].permute(0, 3, 1, 2)
        elif p == "cutout":
            ch, cw = int(h * 0.2 + 0.5), int(w * 0.2 + 0.5)
            ox, oy = next(draws), next(draws)
            gb, gx, gy = torch.meshgrid(torch.arange(b), torch.arange(ch), torch.arange(cw), indexing="ij")
            gx = torch.clamp(gx + ox - ch // 2, min=0, max=h - 1)
            gy = torch.clamp(gy + oy - cw // 2, min=0, max=w - 1)
            mask = torch.ones(b, h, w, dtype=x.dtype)
            mask[gb, gx, gy] = 0
            x = x * mask.unsqueeze(1)
        else:
            raise KeyError(p)
    return x.contiguous()


def disc_multiscale(sd, prefix, x, alpha=1., first_downsample=False, draws=None):
    """Discriminator_MultiScale.forward (discriminator.py:502-585), stddev_group 0; `draws`: DiffAugment on the input
    (discriminator.py:507-508) with these recorded draws."""
    if draws is not None:
        x = diff_augment(x, draws)
    size = x.shape[-1]
    ls = int(math.log(size, 2))
    cur = conv_layer(sd, f"{prefix}conv_in.{2 ** ls}.", x, 1)
    cur = res_block(sd, f"{prefix}convs.{2 ** ls}.", cur, first_downsample)
    if alpha < 1:
        dn = F.interpolate(x, scale_factor=0.5, mode="bilinear")
        dn = conv_layer(sd, f"{prefix}conv_in.{2 ** (ls - 1)}.", dn, 1)
        out = alpha * cur + (1 - alpha) * dn
    else:
        out = cur
    for i in range(ls - 1, 2, -1):
        out = res_block(sd, f"{prefix}convs.{2 ** i}.", out, first_downsample)
    out = conv_layer(sd, f"{prefix}final_conv.", out, 3)
    out = out.view(out.shape[0], -1)
    out = equal_linear(sd, f"{prefix}space_linear.", out, activation=True)
    return equal_linear(sd, f"{prefix}out_linear.", out)


def discriminator_forward(sd, x, alpha=1., use_aux_disc=False, draws=None):
    """Discriminator_MultiScale_Aux.forward (discriminator.py:647-664).  `draws` (diffaug=True): the recorded random
    tensors, main discriminator's first."""
    it = iter(draws) if draws is not None else None
    if use_aux_disc:
        b = x.shape[0] // 2
        m = disc_multiscale(sd, "main_disc.", x[:b], alpha, first_downsample=False, draws=it)
        a = disc_multiscale(sd, "aux_disc.", x[b:], alpha, first_downsample=True, draws=it)
        return torch.cat([m, a], dim=0)
    return disc_multiscale(sd, "main_disc.", x, alpha, first_downsample=False, draws=it)
