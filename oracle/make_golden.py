"""TEST INFRASTRUCTURE — mint golden fixtures from the UNMODIFIED reference.

Run in the build container only (`python oracle/make_golden.py`): it imports
/root/reference through oracle/ref_shim.py, runs the reference generator / discriminator on CPU
with every torch.rand / torch.randn draw captured, and writes small fixtures to tests/golden/.
Weights are NOT stored: the product modules reproduce the reference's initial state_dict
bit-for-bit under the same torch seed (checked here and stored as per-key checksums).
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import yaml

from oracle import ref_shim

ref_shim.install()

from exp.cips3d.models import generator as ref_gen          # noqa: E402
from exp.cips3d.models import discriminator as ref_disc     # noqa: E402
from exp.pigan import pigan_utils                           # noqa: E402
from exp.comm import comm_utils                             # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CFG = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, "exp/cips3d/configs/ffhq_exp.yaml")))


def g_cfg():
    c = dict(CFG["G_cfg_3D2D"])
    c.pop("register_modules"); c.pop("name")
    return c


def d_cfg():
    c = dict(CFG["D_cfg"])
    c.pop("register_modules"); c.pop("name")
    return c


class Capture:
    """Record torch.rand / torch.randn / torch.randperm / torch.randint results in call order."""

    def __enter__(self):
        self.draws = []
        self._rand, self._randn, self._randperm, self._randint = torch.rand, torch.randn, torch.randperm, torch.randint

        def randint(*a, **k):
            t = self._randint(*a, **k); self.draws.append(("randint", t.clone())); return t
        torch.randint = randint

        def randperm(*a, **k):
            t = self._randperm(*a, **k); self.draws.append(("randperm", t.clone())); return t
        torch.randperm = randperm

        def rand(*a, **k):
            t = self._rand(*a, **k); self.draws.append(("rand", t.clone())); return t

        def randn(*a, **k):
            t = self._randn(*a, **k); self.draws.append(("randn", t.clone())); return t

        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn, torch.randperm, torch.randint = self._rand, self._randn, self._randperm, self._randint


def pack_gate(g):
    """bool tensor -> dict(shape, bits): flat little-endian bit packing (bit i&7 of byte i>>3); for a (b,n,C) head
    activation with C % 8 == 0 this is the kernels' gate bit-plane layout (include/cips3d_hip.h: gate_bits)."""
    import numpy as np
    return dict(shape=tuple(g.shape), bits=torch.from_numpy(np.packbits(g.reshape(-1).numpy(), bitorder="little")))


class HeadGates:
    """Record the gate (input > 0) of every LeakyReLU of the reference's CIPS head (SinBlock.act1 / act2,
    generator.py:922, 936) in call order."""

    def __init__(self, G):
        self.gates, self.handles = [], []
        for name, m in G.inr_net.named_modules():
            if isinstance(m, torch.nn.LeakyReLU):
                self.handles.append(m.register_forward_pre_hook(lambda mod, inp: self.gates.append(pack_gate(inp[0].detach() > 0))))

    def close(self):
        for h in self.handles:
            h.remove()


class LeakyGates:
    """Record the gate of every F.leaky_relu call (the discriminator: all of them are FusedLeakyReLU /
    fused_leaky_relu, fused_act.py:51-86) in call order."""

    def __enter__(self):
        import torch.nn.functional as F
        self.gates, self._orig = [], F.leaky_relu

        def leaky_relu(input, negative_slope=0.01, inplace=False):
            self.gates.append(pack_gate(input.detach() > 0))
            return self._orig(input, negative_slope, inplace)
        F.leaky_relu = leaky_relu
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.leaky_relu = self._orig


def save_gates(tag, gates, check_key, value):
    """gates-only minting: the rerun must reproduce the committed fixture bit for bit"""
    fix = torch.load(os.path.join(OUT, f"{tag}.pt"), map_location="cpu", weights_only=False)
    assert torch.equal(fix[check_key], value), f"{tag}: rerun does not reproduce the committed fixture"
    path = os.path.join(OUT, f"gates_{tag}.pt")
    torch.save(dict(tag=tag, gates=gates), path)
    print("gates", tag, "->", path, os.path.getsize(path) // 1024, "KiB", len(gates), "layers")


def checksums(sd):
    return {k: (float(v.double().sum()), float(v.double().abs().sum())) for k, v in sd.items()}


def grad_digest(named_params, stride=97):
    d = {}
    for name, p in named_params:
        if p.grad is None:
            d[name] = None
            continue
        g = p.grad.detach().reshape(-1)
        d[name] = dict(norm=float(g.double().norm()), n=g.numel(),
                       sample=g[::stride].clone() if g.numel() > 70000 else g.clone(),
                       stride=stride if g.numel() > 70000 else 1)
    return d


def make_generator_part_case(tag, seed, b, img_size, S, hier, nerf_noise, aux, grad_points, mint="fixture"):
    """part_grad_forward (generator.py:1536-1657): gradients through a random pixel subset only."""
    torch.manual_seed(seed)
    G = ref_gen.GeneratorNerfINR(**g_cfg(), device="cpu")
    sums = checksums(G.state_dict())
    torch.manual_seed(seed + 1)
    zs = G.get_zs(b)
    kw = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=S, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=hier, psi=1., sample_dist="gaussian")
    hg = HeadGates(G)
    with Capture() as cap:
        imgs, pitch_yaw = G(zs, img_size=img_size, nerf_noise=nerf_noise, return_aux_img=aux,
                            grad_points=grad_points, forward_points=None, **kw)
    hg.close()
    if mint == "gates":
        return save_gates(tag, hg.gates, "imgs", imgs.detach())
    per = (["noise_c", "u"] if hier else []) + ["noise_f"]
    names = ["jitter", "theta", "phi", "rand_idx"] + [n + "_grad" for n in per] + [n + "_rest" for n in per]
    assert len(cap.draws) == len(names), (len(cap.draws), names)
    rand = {n: t for n, (_, t) in zip(names, cap.draws)}
    torch.manual_seed(4321)
    G0 = torch.randn_like(imgs) / imgs.numel()
    (imgs * G0).sum().backward()
    fix = dict(tag=tag, seed=seed, b=b, img_size=img_size, S=S, hier=hier, nerf_noise=nerf_noise, aux=aux,
               freeze=False, grad_points=grad_points, G_kwargs=kw, state_checksums=sums,
               zs={k: v.clone() for k, v in zs.items()}, rand=rand, G0=G0, imgs=imgs.detach().clone(),
               pitch_yaw=pitch_yaw.detach().clone(), grads=grad_digest(G.named_parameters()))
    path = os.path.join(OUT, f"{tag}.pt")
    torch.save(fix, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "imgs", tuple(imgs.shape))


def make_generator_case(tag, seed, b, img_size, S, hier, nerf_noise, aux, freeze=False, mint="fixture"):
    torch.manual_seed(seed)
    cls = ref_gen.GeneratorNerfINR_freeze_NeRF if freeze else ref_gen.GeneratorNerfINR
    G = cls(**g_cfg(), device="cpu")
    sums = checksums(G.state_dict())
    torch.manual_seed(seed + 1)
    zs = G.get_zs(b)
    kw = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=S, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=hier, psi=1., sample_dist="gaussian")
    rec = {}
    o_siren, o_fi, o_sp, o_rays = G.siren.forward, pigan_utils.fancy_integration, pigan_utils.sample_pdf, \
        comm_utils.get_world_points_and_direction
    rec["siren"], rec["fi"], rec["sp"] = [], [], []

    def siren_fwd(*a, **k):
        r = o_siren(*a, **k); rec["siren"].append(r.detach().clone()); return r

    def fi(*a, **k):
        r = o_fi(*a, **k); rec["fi"].append(tuple(t.detach().clone() for t in r)); return r

    def sp(*a, **k):
        r = o_sp(*a, **k); rec["sp"].append(r.detach().clone()); return r

    def rays(*a, **k):
        r = o_rays(*a, **k); rec["rays"] = tuple(t.detach().clone() for t in r); return r

    G.siren.forward = siren_fwd
    pigan_utils.fancy_integration = fi
    pigan_utils.sample_pdf = sp
    comm_utils.get_world_points_and_direction = rays
    hg = HeadGates(G)
    try:
        with Capture() as cap:
            imgs, pitch_yaw = G(zs, img_size=img_size, nerf_noise=nerf_noise, return_aux_img=aux,
                                grad_points=None, forward_points=None, **kw)
    finally:
        hg.close()
        G.siren.forward = o_siren
        pigan_utils.fancy_integration, pigan_utils.sample_pdf = o_fi, o_sp
        comm_utils.get_world_points_and_direction = o_rays
    if mint == "gates":
        return save_gates(tag, hg.gates, "imgs", imgs.detach())
    draws = cap.draws
    names = ["jitter", "theta", "phi"] + (["noise_c", "u"] if hier else []) + ["noise_f"]
    assert len(draws) == len(names), (len(draws), names)
    rand = {n: t for n, (_, t) in zip(names, draws)}
    torch.manual_seed(4321)
    G0 = torch.randn_like(imgs) / imgs.numel()
    (imgs * G0).sum().backward()
    n = img_size * img_size
    fix = dict(tag=tag, seed=seed, b=b, img_size=img_size, S=S, hier=hier, nerf_noise=nerf_noise, aux=aux,
               freeze=freeze, G_kwargs=kw, state_checksums=sums, zs={k: v.clone() for k, v in zs.items()},
               rand=rand, G0=G0, imgs=imgs.detach().clone(), pitch_yaw=pitch_yaw.detach().clone(),
               points=rec["rays"][0].reshape(b, n, S, 3), dirs=rec["rays"][3], origins=rec["rays"][2],
               z=rec["rays"][4], coarse=rec["siren"][0].reshape(b, n, S, 33),
               fine=rec["siren"][1].reshape(b, n, S, 33) if hier else None,
               fine_z=rec["sp"][0] if hier else None,
               coarse_weights=rec["fi"][0][2] if hier else None,
               pixels_fea=rec["fi"][-1][0], depth=rec["fi"][-1][1], weights=rec["fi"][-1][2],
               grads=grad_digest(G.named_parameters()))
    path = os.path.join(OUT, f"{tag}.pt")
    torch.save(fix, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "imgs", tuple(imgs.shape))


def make_generator_eval_case(tag, seed, b, img_size, S, hier, psi, forward_points, nerf_noise, aux, clamp_mode,
                             last_back, white_back, camera=False):
    """Inference path: truncation psi < 1 through generate_avg_frequencies (generator.py:1320-1323,
    1804-1817), the staged no-grad forward in chunks of `forward_points` pixels (generator.py:1406-1473) with its
    per-image / per-chunk draw order, the composite's clamp_mode / last_back / white_back options, and — with
    `camera` — forward_camera_pos_and_lookup (generator.py:1828-1951) with an explicit camera and up vector."""
    torch.manual_seed(seed)
    G = ref_gen.GeneratorNerfINR(**g_cfg(), device="cpu")
    G.eval()
    sums = checksums(G.state_dict())
    torch.manual_seed(seed + 1)
    zs = G.get_zs(b)
    n = img_size * img_size
    E = 2 * S if hier else S
    kw = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=S, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=hier, psi=psi, sample_dist="gaussian", clamp_mode=clamp_mode,
              last_back=last_back, white_back=white_back)
    cam = None
    if camera:
        g = torch.Generator().manual_seed(seed + 2)
        pos = torch.nn.functional.normalize(torch.randn(b, 3, generator=g), dim=-1)
        cam = dict(camera_pos=pos, camera_lookup=-pos + 0.05 * torch.randn(b, 3, generator=g),
                   up_vector=torch.nn.functional.normalize(torch.tensor([[0.1, 1.0, 0.05]]), dim=-1)[0 if forward_points is None else slice(None)])
    with Capture() as cap, torch.no_grad():
        if camera:
            kw.update(h_mean=math.pi * 0.5, v_mean=math.pi * 0.5)
            imgs, pitch_yaw = G.forward_camera_pos_and_lookup(zs, img_size=img_size, nerf_noise=nerf_noise, return_aux_img=aux,
                                                              grad_points=None, forward_points=forward_points, **kw, **cam)
        else:
            imgs, pitch_yaw = G(zs, img_size=img_size, nerf_noise=nerf_noise, return_aux_img=aux, grad_points=None,
                                forward_points=forward_points, **kw)
    draws = list(cap.draws)
    fix_avg = None
    if psi < 1:
        (_, azn), (_, azi) = draws[0], draws[1]
        assert azn.shape == (10000, 256) and azi.shape == (10000, 512)
        # 30 MB of latents are not stored: they are the next two CPU draws after get_zs(b) under seed + 1 (the test
        # regenerates them and checks these checksums); the reference's averaged styles are stored as the answer
        fix_avg = dict(z_checksums=checksums(dict(z_nerf=azn, z_inr=azi)),
                       styles={k: v.detach().clone() for k, v in G.avg_styles.items()})
        draws = draws[2:]
    # staged order: per image [jitter, (theta, phi)], then per chunk [noise_c, u] (if hierarchical) and noise_f
    js, ths, phs, ncs, us, nfs = [], [], [], [], [], []
    it = iter(draws)
    if forward_points is not None:
        for _ in range(b):
            js.append(next(it)[1])
            if not camera:
                ths.append(next(it)[1]); phs.append(next(it)[1])
            head = 0
            while head < n:
                if hier:
                    ncs.append(next(it)[1]); us.append(next(it)[1])
                nfs.append(next(it)[1])
                head += forward_points
        rand = dict(jitter=torch.cat(js, 0), noise_f=torch.cat(nfs, 1).reshape(b, n, E, 1))
        if not camera:
            rand.update(theta=torch.cat(ths, 0), phi=torch.cat(phs, 0))
        if hier:
            rand.update(noise_c=torch.cat(ncs, 1).reshape(b, n, S, 1), u=torch.cat(us, 0))
    else:
        names = ["jitter"] + ([] if camera else ["theta", "phi"]) + (["noise_c", "u"] if hier else []) + ["noise_f"]
        rand = {nm: next(it)[1] for nm in names}
    assert next(it, None) is None, "unconsumed reference draws"
    fix = dict(tag=tag, seed=seed, b=b, img_size=img_size, S=S, hier=hier, nerf_noise=nerf_noise, aux=aux, freeze=False,
               forward_points=forward_points, camera=cam, G_kwargs=kw, state_checksums=sums,
               zs={k: v.clone() for k, v in zs.items()}, avg=fix_avg, rand=rand, imgs=imgs.detach().clone(),
               pitch_yaw=pitch_yaw.detach().clone())
    path = os.path.join(OUT, f"{tag}.pt")
    torch.save(fix, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "imgs", tuple(imgs.shape))


def make_diffaug_case():
    """exp/cips3d/models/diffaug.py:9-85: output and input gradient under recorded draws."""
    from exp.cips3d.models import diffaug as ref_da
    torch.manual_seed(17)
    cases = []
    for shape, policy in [((3, 3, 16, 16), "color,translation,cutout"), ((2, 3, 10, 12), "translation,cutout"),
                          ((2, 3, 8, 8), "color")]:
        x = (torch.rand(*shape) * 2 - 1).requires_grad_(True)
        with Capture() as cap:
            y = ref_da.DiffAugment(x, policy=policy)
        g0 = torch.randn_like(y)
        gx, = torch.autograd.grad((y * g0).sum(), x)
        cases.append(dict(x=x.detach().clone(), policy=policy, draws=[(k, t) for k, t in cap.draws], y=y.detach().clone(),
                          g0=g0, gx=gx.clone()))
    path = os.path.join(OUT, "diffaug_cases.pt")
    torch.save(cases, path)
    print("diffaug ->", path, os.path.getsize(path) // 1024, "KiB")


def make_discriminator_case(tag, seed, b, size, alpha, use_aux, diffaug=False, mint="fixture"):
    torch.manual_seed(seed)
    cfg = d_cfg()
    cfg["diffaug"] = diffaug
    D = ref_disc.Discriminator_MultiScale_Aux(**cfg)
    sums = checksums(D.state_dict())
    torch.manual_seed(seed + 1)
    x = (torch.rand(b * (2 if use_aux else 1), 3, size, size) * 2 - 1).requires_grad_(True)
    with Capture() as cap, LeakyGates() as lg:
        out, _, _ = D(x, alpha=alpha, use_aux_disc=use_aux)
    if mint == "gates":
        return save_gates(tag, lg.gates, "out", out.detach())
    # R1 path of train.py:385-409
    grad_real, = torch.autograd.grad(outputs=out.sum(), inputs=x, create_graph=True)
    pen = grad_real.flatten(1).pow(2).sum(1)
    loss = torch.nn.functional.softplus(-out).mean() + 0.5 * 10. * pen.mean()
    loss.backward()
    fix = dict(tag=tag, seed=seed, b=b, size=size, alpha=alpha, use_aux=use_aux, state_checksums=sums, diffaug=diffaug,
               draws=[(k, t) for k, t in cap.draws],
               x=x.detach().clone(), out=out.detach().clone(), grad_real=grad_real.detach().clone(),
               loss=float(loss), grads=grad_digest(D.named_parameters(), stride=997))
    path = os.path.join(OUT, f"{tag}.pt")
    torch.save(fix, path)
    print(tag, "->", path, os.path.getsize(path) // 1024, "KiB", "out", out.detach().flatten().tolist())


def make_camera_cases():
    """sample_camera_positions (comm_utils.py:451-535), every distribution, under fixed torch / Python seeds on the CPU
    generator: the test re-seeds and must reproduce the draws."""
    import random
    cases = []
    for i, mode in enumerate(["uniform", "normal", "gaussian", "hybrid", "hybrid", "hybrid", "truncated_gaussian",
                              "spherical_uniform", "mean"]):
        seed = 100 + i
        torch.manual_seed(seed); random.seed(seed)
        o, phi, theta = comm_utils.sample_camera_positions("cpu", bs=5, r=1.3, horizontal_stddev=0.3, vertical_stddev=0.155,
                                                           horizontal_mean=1.4, vertical_mean=1.7, mode=mode)
        cases.append(dict(mode=mode, seed=seed, origin=o, phi=phi, theta=theta))
    path = os.path.join(OUT, "camera_cases.pt")
    torch.save(cases, path)
    print("camera ->", path, os.path.getsize(path) // 1024, "KiB")


def make_pigan_cases():
    """Second, independent pin: the original pi-GAN implementations in
    piGAN_lib/generators/volumetric_rendering.py — fancy_integration (:18-55, dim_rgb = 3), sample_pdf (:205-246),
    get_initial_rays_trig + transform_sampled_points (:58-116) — run under captured draws.  exp/ carries its own copies of
    these functions (exp/pigan/pigan_utils.py, exp/comm/comm_utils.py); the oracle and the HIP kernels are checked
    against BOTH lineages."""
    import importlib
    vr = importlib.import_module("piGAN_lib.generators.volumetric_rendering")
    torch.manual_seed(41)
    out = dict(integrate=[], sample_pdf=[], rays=[])
    b, n, S = 2, 40, 7
    for clamp, noise_std, lb, wb in [("relu", 0.0, False, False), ("relu", 0.4, True, False), ("softplus", 0.2, False, True),
                                     ("relu", 0.3, True, True)]:
        rgb_sigma = torch.randn(b, n, S, 4) * torch.tensor([1., 1., 1., 4.])
        z = torch.sort(0.88 + 0.24 * torch.rand(b, n, S, 1), dim=-2)[0]
        with Capture() as cap:
            rgb, depth, w = vr.fancy_integration(rgb_sigma, z, "cpu", noise_std=noise_std, last_back=lb, white_back=wb,
                                                 clamp_mode=clamp)
        (_, noise), = cap.draws
        out["integrate"].append(dict(rgb_sigma=rgb_sigma, z=z, noise=noise, noise_std=noise_std, clamp=clamp, last_back=lb,
                                     white_back=wb, rgb=rgb.clone(), depth=depth.clone(), weights=w.clone()))
    for R, S_ in [(50, 12), (33, 5)]:
        zz = torch.sort(0.88 + 0.24 * torch.rand(R, S_), dim=-1)[0]
        bins = 0.5 * (zz[:, :-1] + zz[:, 1:])
        weights = torch.rand(R, S_ - 2) ** 3
        with Capture() as cap:
            smp = vr.sample_pdf(bins, weights, S_, det=False)
        (_, u), = cap.draws
        out["sample_pdf"].append(dict(z=zz, bins=bins, weights=weights, u=u, samples=smp.clone()))
    for (bb, img, S_) in [(2, 6, 5), (1, 9, 4)]:
        pts, zv, dcam = vr.get_initial_rays_trig(bb, S_, "cpu", fov=12, resolution=(img, img), ray_start=0.88, ray_end=1.12)
        with Capture() as cap:
            tp, tz, td, to, pitch, yaw = vr.transform_sampled_points(pts, zv, dcam, "cpu", h_stddev=0.3, v_stddev=0.155,
                                                                     h_mean=math.pi * 0.5, v_mean=math.pi * 0.5, mode="normal")
        (_, jit), (_, th), (_, ph) = cap.draws
        out["rays"].append(dict(b=bb, img=img, S=S_, jitter=jit, theta=th, phi=ph, points=tp.clone(), z=tz.clone(), dirs=td.clone(),
                                origins=to.clone(), pitch=pitch.clone(), yaw=yaw.clone()))
    path = os.path.join(OUT, "pigan_cases.pt")
    torch.save(out, path)
    print("pigan ->", path, os.path.getsize(path) // 1024, "KiB")


def make_op_cases():
    """Known-answer vectors for the two native ops from the reference's own restatement
    (upfirdn2d_native, upfirdn2d.py:152-186) and the kernel's switch (fused_bias_act_kernel.cu:36-47)."""
    torch.manual_seed(7)
    k = torch.tensor([1., 3., 3., 1.]); k = k[None] * k[:, None]; k = k / k.sum()
    cases = []
    for (shape, up, down, pad) in [((3, 9, 11, 1), 1, 1, (2, 2)), ((2, 8, 8, 1), 1, 1, (1, 1)),
                                   ((2, 6, 7, 2), 2, 1, (2, 1)), ((2, 9, 9, 1), 1, 2, (1, 1)),
                                   ((1, 5, 5, 3), 2, 2, (0, 0))]:
        x = torch.randn(*shape)
        y = ref_shim.upfirdn2d_native(x, k, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        cases.append(dict(x=x, k=k, up=up, down=down, pad=pad, y=y.contiguous()))
    path = os.path.join(OUT, "upfirdn2d_cases.pt")
    torch.save(cases, path)
    print("upfirdn2d ->", path, os.path.getsize(path) // 1024, "KiB")


def main(mint):
    """mint = "fixture": everything; "gates": only tests/golden/gates_<tag>.pt (the LeakyReLU gates of the reference's
    run of every training-path case; each rerun is checked bit for bit against the committed fixture)."""
    os.makedirs(OUT, exist_ok=True)
    G, D = make_generator_case, make_discriminator_case
    for m in (["fixture", "gates"] if mint == "fixture" else ["gates"]):
        G("g_r16_hier", seed=1234, b=2, img_size=16, S=6, hier=True, nerf_noise=0.0, aux=True, mint=m)
        G("g_r8_flat_noise", seed=0, b=2, img_size=8, S=4, hier=False, nerf_noise=0.3, aux=False, mint=m)
        G("g_r8_hier_noise", seed=5, b=1, img_size=8, S=5, hier=True, nerf_noise=0.25, aux=False, mint=m)
        G("g_r8_freeze", seed=3, b=2, img_size=8, S=4, hier=True, nerf_noise=0.0, aux=False, freeze=True, mint=m)
        make_generator_part_case("g_r16_part", seed=21, b=2, img_size=16, S=5, hier=True, nerf_noise=0.2, aux=True,
                                 grad_points=96, mint=m)
        make_generator_part_case("g_r16_part_odd", seed=22, b=2, img_size=16, S=4, hier=True, nerf_noise=0.0, aux=False,
                                 grad_points=100, mint=m)
        D("d_r16", seed=11, b=2, size=16, alpha=1.0, use_aux=False, mint=m)
        D("d_r16_aux_alpha", seed=12, b=2, size=16, alpha=0.5, use_aux=True, mint=m)
        D("d_r16_diffaug", seed=13, b=2, size=16, alpha=0.7, use_aux=True, diffaug=True, mint=m)
    if mint == "fixture":
        make_rest()


def make_rest():
    make_generator_eval_case("g_r8_eval_psi_staged", seed=31, b=2, img_size=8, S=4, hier=True, psi=0.7, forward_points=24,
                             nerf_noise=0.0, aux=True, clamp_mode="relu", last_back=True, white_back=False)
    make_generator_eval_case("g_r8_eval_camera", seed=32, b=2, img_size=8, S=5, hier=True, psi=1.0, forward_points=None,
                             nerf_noise=0.15, aux=False, clamp_mode="softplus", last_back=False, white_back=True, camera=True)
    make_generator_eval_case("g_r8_eval_camera_staged", seed=33, b=1, img_size=8, S=4, hier=True, psi=1.0, forward_points=40,
                             nerf_noise=0.1, aux=True, clamp_mode="relu", last_back=False, white_back=False, camera=True)
    make_diffaug_case()
    make_camera_cases()
    make_op_cases()
    make_pigan_cases()


if __name__ == "__main__":
    if "pigan" in sys.argv[1:]:
        os.makedirs(OUT, exist_ok=True)
        make_pigan_cases()
    else:
        main("gates" if "gates" in sys.argv[1:] else "fixture")
