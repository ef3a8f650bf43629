"""GPU: parity at BASELINE.json's REAL configurations, not toy sizes — the HIP path
against the CPU oracle run on the GPU box's host cores with identical weights, latents and random draws:

  C2  r64, 24 SIREN evaluations per ray, batch 32: S=24 flat and S=12+12 hierarchical, every image of the batch
      (oracle under no_grad, image by image);
  C1  r32, S=12 hierarchical, batch 4, aux image: forward AND every parameter gradient;
  C3  r128 geometry, S=12+12, an image pair: forward;
  C4  r256, S=24+24 (E=48), GeneratorNerfINR_freeze_NeRF, an image pair: forward and the INR-side gradients (the
      per-GPU batch of the 8-GPU stages: the weight-gradient GEMMs split the pixel range, ops.py InrHeadX3Function);
  D   Discriminator_MultiScale_Aux at 64^2 and 256^2, main + aux branch, fade-in alpha < 1: logits, R1 input gradient
      (double-backward graph) and every parameter gradient of the full d_loss — the 256-/128-channel layers
      (convs.64/128/256, the aux branch's 256-wide stack) and every conv dispatch branch (implicit GEMM, folded small
      planes, streaming RGB convs) with the real channel tables (discriminator.py:441-451, 621-631).

Gradients are compared for the same LeakyReLU gates (recorded from the oracle's run and pinned into the HIP path, see
test_gpu_generator.py); bars: images / logits 1e-3 (north_star), gradients 5e-4."""
import pytest
import torch

from conftest import seeded_generator, max_rel, pack_bitplane, D_CFG
from oracle import cips3d_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-3
GRAD_TOL = 5e-4
KW = dict(fov=12, ray_start=0.88, ray_end=1.12, h_stddev=0.3, v_stddev=0.155)


def _draws(g, b, img, S, hier):
    n = img * img
    E = 2 * S if hier else S
    zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
    rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g),
                phi=torch.randn(b, 1, generator=g), noise_c=torch.randn(b, n, S, 1, generator=g),
                u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, E, 1, generator=g))
    return zs, rand


def _slice(zs, rand, i, b, n, S):
    z1 = {k: v[i:i + 1] for k, v in zs.items()}
    r1 = {k: (v[i:i + 1] if k != "u" else v.view(b, n, S)[i].reshape(n, S)) for k, v in rand.items()}
    return z1, r1


def _product_forward(G, zs, rand, d, img, S, hier, aux=False, pin=None, nerf_noise=0.0, rec=None):
    from cips3d_amd import ops
    with ops.gate_debug(pin=pin, rec=rec):
        imgs, _ = G({k: v.to(d) for k, v in zs.items()}, img_size=img, num_steps=S, hierarchical_sample=hier,
                    sample_dist="gaussian", nerf_noise=nerf_noise, return_aux_img=aux, grad_points=None,
                    forward_points=None, rand_override={k: v.to(d) for k, v in rand.items()}, **KW)
    return imgs


@pytest.mark.parametrize("S,hier", [(24, False), (12, True)])
def test_c2_full_batch_forward_vs_oracle(S, hier):
    """BASELINE configs[1]: r64, 24 evaluations per ray, batch 32 — all 32 images against the oracle."""
    d = torch.device("cuda:0")
    b, img = 32, 64
    n = img * img
    g = torch.Generator().manual_seed(2024 + S)
    zs, rand = _draws(g, b, img, S, hier)
    Gc = seeded_generator(1234)
    sd = dict(Gc.named_parameters())
    Gd = seeded_generator(1234, device=d)
    with torch.no_grad():
        imgs = _product_forward(Gd, zs, rand, d, img, S, hier, nerf_noise=0.3).cpu()
        worst = 0.0
        for i in range(b):
            z1, r1 = _slice(zs, rand, i, b, n, S)
            ref = orc.generator_forward(sd, z1, r1, img, KW["fov"], KW["ray_start"], KW["ray_end"], S, KW["h_stddev"],
                                        KW["v_stddev"], hier, nerf_noise=0.3)["imgs"]
            e = max_rel(imgs[i:i + 1], ref)
            worst = max(worst, e)
            assert e < TOL, (i, e)
    print(f"C2 b=32 r64 S={S} hier={hier} (nerf_noise 0.3): worst image max_rel vs oracle {worst:.3e}")


def _grad_compare(named_params, ref_grads, what, ref64=None, tol=None):
    """every parameter gradient against the fp32 oracle's at GRAD_TOL.  With `ref64` (the same evaluation in fp64) the bar of
    a parameter is max(GRAD_TOL, 4 x the fp32 oracle's own distance from the fp64 gradient), both measured against fp64: an
    ill-conditioned gradient is held to the reference's own accuracy, not to a tolerance its fp32 arithmetic does not meet."""
    worst = ("", 0.0, 0.0)
    loose = []
    n_used = 0
    GRAD_TOL = tol if tol is not None else globals()["GRAD_TOL"]
    for name, p in named_params:
        r = ref_grads.get(name)
        if r is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        n_used += 1
        got = p.grad.detach().cpu().double()
        if ref64 is None:
            e, bar = float((got - r.double()).norm() / r.double().norm().clamp_min(1e-300)), GRAD_TOL
        else:
            x = ref64[name].double()
            e = float((got - x).norm() / x.norm().clamp_min(1e-300))
            e32 = float((r.double() - x).norm() / x.norm().clamp_min(1e-300))
            bar = max(GRAD_TOL, 4 * e32)
            if bar > GRAD_TOL:
                loose.append(f"{name}: product {e:.2e}, fp32 oracle {e32:.2e} from fp64")
        assert e < bar, (name, e, bar)
        if e / bar > worst[1]:
            worst = (name, e / bar, e)
    print(f"{what}: {n_used} parameter gradients, tightest at {worst[1]:.2f} of its bar (rel err {worst[2]:.3e}, {worst[0]})"
          + (f"; ill-conditioned in fp32 (bar = 4 x the oracle's own fp32 error): {loose}" if loose else ""))


def test_c1_forward_backward_vs_oracle():
    """BASELINE configs[0] geometry on the GPU: r32, S = 12 + 12, batch 4, aux image, nerf_noise 0.2 — forward and every
    parameter gradient.  Round 4: like C3 below, the path's three discontinuities are pinned to the oracle's choices
    (LeakyReLU gates, placement of the fine samples, branch of relu(sigma + noise)) after the free-running choices have been
    compared and counted.  The free-running form of this test passed at 6.4e-5 in round 3 and failed at 8.1e-4 on
    `siren.final_layer.weight` (alone) once the SIREN forward computed its sine arguments in revolutions: a handful of
    samples took the other side of a discontinuity, which is an input of the sigma head's gradient, not an error of it."""
    n = _g_forward_backward_vs_oracle("C1 b=4 r32 S=12+12 aux, nerf_noise 0.2", 4, 32, 12, True, True, 0.2, 31, pin_fine=True,
                                      pin_clamp=True, tol=2e-4, free_bar=1e-3)
    assert n == 130, n


def test_c3_r128_pair_forward_vs_oracle():
    """C3 geometry: r128, 24 evaluations per ray (S=12+12), an image pair."""
    d = torch.device("cuda:0")
    b, img, S = 2, 128, 12
    g = torch.Generator().manual_seed(128)
    zs, rand = _draws(g, b, img, S, True)
    Gc = seeded_generator(1234)
    Gd = seeded_generator(1234, device=d)
    with torch.no_grad():
        ref = orc.generator_forward(dict(Gc.named_parameters()), zs, rand, img, KW["fov"], KW["ray_start"], KW["ray_end"], S,
                                    KW["h_stddev"], KW["v_stddev"], True)["imgs"]
        imgs = _product_forward(Gd, zs, rand, d, img, S, True)
    e = max_rel(imgs, ref)
    print(f"C3 geometry b=2 r128 S=12+12: imgs max_rel {e:.3e}")
    assert e < TOL


def test_c4_r256_e48_frozen_nerf_forward_backward_vs_oracle():
    """C4: r256, num_steps 24 + hierarchical (E=48), GeneratorNerfINR_freeze_NeRF (ffhq_exp.yaml:192-210), an image
    pair: forward and the gradients of everything that trains in that stage (INR head, mapping_inr)."""
    import psutil
    d = torch.device("cuda:0")
    # the oracle's autograd state is ~6 GB of host memory per 256^2 image
    b, img, S = (2 if psutil.virtual_memory().available > 64 * 2 ** 30 else 1), 256, 24
    g = torch.Generator().manual_seed(256)
    zs, rand = _draws(g, b, img, S, True)
    G0 = torch.randn(b, 3, img, img, generator=g) / (b * 3 * img * img)
    Gc = seeded_generator(1234, freeze=True)
    tape = orc.GateTape()
    with orc.gate_tape(tape):
        ref = orc.generator_forward(dict(Gc.named_parameters()), zs, rand, img, KW["fov"], KW["ray_start"], KW["ray_end"], S,
                                    KW["h_stddev"], KW["v_stddev"], True, freeze_nerf=True)
    (ref["imgs"] * G0).sum().backward()
    ref_grads = {k: p.grad for k, p in Gc.named_parameters() if p.grad is not None}
    assert not any(k.startswith(("siren", "mapping_network_nerf", "aux_to_rbg")) for k in ref_grads)
    ref_imgs = ref["imgs"].detach()
    pins = [pack_bitplane(t) for t in tape.rec]
    del ref, tape
    Gd = seeded_generator(1234, freeze=True, device=d)
    imgs = _product_forward(Gd, zs, rand, d, img, S, True, pin=pins)
    e = max_rel(imgs, ref_imgs)
    print(f"C4 b=2 r256 S=24+24 frozen NeRF: imgs max_rel {e:.3e}")
    assert e < TOL
    (imgs * G0.to(d)).sum().backward()
    torch.cuda.synchronize()
    _grad_compare(list(Gd.named_parameters()), ref_grads, "C4 gradients (oracle's gates pinned)")


@pytest.mark.parametrize("size,b,alpha", [(64, 2, 0.6), (256, 1, 0.75)])
def test_discriminator_real_sizes_vs_oracle(size, b, alpha):
    """Discriminator_MultiScale_Aux with the shipped channel tables at 64^2 and 256^2, main + aux branch, fade-in:
    logits, R1 input gradient and all parameter gradients of the d_loss of train.py:385-409 against the oracle."""
    from cips3d_amd import discriminator as dmod
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    d = torch.device("cuda:0")
    torch.manual_seed(4321)
    D = Discriminator_MultiScale_Aux(**D_CFG)
    g = torch.Generator().manual_seed(size)
    x0 = torch.rand(2 * b, 3, size, size, generator=g) * 2 - 1
    sd = dict(D.state_dict())
    sd.update(dict(D.named_parameters()))
    x = x0.clone().requires_grad_(True)
    tape = orc.GateTape()
    with orc.gate_tape(tape):
        out = orc.discriminator_forward(sd, x, alpha=alpha, use_aux_disc=True)
    gr, = torch.autograd.grad(out.sum(), x, create_graph=True)
    loss = torch.nn.functional.softplus(-out).mean() + 0.5 * 10. * gr.flatten(1).pow(2).sum(1).mean()
    loss.backward()
    ref_grads = {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None}
    ref_out, ref_gr, ref_loss = out.detach(), gr.detach(), float(loss)
    D.zero_grad(set_to_none=True)
    Dd = D.to(d)
    xd = x0.to(d).requires_grad_(True)
    with dmod.gate_debug(pin=tape.rec):
        o = Dd(xd, alpha=alpha, use_aux_disc=True)[0]
    gd, = torch.autograd.grad(o.sum(), xd, create_graph=True)
    ld = torch.nn.functional.softplus(-o).mean() + 0.5 * 10. * gd.flatten(1).pow(2).sum(1).mean()
    ld.backward()
    torch.cuda.synchronize()
    e, eg = max_rel(o, ref_out), max_rel(gd, ref_gr)
    print(f"D {size}x{size} b={b} main+aux alpha={alpha}: logits max_rel {e:.3e}, R1 input-gradient max_rel {eg:.3e}, "
          f"loss {float(ld):.6f} vs {ref_loss:.6f}")
    assert e < TOL and eg < GRAD_TOL and abs(float(ld) - ref_loss) < TOL * max(1.0, abs(ref_loss))
    _grad_compare(list(Dd.named_parameters()), ref_grads, f"D {size}x{size} gradients (oracle's gates pinned)")


FREE_AB = None           # scripts/free_running_parity.py: [(label, ops.TRIG_MODE), ...] arms of the free-running run
FREE_RUNNING = {}        # what -> {"final_layer.weight": e, "final_layer.bias": e, "worst": (name, e)}: scripts/free_running_parity.py


def _g_forward_backward_vs_oracle(what, b, img, S, hier, aux, nerf_noise, seed, pin_fine=False, pin_clamp=False, tol=None,
                                  free_bar=None, free_gates=None):
    """forward + every parameter gradient of `(imgs * G0).sum()` against the oracle, the oracle's LeakyReLU gates pinned.
    pin_fine: also the oracle's placement of the resampled (fine) samples — the searchsorted of the importance
    resampling is the path's second discontinuity: a cdf value within rounding of the uniform draw lands a sample in the
    neighbouring bin, and the sigma head's gradient (a sum with heavy cancellation) moves by a finite amount per such
    sample.  The free-running placement is compared first and the differing samples are counted.
    pin_clamp: also the branch the oracle's `relu(sigma + nerf_noise * eps)` took per sample (pigan_utils.py:246-252, the
    third discontinuity; ops.clamp_debug / orc.clamp_tape): the free-running branches are compared first — every sample
    whose branch differs must be ambiguous (|pre-activation| within the SIREN forward's rounding of 0) and they are counted
    — then forward and backward run on the oracle's branches.
    free_gates = (max fraction of differing gates, max ambiguity, gradient bar) (round 6): after the pinned
    comparison the step runs with NOTHING pinned; the head's LeakyReLU gates the product chose are recorded, counted against the
    oracle's, every differing site must be ambiguous (fp64 |pre-activation| below `max ambiguity` x the layer's rms), and every
    gradient is asserted against the fp64 oracle evaluated AT THE PRODUCT'S GATES.
    free_bar (round 5): after the pinned comparison the SAME step runs FREE — fine-sample placement and clamp branches are the
    product's own; only the INR head's LeakyReLU gates stay pinned (they are the head's matter, proven in
    test_gpu_generator.py, and one flipped gate moves the head's gradients, not the sigma head's) — and the sigma head's two
    gradients (siren.final_layer.*: the ones the round-4 split-bf16 forward moved by 4.4e-3 at r128 / 8.1e-4 at C1) are
    asserted against the fp32 oracle at `free_bar`; every other gradient at 10 x GRAD_TOL as a regression bound."""
    from cips3d_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    zs, rand = _draws(g, b, img, S, hier)
    nimg = 2 * b if aux else b
    G0 = torch.randn(nimg, 3, img, img, generator=g) / (nimg * 3 * img * img)
    Gc = seeded_generator(1234)
    tape = orc.GateTape()
    ctape = orc.ClampTape()
    with orc.gate_tape(tape), orc.clamp_tape(ctape if pin_clamp else None):
        ref = orc.generator_forward(dict(Gc.named_parameters()), zs, rand, img, KW["fov"], KW["ray_start"], KW["ray_end"],
                                    S, KW["h_stddev"], KW["v_stddev"], hier, nerf_noise=nerf_noise, return_aux_img=aux,
                                    keep=pin_fine)
    (ref["imgs"] * G0).sum().backward()
    E = 2 * S if hier else S
    ref_clamp = ctape.rec[0].reshape(b * img * img, E).to(torch.uint8) if pin_clamp else None
    ref_pre = ctape.preact[0].reshape(b * img * img, E) if pin_clamp else None
    ref_grads = {k: p.grad for k, p in Gc.named_parameters() if p.grad is not None}
    ref_imgs = ref["imgs"].detach()
    ref_fz = ref["fine_z"].detach().reshape(b * img * img, S) if pin_fine else None
    pins = [pack_bitplane(t) for t in tape.rec]
    gate_shapes = [tuple(t.shape) for t in tape.rec]
    ref64 = None
    if pin_fine:
        # the same network, gates and sample placement in fp64: how far the reference's OWN fp32 arithmetic is from the
        # exact gradient.  The sigma head's gradients (siren.final_layer.*: sums of d sigma over every sample of the batch,
        # with heavy cancellation) are ill-conditioned at this size — the fp32 oracle itself is 1e-3 off there.
        G64 = seeded_generator(1234).double()
        t64 = orc.GateTape(pin=tape.rec)
        c64 = orc.ClampTape(pin=ctape.rec) if pin_clamp else None
        torch.set_default_dtype(torch.float64)
        try:
            with orc.gate_tape(t64), orc.fine_z_pin(ref["fine_z"].detach().double()), orc.clamp_tape(c64):
                r64 = orc.generator_forward(dict(G64.named_parameters()), {k: v.double() for k, v in zs.items()},
                                            {k: v.double() for k, v in rand.items()}, img, KW["fov"], KW["ray_start"], KW["ray_end"],
                                            S, KW["h_stddev"], KW["v_stddev"], hier, nerf_noise=nerf_noise, return_aux_img=aux)
            (r64["imgs"] * G0.double()).sum().backward()
        finally:
            torch.set_default_dtype(torch.float32)
        ref64 = {k: p.grad for k, p in G64.named_parameters() if p.grad is not None}
        del r64, t64
    del ref, tape, ctape
    Gd = seeded_generator(1234, device=d)
    if pin_fine:
        rec = []
        with torch.no_grad(), ops.resample_debug(rec=rec):
            free = _product_forward(Gd, zs, rand, d, img, S, hier, aux=aux, nerf_noise=nerf_noise)
        dz = (rec[0].cpu() - ref_fz).abs()
        moved = int((dz > 1e-4 * (KW["ray_end"] - KW["ray_start"])).sum())
        print(f"{what}: free-running sample placement: {moved} of {dz.numel()} fine samples in another bin than the oracle's "
              f"({moved / dz.numel():.1e}); images max_rel {max_rel(free, ref_imgs):.3e}")
        # a sanity bound on the COUNT of discontinuity crossings (each one is a cdf value within the SIREN forward's ~1e-5 of a
        # uniform draw), not a parity bar: 1.9e-3 of the samples at r128 in round 3, 2.4e-3 with the sine arguments computed in
        # revolutions (round 4; images 1.4e-5 -> 1.7e-5 against the 1e-3 bar)
        assert moved <= 5e-3 * dz.numel() and max_rel(free, ref_imgs) < TOL
    if pin_clamp:
        # free-running branches on the oracle's sample placement: which samples sit on the other side of the clamp, and
        # how close to 0 their pre-activation is (in units of the pre-activations' rms)
        crec = []
        with torch.no_grad(), ops.resample_debug(pin=[ref_fz] if pin_fine else None), ops.clamp_debug(rec=crec):
            _product_forward(Gd, zs, rand, d, img, S, hier, aux=aux, nerf_noise=nerf_noise)
        diff = crec[0].cpu() != ref_clamp
        nflip = int(diff.sum())
        rms = float(ref_pre.double().pow(2).mean().sqrt())
        amb = float(ref_pre[diff].abs().max()) / rms if nflip else 0.0
        print(f"{what}: free-running relu clamp: {nflip} of {diff.numel()} samples on the other branch than the oracle's "
              f"({nflip / diff.numel():.1e}); largest |sigma + noise| among them {amb:.2e} x rms")
        assert nflip <= 2e-4 * diff.numel() and amb < 2e-4, (nflip, amb)
    with ops.resample_debug(pin=[ref_fz] if pin_fine else None), ops.clamp_debug(pin=[ref_clamp] if pin_clamp else None):
        imgs = _product_forward(Gd, zs, rand, d, img, S, hier, aux=aux, pin=pins, nerf_noise=nerf_noise)
    e = max_rel(imgs, ref_imgs)
    print(f"{what}: imgs max_rel {e:.3e}")
    assert imgs.shape == (nimg, 3, img, img) and e < TOL
    (imgs * G0.to(d)).sum().backward()
    torch.cuda.synchronize()
    _grad_compare(list(Gd.named_parameters()), ref_grads, f"{what} gradients (oracle's gates" + (" and fine-sample placement" if pin_fine else "")
                  + (" and relu-clamp branches" if pin_clamp else "") + " pinned)", ref64=ref64, tol=tol)
    if free_gates is not None:
        assert not pin_fine and not pin_clamp, "free_gates: a flat-sampling, noise-free-clamp configuration"
        max_frac, max_amb, gbar = free_gates
        from conftest import unpack_bitplane
        rec = []
        Gd.zero_grad(set_to_none=True)
        imgs_f = _product_forward(Gd, zs, rand, d, img, S, hier, aux=aux, nerf_noise=nerf_noise, rec=rec)
        (imgs_f * G0.to(d)).sum().backward()
        torch.cuda.synchronize()
        assert len(rec) == len(pins) and max_rel(imgs_f, ref_imgs) < TOL
        own = [unpack_bitplane(r.cpu()).reshape(sh) for r, sh in zip(rec, gate_shapes)]
        # fp64 oracle AT THE PRODUCT'S GATES: gradients to compare with, and the exact pre-activation at every gate
        G64 = seeded_generator(1234).double()
        t64 = orc.GateTape(pin=own)
        t64.keep_preact = True
        torch.set_default_dtype(torch.float64)
        try:
            with orc.gate_tape(t64):
                r64 = orc.generator_forward(dict(G64.named_parameters()), {k: v.double() for k, v in zs.items()},
                                            {k: v.double() for k, v in rand.items()}, img, KW["fov"], KW["ray_start"], KW["ray_end"],
                                            S, KW["h_stddev"], KW["v_stddev"], hier, nerf_noise=nerf_noise, return_aux_img=aux)
            (r64["imgs"] * G0.double()).sum().backward()
        finally:
            torch.set_default_dtype(torch.float32)
        t64.done()
        flips, worst_amb, total = 0, 0.0, 0
        for o, p8, y in zip(own, pins, t64.preact):
            diff = o != unpack_bitplane(p8).reshape(o.shape)
            total += o.numel()
            nd = int(diff.sum())
            if nd:
                flips += nd
                worst_amb = max(worst_amb, float(y[diff].max() / y.pow(2).mean().sqrt()))
        print(f"{what}: FREE-RUNNING head gates: {flips} of {total} differ from the oracle's ({flips / total:.1e}); largest fp64 "
              f"|pre-activation| / rms among them {worst_amb:.2e}")
        assert flips <= max_frac * total and worst_amb <= max_amb, (flips, total, worst_amb)
        g64 = {k: p.grad for k, p in G64.named_parameters() if p.grad is not None}
        worst = ("", 0.0)
        for name, p in Gd.named_parameters():
            r = g64.get(name)
            if r is None:
                continue
            e_ = float((p.grad.detach().cpu().double() - r).norm() / r.norm().clamp_min(1e-300))
            if e_ > worst[1]:
                worst = (name, e_)
        print(f"{what}: FREE-RUNNING gradients against the fp64 oracle at the product's gates: worst {worst[1]:.2e} ({worst[0]}), bar {gbar:.0e}")
        assert worst[1] <= gbar, worst
        del r64, t64, G64, own
    if free_bar is not None:
        for label, trig in (FREE_AB or [("", None)]):
            old_trig = ops.TRIG_MODE
            if trig is not None:
                ops.TRIG_MODE = trig
            try:
                Gd.zero_grad(set_to_none=True)
                imgs = _product_forward(Gd, zs, rand, d, img, S, hier, aux=aux, pin=pins, nerf_noise=nerf_noise)
                (imgs * G0.to(d)).sum().backward()
                torch.cuda.synchronize()
            finally:
                ops.TRIG_MODE = old_trig
            errs = {}
            for name, p in Gd.named_parameters():
                r = ref_grads.get(name)
                if r is not None:
                    errs[name] = float((p.grad.detach().cpu().double() - r.double()).norm() / r.double().norm().clamp_min(1e-300))
            head = {k: v for k, v in errs.items() if k.startswith("siren.final_layer.")}
            rest = {k: v for k, v in errs.items() if k not in head}
            wk = max(rest, key=rest.get)
            key = what + (f" [{label}]" if label else "")
            FREE_RUNNING[key] = {"imgs": max_rel(imgs, ref_imgs), "final_layer.weight": head["siren.final_layer.weight"],
                                 "final_layer.bias": head["siren.final_layer.bias"], "worst_other": [wk, rest[wk]]}
            print(f"{key}: FREE-RUNNING (gates pinned; placement and clamp the product's own): siren.final_layer.weight "
                  f"{head['siren.final_layer.weight']:.2e}, .bias {head['siren.final_layer.bias']:.2e} (bar {free_bar:.0e}); worst other "
                  f"gradient {rest[wk]:.2e} ({wk}); images {FREE_RUNNING[key]['imgs']:.2e}")
            if trig is None or trig == 1:
                assert all(v <= free_bar for v in head.values()), head
                assert rest[wk] <= 10 * GRAD_TOL, (wk, rest[wk])
    return len(ref_grads)


def test_c2_headline_geometry_flat_march_forward_backward_vs_oracle():
    """BASELINE configs[1] at the geometry bench.py times: r64, S = 24 FLAT sampling (the fused ray-march kernel and its
    backward: cips_march_fwd_x3 -> cips_composite_bwd + cips_siren_bwd_x3_rays), aux image on (ffhq_exp.yaml:169), batch 4:
    forward and all 130 parameter gradients (generator.py:1659-1762).  Round 6: then the same step with NOTHING pinned — the
    product's own LeakyReLU gates are counted against the oracle's (at most 2e-5 of them may differ, each at a pre-activation
    within 5e-5 of the layer's rms in fp64) and every gradient is held to 2e-4 against the fp64 oracle at those gates."""
    n = _g_forward_backward_vs_oracle("C2 geometry b=4 r64 S=24 flat, aux", 4, 64, 24, False, True, 0.2, 64,
                                      free_gates=(2e-5, 5e-5, 2e-4))
    assert n == 130, n


def test_c2_flat_march_full_nerf_noise_clamp_pinned_forward_backward_vs_oracle():
    """The step-0 value of the noise schedule (train.py:325-327: nerf_noise = max(0, 1 - step / 5000)): r64, S = 24 flat (the
    fused ray-march), nerf_noise 1.0, default split-bf16 mode, an image pair.  With the oracle's relu-clamp branches pinned
     EVERY gradient — including siren.final_layer.*, the sigma head — is inside 2e-4."""
    n = _g_forward_backward_vs_oracle("r64 S=24 flat b=2, nerf_noise 1.0", 2, 64, 24, False, False, 1.0, 6411, pin_clamp=True,
                                      tol=2e-4)
    assert n >= 100, n


def test_c3_r128_pair_forward_backward_vs_oracle():
    """C3 geometry with gradients: r128, S = 12 + 12, aux image, an image pair (the weight-gradient GEMMs contract over
    16 384 pixels per image), nerf_noise 0.1 (round 4; it ran at 0 before).  All three discontinuities of the path are pinned
    to the oracle's choices — LeakyReLU gates, placement of the fine samples, and the branch of `relu(sigma + noise)`
    (pigan_utils.py fancy_integration: the product's split-bf16 SIREN forward carries sigma to ~1e-5 where the oracle's fp32
    carries 1e-7, so at 786 432 samples a handful land on the other side of the clamp and the sigma head's two gradients —
    sums of d sigma with heavy cancellation — moved by 4.4e-3 free-running, profiles/HISTORY.md §0).  Bars against the fp64 evaluation of
    the same network with the same three pins: max(2e-4, 4 x the fp32 oracle's own distance from fp64), see _grad_compare."""
    _g_forward_backward_vs_oracle("C3 geometry b=2 r128 S=12+12, aux, nerf_noise 0.1", 2, 128, 12, True, True, 0.1, 1283,
                                  pin_fine=True, pin_clamp=True, tol=2e-4, free_bar=1e-3)


def _aug_draws(g, nb, size):
    """the seven DiffAugment draws of one Discriminator_MultiScale.forward on nb images (diffaug.py:32-67), as the
    (kind, tensor) records ReplayDraws replays and orc.discriminator_forward consumes"""
    sx = int(size * 0.125 + 0.5)
    c = int(size * 0.2 + 0.5)
    out = [("rand", torch.rand(nb, 1, 1, 1, generator=g)) for _ in range(3)]
    out += [("randint", torch.randint(-sx, sx + 1, [nb, 1, 1], generator=g)) for _ in range(2)]
    out += [("randint", torch.randint(0, size + (1 - c % 2), [nb, 1, 1], generator=g)) for _ in range(2)]
    return out


def test_c5_finetune_step_r256_aux_two_discriminators_vs_oracle():
    """BASELINE configs[4] (finetune_afhq.yaml:38,70,89 with train_aux_img): r256, num_steps 12 + hierarchical (E = 24),
    GeneratorNerfINR_freeze_NeRF, aux image, Discriminator_MultiScale_Aux with diffaug=True, the aux discriminator and a
    fade-in alpha < 1 — the G step and the D step of train.py:334-466 on one image:
      G step: imgs = G(z) (main + aux image), g_loss = softplus(-D(DiffAugment(imgs))).mean(): images, logits and every
              generator gradient of the stage (through both discriminators, the augmentation and the fade-in);
      D step: r_preds on real images (requires_grad), R1 penalty through the double-backward graph, g_preds on the
              generated images: logits, R1 input gradient, loss and every parameter gradient of both discriminators."""
    from conftest import ReplayDraws
    from cips3d_amd import ops, discriminator as dmod
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    d = torch.device("cuda:0")
    b, img, S, alpha = 1, 256, 12, 0.8
    g = torch.Generator().manual_seed(555)
    zs, rand = _draws(g, b, img, S, True)
    real = torch.rand(2 * b, 3, img, img, generator=g) * 2 - 1          # train.py:376-377: real_imgs twice when aux_reg
    aug_g, aug_r, aug_f = (_aug_draws(g, b, img) + _aug_draws(g, b, img) for _ in range(3))      # main + aux disc, per D forward
    torch.manual_seed(4321)
    D = Discriminator_MultiScale_Aux(**dict(D_CFG, diffaug=True))
    sdD = dict(D.state_dict()); sdD.update(dict(D.named_parameters()))
    Gc = seeded_generator(1234, freeze=True)
    F = torch.nn.functional
    # ---------------- oracle: G step
    tape_g, tape_d = orc.GateTape(), orc.GateTape()
    with orc.gate_tape(tape_g):
        ref = orc.generator_forward(dict(Gc.named_parameters()), zs, rand, img, KW["fov"], KW["ray_start"], KW["ray_end"], S,
                                    KW["h_stddev"], KW["v_stddev"], True, nerf_noise=0.0, return_aux_img=True, freeze_nerf=True)
    with orc.gate_tape(tape_d):
        gp = orc.discriminator_forward(sdD, ref["imgs"], alpha=alpha, use_aux_disc=True, draws=[t for _, t in aug_g])
    F.softplus(-gp).mean().backward()
    ref_g_grads = {k: p.grad for k, p in Gc.named_parameters() if p.grad is not None}
    assert not any(k.startswith(("siren", "mapping_network_nerf", "aux_to_rbg")) for k in ref_g_grads)
    ref_imgs, ref_gp = ref["imgs"].detach(), gp.detach()
    pins_g, pins_dg = [pack_bitplane(t) for t in tape_g.rec], tape_d.rec
    D.zero_grad(set_to_none=True)
    del ref, gp, tape_g
    # ---------------- oracle: D step (fake images = the oracle's, so that the D comparison stands on its own)
    x = real.clone().requires_grad_(True)
    tape_dd = orc.GateTape()
    with orc.gate_tape(tape_dd):
        rp = orc.discriminator_forward(sdD, x, alpha=alpha, use_aux_disc=True, draws=[t for _, t in aug_r])
        gr, = torch.autograd.grad(rp.sum(), x, create_graph=True)
        fp = orc.discriminator_forward(sdD, ref_imgs, alpha=alpha, use_aux_disc=True, draws=[t for _, t in aug_f])
    pen = 0.5 * 10. * gr.flatten(1).square().sum(1, keepdim=True) * 1 + 0. * rp           # train.py:397-398
    d_loss = (F.softplus(fp) + F.softplus(-rp) + pen).mean()
    d_loss.backward()
    ref_d_grads = {k: p.grad.clone() for k, p in D.named_parameters() if p.grad is not None}
    ref_rp, ref_gr, ref_fp, ref_dl = rp.detach(), gr.detach(), fp.detach(), float(d_loss)
    D.zero_grad(set_to_none=True)
    del rp, gr, fp, pen, d_loss
    # ---------------- product: G step
    Gd = seeded_generator(1234, freeze=True, device=d)
    Dd = D.to(d)
    for p in Dd.parameters():
        p.requires_grad_(False)                               # train.py:441-442
    imgs = _product_forward(Gd, zs, rand, d, img, S, True, aux=True, pin=pins_g)
    with dmod.gate_debug(pin=pins_dg), ReplayDraws(aug_g):
        gpd = Dd(imgs, alpha=alpha, use_aux_disc=True)[0]
    F.softplus(-gpd).mean().backward()
    torch.cuda.synchronize()
    e_img, e_gp = max_rel(imgs, ref_imgs), max_rel(gpd, ref_gp)
    print(f"C5 G step b=1 r256 S=12+12 frozen NeRF + aux image + aux D + DiffAugment, alpha={alpha}: imgs max_rel {e_img:.3e}, "
          f"logits max_rel {e_gp:.3e}")
    assert imgs.shape == (2 * b, 3, img, img) and e_img < TOL and e_gp < TOL
    _grad_compare(list(Gd.named_parameters()), ref_g_grads, "C5 G-step generator gradients (oracle's gates pinned)")
    del imgs, gpd
    # ---------------- product: D step
    for p in Dd.parameters():
        p.requires_grad_(True)
    xd = real.to(d).requires_grad_(True)
    with dmod.gate_debug(pin=tape_dd.rec), ReplayDraws(aug_r + aug_f):
        rpd = Dd(xd, alpha=alpha, use_aux_disc=True)[0]
        grd, = torch.autograd.grad(rpd.sum(), xd, create_graph=True)
        fpd = Dd(ref_imgs.to(d), alpha=alpha, use_aux_disc=True)[0]
    pend = 0.5 * 10. * grd.flatten(1).square().sum(1, keepdim=True) * 1 + 0. * rpd
    dl = (F.softplus(fpd) + F.softplus(-rpd) + pend).mean()
    dl.backward()
    torch.cuda.synchronize()
    e_r, e_g, e_f = max_rel(rpd, ref_rp), max_rel(grd, ref_gr), max_rel(fpd, ref_fp)
    print(f"C5 D step: real logits {e_r:.3e}, R1 input gradient {e_g:.3e}, fake logits {e_f:.3e}, d_loss {float(dl):.6f} vs {ref_dl:.6f}")
    assert e_r < TOL and e_f < TOL and e_g < GRAD_TOL and abs(float(dl) - ref_dl) < TOL * max(1.0, abs(ref_dl))
    _grad_compare(list(Dd.named_parameters()), ref_d_grads, "C5 D-step gradients, main + aux discriminator (oracle's gates pinned)")
