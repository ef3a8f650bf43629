"""GPU: the drop-in generator (HIP path through the C-ABI) against the golden vectors minted
from the reference and against the CPU oracle, forward and backward."""
import math

import pytest
import torch

from conftest import load_golden, seeded_generator, check_checksums, max_rel, rel_err, G_CFG
from oracle import cips3d_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-3
CASES = ["g_r16_hier", "g_r8_flat_noise", "g_r8_hier_noise", "g_r8_freeze", "g_r16_part",
         "g_r16_part_odd"]   # _part: part_grad_forward (96 of 256 pixels; _odd: 100, not a multiple of the 32-pixel GEMM granule)


@pytest.fixture(params=["f32", "bf16x3"])
def inr_mode(request):
    """Run under both numeric modes: exact fp32 MFMA (head GEMMs and SIREN forward) and the 3-pass split-bf16 MFMA
    path (default)."""
    from cips3d_amd import ops
    old = (ops.INR_MODE, ops.SIREN_FWD_MODE)
    ops.INR_MODE = request.param
    ops.SIREN_FWD_MODE = "f32" if request.param == "f32" else "x3"
    yield request.param
    ops.INR_MODE, ops.SIREN_FWD_MODE = old


@pytest.mark.parametrize("tag", CASES)
def test_generator_matches_reference_golden(tag, inr_mode):
    fix = load_golden(tag)
    d = torch.device("cuda:0")
    G = seeded_generator(fix["seed"], freeze=fix["freeze"], device=d)
    check_checksums({k: v.cpu() for k, v in G.state_dict().items()}, fix["state_checksums"])
    zs = {k: v.to(d) for k, v in fix["zs"].items()}
    rand = {k: v.to(d) for k, v in fix["rand"].items()}
    imgs, pitch_yaw = G(zs, img_size=fix["img_size"], nerf_noise=fix["nerf_noise"], return_aux_img=fix["aux"],
                        grad_points=fix.get("grad_points"), forward_points=None, rand_override=rand, **fix["G_kwargs"])
    torch.cuda.synchronize()
    assert imgs.shape == fix["imgs"].shape
    e = max_rel(imgs, fix["imgs"])
    print(f"{tag} [{inr_mode}]: imgs max_rel vs reference {e:.3e}")
    assert e < TOL
    assert max_rel(pitch_yaw, fix["pitch_yaw"]) < 1e-5
    (imgs * fix["G0"].to(d)).sum().backward()
    torch.cuda.synchronize()
    # Yardstick for gradients: an fp64 evaluation of the (reference-pinned) oracle.  The HIP path must
    # be within 1e-3 of it, or as close to it as the reference's own fp32 gradients are (some tiny
    # noisy cases are ill-conditioned in fp32: 1 - exp(-delta*sigma) with delta*sigma ~ 1e-4).
    G64 = seeded_generator(fix["seed"], freeze=fix["freeze"]).double()
    kw = fix["G_kwargs"]
    dbl = lambda dd: {k: (v.double() if torch.is_floating_point(v) else v) for k, v in dd.items()}
    torch.set_default_dtype(torch.float64)
    try:
        o64 = orc.generator_forward(dict(G64.named_parameters()), dbl(fix["zs"]), dbl(fix["rand"]), fix["img_size"],
                                    kw["fov"], kw["ray_start"], kw["ray_end"], kw["num_steps"], kw["h_stddev"],
                                    kw["v_stddev"], kw["hierarchical_sample"], nerf_noise=fix["nerf_noise"],
                                    return_aux_img=fix["aux"], freeze_nerf=fix["freeze"],
                                    grad_points=fix.get("grad_points"))
    finally:
        torch.set_default_dtype(torch.float32)
    (o64["imgs"] * fix["G0"].double()).sum().backward()
    g64 = {n: p.grad for n, p in G64.named_parameters()}
    # part_grad_forward sends gradients through ~100 pixels per image only: a single flipped LeakyReLU gate then moves
    # EVERY upstream gradient by up to a few per cent (in pure fp64, perturbing the weights by 1e-7 relative — fp32
    # rounding — moves siren.final_layer.bias of g_r16_part_odd by 3.9e-2).  Measure that conditioning floor here:
    # three fp64 runs with the weights jittered at fp32 rounding level.
    floor = {}
    if fix.get("grad_points") is not None:
        for seed in (1, 2, 3):
            Gp = seeded_generator(fix["seed"], freeze=fix["freeze"]).double()
            gen = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                for p_ in Gp.parameters():
                    p_.mul_(1 + 1e-7 * torch.randn(p_.shape, generator=gen, dtype=torch.float64))
            torch.set_default_dtype(torch.float64)
            try:
                op = orc.generator_forward(dict(Gp.named_parameters()), dbl(fix["zs"]), dbl(fix["rand"]), fix["img_size"],
                                           kw["fov"], kw["ray_start"], kw["ray_end"], kw["num_steps"], kw["h_stddev"],
                                           kw["v_stddev"], kw["hierarchical_sample"], nerf_noise=fix["nerf_noise"],
                                           return_aux_img=fix["aux"], freeze_nerf=fix["freeze"], grad_points=fix["grad_points"])
            finally:
                torch.set_default_dtype(torch.float32)
            (op["imgs"] * fix["G0"].double()).sum().backward()
            for n_, p_ in Gp.named_parameters():
                if p_.grad is not None and g64.get(n_) is not None:
                    dev_ = float((p_.grad - g64[n_]).norm() / g64[n_].norm().clamp_min(1e-300))
                    floor[n_] = max(floor.get(n_, 0.0), dev_)
    rows, bad = [], []
    for name, p in G.named_parameters():
        dg = fix["grads"][name]
        if dg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        t64 = g64[name].reshape(-1)
        g = p.grad.reshape(-1).cpu().double()
        e_hip = float((g - t64).norm() / t64.norm().clamp_min(1e-300))
        st = dg["stride"]
        ref32 = dg["sample"].double()
        e_ref = float((ref32 - t64[::st]).norm() / t64[::st].norm().clamp_min(1e-300))
        e_vs_ref = float((g[::st] - ref32).norm() / ref32.norm().clamp_min(1e-300))
        rows.append((name, e_hip, e_ref, e_vs_ref))
        # LeakyReLU gates are discontinuous: with ~1e6 activations per tiny case, about one pre-activation
        # lands within fp32 rounding of 0 and its gate (1 vs 0.2) is arbitrary in ANY fp32 evaluation (the
        # reference's own fp32 gradients show the same jumps vs fp64).  One flipped gate moves an INR-side
        # weight gradient by ~0.8/sqrt(rows*512) relative; allow for it on the parameters behind the gates.
        rows_px = fix["b"] * fix["img_size"] ** 2
        gate_tol = 2.0 / (rows_px * 512) ** 0.5 if ("inr" in name) else 0.0
        # bf16x3 carries pre-activations to ~5e-6 instead of ~3e-7: ~15x more ambiguous gates; measured
        # gradient noise 0.5-1 % on every parameter upstream of the INR head, independent of problem size
        # (forward agreement stays ~3e-6).  DESIGN.md §3 "numerics".
        if inr_mode == "bf16x3":
            gate_tol = max(3e-2, 3 * gate_tol)
        if e_hip > max(TOL, 3 * e_ref, gate_tol, 3 * floor.get(name, 0.0)):
            bad.append((name, e_hip, e_ref))
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/gradtable_{tag}_{inr_mode}.txt", "w") as fh:
        fh.write("name  err_hip_vs_fp64  err_ref32_vs_fp64  err_hip_vs_ref32  max_abs_diff/max_abs  argmax\n")
        for name, p in G.named_parameters():
            if fix["grads"][name] is None:
                continue
            t64 = g64[name].reshape(-1); g = p.grad.reshape(-1).cpu().double()
            diff = (g - t64).abs()
            r = [x for x in rows if x[0] == name][0]
            fh.write(f"{name} {r[1]:.3e} {r[2]:.3e} {r[3]:.3e} {float(diff.max() / t64.abs().max()):.3e} {int(diff.argmax())}\n")
    worst = max(rows, key=lambda r: r[1])
    print(f"{tag} [{inr_mode}]: worst grad err vs fp64 {worst[1]:.3e} (reference fp32 vs fp64 {worst[2]:.3e}, hip vs ref32 "
          f"{worst[3]:.3e}) at {worst[0]}; params checked {len(rows)}")
    assert not bad, bad


def test_generator_rng_draw_order_matches_reference_shapes():
    """Same-device seed parity contract: the wrapper must issue the reference's draws in order."""
    d = torch.device("cuda:0")
    G = seeded_generator(0, device=d)
    calls = []
    o_rand, o_randn = torch.rand, torch.randn

    def rand(*a, **k):
        calls.append(("rand", tuple(a[0]) if isinstance(a[0], (tuple, list, torch.Size)) else a)); return o_rand(*a, **k)

    def randn(*a, **k):
        calls.append(("randn", tuple(a[0]) if isinstance(a[0], (tuple, list, torch.Size)) else a)); return o_randn(*a, **k)

    torch.rand, torch.randn = rand, randn
    try:
        zs = G.get_zs(2)
        with torch.no_grad():
            G(zs, img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=4, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=True, sample_dist="gaussian", nerf_noise=0.)
    finally:
        torch.rand, torch.randn = o_rand, o_randn
    assert calls == [("randn", (2, 256)), ("randn", (2, 512)), ("rand", (2, 64, 4, 1)), ("randn", (2, 1)),
                     ("randn", (2, 1)), ("randn", (2, 64, 4, 1)), ("rand", (128, 4)), ("randn", (2, 64, 8, 1))]


def test_generator_r64_vs_oracle_forward(inr_mode):
    """Headline geometry (64^2, S=24 flat and S=12 hierarchical) at b=1 against the CPU oracle."""
    d = torch.device("cuda:0")
    for S, hier in [(24, False), (12, True)]:
        G = seeded_generator(1234)
        g = torch.Generator().manual_seed(77)
        b, n = 1, 64 * 64
        E = 2 * S if hier else S
        zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
        rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g),
                    phi=torch.randn(b, 1, generator=g), noise_c=torch.randn(b, n, S, 1, generator=g),
                    u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, E, 1, generator=g))
        with torch.no_grad():
            ref = orc.generator_forward(dict(G.named_parameters()), zs, rand, 64, 12, 0.88, 1.12, S, 0.3, 0.155, hier)
        Gd = G.to(d)
        with torch.no_grad():
            imgs, py = Gd({k: v.to(d) for k, v in zs.items()}, img_size=64, fov=12, ray_start=0.88, ray_end=1.12,
                          num_steps=S, h_stddev=0.3, v_stddev=0.155, hierarchical_sample=hier, sample_dist="gaussian",
                          rand_override={k: v.to(d) for k, v in rand.items()})
        e = max_rel(imgs, ref["imgs"])
        print(f"r64 S={S} hier={hier} [{inr_mode}]: imgs max_rel {e:.3e}")
        assert e < TOL


@pytest.mark.parametrize("S,hier", [(24, False), (12, True)])
def test_generator_full_size_properties(S, hier):
    """BASELINE configuration C2 at its full size (64^2, 24 SIREN evaluations per ray, 32 images, default numeric
    mode), through properties that need no oracle run:
      * determinism: the same inputs give bit-identical images and parameter gradients (no atomics anywhere),
      * batch-sharding invariance (what the data-parallel path of SURVEY.md §8e relies on): images 8..15 rendered on
        their own equal rows 8..15 of the batch and the four quarter-batch gradients sum to the batch gradient — to
        fp32 summation-order noise, not bitwise: another batch size selects other GEMM tilings (hipBLASLt in the
        mapping MLPs); for the gradients that noise is amplified by LeakyReLU gate flips, see the comment at the
        assertion,
      * linearity of the backward in the upstream gradient: doubling it doubles every parameter gradient exactly
        (a power of two commutes with fp32 rounding and with the hi/lo bf16 split)."""
    d = torch.device("cuda:0")
    G = seeded_generator(1234, device=d)
    g = torch.Generator().manual_seed(5)
    b, n = 32, 64 * 64
    E = 2 * S if hier else S
    zs = {"z_nerf": torch.randn(b, 256, generator=g).to(d), "z_inr": torch.randn(b, 512, generator=g).to(d)}
    rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g),
                phi=torch.randn(b, 1, generator=g), noise_c=torch.randn(b, n, S, 1, generator=g),
                u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, E, 1, generator=g))
    rand = {k: v.to(d) for k, v in rand.items()}
    G0 = (torch.randn(b, 3, 64, 64, generator=g) / (b * 3 * 64 * 64)).to(d)
    params = [p for p in G.parameters()]

    def run(sl, scale=1.0):
        rs = {k: (v[sl] if k != "u" else v.view(b, n, S)[sl].reshape(-1, S)) for k, v in rand.items()}
        for p in params:
            p.grad = None
        imgs, _ = G({k: v[sl] for k, v in zs.items()}, img_size=64, fov=12, ray_start=0.88, ray_end=1.12, num_steps=S,
                    h_stddev=0.3, v_stddev=0.155, hierarchical_sample=hier, sample_dist="gaussian", rand_override=rs)
        imgs.backward(G0[sl] * scale)
        return imgs.detach().clone(), [None if p.grad is None else p.grad.clone() for p in params]

    full = slice(0, b)
    im1, g1 = run(full)
    im2, g2 = run(full)
    assert torch.isfinite(im1).all() and im1.abs().max() <= 1.0
    assert torch.equal(im1, im2), "forward is not deterministic"
    for a_, b_ in zip(g1, g2):
        assert (a_ is None) == (b_ is None)
        if a_ is not None:
            assert torch.equal(a_, b_), "backward is not deterministic"
    _, g3 = run(full, scale=2.0)
    for a_, c_ in zip(g1, g3):
        if a_ is not None:
            assert torch.equal(a_ * 2, c_), "backward is not linear in the upstream gradient"
    acc = [None if x is None else torch.zeros_like(x, dtype=torch.float64) for x in g1]
    for q in range(4):
        sl = slice(8 * q, 8 * q + 8)
        imq, gq = run(sl)
        e = max_rel(imq, im1[sl])
        assert e < 2e-5, f"images {8 * q}..{8 * q + 7} depend on the rest of the batch ({e:.2e})"
        for a_, x in zip(acc, gq):
            if a_ is not None:
                a_ += x.double()
    worst, rows = 0.0, []
    names = [k for k, _ in G.named_parameters()]
    for nm, a_, x in zip(names, acc, g1):
        if a_ is not None:
            e = rel_err(a_.float(), x)
            rows.append((e, nm))
            worst = max(worst, e)
    # Gradient tolerance: the mapping MLPs (hipBLASLt picks another kernel for 8 rows than for 32) move the styles by
    # ~1e-7 and the images by ~1e-5 (asserted above at 2e-5).  At that perturbation about 1e-5 of the 6.7e7 LeakyReLU
    # gates per layer sit on the other side of zero, each changing one term of a heavily cancelling sum:
    # measured 5-6e-3 on every parameter alike.  A dropped image, chunk or partial sum would
    # show up at >= 0.15.
    print(f"   worst parameters: {sorted(rows, reverse=True)[:3]}")
    print(f"C2 full size S={S} hier={hier}: sum of quarter-batch gradients vs batch gradient, worst rel err {worst:.2e}")
    assert worst < 2e-2


@pytest.mark.parametrize("tag", ["g_r8_eval_psi_staged", "g_r8_eval_camera", "g_r8_eval_camera_staged"])
def test_generator_eval_paths_match_reference_golden(tag, inr_mode):
    """Inference path against vectors minted from the reference: psi truncation through
    generate_avg_frequencies, the staged forward (forward_points, ragged last chunk) with its per-image / per-chunk draw
    order, last_back / white_back / softplus, forward_camera_pos_and_lookup (one-shot: up_vector dropped like the
    reference does; staged: honoured)."""
    from test_oracle_golden import eval_avg_styles
    fix = load_golden(tag)
    d = torch.device("cuda:0")
    G = seeded_generator(fix["seed"], device=d)
    check_checksums(G.state_dict(), fix["state_checksums"])
    kw = dict(fix["G_kwargs"])
    zs = {k: v.to(d) for k, v in fix["zs"].items()}
    rand = {k: v.to(d) for k, v in fix["rand"].items()}
    az = eval_avg_styles(fix, seeded_generator(fix["seed"]))
    if az is not None:                      # the 10 000 averaging latents the reference drew (regenerated on the CPU)
        real_get_zs = G.get_zs
        G.get_zs = lambda n, **k: {k_: v.to(d) for k_, v in az.items()} if n == 10000 else real_get_zs(n, **k)
    common = dict(img_size=fix["img_size"], nerf_noise=fix["nerf_noise"], return_aux_img=fix["aux"], grad_points=None,
                  forward_points=fix["forward_points"], rand_override=rand)
    with torch.no_grad():
        if fix["camera"] is not None:
            cam = {k: v.to(d) for k, v in fix["camera"].items()}
            imgs, py = G.forward_camera_pos_and_lookup(zs, **common, **kw, **cam)
        else:
            imgs, py = G(zs, **common, **kw)
    if az is not None:
        ref = fix["avg"]["styles"]
        for k, v in G.avg_styles.items():
            assert max_rel(v, ref[k]) < 1e-4, k
    e = max_rel(imgs, fix["imgs"])
    print(f"{tag} [{inr_mode}]: imgs max_rel {e:.3e}")
    assert imgs.shape == fix["imgs"].shape and e < TOL
    assert max_rel(py, fix["pitch_yaw"]) < 1e-5 or float(fix["pitch_yaw"].abs().max()) == 0.0
    assert torch.equal(py.cpu() == 0, fix["pitch_yaw"] == 0)


def test_generator_camera_distributions_run():
    """Every camera distribution of comm_utils.sample_camera_positions goes through the HIP path (one-shot and staged);
    the default sample_dist=None is the reference's `assert 0`."""
    d = torch.device("cuda:0")
    G = seeded_generator(2, device=d)
    kw = dict(img_size=8, fov=12, ray_start=0.88, ray_end=1.12, num_steps=4, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=True, nerf_noise=0.)
    zs = G.get_zs(2)
    with torch.no_grad():
        for mode in ["uniform", "normal", "hybrid", "truncated_gaussian", "spherical_uniform", "mean"]:
            for fp in (None, 40):
                imgs, py = G(zs, sample_dist=mode, forward_points=fp, **kw)
                assert imgs.shape == (2, 3, 8, 8) and torch.isfinite(imgs).all() and py.shape == (2, 2), (mode, fp)
                if mode == "mean":
                    assert torch.allclose(py, torch.full_like(py, math.pi / 2))
        with pytest.raises(AssertionError):
            G(zs, **kw)
