"""GPU: the drop-in generator (HIP path through the C-ABI) against the golden vectors minted
from the reference and against the CPU oracle, forward and backward."""
import pytest
import torch

from conftest import load_golden, seeded_generator, check_checksums, max_rel, rel_err, G_CFG
from oracle import cips3d_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-3
CASES = ["g_r16_hier", "g_r8_flat_noise", "g_r8_hier_noise", "g_r8_freeze"]


@pytest.mark.parametrize("tag", CASES)
def test_generator_matches_reference_golden(tag):
    fix = load_golden(tag)
    d = torch.device("cuda:0")
    G = seeded_generator(fix["seed"], freeze=fix["freeze"], device=d)
    check_checksums({k: v.cpu() for k, v in G.state_dict().items()}, fix["state_checksums"])
    zs = {k: v.to(d) for k, v in fix["zs"].items()}
    rand = {k: v.to(d) for k, v in fix["rand"].items()}
    imgs, pitch_yaw = G(zs, img_size=fix["img_size"], nerf_noise=fix["nerf_noise"], return_aux_img=fix["aux"],
                        grad_points=None, forward_points=None, rand_override=rand, **fix["G_kwargs"])
    torch.cuda.synchronize()
    assert imgs.shape == fix["imgs"].shape
    e = max_rel(imgs, fix["imgs"])
    print(f"{tag}: imgs max_rel vs reference {e:.3e}")
    assert e < TOL
    assert max_rel(pitch_yaw, fix["pitch_yaw"]) < 1e-5
    (imgs * fix["G0"].to(d)).sum().backward()
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for name, p in G.named_parameters():
        dg = fix["grads"][name]
        if dg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        g = p.grad.reshape(-1).cpu()
        got = g[::dg["stride"]] if dg["stride"] > 1 else g
        en = abs(float(g.double().norm()) - dg["norm"]) / max(dg["norm"], 1e-30)
        es = float((got - dg["sample"]).double().norm() / dg["sample"].double().norm().clamp_min(1e-30))
        if es > worst[1]:
            worst = (name, es)
        assert en < TOL and es < 5 * TOL, (name, en, es)
    print(f"{tag}: worst grad sample rel err {worst[1]:.3e} at {worst[0]}")


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


def test_generator_r64_vs_oracle_forward():
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
        print(f"r64 S={S} hier={hier}: imgs max_rel {e:.3e}")
        assert e < TOL
