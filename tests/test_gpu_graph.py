"""cips3d_amd.graph.CapturedStep: a captured generator forward + backward replays to the eager step's gradients."""
import pytest
import torch

from conftest import load_golden, seeded_generator

pytestmark = pytest.mark.gpu


def test_captured_generator_step_replays_the_eager_gradients():
    from cips3d_amd.graph import capture
    fix = load_golden("g_r8_flat_noise")
    d = torch.device("cuda:0")
    G = seeded_generator(fix["seed"], freeze=fix["freeze"], device=d)
    zs = {k: v.to(d) for k, v in fix["zs"].items()}
    rand = {k: v.to(d) for k, v in fix["rand"].items()}
    G0 = fix["G0"].to(d)
    params = [p for p in G.parameters() if p.requires_grad]
    img_buf = torch.zeros_like(fix["imgs"], device=d)     # static output buffer: the closure copies into it (a tensor the closure
                                                          # merely keeps a reference to would be freed by the next call)

    def step():
        for p in params:
            p.grad = None
        imgs, _ = G(zs, img_size=fix["img_size"], nerf_noise=fix["nerf_noise"], return_aux_img=fix["aux"], grad_points=None,
                    forward_points=None, rand_override=rand, **fix["G_kwargs"])
        (imgs * G0).sum().backward()
        img_buf.copy_(imgs.detach())

    step()
    torch.cuda.synchronize()
    eager_imgs = img_buf.clone()
    eager = [None if p.grad is None else p.grad.detach().clone() for p in params]
    cs = capture(step, warmup=1, params=params)
    for _ in range(2):                       # the second replay overwrites the first one's results in place
        cs()
    torch.cuda.synchronize()
    assert torch.equal(img_buf, eager_imgs)
    for p, g in zip(params, eager):
        assert (p.grad is None) == (g is None)
        if g is not None:
            assert torch.allclose(p.grad, g, rtol=1e-6, atol=1e-9), float((p.grad - g).abs().max())


def test_captured_discriminator_step_with_the_auxiliary_stream_replays_the_eager_gradients():
    """Discriminator_MultiScale_Aux forks its auxiliary network onto a side stream inside forward(): under capture the fork and the
    join (and autograd's per-stream backward, R1 double backward included) must all land in the one graph."""
    import torch.nn.functional as F
    from cips3d_amd.graph import capture
    from cips3d_amd import discriminator as dmod
    from conftest import D_CFG
    d = torch.device("cuda:0")
    assert dmod.AUX_SIDE_STREAM
    torch.manual_seed(3)
    D = dmod.Discriminator_MultiScale_Aux(**D_CFG).to(d)
    g = torch.Generator().manual_seed(1)
    x_static = torch.randn(4, 3, 16, 16, generator=g).to(d)
    params = list(D.parameters())

    def step():
        for p in params:
            p.grad = None
        x = x_static.detach().requires_grad_(True)
        out, _, _ = D(x, alpha=1.0, use_aux_disc=True)
        gr, = torch.autograd.grad(outputs=out.sum(), inputs=x, create_graph=True)
        (F.softplus(-out).mean() + 5.0 * gr.flatten(1).pow(2).sum(1).mean()).backward()

    step()
    torch.cuda.synchronize()
    eager = [None if p.grad is None else p.grad.detach().clone() for p in params]
    cs = capture(step, warmup=1, params=params)
    cs(); cs()
    torch.cuda.synchronize()
    n = 0
    for p, e in zip(params, eager):
        assert (p.grad is None) == (e is None)
        if e is not None:
            n += 1
            assert torch.allclose(p.grad, e, rtol=1e-5, atol=1e-8), float((p.grad - e).abs().max())
    assert n > 20
