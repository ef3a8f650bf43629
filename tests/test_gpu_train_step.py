"""GPU: one full GAN step through the drop-in modules exactly as exp/cips3d/scripts/train.py drives them
(D step with R1 double-backward on real images, G step through the frozen D), checking that every parameter the
reference would update receives a finite gradient and that an Adam step changes the outputs."""
import pytest
import torch

from conftest import G_CFG, D_CFG

pytestmark = pytest.mark.gpu


def test_one_gan_step_like_train_py():
    from cips3d_amd.generator import GeneratorNerfINR
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
    D = Discriminator_MultiScale_Aux(**D_CFG).to(d)
    opt_G = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.0, 0.999))
    opt_D = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.0, 0.999))
    kw = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=6, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=True, psi=1., sample_dist="gaussian")
    b, img = 2, 16
    real = (torch.rand(b, 3, img, img, device=d) * 2 - 1)
    # ---- D step (train.py:334-437): G under no_grad with aux image, R1 on reals ----
    for p in G.parameters(): p.requires_grad_(False)
    with torch.no_grad():
        zs = G.get_zs(b)
        gen, _ = G(zs, img_size=img, nerf_noise=0.5, return_aux_img=True, forward_points=None, grad_points=None, **kw)
    assert gen.shape == (2 * b, 3, img, img) and torch.isfinite(gen).all()
    real2 = torch.cat([real, real]).requires_grad_(True)
    r_preds, _, _ = D(real2, alpha=1.0, use_aux_disc=True)
    grad_real, = torch.autograd.grad(outputs=r_preds.sum(), inputs=real2, create_graph=True)
    pen = grad_real.flatten(1).pow(2).sum(1)
    g_preds, _, _ = D(gen, alpha=1.0, use_aux_disc=True)
    d_loss = (torch.nn.functional.softplus(g_preds) + torch.nn.functional.softplus(-r_preds) + 0.5 * 10. * pen.view(-1, 1)).mean()
    opt_D.zero_grad(); d_loss.backward()
    used = [n for n, p in D.named_parameters() if p.grad is not None]
    assert all(torch.isfinite(p.grad).all() for p in D.parameters() if p.grad is not None)
    assert any("aux_disc.convs.16" in n for n in used) and any("main_disc.convs.16" in n for n in used)
    torch.nn.utils.clip_grad_norm_(D.parameters(), 10.)
    opt_D.step()
    # ---- G step (train.py:440-491) ----
    for p in G.parameters(): p.requires_grad_(True)
    for p in D.parameters(): p.requires_grad_(False)
    zs = G.get_zs(b)
    imgs, _ = G(zs, img_size=img, nerf_noise=0.5, return_aux_img=True, grad_points=img * img, forward_points=None, **kw)
    preds, _, _ = D(imgs, alpha=1.0, use_aux_disc=True)
    g_loss = torch.nn.functional.softplus(-preds).mean()
    opt_G.zero_grad(); g_loss.backward()
    missing = [n for n, p in G.named_parameters() if p.grad is None and "norm" not in n and not any(f"to_rgbs.{k}." in n for k in ("4", "8", "16"))]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in G.parameters() if p.grad is not None)
    with torch.no_grad():
        before = G.siren.network[1].linear.weight.clone()
    torch.nn.utils.clip_grad_norm_(G.parameters(), 10.)
    opt_G.step()
    assert not torch.equal(before, G.siren.network[1].linear.weight)
