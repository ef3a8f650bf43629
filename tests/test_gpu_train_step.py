"""GPU: full GAN training steps through the drop-in modules exactly as exp/cips3d/scripts/train.py:334-491 drives them —
D step (G under no_grad with the aux image, R1 penalty through the double-backward graph on the real images, gradient
clip, Adam), G step through the frozen D (clip, Adam, EMA into G_ema) — with the fused step tail
(cips3d_amd.optim.FusedClipAdamEMA), against the SAME loop on the CPU oracle with torch.optim.Adam(betas=(0, 0.999)),
torch.nn.utils.clip_grad_norm_ and the reference's EMA rule (exp/comm/comm_model_utils.py:97-118).  K = 3 steps: every
tensor of G, D and G_ema is compared after the last step.

How two Adam trajectories are compared.  With beta1 = 0 the update is lr * g / (sqrt(v_hat) + eps): for a single element it
is +-lr whatever the size of g, so an element whose gradient is within rounding of zero may move by up to 2 lr apart in
two correct fp32 evaluations (the same ill-conditioning the reference's own run has) — sparse, bounded outliers.  The bars:
  * per network (G, D, G_ema): relative L2 error over all parameters < 1e-4;
  * per tensor: ||got - ref||_2 <= 1e-4 ||ref||_2 + 0.1 K lr sqrt(numel) — a missed / doubled step, a wrong learning rate,
    clip factor or EMA coefficient moves EVERY element by a fraction of lr and breaks this at once, a handful of
    sign-fragile elements does not; the second term only matters for the zero-initialised biases;
  * per element: the trivial bound 2 K lr, and fewer than 0.2 % of all elements off by more than lr / 2.
The LeakyReLU gates of every forward are pinned to the oracle's (tests/test_gpu_generator.py explains why)."""
import copy

import pytest
import torch

from conftest import D_CFG, seeded_generator, pack_bitplane, max_rel
from oracle import cips3d_oracle as orc

pytestmark = pytest.mark.gpu
KW = dict(fov=12, ray_start=0.88, ray_end=1.12, h_stddev=0.3, v_stddev=0.155)
F = torch.nn.functional


def _draws(g, b, img, S):
    n = img * img
    zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
    rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g),
                phi=torch.randn(b, 1, generator=g), noise_c=torch.randn(b, n, S, 1, generator=g),
                u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, 2 * S, 1, generator=g))
    return zs, rand


def test_three_training_steps_follow_the_oracle_trajectory():
    from cips3d_amd import ops, discriminator as dmod
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    K, b, img, S, noise = 3, 2, 16, 6, 0.5
    lrG, lrD, clip, decay, ema_start = 2e-4, 2e-3, 10.0, 0.999, 1          # ffhq_exp.yaml:157-171; EMA from step 1 on
    g = torch.Generator().manual_seed(77)
    real = torch.rand(b, 3, img, img, generator=g) * 2 - 1
    steps = [(_draws(g, b, img, S), _draws(g, b, img, S)) for _ in range(K)]

    # ---------------------------------------------------------------- oracle trajectory (CPU)
    Gc = seeded_generator(1234)
    Gc_ema = copy.deepcopy(Gc)
    torch.manual_seed(4321)
    Dc = Discriminator_MultiScale_Aux(**D_CFG)
    Gd = copy.deepcopy(Gc).to(d); Gd.device = d
    Gd_ema = copy.deepcopy(Gd)
    Dd = copy.deepcopy(Dc).to(d)
    init = {"G": {k: v.detach().clone() for k, v in Gc.named_parameters()}, "D": {k: v.detach().clone() for k, v in Dc.named_parameters()}}
    sdG = dict(Gc.named_parameters())
    sdD = dict(Dc.state_dict()); sdD.update(dict(Dc.named_parameters()))
    oG = torch.optim.Adam(Gc.parameters(), lr=lrG, betas=(0.0, 0.999))
    oD = torch.optim.Adam(Dc.parameters(), lr=lrD, betas=(0.0, 0.999))
    def gen(sd, zs, rand):
        return orc.generator_forward(sd, zs, rand, img, KW["fov"], KW["ray_start"], KW["ray_end"], S, KW["h_stddev"],
                                     KW["v_stddev"], True, nerf_noise=noise, return_aux_img=True)["imgs"]

    pins, ref_losses = [], []
    for step, ((zs_d, rand_d), (zs_g, rand_g)) in enumerate(steps):
        # D step (train.py:334-437)
        for p in Gc.parameters(): p.requires_grad_(False)
        for p in Dc.parameters(): p.requires_grad_(True)
        t_g1, t_d = orc.GateTape(), orc.GateTape()
        with torch.no_grad(), orc.gate_tape(t_g1):
            fake = gen(sdG, zs_d, rand_d)
        x = torch.cat([real, real]).requires_grad_(True)
        with orc.gate_tape(t_d):
            rp = orc.discriminator_forward(sdD, x, alpha=1.0, use_aux_disc=True)
            gr, = torch.autograd.grad(rp.sum(), x, create_graph=True)
            fp = orc.discriminator_forward(sdD, fake, alpha=1.0, use_aux_disc=True)
        d_loss = (F.softplus(fp) + F.softplus(-rp) + 0.5 * 10. * gr.flatten(1).square().sum(1, keepdim=True) + 0. * rp).mean()
        oD.zero_grad(set_to_none=True)
        d_loss.backward()
        torch.nn.utils.clip_grad_norm_(Dc.parameters(), clip)
        oD.step()
        # G step (train.py:439-491)
        for p in Gc.parameters(): p.requires_grad_(True)
        for p in Dc.parameters(): p.requires_grad_(False)
        t_g2, t_dg = orc.GateTape(), orc.GateTape()
        with orc.gate_tape(t_g2):
            imgs = gen(sdG, zs_g, rand_g)
        with orc.gate_tape(t_dg):
            preds = orc.discriminator_forward(sdD, imgs, alpha=1.0, use_aux_disc=True)
        g_loss = F.softplus(-preds).mean()
        oG.zero_grad(set_to_none=True)
        g_loss.backward()
        torch.nn.utils.clip_grad_norm_(Gc.parameters(), clip)
        oG.step()
        if step >= ema_start:                                       # comm_model_utils.py:97-118
            with torch.no_grad():
                for e, p in zip(Gc_ema.parameters(), Gc.parameters()):
                    e.copy_(e * decay + p * (1 - decay))
        pins.append(([pack_bitplane(t) for t in t_g1.rec], list(t_d.rec), [pack_bitplane(t) for t in t_g2.rec], list(t_dg.rec)))
        ref_losses.append((float(d_loss), float(g_loss)))

    # ---------------------------------------------------------------- product trajectory (HIP path, fused step tail)
    fG = FusedClipAdamEMA(Gd.parameters(), lr=lrG, betas=(0.0, 0.999), max_norm=clip, ema_params=Gd_ema.parameters(),
                          ema_decay=decay, ema_start_itr=ema_start)
    fD = FusedClipAdamEMA(Dd.parameters(), lr=lrD, betas=(0.0, 0.999), max_norm=clip)
    gkw = dict(img_size=img, num_steps=S, hierarchical_sample=True, sample_dist="gaussian", nerf_noise=noise,
               return_aux_img=True, forward_points=None, **KW)
    up = lambda t: {k: v.to(d) for k, v in t.items()}
    for step, ((zs_d, rand_d), (zs_g, rand_g)) in enumerate(steps):
        p_g1, p_d, p_g2, p_dg = pins[step]
        for p in Gd.parameters(): p.requires_grad_(False)
        for p in Dd.parameters(): p.requires_grad_(True)
        with torch.no_grad(), ops.gate_debug(pin=p_g1):
            fake, _ = Gd(up(zs_d), grad_points=None, rand_override=up(rand_d), **gkw)
        x = torch.cat([real, real]).to(d).requires_grad_(True)
        with dmod.gate_debug(pin=p_d):
            rp = Dd(x, alpha=1.0, use_aux_disc=True)[0]
            gr, = torch.autograd.grad(rp.sum(), x, create_graph=True)
            fp = Dd(fake, alpha=1.0, use_aux_disc=True)[0]
        d_loss = (F.softplus(fp) + F.softplus(-rp) + 0.5 * 10. * gr.flatten(1).square().sum(1, keepdim=True) + 0. * rp).mean()
        fD.zero_grad()
        d_loss.backward()
        fD.step(itr=step)
        for p in Gd.parameters(): p.requires_grad_(True)
        for p in Dd.parameters(): p.requires_grad_(False)
        with ops.gate_debug(pin=p_g2):
            imgs, _ = Gd(up(zs_g), grad_points=img * img, rand_override=up(rand_g), **gkw)
        with dmod.gate_debug(pin=p_dg):
            preds = Dd(imgs, alpha=1.0, use_aux_disc=True)[0]
        g_loss = F.softplus(-preds).mean()
        fG.zero_grad()
        g_loss.backward()
        fG.step(itr=step)
        dl, gl = ref_losses[step]
        assert abs(float(d_loss) - dl) < 1e-3 * max(1.0, abs(dl)) and abs(float(g_loss) - gl) < 1e-3 * max(1.0, abs(gl)), (step, float(d_loss), dl, float(g_loss), gl)
    torch.cuda.synchronize()

    # ---------------------------------------------------------------- compare every tensor
    def compare(which, got_mod, ref_mod, lr, init_ref):
        worst, num, den, moved, flips, ntot = ("", 0.0), 0.0, 0.0, 0, 0, 0
        for (k, pg), (k2, pr) in zip(got_mod.named_parameters(), ref_mod.named_parameters()):
            assert k == k2
            a, r = pg.detach().cpu().double(), pr.detach().double()
            diff = (a - r).abs()
            n = r.numel()
            assert float(diff.max()) <= 2 * K * lr * 1.01 + 1e-6 * float(r.abs().max()), (which, k, float(diff.max()))
            l2, ref_l2 = float(diff.norm()), float(r.norm())
            bar = 1e-4 * ref_l2 + 0.1 * K * lr * n ** 0.5
            assert l2 <= bar, (which, k, l2, bar)
            if l2 / max(bar, 1e-300) > worst[1]:
                worst = (k, l2 / max(bar, 1e-300))
            num += l2 * l2; den += ref_l2 * ref_l2
            flips += int((diff > 0.5 * lr).sum()); ntot += n
            if init_ref is not None and not torch.equal(pr.detach(), init_ref[k]):
                moved += 1
        g = (num / max(den, 1e-300)) ** 0.5
        print(f"{which}: {moved} tensors updated; global relative L2 error {g:.2e}; tightest tensor at {worst[1]:.2f} of its bar "
              f"({worst[0]}); elements off by more than lr/2: {flips} of {ntot} ({flips / max(ntot, 1):.1e})")
        assert g < 1e-4, (which, g)
        assert flips <= 2e-3 * ntot, (which, flips, ntot)
        return moved

    mg = compare("G after 3 steps", Gd, Gc, lrG, init["G"])
    md = compare("D after 3 steps", Dd, Dc, lrD, init["D"])
    compare("G_ema after 3 steps", Gd_ema, Gc_ema, lrG, None)
    assert mg == 130 and md == 36, (mg, md)      # D at 16 x 16: conv_in.16, convs.16 / 8, final_conv and the linears of both branches
    # the EMA really ran (steps 1 and 2) and differs from both the initial and the current generator
    w0, w, we = init["G"]["siren.network.1.linear.weight"], Gc.siren.network[1].linear.weight.detach(), Gc_ema.siren.network[1].linear.weight.detach()
    assert not torch.equal(we, w0) and not torch.equal(we, w)
    assert max_rel(Gd_ema.siren.network[1].linear.weight, we) < 1e-5
