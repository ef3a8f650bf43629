"""Secondary measurement: one full GAN training step as exp/cips3d/scripts/train.py:334-491 drives it —
D step (G under no_grad with the aux image, R1 double-backward on the reals every `d_reg_every` steps, clip, Adam),
G step through the frozen D (clip, Adam, EMA) — on synthetic "real" images, with the fused step tail
(cips3d_amd.optim.FusedClipAdamEMA).  Prints ms per phase and images/s.  Not the headline metric."""
import argparse, copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--num-steps", type=int, default=12)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--freeze", action="store_true", help="GeneratorNerfINR_freeze_NeRF (the r256 stages, ffhq_exp.yaml:192-210)")
    ap.add_argument("--diffaug", action="store_true", help="DiffAugment in D (r256 stages)")
    ap.add_argument("--no-aux", action="store_true", help="train_aux_img False (C4)")
    ap.add_argument("--torch-optim", action="store_true", help="torch clip_grad_norm_ + Adam + python EMA instead of the fused tail")
    a = ap.parse_args()
    from bench import G_CFG, G_KW                                  # noqa: E402
    from cips3d_amd.generator import GeneratorNerfINR, GeneratorNerfINR_freeze_NeRF
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    torch.manual_seed(1234)
    G = (GeneratorNerfINR_freeze_NeRF if a.freeze else GeneratorNerfINR)(**G_CFG, device=d).to(d); G.device = d
    G_ema = copy.deepcopy(G)
    aux = not a.no_aux
    D = Discriminator_MultiScale_Aux(diffaug=a.diffaug, max_size=1024, channel_multiplier=2, first_downsample=False,
                                     stddev_group=0).to(d)
    b, img, S = a.batch, a.img_size, a.num_steps
    kw = dict(G_KW); kw.update(num_steps=S, hierarchical_sample=True)
    if a.torch_optim:
        oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.0, 0.999)); oD = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.0, 0.999))
    else:
        oG = FusedClipAdamEMA(G.parameters(), lr=2e-4, betas=(0.0, 0.999), max_norm=10.0, ema_params=G_ema.parameters())
        oD = FusedClipAdamEMA(D.parameters(), lr=2e-3, betas=(0.0, 0.999), max_norm=10.0)
    real = torch.rand(b, 3, img, img, device=d) * 2 - 1
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tD = tG = 0.0

    def d_step(it):
        for p in G.parameters(): p.requires_grad_(False)
        for p in D.parameters(): p.requires_grad_(True)
        with torch.no_grad():
            gen, _ = G(G.get_zs(b), img_size=img, nerf_noise=0.5, return_aux_img=aux, forward_points=None, grad_points=None, **kw)
        real2 = (torch.cat([real, real]) if aux else real.clone()).requires_grad_(True)
        r_preds, _, _ = D(real2, alpha=1.0, use_aux_disc=aux)
        if it % 1 == 0:                                             # d_reg_every: 1 (ffhq_exp.yaml)
            grad_real, = torch.autograd.grad(outputs=r_preds.sum(), inputs=real2, create_graph=True)
            pen = 0.5 * 10.0 * grad_real.flatten(1).square().sum(1, keepdim=True)
        else:
            pen = 0.0
        g_preds, _, _ = D(gen, alpha=1.0, use_aux_disc=aux)
        loss = (F.softplus(g_preds) + F.softplus(-r_preds) + pen).mean()
        for p in D.parameters(): p.grad = None
        loss.backward()
        if a.torch_optim:
            torch.nn.utils.clip_grad_norm_(D.parameters(), 10.0); oD.step()
        else:
            oD.step()

    def g_step(it):
        for p in G.parameters(): p.requires_grad_(True)
        for p in D.parameters(): p.requires_grad_(False)
        imgs, _ = G(G.get_zs(b), img_size=img, nerf_noise=0.5, return_aux_img=aux, grad_points=None, forward_points=None, **kw)
        preds, _, _ = D(imgs, alpha=1.0, use_aux_disc=aux)
        loss = F.softplus(-preds).mean()
        for p in G.parameters(): p.grad = None
        loss.backward()
        if a.torch_optim:
            torch.nn.utils.clip_grad_norm_(G.parameters(), 10.0); oG.step()
            with torch.no_grad():
                for e, p in zip(G_ema.parameters(), G.parameters()): e.copy_(e * 0.999 + p * 0.001)
        else:
            oG.step()

    for it in range(a.warmup):
        d_step(it); g_step(it)
    torch.cuda.synchronize()
    for it in range(a.steps):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record(); d_step(it); e1.record(); g_step(it); e2.record()
        torch.cuda.synchronize()
        tD += e0.elapsed_time(e1); tG += e1.elapsed_time(e2)
    n = a.steps
    print(json.dumps({"metric": "full GAN step (D step with R1 + G step), synthetic reals", "img_size": img, "batch": b,
                      "num_steps": S, "hierarchical": True, "aux": aux, "freeze_nerf": a.freeze, "diffaug": a.diffaug, "optimizer": "torch" if a.torch_optim else "fused clip+Adam+EMA",
                      "ms_D_step": round(tD / n, 2), "ms_G_step": round(tG / n, 2), "ms_step": round((tD + tG) / n, 2),
                      "img_per_s": round(b * n / ((tD + tG) * 1e-3), 1)}))


if __name__ == "__main__":
    main()
