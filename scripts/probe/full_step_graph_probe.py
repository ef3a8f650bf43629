"""The full GAN step of bench.full_gan_step (D step with R1 + G step + fused clip / Adam / EMA, C2 geometry) eager against ONE
hipGraph of the whole step (cips3d_amd.graph.capture), same process."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from bench import G_CFG, G_KW
from cips3d_amd.generator import GeneratorNerfINR
from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
from cips3d_amd.optim import FusedClipAdamEMA
from cips3d_amd.graph import capture
dev = torch.device("cuda:0")
b, img, S = 32, 64, 12
torch.manual_seed(1234)
G = GeneratorNerfINR(**G_CFG, device=dev).to(dev); G.device = dev
G_ema = copy.deepcopy(G)
D = Discriminator_MultiScale_Aux(diffaug=False, max_size=1024, channel_multiplier=2, first_downsample=False, stddev_group=0).to(dev)
kw = dict(G_KW); kw.update(num_steps=S, hierarchical_sample=True)
oG = FusedClipAdamEMA(G.parameters(), lr=2e-4, betas=(0.0, 0.999), max_norm=10.0, ema_params=G_ema.parameters(), capture_slots=4)
oD = FusedClipAdamEMA(D.parameters(), lr=2e-3, betas=(0.0, 0.999), max_norm=10.0, capture_slots=4)
real = torch.rand(b, 3, img, img, device=dev) * 2 - 1


def d_step():
    for p in G.parameters(): p.requires_grad_(False)
    for p in D.parameters(): p.requires_grad_(True)
    with torch.no_grad():
        gen, _ = G(G.get_zs(b), img_size=img, nerf_noise=0.5, return_aux_img=True, forward_points=None, grad_points=None, **kw)
    real2 = torch.cat([real, real]).requires_grad_(True)
    r_preds, _, _ = D(real2, alpha=1.0, use_aux_disc=True)
    grad_real, = torch.autograd.grad(outputs=r_preds.sum(), inputs=real2, create_graph=True)
    pen = 0.5 * 10.0 * grad_real.flatten(1).square().sum(1, keepdim=True)
    g_preds, _, _ = D(gen, alpha=1.0, use_aux_disc=True)
    loss = (F.softplus(g_preds) + F.softplus(-r_preds) + pen).mean()
    for p in D.parameters(): p.grad = None
    loss.backward()
    oD.step()


def g_step():
    for p in G.parameters(): p.requires_grad_(True)
    for p in D.parameters(): p.requires_grad_(False)
    imgs, _ = G(G.get_zs(b), img_size=img, nerf_noise=0.5, return_aux_img=True, grad_points=None, forward_points=None, **kw)
    preds, _, _ = D(imgs, alpha=1.0, use_aux_disc=True)
    loss = F.softplus(-preds).mean()
    for p in G.parameters(): p.grad = None
    loss.backward()
    oG.step()


def step():
    d_step(); g_step()


def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_e = timeit(step)
print(f"eager: {t_e:.2f} ms per full GAN step", flush=True)
for p in list(G.parameters()) + list(D.parameters()): p.grad = None
cs = capture(step, warmup=1)
t_g = timeit(cs)
print(f"one hipGraph: {t_g:.2f} ms per full GAN step", flush=True)
t_e2 = timeit(step)
print(f"eager again: {t_e2:.2f} ms", flush=True)
