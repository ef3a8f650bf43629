"""One eager D step (R1 every step, auxiliary discriminator) at C2 shapes, N times: the launch set rocprofv3 traces for
scripts/probe/trace_by_grid.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
dev = torch.device("cuda:0")
torch.manual_seed(0)
b, img = 32, 64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D = Discriminator_MultiScale_Aux(diffaug=False, max_size=1024, channel_multiplier=2, first_downsample=False, stddev_group=0).to(dev)
real = torch.rand(b, 3, img, img, device=dev) * 2 - 1
gen = torch.rand(2 * b, 3, img, img, device=dev) * 2 - 1
def d_step():
    real2 = torch.cat([real, real]).requires_grad_(True)
    r_preds, _, _ = D(real2, alpha=1.0, use_aux_disc=True)
    grad_real, = torch.autograd.grad(outputs=r_preds.sum(), inputs=real2, create_graph=True)
    pen = 0.5 * 10.0 * grad_real.flatten(1).square().sum(1, keepdim=True)
    g_preds, _, _ = D(gen, alpha=1.0, use_aux_disc=True)
    loss = (F.softplus(g_preds) + F.softplus(-r_preds) + pen).mean()
    for p in D.parameters(): p.grad = None
    loss.backward()
for _ in range(2): d_step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n): d_step()
torch.cuda.synchronize()
print(f"D step (fwd real + R1 + fwd fake + backward, no optimizer): {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
