"""Which torch ops make up the "glue" of one eager D step (R1 every step) at C2?  Every kernel launched by an aten op is
attributed to (kernel family, aten op, the chain of enclosing ops / autograd nodes), summed over the step."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
from bench import G_CFG, G_KW
from cips3d_amd.generator import GeneratorNerfINR
from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
dev = torch.device("cuda:0")
torch.manual_seed(0)
b, img = 32, 64
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    d_step()
    torch.cuda.synchronize()
cnt = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU" or not e.kernels or not e.name.startswith("aten::"):
        continue
    if any(c.name.startswith("aten::") and c.kernels for c in (e.cpu_children or [])):
        continue                                  # count the kernels at the innermost aten op only
    us = sum(k.duration for k in e.kernels)
    chain, q = [], e.cpu_parent
    while q is not None:
        n = q.name
        if not n.startswith("aten::"):
            chain.append(n.replace("autograd::engine::evaluate_function: ", "")[:40])
        q = q.cpu_parent
    key = (e.name, str(e.input_shapes)[:44], " < ".join(chain[:2]))
    cnt[key][0] += len(e.kernels); cnt[key][1] += us
    tot[e.name][0] += len(e.kernels); tot[e.name][1] += us
print("== by aten op")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{c:5d} {us/1e3:8.2f} ms  {n}")
print("== by (op, shapes, enclosing node)")
for (n, shp, ch), (c, us) in sorted(cnt.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{c:4d} {us/1e3:7.2f} ms  {n:24s} {shp:44s} {ch}")
print("total aten-kernel ms", sum(v[1] for v in tot.values()) / 1e3)
