"""Which ATen launches remain in one eager D step (C2 shapes) and who issues them: torch.profiler, grouped by op name + Python stack."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    d_step()
    torch.cuda.synchronize()
ev = prof.events()
# every CPU op that directly launched a device kernel / memcpy: attribute to op name + nearest ancestors
byid = {}
cnt = collections.Counter(); tim = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        # climb to the outermost aten op / autograd node
        chain = []
        p = e
        while p is not None and len(chain) < 6:
            chain.append(p.name)
            p = p.cpu_parent
        st = [s for s in (e.stack or []) if "cips3d_amd" in s or "d_step_aten" in s][:2]
        key = (chain[0], " <- ".join(chain[1:4]), " | ".join(s.split("/")[-1] for s in st))
        cnt[key] += len(e.kernels)
        tim[key] += sum(k.duration for k in e.kernels)
tot = sum(cnt.values())
print(f"{tot} device launches in one D step")
for key, c in cnt.most_common(45):
    if key[0].startswith("aten::") or "Memcpy" in key[0] or "copy" in key[0].lower():
        print(f"{c:5d} {tim[key] / 1e3:8.3f} ms  {key[0]:28s} <- {key[1][:90]:90s} {key[2][:110]}")
