"""ATen launches and device copies left in the full GAN step (C2): torch.profiler over 2 steps of bench.full_gan_step, grouped by
op + parent ops + the first cips3d_amd / bench frames of the Python stack."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.full_gan_step(dev, 32, 64, 12, steps=1, warmup=1)
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        st = [s for s in (e.stack or []) if ("cips3d_amd" in s or "bench.py" in s)]
        in_step = any("d_step" in s or "g_step" in s for s in st)
        chain = []; p = e
        while p is not None and len(chain) < 5:
            chain.append(p.name); p = p.cpu_parent
        is_engine = any("autograd::engine" in c for c in chain)
        if not (in_step or is_engine): continue
        if not (chain[0].startswith("aten::") or "emcpy" in chain[0] or "copy" in chain[0].lower()): continue
        key = (chain[0], " <- ".join(chain[1:4])[:80], " | ".join(s.split("/")[-1][:60] for s in st[:3]))
        cnt[key] += len(e.kernels); tim[key] += sum(k.duration for k in e.kernels)
print(f"{sum(cnt.values()) / 2:.0f} ATen launches / copies per step (2 steps profiled)")
for key, c in cnt.most_common(60):
    print(f"{c / 2:6.1f} {tim[key] / 2e3:7.3f} ms  {key[0]:24s} <- {key[1]:80s} {key[2]}")
