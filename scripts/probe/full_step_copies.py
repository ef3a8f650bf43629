"""Who issues the ~140 device copies (__amd_rocclr_copyBuffer) of one full GAN step: aten::copy_ / _to_copy calls by Python stack."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    bench.full_gan_step(dev, 32, 64, 12, steps=1, warmup=1)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::tensor", "aten::lift_fresh", "aten::scalar_tensor"):
        st = [s for s in (e.stack or []) if ("cips3d_amd" in s or "bench.py" in s)]
        if not any("d_step" in s or "g_step" in s for s in st) :
            chain=[]; p=e.cpu_parent
            while p is not None and len(chain)<3: chain.append(p.name); p=p.cpu_parent
            if not any("autograd::engine" in c or "Backward" in c for c in chain): continue
            key=(e.name, str(e.input_shapes)[:60], " <- ".join(chain)[:70])
        else:
            key = (e.name, str(e.input_shapes)[:60], " | ".join(s.split("/")[-1][:70] for s in st[:2]))
        cnt[key] += 1
print("copy-like ATen calls per step (2 steps profiled):")
for k, c in cnt.most_common(50):
    print(f"{c / 2:6.1f}  {k[0]:18s} {k[1]:60s} {k[2]}")
