"""Which torch ops launch the aten kernels of the full GAN step at C2 (bench.full_gan_step: 1 warm-up + 2 steps under the torch
profiler; counts are per step)?  By aten op, input shapes, enclosing autograd node and first cips3d_amd / bench stack frame."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
bench.full_gan_step(dev, 32, 64, 12, steps=1, warmup=1)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.full_gan_step(dev, 32, 64, 12, steps=N - 1, warmup=1)
    torch.cuda.synchronize()
cnt = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
allk = 0
for e in prof.events():
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    if any(c.kernels for c in (e.cpu_children or [])):
        continue
    allk += len(e.kernels)
    if not e.name.startswith("aten::"):
        continue
    chain, q = [], e.cpu_parent
    while q is not None:
        if not q.name.startswith("aten::"):
            chain.append(q.name.replace("autograd::engine::evaluate_function: ", "")[:34])
        q = q.cpu_parent
    st = [s for s in (e.stack or []) if "cips3d_amd" in s or "bench.py" in s]
    key = (e.name, str(e.input_shapes)[:38], " < ".join(chain[:2]), st[0][-58:] if st else "")
    us = sum(k.duration for k in e.kernels)
    cnt[key][0] += len(e.kernels); cnt[key][1] += us
    tot[e.name][0] += len(e.kernels); tot[e.name][1] += us
print(f"all kernel launches per step (incl. the construction / seeding of the step's modules once): {allk / N:.0f}")
print("== aten kernels by op, per step")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:20]:
    print(f"{c / N:7.1f} {us / 1e3 / N:7.2f} ms  {n}")
print("== by (op, shapes, node, frame), per step")
for (n, shp, ch, st), (c, us) in sorted(cnt.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{c / N:6.1f} {us / N:7.1f} us  {n:20s} {shp:38s} {ch:36s} {st}")
