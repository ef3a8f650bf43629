"""which torch ops issue the device-to-device memcpys (and fills) of one eager bench step?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import G_CFG, G_KW
from cips3d_amd.generator import GeneratorNerfINR
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = GeneratorNerfINR(**G_CFG, device=dev).to(dev); G.device = dev
b, img, S = 32, 64, 24
G0 = torch.randn(b, 3, img, img, device=dev) / (b * 3 * img * img)
params = list(G.parameters())
def step():
    zs = G.get_zs(b)
    for p in params: p.grad = None
    imgs, _ = G(zs, img_size=img, num_steps=S, hierarchical_sample=False, nerf_noise=0., return_aux_img=False, grad_points=None, forward_points=None, **G_KW)
    imgs.backward(G0)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.kernels:
        continue
    kn = [k.name for k in e.kernels]
    tag = "memcpy" if any("copyBuffer" in k or "Memcpy" in k for k in kn) else "fill" if any("FillFunctor" in k for k in kn) else None
    if tag is None:
        continue
    chain, q = [], e.cpu_parent
    while q is not None:
        chain.append(q.name[:48]); q = q.cpu_parent
    cnt[(tag, e.name, str(e.input_shapes)[:50], " < ".join(chain[:4]))] += 1
for (tag, n, shp, ch), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d} {tag:6s} {n:22s} {shp:50s} {ch}")
print("total", sum(cnt.values()))
