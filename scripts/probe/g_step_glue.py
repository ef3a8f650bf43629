"""Which torch ops launch the small aten kernels of one eager G fwd+bwd step at C2?  (kernel count and time by aten op,
input shapes and enclosing op / autograd node)"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type.name != "CPU" or not e.kernels or not e.name.startswith("aten::"):
        continue
    if any(c.name.startswith("aten::") and c.kernels for c in (e.cpu_children or [])):
        continue
    chain, q = [], e.cpu_parent
    while q is not None:
        if not q.name.startswith("aten::"):
            chain.append(q.name.replace("autograd::engine::evaluate_function: ", "")[:36])
        q = q.cpu_parent
    st = [s for s in (e.stack or []) if "cips3d_amd" in s or "probe" in s]
    key = (e.name, str(e.input_shapes)[:40], " < ".join(chain[:2]), st[0][-60:] if st else "")
    cnt[key][0] += len(e.kernels); cnt[key][1] += sum(k.duration for k in e.kernels)
for (n, shp, ch, st), (c, us) in sorted(cnt.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{c:4d} {us:7.1f} us  {n:22s} {shp:40s} {ch:40s} {st}")
print("total aten kernels", sum(v[0] for v in cnt.values()), "us", sum(v[1] for v in cnt.values()))
