"""Fused march (C2, no_grad and training forward) under one probe library: python march_desync.py <lib> ; CIPS_X3_MDESYNC from the env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from bench import G_CFG
from cips3d_amd.generator import GeneratorNerfINR
d = torch.device("cuda:0"); torch.manual_seed(0)
G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
b, img, S = 32, 64, 24; n = img * img
style = {k: torch.randn(b, 128, device=d) for k in G.siren.style_dim_dict}
xg = torch.linspace(-1, 1, img, device=d); yg = torch.linspace(1, -1, img, device=d); zg = torch.linspace(0.88, 1.12, S, device=d)
zc = -1.0 / float(torch.tan(torch.tensor(3.14159265 * 12 / 360)))
c2w = torch.eye(4, device=d).repeat(b, 1, 1); c2w[:, 2, 3] = 1.0
jit = torch.rand(b, n, S, device=d)
def ng():
    with torch.no_grad(): return G.siren.march(style, (b, img, img, S, zc, 0.0, 0, 0, False), xg, yg, zg, c2w, jit, None)
def tr():
    return G.siren.march(style, (b, img, img, S, zc, 0.0, 0, 0, True), xg, yg, zg, c2w, jit, None)
def timeit(fn, reps=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
fea = ng()[0]
print(f"lib {os.path.basename(os.path.dirname(sys.argv[1]))} desync {os.environ.get('CIPS_X3_MDESYNC', '0'):>6}  no_grad {min(timeit(ng) for _ in range(3)):7.1f} us  train fwd {min(timeit(tr) for _ in range(3)):7.1f} us  checksum {float(fea.double().sum()):.6f}", flush=True)
