"""Can a small kernel run WHILE the fused SIREN backward holds every CU (160 KiB LDS, 472 of 512 registers per SIMD lane)?
Side stream: N launches of (a) an LDS-free elementwise kernel (ATen add_ on 32x512), (b) an LDS-using row kernel (cips rownorm),
timed alone and while the main stream runs the SIREN backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
b, P = 32, 64 * 64 * 24
def r(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(d).requires_grad_(True)
pts = ((torch.rand(b, P, 3, generator=g) - 0.5) * 0.24).to(d)
g0, g1, gc = [(30 + 5 * torch.randn(b, n, generator=g)).to(d).requires_grad_(True) for n in (128, 128, 64)]
p0, p1, pc = r(b, 128), r(b, 128), r(b, 64)
w0 = r(128, 3, scale=0.3); b0 = r(128, scale=0.1); w1 = r(128, 128, scale=0.01); b1 = r(128, scale=0.1)
ws = r(1, 128, scale=0.01); bs = r(1, scale=0.1); wc = r(64, 128, scale=0.01)
bc = r(64, scale=0.1); wf = r(32, 64, scale=0.05); bf = r(32, scale=0.1)
ops.TRIG_MODE = 1
args = (pts, g0, p0, g1, p1, gc, pc, w0, b0, w1, b1, ws, bs, wc, bc, wf, bf)
df = torch.randn(b, P, 32, device=d); ds = torch.randn(b, P, device=d)
feat, sig = ops.SirenFunction.apply(*args)
def bwd():
    torch.autograd.backward([feat, sig], [df, ds], retain_graph=True)
for _ in range(2): bwd()
torch.cuda.synchronize()
side = torch.cuda.Stream()
x = torch.zeros(32, 512, device=d)
big = torch.zeros(32 * 1024 * 1024 // 4, device=d)       # 32 MB elementwise: a streaming kernel without LDS
gam, bet = torch.ones(512, device=d), torch.zeros(512, device=d)
def small_free(n):
    for _ in range(n): x.add_(1.0)
def small_lds(n):
    with torch.no_grad():
        for _ in range(n): ops.RowNormFunction.apply(x, gam, bet, 3)
def stream_free(n):
    for _ in range(n): big.add_(1.0)
def run(fn, n, with_bwd):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_bwd:
        m0.record(); bwd(); m1.record()
    with torch.cuda.stream(side):
        e0.record(); fn(n); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, (m0.elapsed_time(m1) * 1e3 if with_bwd else 0.0)
for name, fn, n in (("LDS-free 32x512 add_", small_free, 100), ("row kernel with LDS", small_lds, 100), ("LDS-free 32 MB add_", stream_free, 20)):
    run(fn, n, False)
    a, _ = run(fn, n, False)
    c, mb = run(fn, n, True)
    print(f"{name:24s} x{n}: alone {a:8.1f} us   beside the SIREN backward {c:8.1f} us   (backward itself {mb:8.1f} us)")
_, mb = run(lambda n: None, 0, True)
print(f"SIREN backward alone: {mb:.1f} us")
