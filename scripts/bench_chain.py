"""CIPS head forward at C2 (b 32, 64 x 64): one launch per layer against the chain launch (cips_gemm_bf16x3_chain), over
start-phase staggers; interleaved rounds in one process.  Prints forward-only times (training forward: planes and gate
planes written) and the head's forward + backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import G_CFG
from cips3d_amd import ops
from cips3d_amd.generator import GeneratorNerfINR

d = torch.device("cuda:0")
torch.manual_seed(0)
G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
b, n = int(os.environ.get("B", 32)), int(os.environ.get("N", 4096))
fea = torch.randn(b, n, 32, device=d)
w = torch.randn(b, 512, device=d)
up = torch.randn(b, n, 3, device=d)
sd = {k: w.clone().requires_grad_(True) for k in G.inr_net.style_dim_dict}


def fwd():
    return G.inr_net(fea, sd)


def fwd_bwd():
    for p in G.inr_net.parameters():
        p.grad = None
    (fwd() * up).sum().backward()


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cfgs = [("per-layer", False, 0, 1)] + [(f"chain skew {s} x {p}", True, s, p) for s, p in
                                       ((0, 1), (4000, 4), (8000, 4), (16000, 4), (32000, 4), (16000, 2), (8000, 8), (16000, 8))]
res = {c[0]: [] for c in cfgs}
for rnd in range(3):
    for name, chain, skew, ph in cfgs:
        ops.INR_CHAIN, ops.CHAIN_SKEW_CYCLES, ops.CHAIN_PHASES = chain, skew, ph
        res[name].append((timeit(fwd), timeit(fwd_bwd)))
for name in res:
    f = min(x[0] for x in res[name]); fb = min(x[1] for x in res[name])
    print(f"{name:24s} forward {f:7.3f} ms   forward+backward {fb:7.3f} ms   rounds fwd {['%.3f' % x[0] for x in res[name]]}", flush=True)
