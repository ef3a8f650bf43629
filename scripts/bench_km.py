"""K-major x3 GEMM (weight gradients) at the head's dW shape: the grouped pair of a block (2 x 32 images x 512 x 512 x 4096)
and the single problem, round-2 schedule (CIPS_X3_KMV3=0) against the two-register-set schedule; interleaved rounds in one
process, outputs compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops, _lib
d = torch.device("cuda:0")
n, C = 4096, 512
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
lib = _lib.load()
B = 32
torch.manual_seed(0)
ops_ = []
for _ in range(2):
    x = torch.randn(B, n, C, device=d); gq = torch.randn(B, n, C, device=d)
    xP, _ = ops.split_planes(x, want_t=False); gP, _ = ops.split_planes(gq, want_t=False)
    ops_.append((xP, gP))
outs = {m: [torch.empty(B, C, C, device=d) for _ in range(2)] for m in ("0", "1")}
def grouped(m):
    os.environ["CIPS_X3_KMV3"] = m
    ops.gemm_x3_km_grouped([(a, b, o) for (a, b), o in zip(ops_, outs[m])], C, C, n, C, C, B, n * C, n * C)
def single(m):
    os.environ["CIPS_X3_KMV3"] = m
    lib.cips_gemm_bf16x3_set_wide(2)
    ops.gemm_x3_km(ops_[0][0], ops_[0][1], C, C, n, C, C, B, n * C, n * C, outs[m][0])
    lib.cips_gemm_bf16x3_set_wide(-1)
for name, fn, nprob in (("grouped pair", grouped, 2), ("single", single, 1)):
    ts = {"0": [], "1": []}
    for rnd in range(3):
        for m in ("0", "1"):
            ts[m].append(timeit(lambda: fn(m)))
    fn("0"); fn("1"); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(outs["0"][:nprob], outs["1"][:nprob]))
    fl = 2.0 * nprob * B * n * C * C
    t0, t1 = min(ts["0"]), min(ts["1"])
    print(f"{name:13s} round-2 schedule {t0:7.1f} us ({fl/t0/1e6:6.1f} TF, frac {fl/t0/1e6/833.3:.3f})   two-set schedule {t1:7.1f} us "
          f"({fl/t1/1e6:6.1f} TF, frac {fl/t1/1e6/833.3:.3f})   bit-identical {same}   rounds {['%.1f' % t for t in ts['0']]} {['%.1f' % t for t in ts['1']]}")
os.environ.pop("CIPS_X3_KMV3", None)
