"""K-major x3 GEMM at the head's dW shape; B=32 (split-K path in the wide kernel) and B=64 (no split)."""
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
for B in (32, 64):
    x = torch.randn(B, n, C, device=d); gq = torch.randn(B, n, C, device=d)
    xP, _ = ops.split_planes(x, want_t=False); gP, _ = ops.split_planes(gq, want_t=False)
    gw = torch.empty(B, C, C, device=d)
    for mode in (0, 1):
        lib.cips_gemm_bf16x3_set_wide(mode)
        t = timeit(lambda: ops.gemm_x3_km(xP, gP, C, C, n, C, C, B, n * C, n * C, gw))
        print(f"B={B} wide={mode}: {t:8.1f} us  {2.0*B*n*C*C/t/1e6:7.1f} TF")
lib.cips_gemm_bf16x3_set_wide(-1)
