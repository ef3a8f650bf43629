"""x3 NT GEMM main-loop attribution: no-output GEMM at the F shape under CIPS_X3_GDBG = 0 / 1 (no MFMA) / 2 (no loads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = 32, 4096, 512
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C))
print(f"GDBG={os.environ.get('CIPS_X3_GDBG','0')} TILE={os.environ.get('CIPS_X3_TILE','256')}: x3 F no output {t:8.1f} us")

P = lambda *s_: ops.Planes.empty(*s_, device=d)
oP = P(B, n, C); m2 = torch.empty(B, n, C, device=d, dtype=torch.bfloat16)
add = torch.randn(B, n, C, device=d); cu = torch.empty(B, n, C, device=d)
t1 = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, act=1))
t2 = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, act=1, res=xP, mask_out=m2))
t3 = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, mask=m2))
t4 = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, mask=m2, add=add, C_unmasked=cu))
print(f"   F1 {t1:7.1f}  F2(skip) {t2:7.1f}  D2 {t3:7.1f}  D1(skip) {t4:7.1f} us")
