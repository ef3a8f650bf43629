"""Micro-benchmark of the INR GEMM shapes at C2 (B=32, n=4096, 512x512): fp32 MFMA kernel vs bf16x3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = int(os.environ.get("B", 32)), int(os.environ.get("NPIX", 4096)), 512
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
flops = 2.0 * B * n * C * C
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04; out = torch.empty(B, n, C, device=d)
t = timeit(lambda: ops.bmm_nn(x, w, out=out, act=1)); print(f"f32 NN fwd        {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
t = timeit(lambda: ops.bmm_tn(x, out)); print(f"f32 TN dW         {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
P = lambda *s: ops.Planes.empty(*s, device=d)
xP, xT = ops.split_planes(x); wP, wT = ops.split_planes(w)
oP, oT = P(B, n, C), P(B, C, n)
m2 = torch.empty(B, n, C, device=d, dtype=torch.bfloat16)
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, T=oT, ldt=n, strideT=C*n, act=1)); print(f"x3 F (P+T out)    {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, T=oT, ldt=n, strideT=C*n, act=1, res=xP, mask_out=m2)); print(f"x3 F skip         {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, T=oT, ldt=n, strideT=C*n, mask=m2)); print(f"x3 D (mask)       {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
add = torch.randn(B, n, C, device=d); cu = torch.empty(B, n, C, device=d)
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, P=oP, T=oT, ldt=n, strideT=C*n, mask=m2, add=add, C_unmasked=cu)); print(f"x3 D skip         {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
gw = torch.empty(B, C, C, device=d)
t = timeit(lambda: ops.gemm_x3(xT, oT, C, C, n, n, n, B, C*n, C*n, C=gw)); print(f"x3 W (fp32 out)   {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C, C=out)); print(f"x3 plain fp32 out {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
t = timeit(lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n*C, C*C)); print(f"x3 F no output    {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
# longer K at the F shape: A (B,n,2048), B (B,512,2048)
K2 = 2048
x2 = torch.randn(B, n, K2, device=d); w2 = torch.randn(B, C, K2, device=d) * 0.02
x2P, _ = ops.split_planes(x2, want_t=False); w2P, _ = ops.split_planes(w2, want_t=False)
f2 = 2.0 * B * n * C * K2
t = timeit(lambda: ops.gemm_x3(x2P, w2P, n, C, K2, K2, K2, B, n*K2, C*K2), reps=5); print(f"x3 F K=2048 noout {t*1e6:8.1f} us {f2/t/1e12:7.1f} TF")
t = timeit(lambda: ops.gemm_x3(x2P, w2P, n, C, K2, K2, K2, B, n*K2, C*K2, C=out), reps=5); print(f"x3 F K=2048 C out {t*1e6:8.1f} us {f2/t/1e12:7.1f} TF")
t = timeit(lambda: ops.gemm_x3_km(xP, oP, C, C, n, C, C, B, n*C, n*C, gw)); print(f"x3 W k-major (tr) {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF")
