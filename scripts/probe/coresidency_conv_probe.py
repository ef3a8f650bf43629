"""Do LDS-free low-register kernels run BESIDE the implicit-GEMM convolution (2 waves x 240 VGPRs per SIMD: 32 free; 128 of 160 KiB
LDS) and the head NT GEMM?  Side stream: 20 x (32 MB elementwise add_), timed alone and while the main stream loops the matrix kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
import cips3d_amd.discriminator as D
d = torch.device("cuda:0"); torch.manual_seed(0)
B, C, O, HW = 64, 512, 512, 64
x = torch.randn(B, C, HW, HW, device=d); w = torch.randn(O, C, 3, 3, device=d) * 0.02
wP = D._w_planes(w, 1.0); xP = D._nhwc(x)
def conv():
    return ops.conv2d_x3(wP, xP, B, C, HW, HW, O, 3, 3, 1, 1)
n = 4096
xa = torch.randn(32, n, 512, device=d); wa = torch.randn(32, 512, 512, device=d) * 0.04
xaP, _ = ops.split_planes(xa, want_t=False); waP, _ = ops.split_planes(wa, want_t=False)
outP = ops.Planes.empty(32, n, 512, device=d)
def head():
    ops.gemm_x3(xaP, waP, n, 512, 512, 512, 512, 32, n * 512, 512 * 512, P=outP)
big = torch.zeros(32 * 1024 * 1024 // 4, device=d)
side = torch.cuda.Stream()
def stream_free(k):
    for _ in range(k): big.add_(1.0)
def run(fn, reps, with_side, k=20):
    torch.cuda.synchronize()
    m0, m1, e0, e1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    m0.record()
    for _ in range(reps): fn()
    m1.record()
    if with_side:
        side.wait_event(m0)
        with torch.cuda.stream(side):
            e0.record(); stream_free(k); e1.record()
    torch.cuda.synchronize()
    return m0.elapsed_time(m1) * 1e3, (e0.elapsed_time(e1) * 1e3 if with_side else 0.0)
for name, fn, reps in (("conv2d_x3 64x64 512->512 b64", conv, 8), ("head NT GEMM C2", head, 12)):
    for _ in range(3): fn()
    run(fn, reps, False); a, _ = run(fn, reps, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        stream_free(20); e0.record(); stream_free(20); e1.record()
    torch.cuda.synchronize(); s_alone = e0.elapsed_time(e1) * 1e3
    run(fn, reps, True); c, s = run(fn, reps, True)
    print(f"{name:32s} x{reps}: alone {a:8.1f} us | 20 x add_(32 MB) alone {s_alone:7.1f} us | together: matrix {c:8.1f} us, adds {s:8.1f} us  (serial would be {a + s_alone:8.1f})")
