"""Head NT GEMM (gemm_bf16x3_v3) at the C2 shape, the FOUR epilogue flavours the training step launches: forward
(LeakyReLU + gate bits + planes), forward + residual planes + fused ToRGB, backward with the gate, backward with the
planes addend + rank-3 ToRGB term.  Prints us / launch (rounds interleaved) and a SHA-1 of every output so that two
builds of the library can be compared bit for bit:   python scripts/bench_v3_heavy.py [path/to/libcips3d_hip.so]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = int(os.environ.get("B", 32)), int(os.environ.get("NPIX", 4096)), 512
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
rP, _ = ops.split_planes(torch.randn(B, n, C, device=d), want_t=False)          # residual / gated addend planes
gate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
pgate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
rg = torch.randn(B * n, 3, device=d); rw = torch.randn(3, C, device=d); rb = torch.randn(3, device=d)
out = dict(P=ops.Planes.empty(B, n, C, device=d), mo=torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8),
           rgb=torch.zeros(B * n, 3, device=d))
sh = (xP, wP, n, C, C, C, C, B, n * C, C * C)
fl = {
    "fwd": lambda: ops.gemm_x3(*sh, P=out["P"], act=1, mask_out=out["mo"], gate_bits=2),
    "fwd+res+torgb": lambda: ops.gemm_x3_torgb(*sh, out["P"], rw, rb, out["rgb"], False, act=1, res=rP, mask_out=out["mo"], gate_bits=2),
    "bwd gate": lambda: ops.gemm_x3(*sh, P=out["P"], mask=gate, gate_bits=1),
    "bwd addp+rgb": lambda: ops.gemm_x3(*sh, P=out["P"], addp=(rP, pgate), rgb_g=rg, rgb_w=rw, mask=gate, gate_bits=1),
    "bwd addp": lambda: ops.gemm_x3(*sh, P=out["P"], addp=(rP, pgate), mask=gate, gate_bits=1),
}


def timeit(fn, reps=60, warm=60):          # warm: past the ~50-launch DVFS settling window (profiles/r5_power_envelope.txt)
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


flops = 2.0 * B * n * C * C
for name, f in fl.items():
    out["P"].hi.zero_(); out["P"].lo.zero_(); out["mo"].zero_(); out["rgb"].zero_()
    f(); torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in (out["P"].hi, out["P"].lo, out["mo"], out["rgb"]):
        h.update(t.cpu().contiguous().view(torch.uint8).numpy().tobytes())
    ts = [timeit(f) for _ in range(3)]
    t = min(ts)
    print(f"{name:16s} {t:7.1f} us  {flops / t / 1e6:6.1f} TFLOP/s  frac {flops / t / 1e6 / 833.3:.3f}  rounds {['%.1f' % v for v in ts]}  sha1 {h.hexdigest()[:16]}", flush=True)
