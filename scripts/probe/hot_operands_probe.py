"""Is the head NT GEMM's main loop limited by where its operands come from?  1024 tiles of 256 x 256 x 512 (the C2 launch) timed
with no outputs (main loop only) and with the forward epilogue: (i) the C2 operands (A: 32 x 4096 rows, read once from HBM by the
two column tiles of a row block; B: one 512 x 512 weight per image), (ii) every tile on the SAME A row block and the SAME B
(batch stride 0: 1 MiB of operands, L2-resident on every XCD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
C = 512
torch.manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(32, 4096, C, device=d); w = torch.randn(32, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
oP = ops.Planes.empty(32, 4096, C, device=d); bits = torch.empty(32, 4096, C // 8, device=d, dtype=torch.uint8)
xh = ops.Planes(xP.hi[:1, :256].contiguous(), xP.lo[:1, :256].contiguous()); wh = ops.Planes(wP.hi[:1].contiguous(), wP.lo[:1].contiguous())
oh = ops.Planes.empty(512, 256, C, device=d); bh = torch.empty(512, 256, C // 8, device=d, dtype=torch.uint8)
cases = {
    "C2 operands, main loop only": lambda: ops.gemm_x3(xP, wP, 4096, C, C, C, C, 32, 4096 * C, C * C),
    "hot operands, main loop only": lambda: ops.gemm_x3(xh, wh, 256, C, C, C, C, 512, 0, 0),
    "C2 operands, forward epilogue": lambda: ops.gemm_x3(xP, wP, 4096, C, C, C, C, 32, 4096 * C, C * C, P=oP, act=1, mask_out=bits, gate_bits=2),
    "hot operands, forward epilogue (same bytes written)": lambda: ops.gemm_x3(xh, wh, 256, C, C, C, C, 512, 0, 0, P=oh, act=1, mask_out=bh, gate_bits=2),
}
zx = ops.Planes(torch.zeros_like(xP.hi), torch.zeros_like(xP.lo)); zw = ops.Planes(torch.zeros_like(wP.hi), torch.zeros_like(wP.lo))
cases["C2 shapes, ZERO operands, main loop only (power: DVFS give-back)"] = lambda: ops.gemm_x3(zx, zw, 4096, C, C, C, C, 32, 4096 * C, C * C)
cases["C2 shapes, ZERO operands, forward epilogue"] = lambda: ops.gemm_x3(zx, zw, 4096, C, C, C, C, 32, 4096 * C, C * C, P=oP, act=1, mask_out=bits, gate_bits=2)
ts = {k: [] for k in cases}
for rnd in range(3):
    for k, f in cases.items():
        ts[k].append(timeit(f))
for k in cases:
    print(f"{k:70s} {min(ts[k]):7.1f} us   rounds {['%.1f' % t for t in ts[k]]}")
