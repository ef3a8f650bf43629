"""Where does the epilogue time of the head NT GEMM go?  v3 kernel, C2 shape, the tuning switches of gemm_bf16x3_v3.hip
(needs the probe build, scripts/probe/build_tuning.sh: CIPS_X3_V3DBG / V3SKEW / V3PHASES / V3GRID) set in-process between timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops, _lib
lib = _lib.load()
d = torch.device("cuda:0")
B, n, C = 32, 4096, 512
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
add = torch.randn(B, n, C, device=d)
gate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
rg = torch.randn(B * n, 3, device=d); rw = torch.randn(3, C, device=d)
oP = ops.Planes.empty(B, n, C, device=d); mo = torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8); cu = torch.empty(B, n, C, device=d)
rP, _ = ops.split_planes(torch.randn(B, n, C, device=d), want_t=False)
F = {
    "plain": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=oP, act=1, mask_out=mo, gate_bits=2),
    "res": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=oP, act=1, res=rP, mask_out=mo, gate_bits=2),
    "gate": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=oP, mask=gate, gate_bits=1),
    "add": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=oP, add=add, rgb_g=rg, rgb_w=rw, C_unmasked=cu, mask=gate, gate_bits=1),
}
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
CASES = [
    ("base", {}),
    ("non-temporal stores (dbg 4)", {"CIPS_X3_V3DBG": "4"}),
    ("stores into one L2-resident window (dbg 2)", {"CIPS_X3_V3DBG": "2"}),
    ("epilogue without its stores (dbg 1)", {"CIPS_X3_V3DBG": "1"}),
    ("skip epilogue (dbg 8)", {"CIPS_X3_V3DBG": "8"}),
    ("2 start phases, skew 20000 cycles", {"CIPS_X3_V3SKEW": "20000", "CIPS_X3_V3PHASES": "2"}),
    ("half-chip grid", {"CIPS_X3_V3GRID": "128"}),
    ("base again", {}),
]
KEYS = ["CIPS_X3_V3DBG", "CIPS_X3_V3SKEW", "CIPS_X3_V3PHASES", "CIPS_X3_V3GRID", "CIPS_X3_V3TOUCH"]
for name, env in CASES:
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    row = "  ".join(f"{fl} {min(timeit(F[fl]) for _ in range(2)):7.1f}" for fl in F)
    print(f"{name:42s} {row}", flush=True)
for k in KEYS: os.environ.pop(k, None)

ref = {}
for fl in F:
    F[fl](); torch.cuda.synchronize()
    ref[fl] = (oP.hi.clone(), oP.lo.clone())
