"""Where does the epilogue time of the head NT GEMM go?  v3 kernel, C2 shape, the tuning switches of gemm_bf16x3_v3.hip
(CIPS_X3_V3DBG / V3SKEW / V3PHASES / V3GRID) set in-process between timings."""
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
    ("column block innermost (dbg 16)", {"CIPS_X3_V3DBG": "16"}),
    ("column block innermost + nt stores (dbg 20)", {"CIPS_X3_V3DBG": "20"}),
    ("base again", {}),
    ("column block innermost (dbg 16)", {"CIPS_X3_V3DBG": "16"}),
    ("column block innermost + nt stores (dbg 20)", {"CIPS_X3_V3DBG": "20"}),
    ("skip epilogue (dbg 8)", {"CIPS_X3_V3DBG": "8"}),
    ("base 3", {}),
    ("column block innermost (dbg 16)", {"CIPS_X3_V3DBG": "16"}),
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
os.environ["CIPS_X3_V3DBG"] = "16"
for fl in F:
    oP.hi.zero_(); oP.lo.zero_()
    F[fl](); torch.cuda.synchronize()
    print(f"column-block-innermost variant {fl}: bit-identical {torch.equal(oP.hi, ref[fl][0]) and torch.equal(oP.lo, ref[fl][1])}")
os.environ.pop("CIPS_X3_V3DBG", None)
