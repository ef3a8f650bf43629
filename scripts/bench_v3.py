"""Head NT GEMM at the C2 shape (32 x 4096 x 512 x 512): the v3 schedule against the round-2 wide kernel, the four
epilogue flavours of the training step, interleaved rounds in one process; outputs compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops, _lib
lib = _lib.load()
d = torch.device("cuda:0")
B, n, C = int(os.environ.get("B", 32)), int(os.environ.get("NPIX", 4096)), 512
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
res = torch.randn(B, n, C, device=d); rP, _ = ops.split_planes(res, want_t=False)
add = torch.randn(B, n, C, device=d)
gate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
rg = torch.randn(B * n, 3, device=d); rw = torch.randn(3, C, device=d)
P = lambda: ops.Planes.empty(B, n, C, device=d)
bits = lambda: torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8)
cu = lambda: torch.empty(B, n, C, device=d)

def flavours():
    o = dict(P=P(), mo=bits(), cu=cu())
    return o, {
        "plain": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=o["P"], act=1, mask_out=o["mo"], gate_bits=2),
        "res": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=o["P"], act=1, res=rP, mask_out=o["mo"], gate_bits=2),
        "gate": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=o["P"], mask=gate, gate_bits=1),
        "add": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=o["P"], add=add, rgb_g=rg, rgb_w=rw,
                                   C_unmasked=o["cu"], mask=gate, gate_bits=1),
        "add_norgb": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=o["P"], add=add,
                                         C_unmasked=o["cu"], mask=gate, gate_bits=1),
        "noout": lambda: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C),
    }

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

flops = 2.0 * B * n * C * C
oa, fa = flavours(); ob, fb = flavours()
ok = True
for name in fa:
    ops.X3_KERNEL = 3; fa[name]()
    ops.X3_KERNEL = 2; fb[name]()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in ((oa["P"].hi, ob["P"].hi), (oa["P"].lo, ob["P"].lo))) if name != "noout" else True
    if name in ("plain", "res"): same = same and torch.equal(oa["mo"], ob["mo"])
    if name.startswith("add"): same = same and torch.equal(oa["cu"], ob["cu"])
    ok = ok and same
    ts = {3: [], 2: []}
    for rnd in range(3):
        for mode, f in ((3, fa), (2, fb)):
            ops.X3_KERNEL = mode
            ts[mode].append(timeit(f[name]))
    tw, tv = min(ts[3]), min(ts[2])
    print(f"{name:10s} wide {tw:7.1f} us ({flops/tw/1e6:6.1f} TF, frac {flops/tw/1e6/833.3:.3f})   v3 {tv:7.1f} us ({flops/tv/1e6:6.1f} TF, frac {flops/tv/1e6/833.3:.3f})   "
          f"bit-identical {same}   rounds wide {['%.1f' % t for t in ts[3]]} v3 {['%.1f' % t for t in ts[2]]}", flush=True)
ops.X3_KERNEL = 0
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
