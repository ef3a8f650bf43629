"""Head NT GEMM at the C2 shape (32 x 4096 x 512 x 512): the two-workgroups-per-CU kernel (gemm_bf16x3_v4.hip, descriptor
kernel = 4 + 16 * start offset in 1000 cycles) against the 256 x 256-tile v3 kernel, every epilogue flavour of the training step,
interleaved rounds in one process; outputs compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = int(os.environ.get("B", 32)), int(os.environ.get("NPIX", 4096)), 512
SKEWS = [int(s) for s in os.environ.get("SKEWS", "0,12,24,36").split(",")]
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
res = torch.randn(B, n, C, device=d); rP, _ = ops.split_planes(res, want_t=False)
add = torch.randn(B, n, C, device=d); aP, _ = ops.split_planes(add, want_t=False)
gate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
pgate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
rg = torch.randn(B * n, 3, device=d); rw = torch.randn(3, C, device=d)
part = torch.empty(C // 128, B * n, 4, device=d)
G = lambda o, **kw: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=o["P"], **kw)


def flavours():
    o = dict(P=ops.Planes.empty(B, n, C, device=d), mo=torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8),
             part=torch.zeros(C // 128, B * n, 4, device=d))
    return o, {
        "fwd": lambda: G(o, act=1, mask_out=o["mo"], gate_bits=2),
        "fwd+rgbf": lambda: G(o, act=1, mask_out=o["mo"], gate_bits=2, torgb=(rw, o["part"])),
        "fwd+res+rgbf": lambda: G(o, act=1, res=rP, mask_out=o["mo"], gate_bits=2, torgb=(rw, o["part"])),
        "dX": lambda: G(o, mask=gate, gate_bits=1),
        "dX+addp": lambda: G(o, addp=(aP, pgate), rgb_g=rg, rgb_w=rw, mask=gate, gate_bits=1),
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
modes = [("v3", 2)] + [(f"v4 skew {s}k", 4 + 16 * s) for s in SKEWS]
ok = True
for name in fa:
    ops.X3_KERNEL = 2; fa[name]()
    ops.X3_KERNEL = 4; fb[name]()
    torch.cuda.synchronize()
    same = torch.equal(oa["P"].hi, ob["P"].hi) and torch.equal(oa["P"].lo, ob["P"].lo)
    if name.startswith("fwd"): same = same and torch.equal(oa["mo"], ob["mo"])
    if "rgbf" in name: same = same and torch.equal(oa["part"], ob["part"])
    if not same:
        dh = (oa["P"].hi.float() != ob["P"].hi.float())
        rows = dh.any(-1)
        print(f"  MISMATCH {name}: {int(dh.sum())} hi elements differ in {int(rows.sum())} rows; first rows {rows.nonzero()[:6].tolist()}; "
              f"max abs diff {float((oa['P'].float() - ob['P'].float()).abs().max()):.3e}; finite {bool(torch.isfinite(ob['P'].float()).all())}")
    ok = ok and same
    ts = {m: [] for m, _ in modes}
    for rnd in range(3):
        for m, k in modes:
            ops.X3_KERNEL = k
            ts[m].append(timeit((fa if m == "v3" else fb)[name]))
    print(f"{name:14s} " + "   ".join(f"{m} {min(ts[m]):6.1f} us ({flops / min(ts[m]) / 1e6 / 833.3:.3f})" for m, _ in modes) +
          f"   bit-identical {same}", flush=True)
ops.X3_KERNEL = 0
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
