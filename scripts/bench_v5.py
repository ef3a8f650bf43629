"""Head NT GEMM at the C2 shape: kernel 4 (experiment) against the v3 kernel (kernel 2): every epilogue flavour of the training
step + the main loop alone, interleaved rounds in one process; outputs compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = int(os.environ.get("B", 32)), int(os.environ.get("NPIX", 4096)), 512
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
res = torch.randn(B, n, C, device=d); rP, _ = ops.split_planes(res, want_t=False)
add = torch.randn(B, n, C, device=d); aP, _ = ops.split_planes(add, want_t=False)
gate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
pgate = (torch.rand(B, n, C // 8, device=d) * 256).to(torch.uint8)
rg = torch.randn(B * n, 3, device=d); rw = torch.randn(3, C, device=d)
G = lambda o, **kw: ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, **kw)


def flavours():
    o = dict(P=ops.Planes.empty(B, n, C, device=d), mo=torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8),
             part=torch.zeros(C // 128, B * n, 4, device=d))
    return o, {
        "main loop only": lambda: G(o),
        "fwd": lambda: G(o, P=o["P"], act=1, mask_out=o["mo"], gate_bits=2),
        "fwd+rgbf": lambda: G(o, P=o["P"], act=1, mask_out=o["mo"], gate_bits=2, torgb=(rw, o["part"])),
        "fwd+res+rgbf": lambda: G(o, P=o["P"], act=1, res=rP, mask_out=o["mo"], gate_bits=2, torgb=(rw, o["part"])),
        "dX": lambda: G(o, P=o["P"], mask=gate, gate_bits=1),
        "dX+addp": lambda: G(o, P=o["P"], addp=(aP, pgate), rgb_g=rg, rgb_w=rw, mask=gate, gate_bits=1),
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
    for o in (oa, ob):
        o["P"].hi.fill_(float("nan")); o["P"].lo.fill_(float("nan")); o["mo"].zero_(); o["part"].zero_()
    ops.X3_KERNEL = 2; fa[name]()
    ops.X3_KERNEL = 4; fb[name]()
    torch.cuda.synchronize()
    same = True
    if name != "main loop only":
        same = torch.equal(oa["P"].hi.view(torch.int16), ob["P"].hi.view(torch.int16)) and torch.equal(oa["P"].lo.view(torch.int16), ob["P"].lo.view(torch.int16))
        if name.startswith("fwd"): same = same and torch.equal(oa["mo"], ob["mo"])
        if "rgbf" in name: same = same and torch.equal(oa["part"], ob["part"])
        if not same:
            dh = oa["P"].hi.view(torch.int16) != ob["P"].hi.view(torch.int16)
            rows = dh.any(-1)
            print(f"  MISMATCH {name}: {int(dh.sum())} hi elements in {int(rows.sum())} rows, first {rows.nonzero()[:4].tolist()}; finite {bool(torch.isfinite(ob['P'].float()).all())}")
    ok = ok and same
    ts = {2: [], 4: []}
    for rnd in range(3):
        for k, f in ((2, fa), (4, fb)):
            ops.X3_KERNEL = k
            ts[k].append(timeit(f[name]))
    print(f"{name:15s} v3 {min(ts[2]):6.1f} us ({flops / min(ts[2]) / 1e6 / 833.3:.3f})   kernel 4 {min(ts[4]):6.1f} us ({flops / min(ts[4]) / 1e6 / 833.3:.3f})   "
          f"bit-identical {same}   rounds {['%.1f' % t for t in ts[2]]} {['%.1f' % t for t in ts[4]]}", flush=True)
ops.X3_KERNEL = 0
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
