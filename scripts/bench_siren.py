"""SIREN forward + backward microbench at the headline shape (b=32, P=64*64*24 points): HIP-event timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops

def main():
    b = int(os.environ.get("B", 32)); P = int(os.environ.get("P", 64 * 64 * 24)); reps = int(os.environ.get("REPS", 5))
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    def r(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(d).requires_grad_(True)
    pts = ((torch.rand(b, P, 3, generator=g) - 0.5) * 0.24).to(d)
    g0, g1, gc = [(30 + 5 * torch.randn(b, n, generator=g)).to(d).requires_grad_(True) for n in (128, 128, 64)]
    p0, p1, pc = r(b, 128), r(b, 128), r(b, 64)
    w0 = r(128, 3, scale=0.3); b0 = r(128, scale=0.1); w1 = r(128, 128, scale=0.01); b1 = r(128, scale=0.1)
    ws = r(1, 128, scale=0.01); bs = r(1, scale=0.1); wc = r(64, 128, scale=0.01)
    bc = r(64, scale=0.1); wf = r(32, 64, scale=0.05); bf = r(32, scale=0.1)
    ops.TRIG_MODE = 1
    args = (pts, g0, p0, g1, p1, gc, pc, w0, b0, w1, b1, ws, bs, wc, bc, wf, bf)
    df = torch.randn(b, P, 32, device=d); ds = torch.randn(b, P, device=d)
    def step():
        feat, sig = ops.SirenFunction.apply(*args)
        torch.autograd.backward([feat, sig], [df, ds])
    for _ in range(2): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): step()
    e1.record(); torch.cuda.synchronize()
    print(f"siren fwd+bwd b={b} P={P} fwd={ops.SIREN_FWD_MODE} bwd={ops.SIREN_BWD_MODE}: {e0.elapsed_time(e1) / reps:.3f} ms/iter")
    with torch.no_grad():
        for _ in range(2): ops.SirenFunction.apply(*args)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): ops.SirenFunction.apply(*args)
        e1.record(); torch.cuda.synchronize()
    print(f"  forward only: {e0.elapsed_time(e1) / reps:.3f} ms")
    if os.environ.get("CIPS_X3_PROF") and hasattr(__import__("cips3d_amd._lib", fromlist=["x"]).load(), "cips_siren_bwd_x3_prof"):      # probe build only
        import ctypes, numpy as np
        from cips3d_amd import _lib
        buf = np.zeros((8, 4, 16), dtype=np.uint64)
        _lib.load().cips_siren_bwd_x3_prof(buf.ctypes.data_as(ctypes.c_void_p))
        names = ["L0+L1", "film2", "Lc+filmc", "Gf", "dhc+dac", "Gc", "dh2+da2+dh1", "end(da1,h1)", "G1"]
        for w in range(4):
            d = np.diff(buf[2:7, w, :10].astype(np.int64), axis=1).mean(0)
            tot = (buf[3:8, w, 0].astype(np.int64) - buf[2:7, w, 0].astype(np.int64)).mean()
            g = np.diff(buf[2:7, w, 10:15].astype(np.int64), axis=1).mean(0)
            print(f"   G1 sub-phase 1: stage={int(g[0])} barrier={int(g[1])} mfma={int(g[2])} barrier={int(g[3])}")
            print(f"wave {w}: " + " ".join(f"{n}={int(x)}" for n, x in zip(names, d)) + f" | round={int(tot)} (s_memtime ticks, 100 MHz?)")

if __name__ == "__main__":
    main()
