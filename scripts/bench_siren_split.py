"""Split SIREN backward (chain up to da2 + contraction-over-points tail) against the one-kernel form: the 16 gradients compared,
then HIP-event timing of the backward alone at the C2 shape (b = 32, P = 64*64*24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops

d = torch.device("cuda:0")
NAMES = ("dg0", "dp0", "dg1", "dp1", "dgc", "dpc", "dw0", "db0", "dw1", "db1", "dws", "dbs", "dwc", "dbc", "dwf", "dbf")


def inputs(b, P, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(d)
    pts = ((torch.rand(b, P, 3, generator=g) - 0.5) * 0.24).to(d)
    t = dict(g0=(30 + 5 * torch.randn(b, 128, generator=g)).to(d), p0=r(b, 128), g1=(30 + 5 * torch.randn(b, 128, generator=g)).to(d),
             p1=r(b, 128), gc=(30 + 5 * torch.randn(b, 64, generator=g)).to(d), pc=r(b, 64), w0=r(128, 3, scale=0.3),
             b0=r(128, scale=0.1), w1=r(128, 128, scale=0.01), b1=r(128, scale=0.1), ws=r(1, 128, scale=0.01), bs=r(1, scale=0.1),
             wc=r(64, 128, scale=0.01), bc=r(64, scale=0.1), wf=r(32, 64, scale=0.05), bf=r(32, scale=0.1))
    df = torch.randn(b, P, 32, generator=g).to(d); ds = torch.randn(b, P, generator=g).to(d)
    return pts, t, df, ds


def run(split, pts, t, df, ds):
    ops.SIREN_BWD_SPLIT = split
    b, P, _ = pts.shape
    return ops._siren_backward({k: t[k] for k in ops._SIREN_NAMES}, df, ds, b, P, points=pts)


for trig in (1, 0):
    ops.TRIG_MODE = trig
    for (b, P) in ((2, 2048 + 96), (3, 128 * 7 + 5), (1, 4096 * 3), (4, 64 * 64 * 24)):
        pts, t, df, ds = inputs(b, P, seed=b)
        A = run(False, pts, t, df, ds); Bs = run(True, pts, t, df, ds)
        torch.cuda.synchronize()
        worst = max(float((x - y).norm() / x.norm().clamp_min(1e-30)) for x, y in zip(A, Bs))
        bad = [(n, f"{float((x - y).norm() / x.norm().clamp_min(1e-30)):.2e}") for n, x, y in zip(NAMES, A, Bs)
               if float((x - y).norm() / x.norm().clamp_min(1e-30)) > 1e-4 or not bool(torch.isfinite(y).all())]
        print(f"trig {trig} b={b} P={P}: worst relative difference split vs one kernel {worst:.2e}  {bad if bad else 'ok'}", flush=True)

ops.TRIG_MODE = 1
b, P = 32, 64 * 64 * 24
pts, t, df, ds = inputs(b, P)
for rnd in range(3):
    for split in (False, True):
        for _ in range(2): run(split, pts, t, df, ds)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(split, pts, t, df, ds)
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} {'split     ' if split else 'one kernel'}: {e0.elapsed_time(e1) / 5:.3f} ms per backward (kernels + partial sums + finalize)", flush=True)
