"""Per-workgroup time of the SIREN backward kernels against the number of busy CUs: P = 98304 points per image (24 workgroups per
image, 32 rounds of 128 points each), b images -> 24 b workgroups on 256 CUs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = torch.device("cuda:0")
P = 64 * 64 * 24


def inputs(b, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(d)
    pts = ((torch.rand(b, P, 3, generator=g) - 0.5) * 0.24).to(d)
    t = dict(g0=(30 + 5 * torch.randn(b, 128, generator=g)).to(d), p0=r(b, 128), g1=(30 + 5 * torch.randn(b, 128, generator=g)).to(d),
             p1=r(b, 128), gc=(30 + 5 * torch.randn(b, 64, generator=g)).to(d), pc=r(b, 64), w0=r(128, 3, scale=0.3),
             b0=r(128, scale=0.1), w1=r(128, 128, scale=0.01), b1=r(128, scale=0.1), ws=r(1, 128, scale=0.01), bs=r(1, scale=0.1),
             wc=r(64, 128, scale=0.01), bc=r(64, scale=0.1), wf=r(32, 64, scale=0.05), bf=r(32, scale=0.1))
    return pts, t, torch.randn(b, P, 32, generator=g).to(d), torch.randn(b, P, generator=g).to(d)


ops.TRIG_MODE = 1
for b in (1, 2, 4, 5, 8, 10, 11, 16, 21, 32):
    pts, t, df, ds = inputs(b)
    out = []
    for split in (False, True):
        ops.SIREN_BWD_SPLIT = split
        f = lambda: ops._siren_backward({k: t[k] for k in ops._SIREN_NAMES}, df, ds, b, P, points=pts)
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5 * 1e3)
    wgs = 24 * b
    waves = -(-wgs // 256)
    print(f"b={b:2d}: {wgs:4d} workgroups ({waves} per CU at most)   one kernel {out[0]:7.1f} us ({out[0] / waves:6.1f} per workgroup generation)   "
          f"split {out[1]:7.1f} us ({out[1] / waves:6.1f})", flush=True)
