"""Probe build only (scripts/probe/build_tuning.sh; CIPS_X3_PROF=1): s_memtime stamps (100 MHz) of workgroup (0,0) of the SIREN backward
chain kernel, per phase, for the one-kernel and the split form, at a part-filled chip (b = 4: 96 workgroups) and the C2 batch."""
import os, sys, ctypes
os.environ["CIPS_X3_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cips3d_amd import ops, _lib
d = torch.device("cuda:0"); P = 64 * 64 * 24
lib = _lib.load()
lib.cips_siren_bwd_x3_prof.argtypes = [ctypes.c_void_p]


def inputs(b, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(d)
    pts = ((torch.rand(b, P, 3, generator=g) - 0.5) * 0.24).to(d)
    t = dict(g0=(30 + 5 * torch.randn(b, 128, generator=g)).to(d), p0=r(b, 128), g1=(30 + 5 * torch.randn(b, 128, generator=g)).to(d),
             p1=r(b, 128), gc=(30 + 5 * torch.randn(b, 64, generator=g)).to(d), pc=r(b, 64), w0=r(128, 3, scale=0.3),
             b0=r(128, scale=0.1), w1=r(128, 128, scale=0.01), b1=r(128, scale=0.1), ws=r(1, 128, scale=0.01), bs=r(1, scale=0.1),
             wc=r(64, 128, scale=0.01), bc=r(64, scale=0.1), wf=r(32, 64, scale=0.05), bf=r(32, scale=0.1))
    return pts, t, torch.randn(b, P, 32, generator=g).to(d), torch.randn(b, P, generator=g).to(d)


names = ["L0+L1", "film2", "Lc+filmc", "dWf", "dhc+dac", "dWc", "dh2(+dh1)", "da1 tail", "dW1"]
ops.TRIG_MODE = 1
for b in (32,):
    pts, t, df, ds = inputs(b)
    for split in (False, True):
        ops.SIREN_BWD_SPLIT = split
        for _ in range(3):
            ops._siren_backward({k: t[k] for k in ops._SIREN_NAMES}, df, ds, b, P, points=pts)
        torch.cuda.synchronize()
        buf = np.zeros((8, 4, 16), dtype=np.uint64)
        lib.cips_siren_bwd_x3_prof(buf.ctypes.data_as(ctypes.c_void_p))
        ts = buf.astype(np.int64)
        rounds = (ts[3:8, :, 0] - ts[2:7, :, 0]).mean(0)                     # ticks per round, per wave
        ph = ts[2:7, :, 0:10]
        if split:                                                            # stamps 7 is not taken: phase 6 ends at stamp 8
            ph = ph.copy(); ph[:, :, 7] = ph[:, :, 8]
        dd = np.diff(ph, axis=2).mean(0)
        print(f"b={b} {'split chain' if split else 'one kernel '}: round {rounds.mean() * 10:.0f} ns  | " +
              " ".join(f"{n}={int(x) * 10}" for n, x in zip(names, dd[0])) + "  (ns, wave 0)", flush=True)
