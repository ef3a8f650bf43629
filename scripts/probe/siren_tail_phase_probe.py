"""Probe build only (scripts/probe/build_tuning.sh; CIPS_X3_PROF=2): s_memtime stamps of workgroup (0,0) of the SIREN backward tail
kernel, tiles 8..11, all 8 waves: wait for the tile's DMA | barrier | DMA issue | contraction over points | data chain."""
import os, sys, ctypes
os.environ["CIPS_X3_PROF"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cips3d_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
d = torch.device("cuda:0"); P = 64 * 64 * 24
lib = _lib.load()
lib.cips_siren_bwd_x3_prof.argtypes = [ctypes.c_void_p]
b = int(os.environ.get("B", 32))
g = torch.Generator().manual_seed(0)
r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(d)
pts = ((torch.rand(b, P, 3, generator=g) - 0.5) * 0.24).to(d)
t = dict(g0=(30 + 5 * torch.randn(b, 128, generator=g)).to(d), p0=r(b, 128), g1=(30 + 5 * torch.randn(b, 128, generator=g)).to(d),
         p1=r(b, 128), gc=(30 + 5 * torch.randn(b, 64, generator=g)).to(d), pc=r(b, 64), w0=r(128, 3, scale=0.3),
         b0=r(128, scale=0.1), w1=r(128, 128, scale=0.01), b1=r(128, scale=0.1), ws=r(1, 128, scale=0.01), bs=r(1, scale=0.1),
         wc=r(64, 128, scale=0.01), bc=r(64, scale=0.1), wf=r(32, 64, scale=0.05), bf=r(32, scale=0.1))
df = torch.randn(b, P, 32, generator=g).to(d); ds = torch.randn(b, P, generator=g).to(d)
ops.TRIG_MODE = 1; ops.SIREN_BWD_SPLIT = True
for _ in range(3):
    ops._siren_backward({k: t[k] for k in ops._SIREN_NAMES}, df, ds, b, P, points=pts)
torch.cuda.synchronize()
buf = np.zeros((4, 8, 16), dtype=np.uint64)
lib.cips_siren_bwd_x3_prof(buf.ctypes.data_as(ctypes.c_void_p))
ts = buf.astype(np.int64)
names = ["wait DMA", "barrier", "DMA issue", "contraction", "data chain"]
for w in range(8):
    dd = np.diff(ts[:, w, 0:6], axis=1).mean(0)
    per = (ts[1:, w, 0] - ts[:-1, w, 0]).mean()
    print(f"wave {w}: tile period {per:.0f} ticks | " + " ".join(f"{n}={x:.0f}" for n, x in zip(names, dd)))
