"""Phase timestamps of the fused march (probe build, CIPS_X3_MPROF=1): workgroup (0,0), all 8 waves, samples 8..11.
python march_phases.py <probe lib>;  CIPS_X3_MONE=1 for one wave per SIMD, CIPS_X3_MDESYNC=<cycles> for a late second wave."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CIPS_X3_MPROF"] = "1"
import torch
from cips3d_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from bench import G_CFG
from cips3d_amd.generator import GeneratorNerfINR
d = torch.device("cuda:0"); torch.manual_seed(0)
G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
b, img, S = 32, 64, 24; n = img * img
style = {k: torch.randn(b, 128, device=d) for k in G.siren.style_dim_dict}
xg = torch.linspace(-1, 1, img, device=d); yg = torch.linspace(1, -1, img, device=d); zg = torch.linspace(0.88, 1.12, S, device=d)
zc = -1.0 / float(torch.tan(torch.tensor(3.14159265 * 12 / 360)))
c2w = torch.eye(4, device=d).repeat(b, 1, 1); c2w[:, 2, 3] = 1.0
jit = torch.rand(b, n, S, device=d)
def ng():
    with torch.no_grad(): return G.siren.march(style, (b, img, img, S, zc, 0.0, 0, 0, False), xg, yg, zg, c2w, jit, None)
for _ in range(20): ng()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ng()
e1.record(); torch.cuda.synchronize()
buf = np.zeros((4, 8, 8), dtype=np.uint64)
rc = _lib.load().cips_march_x3_prof(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.astype(np.int64)
print(f"MONE={os.environ.get('CIPS_X3_MONE', '0')} MDESYNC={os.environ.get('CIPS_X3_MDESYNC', '0')}  {e0.elapsed_time(e1) / 20 * 1e3:.1f} us / launch  (rc {rc})")
names = ["L0 sines", "W1 mfma", "FiLM1 sines", "Wc mfma", "FiLMc sines", "Wf mfma", "out+composite"]
print("  wave  start(rel w0)  " + "  ".join(f"{x:>13s}" for x in names) + "   step total   to next start")
t0 = t[0, 0, 0]
for w in range(8):
    if t[1, w, 0] == 0: continue
    for s in (1, 2):
        r = t[s, w]
        nxt = t[s + 1, w, 0] - r[0]
        print(f"  {w} s{8 + s}  {r[0] - t0:10d}     " + "  ".join(f"{r[i + 1] - r[i]:13d}" for i in range(7)) + f"   {r[7] - r[0]:8d}   {nxt:8d}")
