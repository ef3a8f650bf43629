"""A/B of the skip branch's blurs (down 2 forward, up 2 backward) between two builds of the library: us per call and a SHA-1 of
every output.  python scripts/probe/blur_pairs_ab.py [path/to/lib.so]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from cips3d_amd import ops, discriminator as dm
d = torch.device("cuda:0")
k = dm.make_kernel([1, 3, 3, 1]).to(d); kf = torch.flip(k, [0, 1]).contiguous()
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def sha(x): return hashlib.sha1(x.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]
for (mj, h) in [(32 * 512, 64), (32 * 512, 32), (32 * 256, 64), (32 * 512, 16), (32 * 512, 8), (4 * 128, 256), (7, 4)]:
    x = torch.randn(mj, h, h, 1, device=d, generator=torch.Generator(device=d).manual_seed(h + mj))
    g = torch.randn(mj, h // 2, h // 2, 1, device=d, generator=torch.Generator(device=d).manual_seed(h))
    dn = lambda: ops.upfirdn2d_op(x, k, 1, 1, 2, 2, 1, 1, 1, 1)
    up = lambda: ops.upfirdn2d_op(g, kf, 2, 2, 1, 1, 2, 1, 2, 1)
    print(f"planes {mj:6d} x {h:3d}^2: down2 {t(dn):7.1f} us {sha(dn())}   up2 {t(up):7.1f} us {sha(up())}  out {tuple(up().shape[1:3])}")
