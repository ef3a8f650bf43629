"""upfirdn2d microbench on the discriminator's blur shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops

def main():
    d = torch.device("cuda:0")
    k = torch.tensor([1., 3., 3., 1.]); k = (k[None] * k[:, None]); k = (k / k.sum()).to(d)
    cases = [("64x64 x8192 down1 pad2", 8192, 64, 1, 1, (2, 2)), ("64x64 x16384 down1 pad2", 16384, 64, 1, 1, (2, 2)),
             ("32x32 x16384 down1 pad2", 16384, 32, 1, 1, (2, 2)), ("16x16 x16384 down1 pad2", 16384, 16, 1, 1, (2, 2)),
             ("65x65 x8192 down1 pad(1,1) (bwd)", 8192, 65, 1, 1, (1, 1)),
             ("64x64 x8192 down2 pad1", 8192, 64, 1, 2, (1, 1)), ("32x32 x16384 down2 pad1", 16384, 32, 1, 2, (1, 1)),
             ("32x32 x8192 up2 (bwd of down2)", 8192, 32, 2, 1, (2, 1)), ("256x256 x256 down1 pad2", 256, 256, 1, 1, (2, 2))]
    for name, mj, n, up, down, pad in cases:
        x = torch.randn(mj, n, n, 1, device=d)
        f = lambda: ops.upfirdn2d_op(x, k, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        y = f()
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        gb = (x.numel() + y.numel()) * 4 / 1e9
        print(f"{name:38s} {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s")

if __name__ == "__main__":
    main()
