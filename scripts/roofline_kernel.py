"""Runs ONLY the roofline kernel of bench.py (gemm_bf16x3, modfc 512x512 forward form, C2 shape) a few times —
the target of the separate rocprofv3 --pmc passes that fill `roofline.traffic` (scripts/pmc_roofline.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
b, n = 32, 4096
x = torch.randn(b, n, 512, device=d); w = torch.randn(b, 512, 512, device=d) * 0.04
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False)
oP = ops.Planes.empty(b, n, 512, device=d)
for _ in range(6):
    ops.gemm_x3(xP, wP, n, 512, 512, 512, 512, b, n * 512, 512 * 512, P=oP, act=1)
torch.cuda.synchronize()
