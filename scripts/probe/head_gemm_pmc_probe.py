"""Five launches each of the head NT GEMM (main loop only; forward epilogue) and of the grouped K-major weight-gradient GEMM at the C2
shape, for a counter pass:  BENCH=probe/head_gemm_pmc_probe.py scripts/pmc.sh lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0")
B, n, C = 32, 4096, 512
torch.manual_seed(0)
x = torch.randn(B, n, C, device=d); w = torch.randn(B, C, C, device=d) * 0.04; g = torch.randn(B, n, C, device=d)
xP, _ = ops.split_planes(x, want_t=False); wP, _ = ops.split_planes(w, want_t=False); gP, _ = ops.split_planes(g, want_t=False)
oP = ops.Planes.empty(B, n, C, device=d); bits = torch.zeros(B, n, C // 8, device=d, dtype=torch.uint8)
dW1 = torch.empty(B, C, C, device=d); dW2 = torch.empty(B, C, C, device=d)
for _ in range(5):
    ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C)
for _ in range(5):
    ops.gemm_x3(xP, wP, n, C, C, C, C, B, n * C, C * C, P=oP, act=1, mask_out=bits, gate_bits=2)
for _ in range(5):
    ops.gemm_x3_km_grouped([(xP, gP, dW1), (gP, xP, dW2)], C, C, n, C, C, B, n * C, n * C)
torch.cuda.synchronize()
