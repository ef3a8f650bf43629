"""Runs ONLY the fused ray-march kernel (cips_march_fwd_x3 under no_grad) at C2 a few times: the target of the PMC
passes of scripts/r2_profiles.sh that measure its HBM bytes per ray."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import G_CFG
from cips3d_amd.generator import GeneratorNerfINR
d = torch.device("cuda:0")
torch.manual_seed(0)
G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
b, img, S = 32, 64, 24
n = img * img
style = {k: torch.randn(b, 128, device=d) for k in G.siren.style_dim_dict}
xg = torch.linspace(-1, 1, img, device=d); yg = torch.linspace(1, -1, img, device=d); zg = torch.linspace(0.88, 1.12, S, device=d)
zc = -1.0 / float(torch.tan(torch.tensor(3.14159265 * 12 / 360)))
c2w = torch.eye(4, device=d).repeat(b, 1, 1); c2w[:, 2, 3] = 1.0
jit = torch.rand(b, n, S, device=d)
with torch.no_grad():
    for _ in range(6):
        G.siren.march(style, (b, img, img, S, zc, 0.0, 0, 0, False), xg, yg, zg, c2w, jit, None)
torch.cuda.synchronize()
