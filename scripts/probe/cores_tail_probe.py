"""Stage-1 probe of the co-resident style / ToRGB gradient tail: (1) results against the existing kernels, (2) duration alone
and on a side stream beside the fused SIREN backward (C2 shapes: 18 head layers 512x512 (first 32x512), 6 ToRGB taps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cips3d_amd import ops
d = torch.device("cuda:0"); torch.manual_seed(0)
B, n = 32, 4096
layers = []
for i in range(18):
    cin = 32 if i == 0 else 512
    W = torch.randn(cin, 512, device=d) * 0.05; s = torch.randn(B, cin, device=d) * 0.3
    demod = torch.rand(B, 512, device=d) + 0.5; gwb = torch.randn(B, cin, 512, device=d)
    layers.append((W, s, demod, gwb))
taps = [ops.split_planes(torch.randn(B, n, 512, device=d), want_t=False)[0] for _ in range(6)]
drgb = torch.randn(B * n, 3, device=d)
ref = ops.modfc_prep_bwd_batch(layers); new = ops.modfc_prep_bwd_batch(layers, cores=True)
e = max(float((a - b).abs().max() / a.abs().max()) for (a, _), (b, _) in zip(ref, new)); f = max(float((a - b).abs().max() / a.abs().max()) for (_, a), (_, b) in zip(ref, new))
print(f"modfc prep bwd: dW max rel {e:.2e}  ds max rel {f:.2e}")
rt = ops.torgb_bwd_w_x3_batch(taps, drgb); nt = ops.torgb_bwd_w_x3_batch(taps, drgb, cores=True)
e = max(float((a - b).abs().max() / a.abs().max()) for (a, _), (b, _) in zip(rt, nt)); f = max(float((a - b).abs().max() / a.abs().max()) for (_, a), (_, b) in zip(rt, nt))
print(f"torgb bwd w:    dw max rel {e:.2e}  db max rel {f:.2e}")

# the SIREN backward as the main-stream occupant
g = torch.Generator().manual_seed(0)
P = 64 * 64 * 24
def r(*s, scale=1.0): return (torch.randn(*s, generator=g) * scale).to(d).requires_grad_(True)
pts = ((torch.rand(B, P, 3, generator=g) - 0.5) * 0.24).to(d)
g0, g1, gc = [(30 + 5 * torch.randn(B, m, generator=g)).to(d).requires_grad_(True) for m in (128, 128, 64)]
p0, p1, pc = r(B, 128), r(B, 128), r(B, 64)
w0 = r(128, 3, scale=0.3); b0 = r(128, scale=0.1); w1 = r(128, 128, scale=0.01); b1 = r(128, scale=0.1)
ws = r(1, 128, scale=0.01); bs = r(1, scale=0.1); wc = r(64, 128, scale=0.01); bc = r(64, scale=0.1); wf = r(32, 64, scale=0.05); bf = r(32, scale=0.1)
ops.TRIG_MODE = 1
feat, sig = ops.SirenFunction.apply(pts, g0, p0, g1, p1, gc, pc, w0, b0, w1, b1, ws, bs, wc, bc, wf, bf)
df = torch.randn(B, P, 32, device=d); dsg = torch.randn(B, P, device=d)
def bwd(): torch.autograd.backward([feat, sig], [df, dsg], retain_graph=True)
side = torch.cuda.Stream()
def tail(cores):
    ops.modfc_prep_bwd_batch(layers, cores=cores); ops.torgb_bwd_w_x3_batch(taps, drgb, cores=cores)
def run(cores, with_bwd):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record()
    if with_bwd: bwd()
    ev[1].record()
    side.wait_event(ev[0])
    with torch.cuda.stream(side):
        ev[2].record(); tail(cores); ev[3].record()
    torch.cuda.current_stream().wait_stream(side)
    ev[4].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3, ev[0].elapsed_time(ev[4]) * 1e3
for cores in (False, True):
    for wb in (False, True):
        run(cores, wb); run(cores, wb)
        res = [run(cores, wb) for _ in range(3)]
        m = min(res, key=lambda t: t[2])
        print(f"tail {'co-resident' if cores else 'existing   '}  {'beside the SIREN backward' if wb else 'alone                    '}: backward {m[0]:7.1f} us  tail {m[1]:7.1f} us  both done after {m[2]:7.1f} us")
