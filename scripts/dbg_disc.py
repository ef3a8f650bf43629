import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import D_CFG, max_rel
from oracle import cips3d_oracle as orc
from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
torch.manual_seed(12)
D = Discriminator_MultiScale_Aux(**D_CFG)
sd = dict(D.state_dict())
x = torch.rand(4, 3, 16, 16) * 2 - 1
d = torch.device("cuda:0")
Dd = Discriminator_MultiScale_Aux(**D_CFG); Dd.load_state_dict(D.state_dict()); Dd = Dd.to(d)
for alpha in (1.0, 0.5):
    for aux in (False, True):
        with torch.no_grad():
            ref = orc.discriminator_forward(sd, x, alpha=alpha, use_aux_disc=aux)
            out, _, _ = Dd(x.to(d), alpha=alpha, use_aux_disc=aux)
        print(f"alpha={alpha} aux={aux}: max_rel {max_rel(out, ref):.3e}  ref {ref.flatten().tolist()}  got {out.flatten().tolist()}")
# per-stage for the aux disc
with torch.no_grad():
    a = Dd.aux_disc
    xi = x.to(d)
    c = a.conv_in["16"](xi); cr = orc.conv_layer(sd, "aux_disc.conv_in.16.", x, 1)
    print("aux conv_in16", max_rel(c, cr))
    blk = a.convs["16"]
    c1 = blk.conv1(c); c1r = orc.conv_layer(sd, "aux_disc.convs.16.conv1.", cr, 3, downsample=True)
    print("aux conv1 (down)", max_rel(c1, c1r))
    c2 = blk.conv2(c1); c2r = orc.conv_layer(sd, "aux_disc.convs.16.conv2.", c1r, 3)
    print("aux conv2", max_rel(c2, c2r))
    s = blk.skip(c); sr = orc.conv_layer(sd, "aux_disc.convs.16.skip.", cr, 1, downsample=True, activate=False, bias=False)
    print("aux skip", max_rel(s, sr))
    m = Dd.main_disc
    c = m.conv_in["8"](torch.nn.functional.interpolate(xi, scale_factor=0.5, mode="bilinear"))
    cr = orc.conv_layer(sd, "main_disc.conv_in.8.", torch.nn.functional.interpolate(x, scale_factor=0.5, mode="bilinear"), 1)
    print("main conv_in8 on downsampled", max_rel(c, cr))
print("---- input gradients ----")
for alpha in (1.0, 0.5):
    for aux in (False, True):
        xr = x.clone().requires_grad_(True)
        ref = orc.discriminator_forward(sd, xr, alpha=alpha, use_aux_disc=aux)
        gr, = torch.autograd.grad(ref.sum(), xr)
        xd = x.to(d).requires_grad_(True)
        out, _, _ = Dd(xd, alpha=alpha, use_aux_disc=aux)
        gd, = torch.autograd.grad(out.sum(), xd)
        per = [(gd[i].cpu() - gr[i]).abs().max().item() / gr.abs().max().item() for i in range(4)]
        print(f"alpha={alpha} aux={aux}: grad max_rel {max_rel(gd, gr):.3e} per-image {['%.1e' % p for p in per]}")
print("---- fixture x ----")
from conftest import load_golden
fix = load_golden("d_r16_aux_alpha")
xf = fix["x"]
for cg in (False, True):
    xr = xf.clone().requires_grad_(True)
    ref = orc.discriminator_forward(sd, xr, alpha=0.5, use_aux_disc=True)
    gr, = torch.autograd.grad(ref.sum(), xr, create_graph=cg)
    xd = xf.to(d).requires_grad_(True)
    out, _, _ = Dd(xd, alpha=0.5, use_aux_disc=True)
    gd, = torch.autograd.grad(out.sum(), xd, create_graph=cg)
    per = [(gd[i].detach().cpu() - gr[i].detach()).abs().max().item() / gr.abs().max().item() for i in range(4)]
    print(f"create_graph={cg}: out max_rel {max_rel(out, ref):.2e} grad max_rel {max_rel(gd, gr):.3e} vs golden {max_rel(gd, fix['grad_real']):.3e} per-image {['%.1e' % p for p in per]}")
    diff = (gd.detach().cpu() - gr.detach()).abs()
    idx = diff.flatten().argmax().item()
    print("   argmax idx", idx, "unravel", [int(v) for v in torch.unravel_index(torch.tensor(idx), diff.shape)], "ref", gr.flatten()[idx].item(), "got", gd.flatten()[idx].item())
    print("   frac elems with rel diff>1e-4:", float((diff / gr.abs().max() > 1e-4).float().mean()))
