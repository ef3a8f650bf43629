"""GPU: discriminator (HIP native ops + fp32 MFMA conv GEMM through the C-ABI) against golden vectors
minted from the reference: logits, R1 gradient w.r.t. the input (double-backward graph) and the
parameter gradients of the full d_loss (train.py:385-409)."""
import pytest
import torch

from conftest import load_golden, check_checksums, max_rel, rel_err, D_CFG

pytestmark = pytest.mark.gpu
TOL = 1e-3


def test_conv2d_function_double_backward_matches_torch_cpu():
    from cips3d_amd.discriminator import conv2d
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for (B, C, O, H, k, stride, pad) in [(2, 8, 12, 8, 3, 1, 1), (2, 3, 16, 8, 1, 1, 0), (2, 8, 8, 9, 3, 2, 0),
                                         (1, 4, 8, 7, 1, 2, 0)]:
        x = torch.randn(B, C, H, H, generator=g, requires_grad=True)
        w = torch.randn(O, C, k, k, generator=g, requires_grad=True)
        y = torch.nn.functional.conv2d(x, w, stride=stride, padding=pad)
        up = torch.randn(y.shape, generator=g)
        gx, = torch.autograd.grad((y * up).sum(), x, create_graph=True)
        loss = (gx ** 2).sum() + (y ** 2).sum()
        loss.backward()
        xd = x.detach().to(d).requires_grad_(True); wd = w.detach().to(d).requires_grad_(True)
        yd = conv2d(xd, wd, stride=stride, padding=pad)
        gxd, = torch.autograd.grad((yd * up.to(d)).sum(), xd, create_graph=True)
        ((gxd ** 2).sum() + (yd ** 2).sum()).backward()
        assert max_rel(yd, y) < 1e-5
        assert max_rel(gxd, gx) < 1e-5
        assert rel_err(xd.grad, x.grad) < 1e-4 and rel_err(wd.grad, w.grad) < 1e-4


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_conv2d_x3_eligible_shapes_double_backward(mode, monkeypatch):
    """layers whose contraction lengths are all multiples of 32 (the split-bf16 GEMM path when mode = bf16x3):
    forward, input gradient, and the double-backward graph against torch on the CPU in fp64"""
    from cips3d_amd import discriminator as dm
    monkeypatch.setattr(dm, "CONV_MODE", mode)
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    tol = 3e-5 if mode == "bf16x3" else 1e-5
    # from 16x16 output planes up the bf16x3 mode runs the implicit-GEMM form (forward, stride-1 data gradient): ragged
    # pixel counts (400, 576 of 256-pixel tiles), 3 k-tiles per tap (C = 96), stride 2 without padding, 1x1 stride 2
    for (B, C, O, H, k, stride, pad) in [(3, 32, 64, 16, 3, 1, 1), (2, 64, 32, 17, 3, 2, 0), (2, 32, 32, 16, 1, 1, 0),
                                         (2, 64, 96, 33, 3, 2, 0), (2, 32, 64, 20, 3, 1, 1), (1, 96, 96, 24, 3, 1, 1),
                                         (2, 64, 32, 31, 1, 2, 0),
                                         # small planes: the batch folded into the pixel dimension (4x4, 8x8 outputs)
                                         (8, 64, 32, 4, 3, 1, 1), (4, 32, 64, 9, 3, 2, 0), (2, 32, 32, 8, 1, 1, 0)]:
        x = torch.randn(B, C, H, H, generator=g, dtype=torch.float64, requires_grad=True)
        w = (torch.randn(O, C, k, k, generator=g, dtype=torch.float64) / (C * k * k) ** 0.5).requires_grad_(True)
        y = torch.nn.functional.conv2d(x, w, stride=stride, padding=pad)
        up = torch.randn(y.shape, generator=g, dtype=torch.float64)
        gx, = torch.autograd.grad((y * up).sum(), x, create_graph=True)
        ((gx ** 2).sum() + (y ** 2).sum()).backward()
        xd = x.detach().float().to(d).requires_grad_(True); wd = w.detach().float().to(d).requires_grad_(True)
        yd = dm.conv2d(xd, wd, stride=stride, padding=pad)
        gxd, = torch.autograd.grad((yd * up.float().to(d)).sum(), xd, create_graph=True)
        ((gxd ** 2).sum() + (yd ** 2).sum()).backward()
        assert rel_err(yd, y) < tol and rel_err(gxd, gx) < tol, (mode, C, O, H, k, stride)
        assert rel_err(xd.grad, x.grad) < 3 * tol and rel_err(wd.grad, w.grad) < 3 * tol, (mode, C, O, H, k, stride)


@pytest.mark.parametrize("conv_mode", ["bf16x3", "f32"])
@pytest.mark.parametrize("tag", ["d_r16", "d_r16_aux_alpha", "d_r16_diffaug"])
def test_discriminator_matches_reference_golden(tag, conv_mode, monkeypatch):
    from cips3d_amd import discriminator as dmod
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    monkeypatch.setattr(dmod, "CONV_MODE", conv_mode)
    fix = load_golden(tag)
    d = torch.device("cuda:0")
    torch.manual_seed(fix["seed"])
    D = Discriminator_MultiScale_Aux(**dict(D_CFG, diffaug=fix.get("diffaug", False)))
    check_checksums(D.state_dict(), fix["state_checksums"])
    D = D.to(d)
    x = fix["x"].to(d).requires_grad_(True)
    if fix.get("diffaug"):                 # DiffAugment on the input, with the draws the reference made
        from conftest import ReplayDraws
        with ReplayDraws(fix["draws"]):
            out, _, _ = D(x, alpha=fix["alpha"], use_aux_disc=fix["use_aux"])
    else:
        out, _, _ = D(x, alpha=fix["alpha"], use_aux_disc=fix["use_aux"])
    e = max_rel(out, fix["out"])
    print(f"{tag}: logits max_rel {e:.3e}")
    assert e < TOL
    grad_real, = torch.autograd.grad(outputs=out.sum(), inputs=x, create_graph=True)
    # LeakyReLU gates are discontinuous: a pre-activation within fp32 rounding of 0 may take the other
    # branch than the reference's fp32 run did, which perturbs a local patch of ONE image's gradient by ~1e-2
    # (seen here and between the reference's own fp32 and fp64 runs).  Require: almost all elements agree
    # tightly, no element is far off.
    scale = fix["grad_real"].abs().max()
    diff = (grad_real.detach().cpu() - fix["grad_real"]).abs() / scale
    frac_bad = float((diff > TOL).float().mean())
    per_img = [float((diff[i] > TOL).float().mean()) for i in range(diff.shape[0])]
    print(f"{tag}: R1 input-gradient max_rel {float(diff.max()):.3e}, fraction of elements off by >1e-3: {frac_bad:.4f} "
          f"(per image {[round(f, 3) for f in per_img]})")
    # Images are independent in D, so a flipped gate stays inside its image: at least half of the images must agree
    # tightly everywhere, the perturbed ones must stay small.  Seen only in the DiffAugment case on the split-bf16 convs:
    # next to the zero regions of translation / cutout many pre-activations are tiny, and two of them (convs.16.conv1,
    # one image) change sign under the 1e-5 relative GEMM error; exact zeros stay exact.  The fp32 MFMA convs flip
    # a gate as well now and then (d_r16_aux_alpha, one image): any fp32 evaluation order does.
    assert float(diff.max()) < 5e-2 and frac_bad < 0.1
    assert sum(f < 1e-3 for f in per_img) * 2 >= len(per_img)
    loss = torch.nn.functional.softplus(-out).mean() + 0.5 * 10. * grad_real.flatten(1).pow(2).sum(1).mean()
    assert abs(float(loss.detach()) - fix["loss"]) < TOL * max(1.0, abs(fix["loss"]))
    loss.backward()
    torch.cuda.synchronize()
    worst = ("", 0.0)
    GATE_TOL = 2e-2      # one flipped gate in a 2..4-image batch (see above)
    for name, p in D.named_parameters():
        dg = fix["grads"][name]
        if dg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        g = p.grad.reshape(-1).cpu()
        got = g[::dg["stride"]] if dg["stride"] > 1 else g
        en = abs(float(g.double().norm()) - dg["norm"]) / max(dg["norm"], 1e-30)
        es = float((got - dg["sample"]).double().norm() / dg["sample"].double().norm().clamp_min(1e-30))
        if es > worst[1]:
            worst = (name, es)
        assert en < GATE_TOL and es < GATE_TOL, (name, en, es)
    print(f"{tag}: worst param-grad sample rel err {worst[1]:.3e} at {worst[0]}")
