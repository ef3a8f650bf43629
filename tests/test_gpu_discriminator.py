"""GPU: discriminator (HIP native ops + fp32 MFMA conv GEMM through the C-ABI) against golden vectors
minted from the reference: logits, R1 gradient w.r.t. the input (double-backward graph) and the
parameter gradients of the full d_loss (train.py:385-409)."""
import pytest
import torch

from conftest import load_golden, check_checksums, max_rel, rel_err, D_CFG
from oracle import cips3d_oracle as orc

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


def _d_oracle64(fix, gates):
    """fp64 oracle of the full d_loss (train.py:385-409) with pinned LeakyReLU gates -> logits, grad_real, {grads}"""
    from test_discriminator_cpu import d_loss_grads
    tape = orc.GateTape(pin=gates)
    tape.keep_preact = True
    out, g, grads = d_loss_grads(fix, torch.float64, tape)
    tape.done()
    return out, g, grads, tape


@pytest.mark.parametrize("conv_mode", ["bf16x3", "f32"])
@pytest.mark.parametrize("tag", ["d_r16", "d_r16_aux_alpha", "d_r16_diffaug"])
def test_discriminator_matches_reference_golden(tag, conv_mode, monkeypatch):
    """Logits, R1 input gradient (double-backward graph) and the parameter gradients of the full d_loss against the
    reference, for the SAME LeakyReLU gates, no gate allowance:
      (A) gates pinned to the reference's (tests/golden/gates_*.pt): everything within 1e-3 of the reference;
      (B) free-running: the product's own gates differ from the reference's at a handful of pre-activations that are
          ambiguous at the conv arithmetic's precision, and its gradients match the fp64 oracle at ITS gates."""
    from cips3d_amd import discriminator as dmod
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    from conftest import load_gates
    monkeypatch.setattr(dmod, "CONV_MODE", conv_mode)
    fix = load_golden(tag)
    ref_gates = load_gates(tag)
    d = torch.device("cuda:0")
    torch.manual_seed(fix["seed"])
    D = Discriminator_MultiScale_Aux(**dict(D_CFG, diffaug=fix.get("diffaug", False)))
    check_checksums(D.state_dict(), fix["state_checksums"])
    D = D.to(d)

    def run(pin=None, rec=None):
        for p in D.parameters():
            p.grad = None
        x = fix["x"].to(d).requires_grad_(True)
        with dmod.gate_debug(pin=pin, rec=rec):
            if fix.get("diffaug"):                 # DiffAugment on the input, with the draws the reference made
                from conftest import ReplayDraws
                with ReplayDraws(fix["draws"]):
                    out, _, _ = D(x, alpha=fix["alpha"], use_aux_disc=fix["use_aux"])
            else:
                out, _, _ = D(x, alpha=fix["alpha"], use_aux_disc=fix["use_aux"])
        grad_real, = torch.autograd.grad(outputs=out.sum(), inputs=x, create_graph=True)
        loss = torch.nn.functional.softplus(-out).mean() + 0.5 * 10. * grad_real.flatten(1).pow(2).sum(1).mean()
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().cpu(), grad_real.detach().cpu(), float(loss.detach()), \
            {n: (None if p.grad is None else p.grad.detach().cpu().double()) for n, p in D.named_parameters()}

    def check(out, grad_real, loss, grads, want_out, want_g, want_grads, what, digest):
        e = max_rel(out, want_out)
        eg = max_rel(grad_real, want_g)
        worst = ("", 0.0)
        for name, g in grads.items():
            dg = fix["grads"][name]
            if dg is None:
                assert g is None or float(g.abs().max()) == 0.0, name
                continue
            t = want_grads[name].reshape(-1).double()
            es = float((g.reshape(-1) - t).norm() / t.norm().clamp_min(1e-300))
            if digest:
                v = g.reshape(-1)
                got = v[::dg["stride"]] if dg["stride"] > 1 else v
                es = max(es, float((got - dg["sample"].double()).norm() / dg["sample"].double().norm().clamp_min(1e-300)),
                         abs(float(v.norm()) - dg["norm"]) / max(dg["norm"], 1e-300))
            if es > worst[1]:
                worst = (name, es)
        print(f"{tag} [{conv_mode}] {what}: logits max_rel {e:.3e}, R1 input-gradient max_rel {eg:.3e}, worst "
              f"parameter-gradient error {worst[1]:.3e} at {worst[0]}")
        assert e < TOL and eg < 1e-4 and worst[1] < 1e-4, what      # measured <= 9e-6 on MI355X

    o64, g64, grads64, tape_ref = _d_oracle64(fix, ref_gates)
    # (A) pinned
    out, grad_real, loss, grads = run(pin=ref_gates)
    assert abs(loss - fix["loss"]) < TOL * max(1.0, abs(fix["loss"]))
    assert max_rel(out, fix["out"]) < TOL and max_rel(grad_real, fix["grad_real"]) < TOL
    check(out, grad_real, loss, grads, o64.float(), g64.float(), grads64, "reference gates pinned (vs fp64 oracle + reference digest)", True)
    # (B) free-running
    rec = []
    out, grad_real, loss, grads = run(rec=rec)
    assert max_rel(out, fix["out"]) < TOL and abs(loss - fix["loss"]) < TOL * max(1.0, abs(fix["loss"]))
    own = [g.cpu() for g in rec]
    assert [tuple(g.shape) for g in own] == [tuple(g.shape) for g in ref_gates]
    flips, worst_amb = 0, 0.0
    for a, b, y in zip(own, ref_gates, tape_ref.preact):
        diff = a != b
        if int(diff.sum()):
            flips += int(diff.sum())
            worst_amb = max(worst_amb, float(y[diff].max() / y.pow(2).mean().sqrt()))
    total = sum(g.numel() for g in ref_gates)
    print(f"{tag} [{conv_mode}] free: {flips} of {total} gates differ from the reference's; largest |pre-activation| / rms "
          f"among them {worst_amb:.2e}")
    assert worst_amb < (2e-4 if conv_mode == "bf16x3" else 2e-5) and flips <= 4 + total * 1e-4
    if flips:
        o64, g64, grads64, _ = _d_oracle64(fix, own)
    check(out, grad_real, loss, grads, o64.float(), g64.float(), grads64, "free-running (vs fp64 oracle at the product's gates)", False)


def test_diffaugment_hip_operator_matches_reference_golden():
    """The fused DiffAugment operator (cips_diffaug: forward, adjoint, and — the forward without its constant — the
    adjoint's backward) against (1) the vectors minted from the reference with its recorded draws, every policy of the
    fixture (all three stages, translation + cutout, colour only): output and input gradient; (2) the oracle's op-by-op
    restatement through a double-backward: an R1-style penalty on the input gradient, differentiated w.r.t. a parameter
    that scales the input."""
    from conftest import ReplayDraws
    from cips3d_amd import discriminator as dm
    d = torch.device("cuda:0")
    for case in load_golden("diffaug_cases"):
        x = case["x"].to(d).requires_grad_(True)
        with ReplayDraws(case["draws"]):
            y = dm.DiffAugment(x, policy=case["policy"])
        assert y.shape == case["y"].shape and max_rel(y, case["y"]) < 1e-6, case["policy"]
        if "color" not in case["policy"]:
            assert torch.equal(y.cpu(), case["y"])                       # shift + hole only: values pass through untouched
        gx, = torch.autograd.grad((y * case["g0"].to(d)).sum(), x)
        assert max_rel(gx, case["gx"]) < 1e-6, case["policy"]

    case = [c for c in load_golden("diffaug_cases") if c["policy"] == "color,translation,cutout"][0]

    def penalty(fused):
        dev = d if fused else torch.device("cpu")
        w = torch.tensor(0.7, device=dev, requires_grad=True)
        xi = case["x"].to(dev).requires_grad_(True)
        if fused:
            with ReplayDraws(case["draws"]):
                yy = dm.DiffAugment(xi * w, policy=case["policy"])
        else:
            yy = orc.diff_augment(xi * w, iter(t for _, t in case["draws"]), case["policy"])
        out = (yy * yy * case["g0"].to(dev)).sum()                       # nonlinear in y: the input gradient depends on w
        g, = torch.autograd.grad(out, xi, create_graph=True)
        pen = g.pow(2).sum()
        gw, = torch.autograd.grad(pen, w)
        return float(out), g.detach().cpu(), float(gw)

    o1, g1, w1 = penalty(True)
    o0, g0, w0 = penalty(False)
    assert abs(o1 - o0) <= 1e-5 * abs(o0) and max_rel(g1, g0) < 1e-5 and abs(w1 - w0) <= 1e-4 * abs(w0), (o1, o0, w1, w0)


@pytest.mark.parametrize("policy", ["translation,color", "cutout,color,translation", "cutout,translation", "color,color"])
def test_diffaugment_applies_stages_in_the_order_listed(policy):
    """diffaug.py:13-16 applies the stages in the order the policy lists them (the product only took subsequences
    of color,translation,cutout): a non-canonical order runs stage by stage on the same operator, with the reference's draw
    order — output and input gradient against the oracle's op-by-op form on the same draws."""
    from conftest import ReplayDraws
    from cips3d_amd import discriminator as dm
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    b, h, w = 3, 20, 24
    x = torch.randn(b, 3, h, w, generator=g)
    g0 = torch.randn(b, 3, h, w, generator=g)
    draws = []
    for st in policy.split(","):
        if st == "color":
            draws += [("rand", torch.rand(b, 1, 1, 1, generator=g)) for _ in range(3)]
        elif st == "translation":
            sx, sy = int(h * 0.125 + 0.5), int(w * 0.125 + 0.5)
            draws += [("randint", torch.randint(-sx, sx + 1, (b, 1, 1), generator=g)), ("randint", torch.randint(-sy, sy + 1, (b, 1, 1), generator=g))]
        else:
            ch, cw = int(h * 0.2 + 0.5), int(w * 0.2 + 0.5)
            draws += [("randint", torch.randint(0, h + (1 - ch % 2), (b, 1, 1), generator=g)),
                      ("randint", torch.randint(0, w + (1 - cw % 2), (b, 1, 1), generator=g))]
    xr = x.clone().requires_grad_(True)
    yr = orc.diff_augment(xr, iter(t for _, t in draws), policy)
    gr, = torch.autograd.grad((yr * g0).sum(), xr)
    xd = x.to(d).requires_grad_(True)
    with ReplayDraws(draws):
        y = dm.DiffAugment(xd, policy=policy)
    gx, = torch.autograd.grad((y * g0.to(d)).sum(), xd)
    assert max_rel(y, yr) < 1e-6 and max_rel(gx, gr) < 1e-6, (policy, max_rel(y, yr), max_rel(gx, gr))


def test_diffaug_sums_at_256_match_oracle():
    """the per-image sums of cips_diffaug are reduced in 32 slices per image: brightness / contrast means at 256 x 256
    against the oracle's op-by-op restatement (CPU) with the same draws"""
    from conftest import ReplayDraws
    from cips3d_amd import discriminator as dm
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    b, h, w = 3, 256, 256
    x = torch.randn(b, 3, h, w, generator=g)
    sx, sy, ch, cw = int(h * 0.125 + 0.5), int(w * 0.125 + 0.5), int(h * 0.2 + 0.5), int(w * 0.2 + 0.5)
    draws = [("rand", torch.rand(b, 1, 1, 1, generator=g)) for _ in range(3)]
    draws += [("randint", torch.randint(-sx, sx + 1, [b, 1, 1], generator=g)), ("randint", torch.randint(-sy, sy + 1, [b, 1, 1], generator=g)),
              ("randint", torch.randint(0, h + (1 - ch % 2), [b, 1, 1], generator=g)),
              ("randint", torch.randint(0, w + (1 - cw % 2), [b, 1, 1], generator=g))]
    with ReplayDraws(draws):
        y = dm.DiffAugment(x.to(d), policy="color,translation,cutout")
    ref = orc.diff_augment(x, iter(t for _, t in draws), "color,translation,cutout")
    assert max_rel(y, ref) < 2e-6


@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (3, 5, 65, 65), (1, 4, 256, 256), (2, 3, 7, 5), (4, 512, 4, 4)])
def test_lrelu_backward_with_bias_gradient_in_one_pass(shape):
    """cips_lrelu_bwd_bias against the two-step form it replaces (cips_fused_bias_act act 3 grad 1, then a torch
    reduction): the gated gradient bit for bit, the bias gradient to summation order"""
    from cips3d_amd import ops
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    grad = torch.randn(*shape, generator=g).to(d)
    out = torch.randn(*shape, generator=g).to(d)
    gin, gb = ops.lrelu_bwd_bias(grad, out, 0.2, 2 ** 0.5)
    ref = ops.fused_bias_act(grad, grad.new_empty(0), out, 3, 1, 0.2, 2 ** 0.5)
    assert torch.equal(gin, ref)
    want = ref.double().sum((0, 2, 3))
    assert gb.shape == (shape[1],) and float((gb.double() - want).abs().max()) <= 1e-5 * float(ref.abs().sum((0, 2, 3)).max())


@pytest.mark.parametrize("cfg", [(5, 32, 64, 16, 3, 1, 1, 3), (5, 32, 64, 16, 3, 1, 1, 7), (3, 64, 32, 33, 3, 2, 0, 5),
                                 (2, 32, 32, 16, 1, 1, 0, 1), (4, 64, 64, 12, 3, 1, 1, 13)])
def test_conv_weight_gradient_ragged_chunks(cfg):
    """cips_conv2d_x3_wgrad cuts the B*Ho*Wo contraction into nchunks nearly equal k-tile ranges (chunk lengths differ by
    one 32-row k-tile when nchunks does not divide the k-tile count): every chunk count must give the same gradient"""
    from cips3d_amd import ops
    B, C, O, H, k, stride, pad, nch = cfg
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31 + nch)
    x = torch.randn(B, C, H, H, generator=g, dtype=torch.float64, requires_grad=False)
    w = (torch.randn(O, C, k, k, generator=g, dtype=torch.float64) / (C * k * k) ** 0.5).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, w, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * dy).sum().backward()
    Ho = y.shape[2]
    assert (B * Ho * Ho) % 32 == 0
    dw = ops.conv2d_x3_wgrad(ops.split_planes_nhwc(dy.float().to(d)), ops.split_planes_nhwc(x.float().to(d)), B, C, H, H, O, k, k,
                             stride, pad, scale=0.5, nch=nch)
    assert dw is not None and rel_err(dw, 0.5 * w.grad) < 3e-5, (cfg, float(rel_err(dw, 0.5 * w.grad)))


@pytest.mark.parametrize("cfg", [(2, 64, 96, 16, 3, 1, 1, 4), (3, 32, 32, 16, 3, 1, 1, 3), (2, 64, 64, 17, 3, 2, 0, 5),
                                 (2, 96, 32, 16, 1, 1, 0, 3), (1, 32, 64, 24, 3, 1, 1, 7), (4, 512, 512, 32, 3, 1, 1, 2),
                                 (2, 64, 320, 32, 1, 1, 0, 1),
                                 # output planes under 256 pixels: the batch folded into the pixel dimension (round 6)
                                 (8, 64, 96, 8, 3, 1, 1, 4), (32, 64, 64, 4, 3, 1, 1, 3), (6, 64, 64, 9, 3, 2, 0, 2), (5, 96, 32, 8, 1, 1, 0, 1),
                                 (3, 32, 32, 12, 3, 1, 1, 2)])
def test_implicit_conv_split_contraction(cfg):
    """cips_conv2d_x3 with ksplit > 1 (the contraction cut into ragged k-tile ranges computed by different workgroups,
    partial planes summed into y): same result as the unsplit launch to fp32 summation order, and as torch in fp64"""
    from cips3d_amd import ops
    B, C, O, H, k, stride, pad, ks = cfg
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17 + ks)
    x = torch.randn(B, C, H, H, generator=g, dtype=torch.float64)
    w = torch.randn(O, C, k, k, generator=g, dtype=torch.float64) / (C * k * k) ** 0.5
    y = torch.nn.functional.conv2d(x, w, stride=stride, padding=pad)
    wP, _ = ops.split_planes(w.float().to(d).permute(0, 2, 3, 1).reshape(1, O, k * k * C).contiguous(), want_p=True, want_t=False)
    xP = ops.split_planes_nhwc(x.float().to(d))
    y1 = ops.conv2d_x3(wP, xP, B, C, H, H, O, k, k, stride, pad, ksplit=1)
    y2 = ops.conv2d_x3(wP, xP, B, C, H, H, O, k, k, stride, pad, ksplit=ks)
    y3 = ops.conv2d_x3(wP, xP, B, C, H, H, O, k, k, stride, pad)                 # the launcher's own choice
    assert rel_err(y1, y) < 3e-5 and rel_err(y2, y) < 3e-5 and rel_err(y3, y) < 3e-5
    assert rel_err(y2, y1) < 2e-6


@pytest.mark.parametrize("cfg", [(2, 64, 64, 32, False), (2, 32, 96, 32, True), (3, 64, 32, 16, False), (2, 32, 64, 64, True),
                                 (2, 32, 32, 8, False)])
def test_conv_layer_with_activation_in_the_gemm_epilogue(cfg, monkeypatch):
    """ConvLayer = [Blur,] EqualConv2d, FusedLeakyReLU (discriminator.py:134-222) with bias + LeakyReLU * sqrt(2) applied in
    the implicit-GEMM epilogue (also through the split-contraction launch: 16 x 16 planes) against the three separate
    kernels: output, input gradient, and the R1-style double backward into input, weight and bias"""
    from cips3d_amd import discriminator as dm
    B, C, O, H, down = cfg
    d = torch.device("cuda:0")
    torch.manual_seed(5)
    layer = dm.ConvLayer(C, O, 3, downsample=down).to(d)
    with torch.no_grad():
        layer.flrelu.bias.copy_(torch.randn(O, device=d) * 0.3)
    x0 = torch.randn(B, C, H, H, device=d)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(dm, "_CONV_ACT_FUSED", fused)
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = layer(x)
        up = torch.randn(y.shape, device=d, generator=torch.Generator(device=d).manual_seed(1))
        gx, = torch.autograd.grad((y * up).sum(), x, create_graph=True)
        ((gx ** 2).sum() + (y ** 2).sum()).backward()
        res[fused] = (y.detach(), gx.detach(), x.grad.clone(), layer.equal_conv.weight.grad.clone(), layer.flrelu.bias.grad.clone())
    for a, b, what in zip(res[True], res[False], ("y", "dx", "x.grad", "w.grad", "bias.grad")):
        assert a.shape == b.shape and rel_err(a, b) < 1e-5, (what, float(rel_err(a, b)))
    assert H < 16 or float((res[True][0] - res[False][0]).abs().max()) <= 1e-6 * float(res[False][0].abs().max())


@pytest.mark.parametrize("B,K,O,bias,act", [(6, 8192, 512, True, True), (6, 512, 1, True, False), (5, 64, 8, False, False),
                                            (3, 2048, 4, True, False)])
def test_equal_linear_hip_forward_backward_double_backward(B, K, O, bias, act):
    """EqualLinear on the HIP library (cips_equal_linear: chunked exact-fp32 GEMM for the 8192 -> 512 layer, streaming
    kernels for 512 -> 1) against the reference formula (discriminator.py:254-288) evaluated in fp64: output, input and
    parameter gradients, and an R1-style second-order gradient (penalty on the input gradient, differentiated w.r.t. the
    weight) through the three-form closure _EqLinFwd / _EqLinDx / _EqLinDw."""
    from cips3d_amd.discriminator import EqualLinear
    import math
    d = torch.device("cuda:0")
    torch.manual_seed(B + K + O)
    lin = EqualLinear(K, O, bias=bias, bias_init=0.3 if bias else 0, activation="fused_lrelu" if act else None).to(d)
    x = torch.randn(B, K, device=d, requires_grad=True)
    c = torch.randn(B, O, device=d)

    def ref(xr, w, bvec):
        out = xr @ (w * lin.scale).t()
        if act:
            return torch.nn.functional.leaky_relu(out + bvec * lin.lr_mul, 0.2) * math.sqrt(2)
        return out + (bvec * lin.lr_mul if bvec is not None else 0)

    y = lin(x)
    gx, = torch.autograd.grad((y * c).sum(), x, create_graph=True)
    pen = gx.pow(2).sum()
    gw_pen, = torch.autograd.grad(pen, lin.weight, retain_graph=True)
    grads = torch.autograd.grad((y * c).sum() + 0.1 * pen, [x, lin.weight] + ([lin.bias] if bias else []))
    xr = x.detach().double().requires_grad_(True); wr = lin.weight.detach().double().requires_grad_(True)
    br = lin.bias.detach().double().requires_grad_(True) if bias else None
    yr = ref(xr, wr, br)
    gxr, = torch.autograd.grad((yr * c.double()).sum(), xr, create_graph=True)
    penr = gxr.pow(2).sum()
    gw_penr, = torch.autograd.grad(penr, wr, retain_graph=True)
    gradsr = torch.autograd.grad((yr * c.double()).sum() + 0.1 * penr, [xr, wr] + ([br] if bias else []))
    assert max_rel(y, yr) < 2e-6 and max_rel(gx, gxr) < 2e-6, (max_rel(y, yr), max_rel(gx, gxr))
    assert max_rel(gw_pen, gw_penr) < 1e-5, max_rel(gw_pen, gw_penr)
    for a, b in zip(grads, gradsr):
        assert max_rel(a, b) < 1e-5, max_rel(a, b)


def test_d_step_captured_in_a_hipgraph_replays_like_eager():
    """A D step (forward, R1 double-backward, FusedClipAdamEMA) captured once and replayed three times gives the parameters
    of three eager steps (the weight-plane cache was keyed on the parameter version, which a replay never bumps —
    replays would have convolved with the planes of the captured step's weights —, and planes built during a capture were
    stored in the global cache although they live in the graph's private pool).  After the replays an EAGER forward on the
    replay-updated weights must also see the current weights."""
    from cips3d_amd import discriminator as dmod
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, 3, 16, 16, generator=g).to(d) for _ in range(4)]

    def make():
        torch.manual_seed(11)
        D = Discriminator_MultiScale_Aux(**D_CFG).to(d)
        opt = FusedClipAdamEMA(list(D.parameters()), lr=2e-3, betas=(0.0, 0.999), max_norm=10.0, capture_slots=2)
        return D, opt

    def d_step(D, opt, x_static):
        x = x_static.detach().requires_grad_(True)
        out, _, _ = D(x, alpha=1.0, use_aux_disc=False)
        gr, = torch.autograd.grad(outputs=out.sum(), inputs=x, create_graph=True)
        loss = torch.nn.functional.softplus(-out).mean() + 5.0 * gr.flatten(1).pow(2).sum(1).mean()
        loss.backward()
        opt.step()

    def grads_to_static_zero(D):
        for p in D.parameters():
            p.grad = torch.zeros_like(p)          # static gradient buffers: backward accumulates into them

    # eager twin: warm-up step on xs[0], then three steps
    De, oe = make()
    for i in range(4):
        grads_to_static_zero(De)
        d_step(De, oe, xs[i])
    torch.cuda.synchronize()
    # captured: the same warm-up eagerly (on a side stream, as torch asks), then ONE capture, replayed on xs[1..3]
    Dg, og = make()
    x_static = xs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        grads_to_static_zero(Dg)
        d_step(Dg, og, x_static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in Dg.parameters():
        p.grad.zero_()
    graph = torch.cuda.CUDAGraph()
    n_cache_before = sum(len(v[1]) for v in dmod._WCACHE.values())
    snap = [p.detach().clone() for p in Dg.parameters()]
    with torch.cuda.graph(graph):
        for p in Dg.parameters():
            p.grad.zero_()
        d_step(Dg, og, x_static)
    # nothing built during the capture may have entered the global cache (the planes live in the graph's pool)
    assert sum(len(v[1]) for v in dmod._WCACHE.values()) <= n_cache_before
    for p, s_ in zip(Dg.parameters(), snap):
        assert torch.equal(p.detach(), s_)        # capture does not execute
    for i in (1, 2, 3):
        x_static.copy_(xs[i])
        graph.replay()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, a), b in zip(De.named_parameters(), Dg.parameters()):
        if a.grad is None or not float(a.grad.abs().max()):
            continue
        # Adam with beta1 = 0 moves an element by about +-lr per step whatever the gradient's size: tensors are tight,
        # single elements only bounded (tests/test_gpu_train_step.py) — both runs execute the same kernels on the same
        # inputs here, so they agree far below that
        err = float((a - b).abs().max())
        worst = max(worst, err)
        assert err <= 1e-6 + 1e-5 * float(a.abs().max()), (n, err)
    # an eager forward after the replays: its planes must come from the CURRENT weights (the version counter did not move)
    dmod.invalidate_weight_cache(Dg)              # documented duty of a caller that replays an optimizer step
    with torch.no_grad():
        oe_, _, _ = De(xs[0], alpha=1.0, use_aux_disc=False)
        og_, _, _ = Dg(xs[0], alpha=1.0, use_aux_disc=False)
    assert max_rel(og_, oe_) < 1e-5, (max_rel(og_, oe_), worst)


@pytest.mark.parametrize("cfg", [(2, 64, 96, 17, 3), (2, 64, 64, 33, 3), (3, 32, 64, 65, 3), (2, 96, 64, 18, 3), (2, 64, 64, 20, 4),
                                 (1, 512, 512, 65, 3), (2, 64, 64, 16, 2)])
def test_stride2_data_gradient_as_parity_subconvolutions(cfg):
    """cips_conv2d_x3_dgrad_s2: the data gradient of a stride-2 unpadded convolution as four stride-1
    sub-convolutions over dy, one per parity class of the input pixel, in one launch — no dcol, no col2im.  Against torch's
    conv_transpose in fp64 (3e-5: the split-bf16 class) and against the K-major GEMM + col2im path it replaces (same class);
    then the Blur's transpose straight from the parity blocks (cips_upfirdn2d_parity) against cips_upfirdn2d on the
    interleaved tensor: bit-identical."""
    from cips3d_amd import ops
    from cips3d_amd import discriminator as dm
    B, C, O, H, k = cfg
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H + C + k)
    Ho = (H - k) // 2 + 1
    w = torch.randn(O, C, k, k, generator=g, dtype=torch.float64) / (C * k * k) ** 0.5
    dy = torch.randn(B, O, Ho, Ho, generator=g, dtype=torch.float64)
    want = torch.nn.functional.conv_transpose2d(dy, w, stride=2, output_padding=(H - k) % 2)
    assert want.shape == (B, C, H, H)
    banks, w_off = ops.dgrad_s2_banks(w.float().to(d))
    dyP = ops.split_planes_nhwc(dy.float().to(d))
    dxp, out_off = ops.conv2d_x3_dgrad_s2(banks, w_off, dyP, B, C, H, H, O, k, k)
    got = ops.parity_to_nchw(dxp, out_off, B, C, H, H)
    e_new = rel_err(got, want)
    try:
        e_old = rel_err(dm._conv_bwd_data(dy.float().to(d), w.float().to(d), (B, C, H, H), 2, 0), want)
    except RuntimeError:                       # shapes the GEMM + col2im path has no kernel for (odd pixel counts)
        e_old = None
    print(f"dgrad s2 {cfg}: parity {e_new:.2e}  col2im path {e_old if e_old is None else format(e_old, '.2e')}")
    assert e_new < 3e-5 and (e_old is None or e_new < 2 * e_old + 1e-6)
    # the padding elements of every block are zeros (they are written, not left uninitialised)
    nps = ops.dgrad_s2_layout(H, H)
    for a in range(2):
        for b in range(2):
            hs, ws = (H - a + 1) // 2, (H - b + 1) // 2
            i = 2 * a + b
            blk = dxp[out_off[i]:out_off[i] + B * C * nps[i]].view(B * C, nps[i])
            assert not bool(blk[:, hs * ws:].any())
    kern = dm.make_kernel([1, 3, 3, 1]).to(d)
    kf = torch.flip(kern, [0, 1]).contiguous()
    a1 = ops.upfirdn2d_parity(dxp, out_off, kf, B * C, H, H, 1, 1, 1, 1)
    a0 = ops.upfirdn2d_op(got.reshape(B * C, H, H, 1), kf, 1, 1, 1, 1, 1, 1, 1, 1).view(B * C, H - 1, H - 1)
    assert torch.equal(a1, a0)
    if cfg == (2, 64, 96, 17, 3):
        # a contraction of one k-tile (O = 32, the single-tap class) is refused, loudly: the caller keeps the col2im path
        b32, o32 = ops.dgrad_s2_banks(w[:32].float().to(d))
        with pytest.raises(RuntimeError):
            ops.conv2d_x3_dgrad_s2(b32, o32, ops.split_planes_nhwc(dy[:, :32].float().to(d)), B, C, H, H, 32, k, k)


@pytest.mark.parametrize("cfg", [(2, 64, 64, 64, False), (2, 64, 128, 32, False), (2, 64, 64, 32, True), (2, 64, 64, 16, False),
                                 (2, 32, 32, 64, True)])
def test_resblock_with_the_blur_folded_into_its_convolutions(cfg, monkeypatch):
    """ResBlock (discriminator.py:224-252) with FOLD_BLUR — the Blur inside the convolution Functions, the stride-2
    data gradient as parity sub-convolutions with the Blur's transpose read from the parity blocks, no UpFirDn2d node —
    against the same block with the Blur as its own op: output, R1-style input gradient, and the double backward into input,
    weights and biases.  The forward is bit-identical (same planes, same GEMM); gradients agree to the split-bf16 class."""
    from cips3d_amd import discriminator as dm
    B, C, O, H, first = cfg
    d = torch.device("cuda:0")
    torch.manual_seed(11)
    blk = dm.ResBlock(C, O, first_downsample=first).to(d)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, dm.FusedLeakyReLU):
                m.bias.copy_(torch.randn_like(m.bias) * 0.3)
    x0 = torch.randn(B, C, H, H, device=d)
    res = {}
    for fold in (True, False):
        monkeypatch.setattr(dm, "FOLD_BLUR", fold)
        blk.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        up = torch.randn(y.shape, device=d, generator=torch.Generator(device=d).manual_seed(1))
        gx, = torch.autograd.grad((y * up).sum(), x, create_graph=True)
        ((gx ** 2).sum() + (y ** 2).sum()).backward()
        res[fold] = [y.detach(), gx.detach(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
    names = ["y", "dx", "x.grad"] + [n for n, _ in blk.named_parameters()]
    assert torch.equal(res[True][0], res[False][0])
    for a, b, what in zip(res[True], res[False], names):
        assert a.shape == b.shape and rel_err(a, b) < 2e-5, (what, float(rel_err(a, b)))


def test_batched_weight_planes_are_the_per_layer_ones_bit_for_bit():
    """cips_conv_weight_prep_batch (round 6): the forward bank, the flipped / transposed bank and the four parity banks of every
    convolution weight of a network in one launch per 24 layers, against the per-layer builders they replace (multiply, permuting
    copy, flip, split_planes; ops.dgrad_s2_banks): identical bf16 planes; and the cache serves them until the weight changes."""
    from cips3d_amd import ops
    from cips3d_amd import discriminator as dm
    d = torch.device("cuda:0")
    torch.manual_seed(3)
    D = dm.Discriminator_MultiScale(**dict(D_CFG)).to(d)
    layers = D._conv_layers(6)
    dm.invalidate_weight_cache(D)
    dm.prepare_weight_planes(layers)
    n = 0
    for conv, alt in layers:
        w, sc = conv.weight, conv.scale
        O, C, kh, kw = w.shape
        ent = dm._WCACHE[id(w)][1]
        got = ent[("fwd", float(sc))][2]
        want = dm._w_planes_raw((w.detach() * sc))
        assert torch.equal(got.hi.view(torch.int16).reshape(-1), want.hi.view(torch.int16).reshape(-1))
        assert torch.equal(got.lo.view(torch.int16).reshape(-1), want.lo.view(torch.int16).reshape(-1))
        if alt == "flipT":
            g2 = ent[("flipT", float(sc))][2]
            w2 = dm._w_planes_raw((w.detach() * sc).flip(2, 3).transpose(0, 1))
        else:
            g2, offs = ent[("s2banks", float(sc))][2]
            w2, offs2 = ops.dgrad_s2_banks(w.detach() * sc)
            assert list(offs) == list(offs2)
        assert torch.equal(g2.hi.view(torch.int16).reshape(-1), w2.hi.view(torch.int16).reshape(-1)), (alt, tuple(w.shape))
        assert torch.equal(g2.lo.view(torch.int16).reshape(-1), w2.lo.view(torch.int16).reshape(-1)), (alt, tuple(w.shape))
        n += 1
    assert n == 13
    # served from the cache: the lookups of the convolution Functions return the very objects
    conv, _ = layers[0]
    assert dm._w_planes(conv.weight, conv.scale) is dm._WCACHE[id(conv.weight)][1][("fwd", float(conv.scale))][2]
    with torch.no_grad():
        conv.weight.mul_(1.5)                          # version bump: stale
    before = dm._WCACHE[id(conv.weight)][1][("fwd", float(conv.scale))][2]
    dm.prepare_weight_planes(layers)
    assert dm._WCACHE[id(conv.weight)][1][("fwd", float(conv.scale))][2] is not before


@pytest.mark.parametrize("cfg", [(2, 64, 64, 32, False), (2, 64, 96, 32, True), (4, 64, 64, 16, True), (2, 32, 64, 64, False), (8, 64, 64, 8, False)])
def test_gated_gradient_as_planes_only(cfg, monkeypatch):
    """ConvLayer (EqualConv2d + FusedLeakyReLU) with its gated gradient written as NHWC planes only (cips_lrelu_bwd_bias_nhwc, round 6)
    against the fp32 gradient + split path: the planes are the same bit for bit, so input gradient, R1-style double backward,
    weight and bias gradients are identical up to the bias sums' order; and a planes-only gradient that reaches a value-reading
    path raises instead of reading zeros."""
    from cips3d_amd import ops
    from cips3d_amd import discriminator as dm
    B, C, O, H, down = cfg
    d = torch.device("cuda:0")
    torch.manual_seed(7)
    layer = dm.ConvLayer(C, O, 3, downsample=down).to(d)
    with torch.no_grad():
        layer.flrelu.bias.copy_(torch.randn(O, device=d) * 0.3)
    x0 = torch.randn(B, C, H, H, device=d)
    res = {}
    for po in (True, False):
        monkeypatch.setattr(dm, "PLANES_ONLY_GRADIENT", po)
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = layer(x)
        up = torch.randn(y.shape, device=d, generator=torch.Generator(device=d).manual_seed(1))
        gx, = torch.autograd.grad((y * up).sum(), x, create_graph=True)
        ((gx ** 2).sum() + (y ** 2).sum()).backward()
        res[po] = (gx.detach(), x.grad.clone(), layer.equal_conv.weight.grad.clone(), layer.flrelu.bias.grad.clone())
    for a, b, what in zip(res[True], res[False], ("dx", "x.grad", "w.grad", "bias.grad")):
        if what == "bias.grad":
            assert rel_err(a, b) < 1e-6, (what, float(rel_err(a, b)))
        else:
            assert torch.equal(a, b), (what, float(rel_err(a, b)))
    # the kernel against the two it replaces, and the guard
    g = torch.randn(B, O, 16, 16, device=d); r = torch.randn(B, O, 16, 16, device=d)
    P, gb = ops.lrelu_bwd_bias_nhwc(g, r, 0.2, 2 ** 0.5)
    gin, gb0 = ops.lrelu_bwd_bias(g, r, 0.2, 2 ** 0.5)
    R = ops.split_planes_nhwc(gin)
    assert torch.equal(P.hi.view(torch.int16), R.hi.view(torch.int16)) and torch.equal(P.lo.view(torch.int16), R.lo.view(torch.int16))
    assert rel_err(gb, gb0) < 1e-6
    ph, _ = dm.FusedLeakyReLUFunctionBackward.apply(g, r, 0.2, 2 ** 0.5, True)
    with pytest.raises(RuntimeError):
        dm._dense(ph)
    with pytest.raises(RuntimeError):
        dm._conv_bwd_data(ph, torch.randn(O, 3, 1, 1, device=d), (B, 3, 16, 16), 1, 0)
