"""Per-layer roofline of the discriminator's convolutions (round-2 verdict item 6): every 3x3 / 1x1 EqualConv2d shape of
Discriminator_MultiScale_Aux (main + aux branch) at a given resolution and batch, timed on the implicit-GEMM kernels the
step uses — forward (cips_conv2d_x3), stride-1 data gradient (same kernel on the flipped filter bank: timed as a forward of
the transposed channel counts) and weight gradient (cips_conv2d_x3_wgrad) — against the split-bf16 roof
(2500 TFLOP/s dense bf16 / 3 passes).  Writes one JSON document (profiles/r3_d_conv_roofline.json when run by
scripts/r3_profiles.sh)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops
from cips3d_amd.discriminator import Discriminator_MultiScale_Aux

ROOF = 2500.0 / 3.0


def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch b of the step (D sees 2b images per forward with the aux image)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    F = torch.nn.functional
    torch.manual_seed(0)
    D = Discriminator_MultiScale_Aux(diffaug=False, max_size=1024, channel_multiplier=2, first_downsample=False, stddev_group=0).to(d)
    # record every implicit-GEMM launch of one D step (two forwards, backward, R1 double-backward: train.py:385-409)
    calls = {"conv2d_x3": {}, "conv2d_x3_wgrad": {}}
    orig_f, orig_w = ops.conv2d_x3, ops.conv2d_x3_wgrad

    def rec_f(wP, xP, B, C, H, W, O, kh, kw, stride, pad, *args, **kwargs):
        k = (B, C, H, W, O, kh, kw, stride, pad)
        calls["conv2d_x3"][k] = calls["conv2d_x3"].get(k, 0) + 1
        return orig_f(wP, xP, B, C, H, W, O, kh, kw, stride, pad, *args, **kwargs)

    def rec_w(dyP, xP, B, C, H, W, O, kh, kw, stride, pad, *args, **kwargs):
        k = (B, C, H, W, O, kh, kw, stride, pad)
        calls["conv2d_x3_wgrad"][k] = calls["conv2d_x3_wgrad"].get(k, 0) + 1
        return orig_w(dyP, xP, B, C, H, W, O, kh, kw, stride, pad, *args, **kwargs)
    ops.conv2d_x3, ops.conv2d_x3_wgrad = rec_f, rec_w
    try:
        b, img = a.batch, a.img_size
        real = (torch.rand(2 * b, 3, img, img, device=d) * 2 - 1).requires_grad_(True)
        fake = torch.rand(2 * b, 3, img, img, device=d) * 2 - 1
        rp = D(real, alpha=1.0, use_aux_disc=True)[0]
        gr, = torch.autograd.grad(rp.sum(), real, create_graph=True)
        fp = D(fake, alpha=1.0, use_aux_disc=True)[0]
        (F.softplus(fp) + F.softplus(-rp) + 5.0 * gr.flatten(1).square().sum(1, keepdim=True)).mean().backward()
        torch.cuda.synchronize()
    finally:
        ops.conv2d_x3, ops.conv2d_x3_wgrad = orig_f, orig_w
    rows, tot = [], {"conv2d_x3": [0.0, 0.0], "conv2d_x3_wgrad": [0.0, 0.0]}
    for kind in ("conv2d_x3", "conv2d_x3_wgrad"):
        for (B, C, H, W, O, kh, kw, stride, pad), n in sorted(calls[kind].items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][4] * kv[0][2] * kv[0][3]):
            Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
            flops = 2.0 * B * O * C * kh * kw * Ho * Wo
            x = torch.randn(B, C, H, W, device=d); w = torch.randn(O, C, kh, kw, device=d) * 0.05
            xP = ops.split_planes_nhwc(x)
            if kind == "conv2d_x3":
                wP, _ = ops.split_planes(w.permute(0, 2, 3, 1).reshape(1, O, kh * kw * C).contiguous(), want_p=True, want_t=False)
                t = timeit(lambda: ops.conv2d_x3(wP, xP, B, C, H, W, O, kh, kw, stride, pad))
            else:
                dyP = ops.split_planes_nhwc(torch.randn(B, O, Ho, Wo, device=d))
                t = timeit(lambda: ops.conv2d_x3_wgrad(dyP, xP, B, C, H, W, O, kh, kw, stride, pad))
            rows.append({"kernel": kind, "B": B, "C": C, "H": H, "W": W, "O": O, "k": kh, "stride": stride, "pad": pad,
                         "launches_per_D_step": n, "gflop": round(flops / 1e9, 2), "us": round(t * 1e6, 1),
                         "tflops": round(flops / t / 1e12, 1), "frac_of_roof": round(flops / t / 1e12 / ROOF, 3),
                         "ms_per_D_step": round(n * t * 1e3, 3)})
            tot[kind][0] += n * flops; tot[kind][1] += n * t
    doc = {"what": "every implicit-GEMM convolution launch of one D step (two forwards of 2b images, backward, R1 double-backward) of "
                   "Discriminator_MultiScale_Aux, main + aux branch: cips_conv2d_x3 (forward / stride-1 data gradient / their "
                   "double-backward forms) and cips_conv2d_x3_wgrad, each unique shape timed alone (10 launches, events); "
                   "split-bf16 roof = 2500 / 3 TFLOP/s",
           "img_size": a.img_size, "batch": a.batch, "roof_tflops": round(ROOF, 1), "shapes": rows,
           "total": {k: {"gflop_per_D_step": round(v[0] / 1e9, 1), "ms_per_D_step": round(v[1] * 1e3, 3),
                         "frac_of_roof": round(v[0] / v[1] / 1e12 / ROOF, 3) if v[1] else None} for k, v in tot.items()}}
    if a.out:
        open(a.out, "w").write(json.dumps(doc, indent=1))
    for r in rows:
        print(r)
    print(json.dumps(doc["total"]))


if __name__ == "__main__":
    main()
