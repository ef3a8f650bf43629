"""Bandwidth of the discriminator step's streaming kernels at the shapes of the C2 GAN step (2b = 64 images), each launch timed
alone with HIP events -> profiles/r4_stream_kernels.json (bytes, us, TB/s, fraction of 8 TB/s).  VERDICT r3 next-7 asks the
three kernels below for >= 40 % of 8 TB/s: upfirdn2d_direct_kernel<1,1,16> (the 4x4 blur), col2im_fixed_kernel<3,3,2,0>
(stride-2 3x3 data gradient), split_nhwc_kernel (NCHW fp32 -> NHWC split planes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cips3d_amd import ops


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    d = torch.device("cuda:0")
    rows = []

    def rec(kernel, shape, nbytes, us):
        tbs = nbytes / us / 1e6
        rows.append({"kernel": kernel, "shape": shape, "bytes": int(nbytes), "us": round(us, 1), "TBps": round(tbs, 2), "frac_of_8TBps": round(tbs / 8, 3)})
        print(f"{kernel:44s} {shape:38s} {nbytes / 1e6:8.1f} MB {us:8.1f} us {tbs:5.2f} TB/s ({tbs / 8:.0%})", flush=True)

    k = torch.tensor([1., 3., 3., 1.]); k = (k[None] * k[:, None]); k = (k / k.sum()).to(d)
    for (B, C, H) in [(64, 512, 64), (64, 256, 64), (64, 512, 32), (64, 512, 16), (8, 128, 256)]:
        x = torch.randn(B, C, H, H, device=d)
        us = timeit(lambda: ops.split_planes_nhwc(x))
        rec("split_nhwc_kernel", f"({B},{C},{H},{H}) fp32 -> NHWC planes", x.numel() * 8, us)
    for (mj, n, down, pad) in [(64 * 512, 64, 1, (2, 2)), (64 * 256, 64, 1, (2, 2)), (64 * 512, 32, 1, (2, 2)), (64 * 512, 65, 1, (1, 1)), (64 * 512, 64, 2, (1, 1))]:
        x = torch.randn(mj, n, n, 1, device=d)
        y = ops.upfirdn2d_op(x, k, 1, 1, down, down, pad[0], pad[1], pad[0], pad[1])
        us = timeit(lambda: ops.upfirdn2d_op(x, k, 1, 1, down, down, pad[0], pad[1], pad[0], pad[1]))
        rec(f"upfirdn2d_direct_kernel<{down},1,*>", f"{mj} planes {n}x{n} down{down} pad{pad}", (x.numel() + y.numel()) * 4, us)
    for (B, C, H) in [(64, 512, 65), (64, 256, 65), (64, 512, 33)]:
        Ho = (H - 3) // 2 + 1
        col = torch.randn(B, C * 9, Ho * Ho, device=d)
        us = timeit(lambda: ops.col2im(col, B, C, H, H, 3, 3, 2, 0))
        rec("col2im_fixed_kernel<3,3,2,0>", f"dcol ({B},{C * 9},{Ho * Ho}) -> dx ({B},{C},{H},{H})", (col.numel() + B * C * H * H) * 4, us)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r4_stream_kernels.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
