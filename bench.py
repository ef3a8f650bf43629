#!/usr/bin/env python
"""bench.py — BASELINE.json metric: rendered images / second, generator forward + backward,
FFHQ geometry, synthetic latents, on N MI355X (one process per GPU, RCCL gradient all-reduce).

Workload at N=1 (config.workload): BASELINE configs[1] = C2 of SURVEY.md §8:
  img_size 64, 24 SIREN evaluations per ray, per-GPU batch 32, fp32.
  Default E=24 as S=24 / hierarchical off (SURVEY §8 "primary"); --hier gives S=12 + 12 resampled.
One step = G(zs, ...) forward on fresh latents + imgs.backward(G0) populating every generator
parameter gradient the reference populates (+ gradient all-reduce when N > 1).
Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for the roofline / cpu_baseline legs.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
G_KW = dict(fov=12, ray_start=0.88, ray_end=1.12, h_stddev=0.3, v_stddev=0.155, psi=1., sample_dist="gaussian")
G_CFG = dict(
    z_dim=256,
    nerf_cfg=dict(in_dim=3, hidden_dim=128, hidden_layers=2, rgb_dim=32, style_dim=128),
    mapping_nerf_cfg=dict(z_dim=256, hidden_dim=128, base_layers=4, head_layers=0),
    inr_cfg=dict(input_dim=32, style_dim=512, hidden_dim=512, pre_rgb_dim=3),
    mapping_inr_cfg=dict(z_dim=512, hidden_dim=512, base_layers=8, head_layers=0, add_norm=True, norm_out=True),
)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--num-steps", type=int, default=None, help="coarse samples per ray")
    ap.add_argument("--hier", action="store_true", help="hierarchical sampling (S coarse + S fine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def gemm_roofline(dev, b, n):
    """Dominant kernel: gemm_f32_kernel (the 512x512 modulated-FC layer GEMM, 34 launches per
    fwd+bwd step).  Timed live with events on the launch stream (torch's current stream)."""
    from cips3d_amd import ops
    x = torch.randn(b, n, 512, device=dev)
    w = torch.randn(b, 512, 512, device=dev) * 0.04
    out = torch.empty(b, n, 512, device=dev)
    for _ in range(3):
        ops.bmm_nn(x, w, out=out, act=1)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.bmm_nn(x, w, out=out, act=1)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    flops = 2.0 * b * n * 512 * 512
    ach = flops / t / 1e12
    return {"bound": "mfma", "kernel": "gemm_f32_kernel<false,false> (modfc 512x512, act=lrelu)",
            "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "launch_us": round(t * 1e6, 1), "flops_per_launch": flops}


def cpu_baseline(img_size, S, hier):
    """Oracle (CPU restatement of the reference path, kind 'port') on this host's cores, bounded
    sample: b=1 image at the bench geometry, 1 warm-up + 2 timed fwd+bwd."""
    from oracle import cips3d_oracle as orc
    from cips3d_amd.generator import GeneratorNerfINR
    torch.manual_seed(1234)
    G = GeneratorNerfINR(**G_CFG, device="cpu")
    sd = dict(G.named_parameters())
    b, n = 1, img_size * img_size
    E = 2 * S if hier else S
    g = torch.Generator().manual_seed(1)

    def one():
        zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
        rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g),
                    phi=torch.randn(b, 1, generator=g), noise_c=torch.randn(b, n, S, 1, generator=g),
                    u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, E, 1, generator=g))
        G.zero_grad()
        out = orc.generator_forward(sd, zs, rand, img_size, 12, 0.88, 1.12, S, 0.3, 0.155, hier)
        out["imgs"].backward(torch.ones_like(out["imgs"]) / out["imgs"].numel())
    one()
    t0 = time.time()
    reps = 2
    for _ in range(reps):
        one()
    dt = (time.time() - t0) / reps
    return {"value": round(b / dt, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle G fwd+bwd, {img_size}x{img_size}, E={E} evals/ray, b=1, {reps} timed iters"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from cips3d_amd.generator import GeneratorNerfINR
    from cips3d_amd.distributed import allreduce_grads

    S = a.num_steps if a.num_steps is not None else (12 if a.hier else 24)
    torch.manual_seed(1234)                       # identical initial weights on every rank
    G = GeneratorNerfINR(**G_CFG, device=dev).to(dev)
    G.device = dev
    torch.manual_seed(1234 + rank)                # per-rank latents / cameras (train.py:221)
    b, img = a.batch, a.img_size
    G0 = torch.randn(b, 3, img, img, device=dev) / (b * 3 * img * img)
    params = list(G.parameters())

    def step():
        zs = G.get_zs(b)
        for p in params:
            p.grad = None
        imgs, _ = G(zs, img_size=img, num_steps=S, hierarchical_sample=a.hier, nerf_noise=0.,
                    return_aux_img=False, grad_points=None, forward_points=None, **G_KW)
        imgs.backward(G0)
        if world > 1:
            allreduce_grads(params)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / a.steps * 1e3
    value = world * b * a.steps / dt
    E = 2 * S if a.hier else S
    line = {
        "metric": "rendered imgs/sec (G fwd+bwd)", "value": round(value, 2), "unit": "img/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"FFHQ r{img}, {E} SIREN evals/ray (num_steps {S}, hierarchical {a.hier}), "
                               f"batch {b}/GPU, G fwd+bwd, all 9 CIPS blocks",
                   "global_batch": world * b, "parallelism": f"dp{world}"},
    }
    if rank == 0:
        if not a.no_roofline:
            line["roofline"] = gemm_roofline(dev, b, img * img)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(img, S, a.hier)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
