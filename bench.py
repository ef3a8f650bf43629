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
    ap.add_argument("--no-exact", action="store_true", help="skip the extra exact-fp32-MFMA timing")
    ap.add_argument("--inr-mode", default=None, choices=["bf16x3", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying the captured hipGraph of the step")
    return ap.parse_args()


BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PF headline includes 2:1 sparsity)


def gemm_roofline(dev, b, n, mode):
    """Dominant kernel of the step (>= 54 % of GPU time): the modulated-FC 512x512 layer GEMM of the CIPS
    head, forward form, timed live with events on the launch stream (torch's current stream).
    bf16x3 mode: gemm_bf16x3_kernel — every fp32 product is 3 bf16 MFMA passes, so the roof for ALGORITHMIC
    (fp32-equivalent) flops is the dense bf16 MFMA peak / 3.  f32 mode: gemm_f32_kernel on fp32 MFMA."""
    from cips3d_amd import ops
    x = torch.randn(b, n, 512, device=dev)
    w = torch.randn(b, 512, 512, device=dev) * 0.04
    flops = 2.0 * b * n * 512 * 512
    if mode == "bf16x3":
        xP, _ = ops.split_planes(x, want_t=False)
        wP, _ = ops.split_planes(w, want_t=False)
        oP = ops.Planes.empty(b, n, 512, device=dev)
        fn = lambda: ops.gemm_x3(xP, wP, n, 512, 512, 512, 512, b, n * 512, 512 * 512, P=oP, act=1)
        name, peak = "gemm_bf16x3_wide_kernel<0,0,0> (modfc 512x512 fwd: 256x256 tiles, lrelu + split-bf16 planes out)", \
            BF16_MFMA_PEAK_TFLOPS / 3.0
    else:
        out = torch.empty(b, n, 512, device=dev)
        fn = lambda: ops.bmm_nn(x, w, out=out, act=1)
        name, peak = "gemm_f32_kernel<false,false> (modfc 512x512 fwd, act=lrelu)", F32_MFMA_PEAK_TFLOPS
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    ach = flops / t / 1e12
    r = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
         "frac": round(ach / peak, 4), "traffic": None, "launch_us": round(t * 1e6, 1), "flops_per_launch": flops}
    if mode == "bf16x3":
        r["note"] = "algorithmic fp32-equivalent flops; raw bf16 MFMA rate = 3x achieved vs 2500 dense peak"
        # HBM bytes per launch of exactly this kernel/shape from the PMC counters, collected in their own
        # rocprofv3 --pmc passes (scripts/pmc_roofline.sh -> profiles/r1_roofline_pmc.json); null if absent
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_roofline_pmc.json")))
            if b == 32 and n == 4096:
                r["traffic"] = pm["hbm_bytes"]
                r["traffic_algorithmic_bytes"] = pm["algorithmic_bytes"]
        except Exception:
            pass
    return r


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
    # CIPS_BENCH_BACKEND=gloo: functional check of the N > 1 path on a box with fewer GPUs than ranks (ranks share
    # devices, the all-reduce goes through the host); never a measurement
    backend = os.environ.get("CIPS_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from cips3d_amd.generator import GeneratorNerfINR
    from cips3d_amd.distributed import GradAllReducer
    from cips3d_amd import ops
    if a.inr_mode:
        ops.INR_MODE = a.inr_mode
    mode = ops.INR_MODE

    S = a.num_steps if a.num_steps is not None else (12 if a.hier else 24)
    torch.manual_seed(1234)                       # identical initial weights on every rank
    G = GeneratorNerfINR(**G_CFG, device=dev).to(dev)
    G.device = dev
    torch.manual_seed(1234 + rank)                # per-rank latents / cameras (train.py:221)
    b, img = a.batch, a.img_size
    G0 = torch.randn(b, 3, img, img, device=dev) / (b * 3 * img * img)
    params = list(G.parameters())
    reduce_grads = GradAllReducer(params)

    def fwd_bwd():
        zs = G.get_zs(b)
        for p in params:
            p.grad = None
        imgs, _ = G(zs, img_size=img, num_steps=S, hierarchical_sample=a.hier, nerf_noise=0.,
                    return_aux_img=False, grad_points=None, forward_points=None, **G_KW)
        imgs.backward(G0)

    graph = None
    if not a.no_graph:
        # ~570 kernel launches per step: capture latents (graph-safe Philox RNG: fresh draws per replay) + forward +
        # backward once, replay per step; the gradient all-reduce stays outside the graph.  Eager on any capture error.
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    fwd_bwd()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # with a process group up, RCCL's watchdog thread may touch the HIP runtime while this thread captures:
            # only this thread's calls belong to the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if world > 1 else "global"):
                fwd_bwd()
        except Exception as e:                      # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    use_graph = [graph is not None]

    def step():
        if use_graph[0]:
            graph.replay()
        else:
            fwd_bwd()
        if world > 1:
            reduce_grads()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def timed(nsteps, nwarm):
        for _ in range(nwarm):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt = timed(a.steps, a.warmup)
    ms = dt / a.steps * 1e3
    value = world * b * a.steps / dt
    exact = None
    if mode == "bf16x3" and not a.no_exact:
        ops.INR_MODE = "f32"          # same step with the head GEMMs and the SIREN forward on exact fp32 MFMA, for
        fwd_was = ops.SIREN_FWD_MODE  # reference (eager launches)
        ops.SIREN_FWD_MODE = "f32"
        was = use_graph[0]
        use_graph[0] = False
        dte = timed(max(2, a.steps // 2), 1)
        use_graph[0] = was
        ops.INR_MODE = mode
        ops.SIREN_FWD_MODE = fwd_was
        ne = max(2, a.steps // 2)
        exact = {"value": round(world * b * ne / dte, 2), "ms_per_step": round(dte / ne * 1e3, 3),
                 "note": "identical step, INR GEMMs and SIREN forward on v_mfma_f32_32x32x2_f32 (CIPS_INR_MODE=f32 CIPS_SIREN_FWD=f32)"}
    E = 2 * S if a.hier else S
    line = {
        "metric": "rendered imgs/sec (G fwd+bwd)", "value": round(value, 2), "unit": "img/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if mode == "f32" else "f32 (dense layers as 3-pass split-bf16 MFMA with fp32 accumulate, ~1e-5 rel.; everything else fp32)",
        "data": "synthetic",
        "config": {"workload": f"FFHQ r{img}, {E} SIREN evals/ray (num_steps {S}, hierarchical {a.hier}), "
                               f"batch {b}/GPU, G fwd+bwd, all 9 CIPS blocks",
                   "global_batch": world * b, "parallelism": f"dp{world}", "inr_gemm_mode": mode,
                   "launch": "hipGraph replay" if use_graph[0] else "eager"},
        **({"backend_note": f"{backend} functional check, not a measurement"} if backend != "nccl" and world > 1 else {}),
    }
    if exact:
        line["exact_f32"] = exact
    if rank == 0:
        if not a.no_roofline:
            line["roofline"] = gemm_roofline(dev, b, img * img, mode)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(img, S, a.hier)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
