"""GPU: the drop-in generator and discriminator under torch.nn.parallel.DistributedDataParallel exactly as the
reference's training script wraps them (exp/cips3d/scripts/train.py:41-49 setup_ddp, :235-236
DDP(..., find_unused_parameters=True, broadcast_buffers=False)), through one D step (two D forwards, R1
double-backward, train.py:383-440) and one G step through the frozen D (:441-466).  Two ranks share cuda:0 and
exchange gradients over gloo (the driver's boxes have one GPU; RCCL needs one device per rank).  After each backward
the gradients on every rank must equal the mean of the two ranks' single-process gradients — for every parameter DDP
leaves a gradient on, including those used on no rank (None)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

IMG, B, S = 16, 2, 4
G_KW = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=S, h_stddev=0.3, v_stddev=0.155, hierarchical_sample=True,
            psi=1., sample_dist="gaussian")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _requires_grad(m, flag):
    for p in m.parameters():
        p.requires_grad_(flag)


def _build(dev):
    from conftest import G_CFG, D_CFG
    from cips3d_amd import GeneratorNerfINR, Discriminator_MultiScale_Aux
    torch.manual_seed(1234)                             # same initial weights on every rank (build_model + seed)
    G = GeneratorNerfINR(**G_CFG, device=dev).to(dev); G.device = dev
    D = Discriminator_MultiScale_Aux(**D_CFG).to(dev)
    return G, D


def _steps(G_call, D_call, G, D, data_rank, dev):
    """the D step and the G step of train.py for the data of `data_rank` -> (D grads, G grads) as lists of CPU tensors"""
    torch.manual_seed(777 + data_rank)                  # latents, cameras, jitter and the "real" images of this rank
    real = torch.rand(B, 3, IMG, IMG, device=dev) * 2 - 1
    # ---- TRAIN DISCRIMINATOR (train.py:334-440) ----
    _requires_grad(G, False); _requires_grad(D, True)
    with torch.no_grad():
        gen, _ = G_call(G.get_zs(B), img_size=IMG, nerf_noise=0.1, return_aux_img=True, forward_points=None,
                        grad_points=None, **G_KW)
    real2 = torch.cat([real, real], dim=0).requires_grad_()
    r_preds, _, _ = D_call(real2, alpha=0.8, use_aux_disc=True)
    grad_real, = torch.autograd.grad(outputs=r_preds.sum(), inputs=real2, create_graph=True)
    pen = 0.5 * 10.0 * grad_real.flatten(start_dim=1).square().sum(dim=1, keepdim=True) + 0. * r_preds
    g_preds, _, _ = D_call(gen, alpha=0.8, use_aux_disc=True)
    d_loss = (F.softplus(g_preds) + F.softplus(-r_preds) + pen).mean()
    for p in D.parameters():
        p.grad = None
    d_loss.backward()
    d_grads = [None if p.grad is None else p.grad.detach().cpu().clone() for p in D.parameters()]
    # ---- TRAIN GENERATOR (train.py:441-466) ----
    _requires_grad(G, True); _requires_grad(D, False)
    imgs, _ = G_call(G.get_zs(B), img_size=IMG, nerf_noise=0.1, return_aux_img=True, grad_points=None,
                     forward_points=None, **G_KW)
    preds, _, _ = D_call(imgs.to(torch.float32), alpha=0.8, use_aux_disc=True)
    g_loss = F.softplus(-preds).mean()
    for p in G.parameters():
        p.grad = None
    g_loss.backward()
    g_grads = [None if p.grad is None else p.grad.detach().cpu().clone() for p in G.parameters()]
    torch.cuda.synchronize()
    return d_grads, g_grads


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # setup_ddp (train.py:41-49) with gloo
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from torch.nn.parallel import DistributedDataParallel as DDP
    G, D = _build(dev)
    G_ddp = DDP(G, device_ids=[0], find_unused_parameters=True, broadcast_buffers=False)
    D_ddp = DDP(D, device_ids=[0], find_unused_parameters=True, broadcast_buffers=False)
    G_ddp.module.set_device(dev)
    d_grads, g_grads = _steps(G_ddp, D_ddp, G, D, rank, dev)
    q.put((rank, [None if t is None else t.numpy() for t in d_grads], [None if t is None else t.numpy() for t in g_grads]))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_wrapped_generator_and_discriminator_train_steps():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, _free_port_shared(), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, dg, gg = q.get(timeout=180)
            res[r] = (dg, gg)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    # single-process gradients for each rank's data, then their mean
    dev = torch.device("cuda", 0)
    G, D = _build(dev)
    singles = [_steps(G, D, G, D, r, dev) for r in range(world)]

    def mean(ts):
        if all(t is None for t in ts):
            return None
        return sum(torch.zeros_like(next(x for x in ts if x is not None)) if t is None else t for t in ts) / len(ts)

    for which, params in ((0, list(D.named_parameters())), (1, list(G.named_parameters()))):
        worst, used = 0.0, 0
        for i, (name, _) in enumerate(params):
            want = mean([singles[r][which][i] for r in range(world)])
            for r in range(world):
                got = res[r][which][i]
                if want is None:
                    assert got is None or float(abs(got).max()) == 0.0, (name, "unused parameter got a gradient")
                    continue
                assert got is not None, name
                got = torch.from_numpy(got)
                e = float((got.double() - want.double()).norm() / want.double().norm().clamp_min(1e-30))
                worst = max(worst, e)
                assert e < 1e-5, (name, r, e)
            used += want is not None
        print(f"{'D' if which == 0 else 'G'} under DDP: {used} of {len(params)} parameters carry a gradient; worst "
              f"difference from the mean of the single-process gradients {worst:.2e}")


_PORT = []


def _free_port_shared():
    if not _PORT:
        _PORT.append(_free_port())
    return _PORT[0]


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from cips3d_amd.distributed import GradAllReducer
    out = {}
    for mode in ("classic", "overlap"):
        G, _ = _build(dev)
        params = list(G.parameters())
        red = GradAllReducer(params, bucket_mb=2.0, overlap=(mode == "overlap"))
        steps = []
        for step in range(3):
            torch.manual_seed(500 + 10 * step + rank)
            for p in params:
                p.grad = None
            imgs, _ = G(G.get_zs(B), img_size=IMG, nerf_noise=0.0, return_aux_img=(step == 2), forward_points=None,
                        grad_points=None, **G_KW)               # step 2: the aux branch adds parameters on every rank
            imgs.square().mean().backward()
            red()
            torch.cuda.synchronize()
            red._check_pending(block=True)
            steps.append(([None if p.grad is None else p.grad.detach().cpu().numpy() for p in params],
                          red.last_launched_early, len(red._buckets)))
        out[mode] = steps
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gradient_buckets_on_gpu_equal_classic():
    """GradAllReducer(overlap=True) on GPU tensors (two ranks share cuda:0, gloo): buckets issued from autograd hooks on a
    side stream while the generator's backward is still running must leave the same averaged gradients as the classic
    after-backward reduce, on the planning step, in steady state and when the presence pattern changes"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, out = q.get(timeout=180)
            res[r] = out
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    import numpy as np
    for r in range(world):
        for step in range(3):
            gc, _, _ = res[r]["classic"][step]
            go, early, nb = res[r]["overlap"][step]
            for a, b in zip(gc, go):
                assert (a is None) == (b is None), (r, step)
                if a is not None:
                    assert np.array_equal(a, b), (r, step, float(abs(a - b).max()))
            if step == 1:
                assert nb >= 3 and early >= nb - 1, (early, nb)
        for a, b in zip(res[0]["overlap"][step][0], res[1]["overlap"][step][0]):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))


def test_overlap_bucket_waits_for_every_stream_its_gradients_were_produced_on(monkeypatch):
    """with the INR mapping MLP on a side stream the gradients of one bucket come from two streams, and autograd
    joins them only when backward() returns — a bucket issued from a hook must wait for ALL of them, not for the stream the
    last hook happened to run on.  One process, the collective replaced by a stub (x2 in place on the issuing stream = the
    SUM over two identical ranks), so the stream ordering is the only thing under test: parameter `a` is used on a side
    stream behind a ~50 ms spin kernel in its backward, parameter `b` on the main stream and its hook fires last."""
    from cips3d_amd import distributed as dmod
    d = torch.device("cuda:0")

    class _Work:
        def wait(self):
            return True

    class _Dist:
        ReduceOp = dist.ReduceOp
        calls = []

        @staticmethod
        def is_available(): return True
        @staticmethod
        def is_initialized(): return True
        @staticmethod
        def get_world_size(group=None): return 2

        @staticmethod
        def all_reduce(t, op=None, group=None, async_op=False):
            _Dist.calls.append((tuple(t.shape), str(op)))
            if op == dist.ReduceOp.SUM:
                t.mul_(2)
            return _Work()

    monkeypatch.setattr(dmod, "dist", _Dist)

    class _Slow(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            torch.cuda._sleep(100_000_000)        # ~50 ms on the stream of the forward (the side stream)
            return g * 3.0

    a = torch.nn.Parameter(torch.ones(1 << 16, device=d))
    b = torch.nn.Parameter(torch.ones(1 << 16, device=d))
    red = dmod.GradAllReducer([a, b], bucket_mb=64.0, overlap=True)
    side = torch.cuda.Stream()
    n_streams = []
    for step in range(3):                          # step 0 plans (classic path); steps 1-2 issue from hooks
        a.grad = b.grad = None
        main = torch.cuda.current_stream()
        yb = (b * 2.0).sum()                       # created first: its backward runs last
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ya = _Slow.apply(a * 5.0).sum()
        main.wait_stream(side)
        (ya + yb).backward()
        if step:
            n_streams.append(len(red._streams[0]))
        red()
        torch.cuda.synchronize()
        assert torch.equal(a.grad, torch.full_like(a, 15.0)), (step, float(a.grad[0]))
        assert torch.equal(b.grad, torch.full_like(b, 2.0)), (step, float(b.grad[0]))
        if step:
            assert red.last_launched_early == 1    # the bucket WAS issued from the hooks, during backward
    assert all(n >= 1 for n in n_streams)


def test_bench_two_ranks_gloo_functional_run_reports_the_exchange():
    """`bench.py --gpus 2` as the driver launches N > 1 runs (self-spawned torch.distributed.run ranks), here over gloo with
    both ranks on the box's one GPU (CIPS_BENCH_BACKEND=gloo: a functional check, never a measurement): the JSON line must
    carry the fields the first real 8-GPU run needs — ranks, the reduce form, and the event-timed all-reduce (ms, bytes
    per rank, bus bandwidth, fraction of the step) — so that run yields the SURVEY §8(e) numbers with no code change."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CIPS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--img-size", "16",
                          "--batch", "2", "--num-steps", "4", "--no-cpu-baseline", "--no-roofline", "--no-exact", "--no-full-step"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4
    assert line["config"]["rccl_ranks"] == 0 and "functional check" in line["backend_note"]          # gloo: not RCCL, and says so
    assert line["config"]["grad_reduce"].startswith("flat-bucket all-reduce")
    ar = line["allreduce"]
    # presence and type of the fields only: two ranks time-slice ONE GPU here, so no clock-derived value has a bound
    # (`frac_of_step < 1` failed at 1.03 on the driver's box and -x blanked the parity rows)
    assert ar["bytes_per_rank"] > 40e6
    for k in ("ms_median", "ms_min", "ms_max", "bus_GBps", "frac_of_step", "step_ms_median_events"):
        assert isinstance(ar[k], (int, float)), k
    assert isinstance(line["ms_per_step_median"], (int, float)) and isinstance(line["value"], (int, float))


def test_full_gan_step_two_ranks_gloo_reduces_both_gradient_sets():
    """scripts/bench_full_step.py --gpus 2 (functional, gloo, both ranks on the box's one GPU): the full GAN step of
    train.py:334-491 with BOTH exchanges — the discriminator's gradients after the D backward and the generator's after the G
    backward (train.py:235-236 wraps both networks) — through GradAllReducer and the fused optimizer tail.  Byte
    counts are the networks' at this stage, and after the steps the replicas (G, D, G_ema) are still identical on both
    ranks: same initial weights + the same averaged gradients."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CIPS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "bench_full_step.py"), "--gpus", "2", "--img-size", "16",
                          "--batch", "2", "--num-steps", "4", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 2 and line["replicas_identical"] is True
    assert 40e6 < line["allreduce_bytes_G"] < 50e6           # 130 of 172 generator parameters
    assert 10e6 < line["allreduce_bytes_D"] < 151e6          # the 16x16 stage of both discriminators
    assert "functional check" in line["backend"]
