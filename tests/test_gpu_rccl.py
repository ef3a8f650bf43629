"""RCCL on the one GPU a test box has: a process group of ONE rank over backend "nccl" (= RCCL on ROCm).
No scaling claim — what these tests execute is the code an N > 1 run needs and no earlier round ever ran on RCCL:
communicator bring-up with `device_id` (train.py:41-49), the thread-local hipGraph capture of the step beside RCCL's watchdog
thread, GradAllReducer's presence exchange on its control stream, the bucket all-reduce + copy-back + plan checksum on RCCL's
streams, and the late-gradient path of the overlapped form on GPU streams.  Collected last (tests/conftest.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")


def _last_json(out):
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_under_torchrun_one_rank_runs_the_exchange_over_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`, the driver's N > 1 launch line with N = 1: the
    step is captured (thread-local) with the communicator up, replayed, and every step's gradients go through the one-rank
    all-reduce.  The line must say so: rccl_ranks 1, an `allreduce` block with the generator's ~45 MB, hipGraph replay."""
    import bench
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(bench._free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "3",
           "--no-cpu-baseline", "--no-roofline", "--no-exact", "--no-full-step", "--no-other-configs"]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1
    assert line["config"]["launch"] == "hipGraph replay", out.stderr[-2000:]
    assert line["config"]["grad_reduce"].startswith("flat-bucket all-reduce")
    ar = line["allreduce"]
    assert 40e6 < ar["bytes_per_rank"] < 50e6                  # 130 of the generator's 172 parameters receive a gradient
    for k in ("ms_median", "ms_min", "ms_max", "frac_of_step"):
        assert isinstance(ar[k], (int, float)), k              # presence only: no assertion on a clock-derived value
    print(f"one-rank RCCL: {line['ms_per_step']} ms / step, exchange median {ar['ms_median']} ms ({ar['frac_of_step']} of the step)")


def test_full_gan_step_one_rank_rccl_reduces_both_gradient_sets():
    """scripts/bench_full_step.py --rccl: both exchanges of the GAN step (D's after the D backward, G's after the G backward,
    train.py:235-236) through a one-rank RCCL group, eager launches, fused optimizer tail."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_full_step.py"), "--rccl", "--img-size", "16", "--batch", "2",
                          "--num-steps", "4", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out)
    assert line["ranks"] == 1 and line["replicas_identical"] is True and line["backend"] == "nccl"
    assert 40e6 < line["allreduce_bytes_G"] < 50e6 and 10e6 < line["allreduce_bytes_D"] < 151e6


_LATE = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cips3d_amd.distributed import GradAllReducer
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
class Chain(torch.nn.Module):
    def __init__(s):
        super().__init__(); s.a = torch.nn.Linear(64, 256); s.b = torch.nn.Linear(256, 256); s.c = torch.nn.Linear(256, 8)
    def forward(s, x, detach_a):
        h = torch.tanh(s.a(x))
        if detach_a: h = h.detach()
        return s.c(torch.tanh(s.b(h)))
res = {}
for mode in ("classic", "overlap"):
    torch.manual_seed(0)
    net = Chain().to(dev); params = list(net.parameters())
    red = GradAllReducer(params, bucket_mb=0.05, overlap=(mode == "overlap"), single_rank_exchange=True)
    steps = []
    for step in range(4):
        for p in params: p.grad = None
        x = torch.randn(512, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(500 + step))
        net(x, detach_a=step in (0, 2)).square().mean().backward()        # steps 1, 3: a.* arrive after their bucket went out
        red(); red._check_pending(block=True)
        steps.append([None if p.grad is None else p.grad.clone() for p in params])
    res[mode] = steps
    if mode == "overlap":
        # a second backward before the reduction must be refused, not mis-reduced (ADVICE r5)
        for p in params: p.grad = None
        x = torch.randn(512, 64, device=dev)
        net(x, False).square().mean().backward()
        try:
            net(x, False).square().mean().backward(); print("NO_RAISE")
        except RuntimeError as e:
            print("RAISED" if "twice" in str(e) else "OTHER " + str(e))
ok = True
for step in range(4):
    for k, (a, b) in enumerate(zip(res["classic"][step], res["overlap"][step])):
        # one rank: the mean is the gradient itself; the overlapped form must deliver it bit for bit, late gradients included.
        # A parameter nobody produced a gradient for (a.* at steps 0, 2) is None (classic) or zero (overlap: part of the plan).
        if a is None: ok = ok and (b is None or not bool(b.any()))
        else: ok = ok and b is not None and torch.equal(a, b)
print("LATE_OK" if ok else "LATE_MISMATCH")
dist.destroy_process_group()
'''


def test_overlapped_reducer_late_gradient_and_second_backward_on_rccl_streams(tmp_path):
    """The overlapped reducer on GPU streams over RCCL (one rank): a gradient that arrives after its bucket was issued is
    delivered (private zeros on the wire, copy-back after the late accumulate: ADVICE r5 — only gloo on CPU ran this before),
    and a second backward before the reduction raises."""
    f = tmp_path / "late.py"
    f.write_text(_LATE)
    env = _env(); env["MASTER_PORT"] = str(__import__("bench")._free_port())
    out = subprocess.run([sys.executable, str(f), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "LATE_OK" in out.stdout and "RAISED" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
