"""Secondary measurement: one full GAN training step as exp/cips3d/scripts/train.py:334-491 drives it
(bench.full_gan_step; bench.py reports the C2 case in its own JSON line under "full_step").  Prints ms per phase and
images/s for any geometry.  Not the headline metric."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img-size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--num-steps", type=int, default=12)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--freeze", action="store_true", help="GeneratorNerfINR_freeze_NeRF (the r256 stages, ffhq_exp.yaml:192-210)")
    ap.add_argument("--diffaug", action="store_true", help="DiffAugment in D (r256 stages)")
    ap.add_argument("--no-aux", action="store_true", help="train_aux_img False (C4)")
    ap.add_argument("--torch-optim", action="store_true", help="torch clip_grad_norm_ + Adam + python EMA instead of the fused tail")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: one rank per GPU over RCCL (CIPS_BENCH_BACKEND=gloo: functional "
                    "check with the ranks sharing the visible devices), both gradient sets all-reduced every step")
    ap.add_argument("--rccl", action="store_true", help="N = 1: run both gradient exchanges through a one-rank RCCL process group")
    a = ap.parse_args()
    import bench
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(bench._free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("CIPS_BENCH_BACKEND", "nccl")
    pg = world > 1 or a.rccl
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(bench._free_port())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit(f"--gpus {world} needs {world} GPUs (CIPS_BENCH_BACKEND=gloo: functional check on fewer)")
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        torch.distributed.init_process_group(backend, rank=rank, world_size=world,
                                             **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    res = bench.full_gan_step(dev, a.batch, a.img_size, a.num_steps, steps=a.steps, warmup=a.warmup,
                              freeze=a.freeze, diffaug=a.diffaug, aux=not a.no_aux, torch_optim=a.torch_optim)
    if pg:
        res["backend"] = backend if backend == "nccl" else f"{backend} (functional check, not a measurement)"
    if rank == 0:
        print(json.dumps(res))
    if pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
