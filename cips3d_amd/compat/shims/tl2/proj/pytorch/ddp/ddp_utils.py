"""tl2.proj.pytorch.ddp.ddp_utils (train.py:526, 537, 575; gen_images.py:92; the detectron2-style helpers)"""
import os

import torch
import torch.distributed as dist


def d2_synchronize():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def d2_get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return d2_get_rank() == 0


def ddp_init(seed=0, backend=None):
    """gen_images.py:92: (rank, world_size) of a torchrun-style launch; a single process when no launcher variables are set"""
    rank = int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", 0)))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "12355")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    from tl2.proj.pytorch.torch_utils import init_seeds
    init_seeds(seed, rank)
    return rank, world
