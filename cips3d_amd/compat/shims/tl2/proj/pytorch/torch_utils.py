"""tl2.proj.pytorch.torch_utils — what train.py and the model files use (train.py:70, 192-196, 220, 264-277, 335-336,
441-442, 494-495; generator.py:21)."""
import random

import numpy as np
import torch

from cips3d_amd import checkpoint as _ckpt


def init_seeds(seed=0, rank=0, cuda_deterministic=False):
    """train.py:220: per-rank seeding of python / numpy / torch (`seed + rank`, the convention bench.py mirrors)"""
    s = int(seed) + int(rank)
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(s)


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad_(flag)


def print_number_params(models_dict=None, logger=None, **kwargs):
    if not models_dict:
        return
    for name, m in models_dict.items():
        n = sum(p.numel() for p in m.parameters())
        msg = f"{name} params: {n / 1e6:.2f} M"
        (logger.info if logger is not None else print)(msg)


def get_optimizer_lr(optimizer, return_all=False):
    lrs = [g["lr"] for g in optimizer.param_groups]
    return lrs if return_all else lrs[0]


def save_models(save_dir, model_dict, info_msg=None, cfg=None, msg_mode='w'):
    """train.py:70: one `<name>.pth` per entry of model_dict (modules / optimizers -> state_dict, plain dicts as they are):
    the tl2 checkpoint-directory layout gen_images.py:102 and the resume path read back"""
    return _ckpt.save_models(save_dir, model_dict, info_msg=info_msg)


def load_models(save_dir, model_dict, strict=True, rank=0, verbose=True, **kwargs):
    """train.py:264, 276"""
    return _ckpt.load_models(save_dir, model_dict, strict=strict, rank=rank, verbose=verbose)


def ema_accumulate(model1, model2, decay=0.999):
    par1, par2 = dict(model1.named_parameters()), dict(model2.named_parameters())
    for k in par1:
        par1[k].data.mul_(decay).add_(par2[k].data, alpha=1 - decay)
