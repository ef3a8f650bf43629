"""tl2.proj.pytorch.init_func.kaiming_leaky_init (generator.py:22; identical in-tree copies:
exp/cips3d/models/multi_head_mapping.py:22-25, piGAN_lib/siren/siren.py:43-46)"""
import torch


def kaiming_leaky_init(m):
    if m.__class__.__name__.find("Linear") != -1:
        torch.nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
