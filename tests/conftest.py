import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

G_CFG = dict(
    z_dim=256,
    nerf_cfg=dict(in_dim=3, hidden_dim=128, hidden_layers=2, rgb_dim=32, style_dim=128),
    mapping_nerf_cfg=dict(z_dim=256, hidden_dim=128, base_layers=4, head_layers=0),
    inr_cfg=dict(input_dim=32, style_dim=512, hidden_dim=512, pre_rgb_dim=3),
    mapping_inr_cfg=dict(z_dim=512, hidden_dim=512, base_layers=8, head_layers=0, add_norm=True, norm_out=True),
    optim=dict(lr=0.0002, equal_lr=0.001),
)   # exp/cips3d/configs/ffhq_exp.yaml:43-81 (G_cfg_3D2D)

D_CFG = dict(diffaug=False, max_size=1024, channel_multiplier=2, first_downsample=False, stddev_group=0)
# exp/cips3d/configs/ffhq_exp.yaml:89-96 (D_cfg)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), map_location="cpu", weights_only=False)


def load_gates(tag):
    """LeakyReLU gates of the reference's run of golden case `tag` (oracle/make_golden.py gates) -> list of bool
    tensors in call order (head: (b, n, C) per modulated-FC layer; discriminator: the activation's shape)."""
    import numpy as np
    out = []
    for e in load_golden("gates_" + tag)["gates"]:
        n = 1
        for s_ in e["shape"]:
            n *= s_
        bits = np.unpackbits(e["bits"].numpy(), count=n, bitorder="little")
        out.append(torch.from_numpy(bits).bool().reshape(e["shape"]))
    return out


def pack_bitplane(g):
    """bool (..., C) with C % 8 == 0 -> uint8 (..., C/8), bit c&7 of byte c>>3: the kernels' gate bit-plane layout"""
    w = (1 << torch.arange(8, device=g.device)).to(torch.uint8)
    return (g.reshape(*g.shape[:-1], g.shape[-1] // 8, 8).to(torch.uint8) * w).sum(-1).to(torch.uint8)


def unpack_bitplane(p, C=None):
    """uint8 (..., C/8) -> bool (..., C)"""
    sh = torch.arange(8, device=p.device)
    g = ((p.unsqueeze(-1).to(torch.int32) >> sh) & 1).bool().reshape(*p.shape[:-1], p.shape[-1] * 8)
    return g if C is None else g[..., :C]


def seeded_generator(seed, freeze=False, device="cpu"):
    """Product module under the reference's seed: reproduces the reference's initial state_dict
    bit-for-bit (verified against the per-key checksums stored in the golden fixtures)."""
    from cips3d_amd.generator import GeneratorNerfINR, GeneratorNerfINR_freeze_NeRF
    torch.manual_seed(seed)
    cls = GeneratorNerfINR_freeze_NeRF if freeze else GeneratorNerfINR
    G = cls(**G_CFG, device="cpu")
    if device != "cpu":
        G = G.to(device)
        G.device = device
    return G


def check_checksums(sd, sums):
    assert list(sd.keys()) == list(sums.keys())
    for k, v in sd.items():
        s, a = sums[k]
        assert abs(float(v.double().sum()) - s) <= 1e-9 * max(1.0, abs(a)), k
        assert abs(float(v.double().abs().sum()) - a) <= 1e-9 * max(1.0, abs(a)), k


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    """max |a-b| / max|b| — the 'relative fp32' measure used by the parity bars."""
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


class ReplayDraws:
    """Make torch.rand / torch.randint return recorded tensors (in order, moved to the requested device): lets the
    product's DiffAugment consume exactly the draws the reference made when a golden fixture was minted."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        import torch
        self._rand, self._randint = torch.rand, torch.randint
        it = iter(self.draws)

        def take(kind, device):
            k, t = next(it)
            assert k == kind, (k, kind)
            return t.to(device) if device is not None else t

        torch.rand = lambda *a, **k: take("rand", k.get("device"))
        torch.randint = lambda *a, **k: take("randint", k.get("device"))
        self._it = it
        return self

    def __exit__(self, *exc):
        import torch
        torch.rand, torch.randint = self._rand, self._randint
        if exc[0] is None:
            assert next(self._it, None) is None, "recorded draws left over"


# Collection order of the GPU set (the driver runs `pytest -x -m gpu`): kernel parity first, then the operator / module /
# golden-fixture tests, then whole-step tests, and everything that spawns processes or runs bench.py LAST — a late flake
# in a functional test must never blank the parity rows again.
_ORDER = ["test_gpu_kernels", "test_gpu_generator", "test_gpu_discriminator", "test_gpu_optim", "test_gpu_eval_path",
          "test_gpu_real_configs", "test_gpu_train_step", "test_gpu_ddp"]


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        if mod == "test_gpu_rccl":               # brings up RCCL communicators in child processes: the very last
            return len(_ORDER) + 1
        return _ORDER.index(mod) if mod in _ORDER else (len(_ORDER) if mod.startswith("test_gpu") else -1)
    items.sort(key=key)          # stable: order inside a file is kept
