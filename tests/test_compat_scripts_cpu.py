"""CPU: the product-side stand-ins for the reference's un-vendored dependencies (cips3d_amd/compat/shims: tl2, torchvision,
easydict, streamlit, torch_fidelity) — SURVEY.md section 8 f-4, "on-disk formats": the tl2 command line + YAML `base:`
inheritance, the StyleGAN-style dataset zip, the checkpoint-directory layout.  Where /root/reference exists (the build
container) the UNMODIFIED reference files run on top of them: `scripts/dataset_tool.py` writes the archive the reader is
tested on, the reference's own YAML is resolved, and `exp/cips3d/scripts/train.py` imports with every name it needs.
(`train()` itself needs a GPU and NCCL: the product has no CPU path, and /root/reference does not travel to the GPU box —
tests/test_gpu_formats.py drives the same shim entry points on the GPU with the product's own modules.)"""
import json
import os
import subprocess
import sys
import zipfile

import numpy as np
import pytest
import torch
import yaml
from PIL import Image

from conftest import ROOT

SHIMS = os.path.join(ROOT, "cips3d_amd", "compat", "shims")
REF = "/root/reference"
has_ref = os.path.isdir(os.path.join(REF, "exp", "cips3d"))


@pytest.fixture()
def shims(monkeypatch):
    """the shims at the END of sys.path, their modules dropped again afterwards (other tests install oracle/ref_shim.py's)"""
    before = set(sys.modules)
    for name in [m for m in sys.modules if m.split(".")[0] in ("tl2", "torchvision", "easydict", "streamlit", "torch_fidelity")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(SHIMS)
    yield SHIMS
    for name in set(sys.modules) - before:
        if name.split(".")[0] in ("tl2", "torchvision", "easydict", "streamlit", "torch_fidelity", "exp"):
            sys.modules.pop(name, None)


def _write_images(folder, n, res, seed=0):
    rng = np.random.RandomState(seed)
    os.makedirs(folder, exist_ok=True)
    imgs = []
    for i in range(n):
        a = rng.randint(0, 256, size=(res, res, 3), dtype=np.uint8)
        a[:, : res // 2, 0] = i                     # left half tagged: a mirrored copy is recognisable
        Image.fromarray(a).save(os.path.join(folder, f"face{i:03d}.png"))
        imgs.append(a)
    return imgs


def _write_archive(path, imgs, labels=None):
    """an archive in dataset_tool.py's layout (scripts/dataset_tool.py:527-540) without the reference: used when it is absent"""
    import io
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        recs = []
        for i, a in enumerate(imgs):
            name = f"{i // 1000:05d}/img{i:08d}.png"
            buf = io.BytesIO()
            Image.fromarray(a).save(buf, format="png", compress_level=0, optimize=False)
            z.writestr(name, buf.getvalue())
            recs.append([name, labels[i]] if labels is not None else None)
        z.writestr("dataset.json", json.dumps({"labels": recs if labels is not None else None}))


def test_tl2_command_line_yaml_inheritance_and_opts(shims, tmp_path, monkeypatch):
    from tl2.launch import launch_utils as lu
    cfg = {
        "common": {"G_kwargs": {"fov": 12, "num_steps": 12}, "betas": [0, 0.999]},
        "train_a": {"base": "common", "seed": 1234, "batch_size": 4, "img_size": 32, "diffaug": False,
                    "D_cfg": {"name": "D", "diffaug": False, "max_size": 1024}, "G_cfg": {"name": "G"}},
        "train_b": {"base": "train_a", "img_size": 256, "diffaug": True, "D_cfg": {"diffaug": True},
                    "G_cfg": {"name": "G_freeze"}},
    }
    y = tmp_path / "exp.yaml"
    y.write_text(yaml.safe_dump(cfg))
    out = tmp_path / "results" / "run"
    monkeypatch.setattr(sys, "argv", ["train.py", "--port", "1", "--tl_config_file", str(y), "--tl_command", "train_b", "--tl_outdir",
                                      str(out), "--tl_opts", "batch_size", "8", "G_kwargs.num_steps", "24", "new.nested.key", "[1, 2]"])
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument("--port", type=str, default="12355")
    parser.add_argument("--seed", type=int, default=0)
    g = lu.update_parser_defaults_from_yaml(parser)
    assert g is lu.global_cfg
    assert g.img_size == 256 and g.diffaug is True and g.seed == 1234                       # own keys over two levels of `base:`
    assert g.D_cfg.diffaug is True and g.D_cfg.max_size == 1024 and g.D_cfg.name == "D"     # nested dicts merge key by key
    assert g.G_cfg.name == "G_freeze" and g.betas == [0, 0.999]
    assert g.batch_size == 8 and g.G_kwargs.num_steps == 24 and g.G_kwargs.fov == 12        # --tl_opts, dotted keys
    assert g.new.nested.key == [1, 2]
    assert g.tl_command == "train_b" and g.tl_outdir == str(out) and g.tl_ckptdir == os.path.join(str(out), "ckptdir")
    assert g.get("missing", 7) == 7 and g.tl_resume is False and g.tl_debug is False
    opt, _ = parser.parse_known_args()
    assert opt.port == "1" and opt.seed == 1234                                              # same-named parser default from the YAML
    assert os.path.isdir(g.tl_ckptdir)
    # train.py:68
    g.dump_to_file_with_command(str(out / "ckpt" / "config_command.yaml"), g.tl_command)
    back = yaml.safe_load(open(out / "ckpt" / "config_command.yaml"))
    assert list(back) == ["train_b"] and back["train_b"]["D_cfg"]["diffaug"] is True and "tl_outdir" not in back["train_b"]
    with pytest.raises(KeyError):
        lu.resolve_command(cfg, "nope")
    with pytest.raises(ValueError):
        lu.resolve_command({"a": {"base": "b"}, "b": {"base": "a"}}, "a")


def test_stylegan_dataset_archive_reader(shims, tmp_path):
    """the zip / folder format, xflip, max_size, resize_resolution, labels, the rank-strided infinite loader, normalisation"""
    from tl2.proj.fvcore import build_model
    from tl2.proj.pytorch.examples.dataset_stylegan3.dataset import get_training_dataloader, to_norm_tensor
    src = tmp_path / "raw"
    imgs = _write_images(str(src), 10, 16)
    arch = str(tmp_path / "faces_16x16.zip")
    if has_ref:
        # the reference's own, unmodified converter writes the archive (scripts/dataset_tool.py:398)
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([REF, ROOT, SHIMS]))
        r = subprocess.run([sys.executable, os.path.join(REF, "scripts", "dataset_tool.py"), "--source", str(src), "--dest", arch],
                           env=env, capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
    else:
        _write_archive(arch, imgs)
    with zipfile.ZipFile(arch) as z:
        names = z.namelist()
    assert "dataset.json" in names and "00000/img00000000.png" in names and "00000/img00000009.png" in names
    data_cfg = {"register_modules": ["tl2.proj.pytorch.examples.dataset_stylegan3.dataset"], "name": "ImageFolderDataset_of_stylegan",
                "path": arch, "use_labels": False, "max_size": None, "xflip": True, "resize_resolution": None, "random_seed": 0}
    ds = build_model(data_cfg, kwargs_priority=True, resize_resolution=16)                # train.py:300
    assert len(ds) == 20 and ds.image_shape == [3, 16, 16] and ds.resolution == 16 and not ds.has_labels
    for i in (0, 3, 9):
        x, label, idx = ds[i]
        assert x.dtype == np.uint8 and idx == i and label.shape == (0,)
        assert np.array_equal(x, imgs[i].transpose(2, 0, 1))
        xf, _, _ = ds[10 + i]
        assert np.array_equal(xf, imgs[i].transpose(2, 0, 1)[:, :, ::-1])                    # the mirrored second half
    # the same archive unpacked as a folder
    folder = tmp_path / "unz"
    with zipfile.ZipFile(arch) as z:
        z.extractall(str(folder))
    ds_dir = build_model(dict(data_cfg, path=str(folder), xflip=False), kwargs_priority=True)
    assert len(ds_dir) == 10 and np.array_equal(ds_dir[4][0], ds[4][0])
    # max_size: a seeded subset, sorted; resize_resolution: every image resampled
    sub = build_model(dict(data_cfg, xflip=False, max_size=4), kwargs_priority=True)
    sub2 = build_model(dict(data_cfg, xflip=False, max_size=4), kwargs_priority=True)
    assert len(sub) == 4 and all(np.array_equal(sub[i][0], sub2[i][0]) for i in range(4))
    small = build_model(dict(data_cfg, xflip=False), kwargs_priority=True, resize_resolution=8)
    assert small.image_shape == [3, 8, 8] and small[0][0].shape == (3, 8, 8)
    # labels: integer classes -> one-hot
    larch = str(tmp_path / "labelled.zip")
    _write_archive(larch, imgs[:6], labels=[0, 2, 1, 2, 0, 1])
    lab = build_model(dict(data_cfg, path=larch, xflip=False, use_labels=True), kwargs_priority=True)
    assert lab.has_labels and lab.label_dim == 3 and lab[1][1].tolist() == [0.0, 0.0, 1.0]
    # the loader: global batch 4 over 2 ranks -> 2 images per rank and step; the ranks' index streams are the two strided
    # halves of ONE shuffled stream (same seed on every rank), so no image is drawn twice in a step
    from tl2.proj.pytorch.examples.dataset_stylegan3.dataset import InfiniteSampler
    import itertools
    whole = list(itertools.islice(iter(InfiniteSampler(ds, rank=0, num_replicas=1, shuffle=True, seed=0)), 40))
    assert sorted(whole[:20]) != whole[:20] and set(whole) <= set(range(20))
    for rank in range(2):
        it = iter(get_training_dataloader(dataset=ds, rank=rank, num_gpus=2, batch_size=4, num_workers=0, shuffle=True, sampler_seed=0))
        x, _, idx = next(it)
        assert x.shape == (2, 3, 16, 16) and x.dtype == torch.uint8
        got = idx.tolist()
        for _ in range(9):
            got += next(it)[2].tolist()
        assert got == whole[rank::2][:20], rank
        assert np.array_equal(x[0].numpy(), ds[got[0]][0])
    plain = list(itertools.islice(iter(InfiniteSampler(ds, rank=1, num_replicas=2, shuffle=False)), 12))
    assert plain == [1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 1, 3]
    t = to_norm_tensor(x, device="cpu")
    assert t.dtype == torch.float32 and float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    assert torch.equal(t, x.float() / 127.5 - 1.0)
    with pytest.raises(IOError):
        build_model(dict(data_cfg, path=str(tmp_path / "missing.txt")), kwargs_priority=True)


def test_checkpoint_directory_through_torch_utils_and_registry(shims, tmp_path):
    """torch_utils.save_models / load_models (train.py:70, 264) write / read the tl2 checkpoint directory; MODEL_REGISTRY names"""
    from tl2.proj.pytorch import torch_utils
    from tl2.proj.fvcore import MODEL_REGISTRY, build_model
    from tl2 import tl2_utils

    @MODEL_REGISTRY.register(name_prefix="pkg.mod")
    class Net(torch.nn.Linear):
        def __init__(self, i, o, **kw):
            super().__init__(i, o)
    net = build_model({"name": "pkg.mod.Net", "i": 3, "o": 2})
    assert isinstance(net, Net)
    with pytest.raises(KeyError):
        MODEL_REGISTRY.get("pkg.mod.Other")
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    state = {"step": 7, "best_fid": 3.5}
    d = tl2_utils.MaxToKeep.get_named_max_to_keep(name="ckpt", use_circle_number=True).step_and_ret_circle_dir(str(tmp_path / "ckptdir"))
    torch_utils.save_models(save_dir=d, model_dict={"generator": net, "optimizer_G": opt, "state_dict": state})
    tl2_utils.write_info_msg(d, "step 7")
    assert sorted(os.listdir(d)) == ["0info.txt", "generator.pth", "optimizer_G.pth", "state_dict.pth"]
    net2 = Net(3, 2)
    state2 = {"step": 0, "best_fid": float("inf")}
    torch_utils.load_models(save_dir=d, model_dict={"generator": net2, "state_dict": state2}, strict=False, rank=0)
    assert torch.equal(net2.weight, net.weight) and state2 == state
    torch_utils.init_seeds(seed=5, rank=1)
    a = torch.rand(3)
    torch_utils.init_seeds(seed=6, rank=0)
    assert torch.equal(a, torch.rand(3))
    torch_utils.requires_grad(net, False)
    assert not any(p.requires_grad for p in net.parameters())
    assert torch_utils.get_optimizer_lr(opt) == 1e-3


def test_torchvision_utils_stand_in_writes_the_reference_grids(shims, tmp_path):
    from torchvision.utils import make_grid, save_image
    x = torch.rand(5, 3, 8, 8) * 2 - 1
    g = make_grid(x, nrow=3, normalize=True, scale_each=True)
    assert g.shape == (3, 2 * 8 + 3 * 2, 3 * 8 + 4 * 2)
    save_image(x, str(tmp_path / "g.jpg"), nrow=3, normalize=True, scale_each=True)
    assert Image.open(tmp_path / "g.jpg").size == (3 * 8 + 4 * 2, 2 * 8 + 3 * 2)


@pytest.mark.skipif(not has_ref, reason="/root/reference exists only in the build container")
def test_unmodified_reference_scripts_import_and_configure_on_the_shims(tmp_path):
    """exp/cips3d/scripts/{train, gen_images, setup_evaluation, eval_fid}.py — unmodified — import on the stand-ins with every
    name they use at module level, and the reference's own YAML resolves through the tl2 command line: `train_ffhq_high`
    (base: train_ffhq; ffhq_exp.yaml:192-210) with the overrides of exp/cips3d/bash/ffhq_exp/train_ffhq_r32.sh."""
    code = r'''
import sys, argparse
sys.argv = ["train.py", "--port", "12355", "--tl_config_file", "exp/cips3d/configs/ffhq_exp.yaml", "--tl_command", "train_ffhq_high",
            "--tl_outdir", sys.argv[1], "--tl_opts", "batch_size", "4", "img_size", "32", "total_iters", "80000", "G_kwargs.num_steps", "24",
            "betas", "[0.0, 0.999]"]      # torch >= 2 refuses the YAML's mixed int / float `betas: [0, 0.999]` (the reference pins 1.8.2)
import exp.cips3d.scripts.train as T
import exp.cips3d.scripts.gen_images, exp.cips3d.scripts.setup_evaluation, exp.cips3d.scripts.eval_fid
import tl2, torchvision
assert getattr(tl2, "__cips3d_shim__", False) and torchvision.__version__.endswith("shim")
from tl2.launch.launch_utils import global_cfg, update_parser_defaults_from_yaml
parser = argparse.ArgumentParser(); parser.add_argument("--port", type=str, default="0")
update_parser_defaults_from_yaml(parser)
assert T.global_cfg is global_cfg
g = global_cfg
assert g.G_cfg.name == "exp.cips3d.models.generator.GeneratorNerfINR_freeze_NeRF" and g.D_cfg.diffaug is True
assert g.D_cfg.name.endswith("Discriminator_MultiScale_Aux") and g.D_cfg.max_size == 1024      # inherited through `base:`
assert g.img_size == 32 and g.batch_size == 4 and g.total_iters == 80000 and g.G_kwargs.num_steps == 24 and g.G_kwargs.fov == 12
assert g.betas == [0.0, 0.999] and g.gen_lr == 0.0001 and g.train_aux_img is False and g.load_nerf_ema is True
assert g.data_cfg.name == "ImageFolderDataset_of_stylegan" and g.tl_command == "train_ffhq_high"
# the optimisers exactly as train.py:175-190 builds them, on the configuration just resolved
import torch
net = torch.nn.Linear(2, 2)
oG, oD = T.build_optimizer(net, net)
assert oG.param_groups[0]["lr"] == g.gen_lr and oD.param_groups[0]["lr"] == g.disc_lr and tuple(oG.param_groups[0]["betas"]) == (0, 0.999)
for name in ("train", "saved_models", "save_images", "setup_ddp", "build_optimizer"):
    assert callable(getattr(T, name))
print("OK")
'''
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REF, ROOT, SHIMS]))
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "out")], env=env, capture_output=True, text=True, cwd=REF)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-1500:], r.stderr[-3000:])
