"""GPU: the on-disk formats and the script-level plumbing of exp/cips3d/scripts/train.py driven
end to end on the MI355X with the product's modules and the product-side stand-ins of cips3d_amd/compat/shims — in the
order train.py:214-317, 334-491, 70, 264 uses them: tl2 command line + YAML with `base:` inheritance -> global_cfg ->
seeds -> build_model (generator, discriminator with kwargs priority, dataset) -> endless data loader over a StyleGAN-style
zip -> to_norm_tensor -> D step (R1) and G step -> checkpoint directory -> resume.  (/root/reference does not exist on the
GPU box, so the unmodified train.py itself is imported and configured in tests/test_compat_scripts_cpu.py instead.)"""
import copy
import io
import json
import os
import sys
import zipfile

import numpy as np
import pytest
import torch
import yaml
from PIL import Image

from conftest import ROOT, G_CFG, D_CFG

pytestmark = pytest.mark.gpu
SHIMS = os.path.join(ROOT, "cips3d_amd", "compat", "shims")
F = torch.nn.functional


@pytest.fixture()
def shims(monkeypatch):
    before = set(sys.modules)
    for name in [m for m in sys.modules if m.split(".")[0] in ("tl2", "torchvision", "easydict", "streamlit", "torch_fidelity")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.delitem(sys.modules, "cips3d_amd.compat.registry", raising=False)     # re-registers in THIS tl2's registry on import
    monkeypatch.syspath_prepend(SHIMS)
    yield SHIMS
    for name in set(sys.modules) - before:
        if name.split(".")[0] in ("tl2", "torchvision", "easydict", "streamlit", "torch_fidelity"):
            sys.modules.pop(name, None)
    sys.modules.pop("cips3d_amd.compat.registry", None)


def _archive(path, n, res, seed=0):
    rng = np.random.RandomState(seed)
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        for i in range(n):
            buf = io.BytesIO()
            Image.fromarray(rng.randint(0, 256, size=(res, res, 3), dtype=np.uint8)).save(buf, format="png")
            z.writestr(f"{i // 1000:05d}/img{i:08d}.png", buf.getvalue())
        z.writestr("dataset.json", json.dumps({"labels": None}))


def test_train_script_plumbing_on_the_gpu(shims, tmp_path, monkeypatch):
    from tl2.launch.launch_utils import update_parser_defaults_from_yaml, global_cfg
    from tl2.proj.fvcore import build_model
    from tl2.proj.pytorch import torch_utils
    from tl2.proj.pytorch.examples.dataset_stylegan3.dataset import get_training_dataloader, to_norm_tensor
    from tl2.proj.fvcore.checkpoint import Checkpointer
    from torchvision.utils import save_image
    from cips3d_amd.optim import FusedClipAdamEMA
    dev = torch.device("cuda:0")
    arch = str(tmp_path / "faces_32x32.zip")
    _archive(arch, 12, 32)
    reg = "cips3d_amd.compat.registry"
    cfg = {
        "models": {
            "G_cfg": dict(G_CFG, register_modules=[reg], name=f"{reg}.GeneratorNerfINR"),
            "D_cfg": dict(D_CFG, register_modules=[reg], name=f"{reg}.Discriminator_MultiScale_Aux"),
            "data_cfg": {"register_modules": ["tl2.proj.pytorch.examples.dataset_stylegan3.dataset"],
                         "name": "ImageFolderDataset_of_stylegan", "path": arch, "use_labels": False, "max_size": None, "xflip": True,
                         "resize_resolution": None, "random_seed": 0},
            "G_kwargs": dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=4, h_stddev=0.3, v_stddev=0.155,
                             hierarchical_sample=True, sample_dist="gaussian"),
        },
        "train_ffhq": {"base": "models", "seed": 1234, "gen_lr": 0.0002, "disc_lr": 0.002, "betas": [0.0, 0.999], "batch_size": 4,
                       "img_size": 32, "diffaug": False, "r1_lambda": 10.0, "grad_clip": 10, "train_aux_img": True, "num_workers": 0,
                       "total_iters": 2},
        "train_small": {"base": "train_ffhq", "img_size": 16, "G_kwargs": {"num_steps": 6}},
    }
    yml = tmp_path / "exp.yaml"
    yml.write_text(yaml.safe_dump(cfg))
    out = str(tmp_path / "results" / "train_small")
    monkeypatch.setattr(sys, "argv", ["train.py", "--tl_config_file", str(yml), "--tl_command", "train_small", "--tl_outdir", out,
                                      "--tl_opts", "batch_size", "2"])
    update_parser_defaults_from_yaml(parser=None, is_main_process=True)                     # train.py:214
    assert global_cfg.img_size == 16 and global_cfg.batch_size == 2 and global_cfg.G_kwargs.num_steps == 6 and global_cfg.G_kwargs.fov == 12
    torch_utils.init_seeds(seed=global_cfg.seed, rank=0)                                    # :220
    generator = build_model(cfg=global_cfg.G_cfg).to(dev)                                   # :228-229
    generator.device = dev
    discriminator = build_model(cfg=global_cfg.D_cfg, kwargs_priority=True, diffaug=global_cfg.diffaug).to(dev)
    from cips3d_amd.generator import GeneratorNerfINR
    from cips3d_amd.discriminator import Discriminator_MultiScale_Aux
    assert type(generator) is GeneratorNerfINR and type(discriminator) is Discriminator_MultiScale_Aux
    G_ema = copy.deepcopy(generator)                                                        # :230
    dataset = build_model(global_cfg.data_cfg, kwargs_priority=True, resize_resolution=global_cfg.img_size)       # :300
    assert len(dataset) == 24 and dataset.image_shape == [3, 16, 16]
    loader = iter(get_training_dataloader(dataset=dataset, rank=0, num_gpus=1, batch_size=global_cfg.batch_size,
                                          num_workers=global_cfg.num_workers, shuffle=True, sampler_seed=0))
    fG = FusedClipAdamEMA(generator.parameters(), lr=global_cfg.gen_lr, betas=tuple(global_cfg.betas), max_norm=global_cfg.grad_clip,
                          ema_params=G_ema.parameters(), ema_decay=0.999, ema_start_itr=0)
    fD = FusedClipAdamEMA(discriminator.parameters(), lr=global_cfg.disc_lr, betas=tuple(global_cfg.betas), max_norm=global_cfg.grad_clip)
    state_dict = {"cur_fid": float("inf"), "best_fid": float("inf"), "worst_fid": 0, "step": 0}
    w0 = generator.siren.network[1].linear.weight.detach().clone()
    gk = {k: v for k, v in global_cfg.G_kwargs.items()}
    losses = []
    for step in range(global_cfg.total_iters):                                              # :309-491
        imgs, _, _ = next(loader)
        real = to_norm_tensor(imgs, device=dev)
        assert real.shape == (2, 3, 16, 16) and real.dtype == torch.float32
        torch_utils.requires_grad(generator, False); torch_utils.requires_grad(discriminator, True)
        with torch.no_grad():
            zs = generator.get_zs(real.shape[0])
            fake, _ = generator(zs, img_size=global_cfg.img_size, nerf_noise=1.0, return_aux_img=global_cfg.train_aux_img,
                                grad_points=None, forward_points=None, **gk)
        x = (torch.cat([real, real]) if global_cfg.train_aux_img else real).requires_grad_(True)
        rp = discriminator(x, use_aux_disc=global_cfg.train_aux_img, alpha=1.0)[0]
        gr, = torch.autograd.grad(rp.sum(), x, create_graph=True)
        fp = discriminator(fake, use_aux_disc=global_cfg.train_aux_img, alpha=1.0)[0]
        d_loss = (F.softplus(fp) + F.softplus(-rp) + 0.5 * global_cfg.r1_lambda * gr.flatten(1).square().sum(1, keepdim=True)).mean()
        fD.zero_grad(); d_loss.backward(); fD.step(itr=step)
        torch_utils.requires_grad(generator, True); torch_utils.requires_grad(discriminator, False)
        zs = generator.get_zs(real.shape[0])
        gimgs, _ = generator(zs, img_size=global_cfg.img_size, nerf_noise=1.0, return_aux_img=global_cfg.train_aux_img,
                             grad_points=global_cfg.img_size ** 2, forward_points=None, **gk)
        g_loss = F.softplus(-discriminator(gimgs, use_aux_disc=global_cfg.train_aux_img, alpha=1.0)[0]).mean()
        fG.zero_grad(); g_loss.backward(); fG.step(itr=step)
        state_dict["step"] += 1
        losses.append((float(d_loss), float(g_loss)))
    torch.cuda.synchronize()
    assert all(np.isfinite(l).all() for l in losses), losses
    assert not torch.equal(generator.siren.network[1].linear.weight, w0)
    # ---- checkpoint directory (train.py:70: save_models; :264: load_models on resume; gen_images.py:102: Checkpointer)
    model_dict = {"generator": generator, "G_ema": G_ema, "discriminator": discriminator, "state_dict": state_dict}
    resume = os.path.join(global_cfg.tl_ckptdir, "resume")
    torch_utils.save_models(save_dir=resume, model_dict=model_dict)
    global_cfg.dump_to_file_with_command(f"{resume}/config_command.yaml", global_cfg.tl_command)
    assert sorted(f for f in os.listdir(resume) if f.endswith(".pth")) == ["G_ema.pth", "discriminator.pth", "generator.pth", "state_dict.pth"]
    save_image(gimgs.detach(), f"{resume}/0Gz.jpg", nrow=2, normalize=True, scale_each=True)
    assert Image.open(f"{resume}/0Gz.jpg").size[0] > 16
    torch_utils.init_seeds(seed=99, rank=0)
    G2 = build_model(cfg=global_cfg.G_cfg).to(dev); G2.device = dev
    D2 = build_model(cfg=global_cfg.D_cfg, kwargs_priority=True, diffaug=global_cfg.diffaug).to(dev)
    E2 = copy.deepcopy(G2)
    st2 = {"cur_fid": 0, "best_fid": 0, "worst_fid": 0, "step": 0}
    torch_utils.load_models(save_dir=resume, model_dict={"generator": G2, "G_ema": E2, "discriminator": D2, "state_dict": st2},
                            strict=False, rank=0)
    assert st2["step"] == 2
    for a, b in ((generator, G2), (G_ema, E2), (discriminator, D2)):
        for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert k == k2 and torch.equal(v, v2), k
    E3 = build_model(cfg=global_cfg.G_cfg).to(dev)
    Checkpointer(E3).load_state_dict_from_file(f"{resume}/G_ema.pth", rank=0)
    assert all(torch.equal(v, v2) for v, v2 in zip(G_ema.state_dict().values(), E3.state_dict().values()))
    # the resolved configuration written next to the checkpoint selects the same run again
    back = yaml.safe_load(open(f"{resume}/config_command.yaml"))["train_small"]
    assert back["img_size"] == 16 and back["G_cfg"]["name"].endswith("GeneratorNerfINR") and back["G_kwargs"]["num_steps"] == 6
    print(f"2 steps from the zip through the shims: d_loss / g_loss {losses}")
