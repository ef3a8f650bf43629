"""GPU: the evaluation / FID path around the generator: the float -> uint8 quantisation of
gen_images.py:60 on the HIP library (bit-exact against torchvision's arithmetic on identical inputs; end-to-end
against the oracle's image), `gen_images` (file naming, rank interleave, JPEG round trip) and `save_images`
(train.py:87-170: six grids), and the checkpoint directory on the GPU modules."""
import os

import numpy as np
import pytest
import torch

from conftest import seeded_generator
from oracle import cips3d_oracle as orc

pytestmark = pytest.mark.gpu
KW = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=6, h_stddev=0.3, v_stddev=0.155, hierarchical_sample=True,
          sample_dist="gaussian", clamp_mode="relu", nerf_noise=0., white_back=False, last_back=False)


def _tv_u8(x, lo, hi):
    """torchvision.utils.save_image(normalize=True, value_range=(lo, hi)) quantisation, torch-CPU restatement
    (torchvision/utils.py: norm_ip = clamp_, sub_, div_(max(hi - lo, 1e-5)); then mul(255).add_(0.5).clamp_(0, 255))"""
    t = x.clone().clamp_(min=lo, max=hi)
    t.sub_(lo).mul_(1.0 / max(hi - lo, 1e-5))       # ATen CUDA divides by a scalar as a * (1 / b)
    return t.mul(255).add_(0.5).clamp_(0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def test_image_to_u8_bit_exact():
    from cips3d_amd.evaluation import image_to_u8
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.tanh(torch.randn(3, 3, 37, 41, generator=g) * 1.5)
    # values on and next to every rounding boundary of the quantiser, and out-of-range values
    k = torch.arange(0, 256, dtype=torch.float32)
    edge = (k + 0.5) / 255 * 2 - 1
    x.view(-1)[:256] = edge
    x.view(-1)[256:512] = torch.nextafter(edge, torch.full_like(edge, 2.0))
    x.view(-1)[512:768] = torch.nextafter(edge, torch.full_like(edge, -2.0))
    x.view(-1)[768:772] = torch.tensor([-1.5, 1.5, -1.0, 1.0])
    for (lo, hi) in [(-1.0, 1.0), (0.0, 1.0), (-0.7, 0.9)]:
        got = image_to_u8(x.to(d), value_range=(lo, hi)).cpu()
        assert got.shape == (3, 37, 41, 3) and got.dtype == torch.uint8
        assert torch.equal(got, _tv_u8(x, lo, hi)), (lo, hi)


def test_fid_path_pixels_match_oracle_end_to_end():
    """G_ema.eval() -> forward(psi=1, forward_points) -> uint8 pixels, HIP path vs CPU oracle: the floats agree to
    ~1e-5, so a pixel can differ only where it sits within that distance of a quantisation boundary: by one level, in
    a small fraction of the pixels (reported)."""
    from cips3d_amd.evaluation import image_to_u8
    d = torch.device("cuda:0")
    b, img, S = 4, 32, 6
    G = seeded_generator(1234)
    g = torch.Generator().manual_seed(9)
    n = img * img
    zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
    rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g), phi=torch.randn(b, 1, generator=g),
                noise_c=torch.randn(b, n, S, 1, generator=g), u=torch.rand(b * n, S, generator=g),
                noise_f=torch.randn(b, n, 2 * S, 1, generator=g))
    with torch.no_grad():
        ref = orc.generator_forward(dict(G.named_parameters()), zs, rand, img, 12, 0.88, 1.12, S, 0.3, 0.155, True)["imgs"]
        Gd = G.to(d).eval()
        imgs, _ = Gd({k: v.to(d) for k, v in zs.items()}, img_size=img, psi=1, forward_points=256 ** 2,
                     rand_override={k: v.to(d) for k, v in rand.items()}, **KW)
    got = image_to_u8(imgs).cpu().int()
    want = _tv_u8(ref, -1.0, 1.0).int()
    diff = (got - want).abs()
    frac = float((diff > 0).float().mean())
    print(f"FID-path pixels: {frac:.2e} of the uint8 values differ from the oracle's, max difference {int(diff.max())} level")
    assert int(diff.max()) <= 1 and frac < 2e-3


def test_gen_images_and_save_images_and_checkpoint(tmp_path):
    import copy
    from PIL import Image
    from cips3d_amd.evaluation import gen_images, save_images, saved_models, image_to_u8
    from cips3d_amd.checkpoint import load_models
    d = torch.device("cuda:0")
    G = seeded_generator(1234, device=d)
    G_ema = copy.deepcopy(G); G_ema.device = d
    G_kwargs = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=4, h_stddev=0.3, v_stddev=0.155, h_mean=1.5707963,
                    v_mean=1.5707963, hierarchical_sample=True, sample_dist="gaussian", psi=1.)
    fake = str(tmp_path / "fid" / "fake")
    # two "ranks" of a world of 2 write interleaved indices, 7 images requested in batches of 4 -> indices 0..7
    torch.manual_seed(0)
    n0 = gen_images(0, 2, G_ema, G_kwargs, fake, num_imgs=7, img_size=16, batch_size=4, forward_points=100)
    n1 = gen_images(1, 2, G_ema, G_kwargs, fake, num_imgs=7, img_size=16, batch_size=4, forward_points=100)
    files = sorted(os.listdir(fake))
    assert n0 == n1 == 4 and files == [f"{i:05d}.jpg" for i in range(8)]
    # the JPEG holds the quantised pixels of a fresh forward with the same seed (JPEG is lossy: compare loosely)
    torch.manual_seed(0)
    zs = G_ema.get_zs(2)
    meta = dict(G_kwargs, img_size=16, psi=1)
    with torch.no_grad():
        im = G_ema(zs, forward_points=100, **meta)[0]
    u8 = image_to_u8(im).cpu().numpy()
    jpg = np.asarray(Image.open(os.path.join(fake, "00000.jpg")).convert("RGB"))
    assert jpg.shape == (16, 16, 3) and np.abs(jpg.astype(int) - u8[0].astype(int)).mean() < 12
    # save_images: the six grids of train.py:87-170
    out = str(tmp_path / "ckpt")
    fixed_z = G.get_zs(4)
    state = {"step": 10, "best_fid": 1e9}
    saved_models({"generator": G, "G_ema": G_ema, "discriminator": torch.nn.Linear(2, 2), "state_dict": state}, "step: 10", G, G_ema,
                 {k: v for k, v in G_kwargs.items() if k != "psi"} | {"psi": 1.}, fixed_z, 16, out)
    want = {"0Gz.jpg", "0Gz_ema.jpg", "0G_trunc_ema.jpg", "0Gz_tilted.jpg", "0Gz_tilted_ema.jpg", "0G_flip_ema.jpg",
            "generator.pth", "G_ema.pth", "discriminator.pth", "state_dict.pth", "0info.txt"}
    assert set(os.listdir(out)) == want
    grid = Image.open(os.path.join(out, "0Gz.jpg"))
    assert grid.size == (2 * 18 + 2, 4 * 18 + 2)          # 8 images (inr + aux) in rows of 2, padding 2
    G2 = seeded_generator(7, device=d)
    load_models(out, {"generator": G2}, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(G2.state_dict().values(), G.state_dict().values()))
