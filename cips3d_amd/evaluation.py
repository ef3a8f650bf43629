"""Evaluation / image-dump path around the generator: what
exp/cips3d/scripts/gen_images.py:30-69 (the FID path: fake images as JPEG files) and
exp/cips3d/scripts/train.py:87-170 (`save_images`: the sample grids written with every checkpoint) do with
`generator.forward`.  The float image -> uint8 quantisation of the FID path runs on the HIP library
(`cips_image_to_u8`, bit-exact torchvision arithmetic); JPEG encoding is PIL on the host, as torchvision does it.
FID itself (torch-fidelity's Inception network, eval_fid.py:36-50) is outside this repository: the submodule is not
vendored in the reference, so parity is stated on the pixels that reach it."""
import copy
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import check


def image_to_u8(imgs, value_range=(-1.0, 1.0)):
    """(B, C, H, W) fp32 on the GPU -> (B, H, W, C) uint8, quantised exactly like torchvision.utils.save_image(img,
    normalize=True, value_range=value_range) (gen_images.py:60)."""
    if not imgs.is_cuda:
        raise RuntimeError("image_to_u8 needs a tensor on the GPU (no CPU fallback)")
    x = imgs.detach().contiguous().float()
    B, Cc, H, W = x.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.uint8, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.cips_image_to_u8(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), B, Cc, H, W, float(value_range[0]),
                                   float(value_range[1]), st), "cips_image_to_u8")
    return out


def _save_u8(arr_hwc, path, quality=75):
    from PIL import Image
    a = arr_hwc.cpu().numpy()
    im = Image.fromarray(a[..., 0] if a.shape[-1] == 1 else a)
    im.save(path, **({"quality": quality} if path.lower().endswith((".jpg", ".jpeg")) else {}))


def make_grid(imgs, nrow=8, padding=2, normalize=False, value_range=None, scale_each=False, pad_value=0.0):
    """torchvision.utils.make_grid (the subset save_images uses): -> (C, H', W') float grid"""
    t = imgs.detach().float().clone()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if normalize:
        def norm_ip(img, low, high):
            img.clamp_(min=low, max=high)
            img.sub_(low).div_(max(high - low, 1e-5))

        def norm_range(img, vr):
            if vr is not None:
                norm_ip(img, vr[0], vr[1])
            else:
                norm_ip(img, float(img.min()), float(img.max()))
        if scale_each:
            for im in t:
                norm_range(im, value_range)
        else:
            norm_range(t, value_range)
    if t.size(0) == 1:
        return t.squeeze(0)
    nmaps = t.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    h, w = int(t.size(2) + padding), int(t.size(3) + padding)
    grid = t.new_full((t.size(1), h * ymaps + padding, w * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * h + padding, h - padding).narrow(2, x * w + padding, w - padding).copy_(t[k])
            k += 1
    return grid


def save_image(imgs, path, nrow=8, padding=2, normalize=False, value_range=None, scale_each=False):
    """torchvision.utils.save_image: grid -> uint8 (the HIP quantiser for GPU tensors) -> PIL."""
    grid = make_grid(imgs, nrow=nrow, padding=padding, normalize=normalize, value_range=value_range, scale_each=scale_each)
    if grid.is_cuda:
        u8 = image_to_u8(grid.unsqueeze(0), value_range=(0.0, 1.0))[0]
    else:
        u8 = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8)
    _save_u8(u8, path)


@torch.no_grad()
def gen_images(rank, world_size, generator, G_kwargs, fake_dir, num_imgs, img_size, batch_size, forward_points=256 ** 2,
               progress=False):
    """gen_images.py:30-69: `num_imgs` samples of `generator` (normally G_ema) at psi = 1 as
    <fake_dir>/<index:05d>.jpg, interleaved over ranks exactly like the reference (index = batch * batch_size +
    i * world_size + rank).  Returns the number of files this rank wrote."""
    if rank == 0:
        os.makedirs(fake_dir, exist_ok=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
    metadata = copy.deepcopy(dict(G_kwargs))
    batch_gpu = batch_size // world_size
    metadata["img_size"] = img_size
    metadata["psi"] = 1
    generator.eval()
    written = 0
    for idx_b in range((num_imgs + batch_size - 1) // batch_size):
        zs = generator.get_zs(batch_gpu)
        generated = generator(zs, forward_points=forward_points, **metadata)[0]
        u8 = image_to_u8(generated, value_range=(-1.0, 1.0))
        for idx_i in range(u8.shape[0]):
            _save_u8(u8[idx_i], f"{fake_dir}/{idx_b * batch_size + idx_i * world_size + rank:0>5}.jpg")
            written += 1
        if progress and rank == 0:
            print(f"gen_images: {min((idx_b + 1) * batch_size, num_imgs)}/{num_imgs}", flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
    return written


@torch.no_grad()
def save_images(saved_dir, G, G_ema, G_kwargs, fixed_z, img_size, forward_points=256 ** 2):
    """train.py:87-170: the six sample grids written next to every checkpoint (frontal view of G and G_ema, truncated
    G_ema, tilted views, the mirror-symmetry pair), same file names and view parameters."""
    os.makedirs(saved_dir, exist_ok=True)
    G.eval(); G_ema.eval()
    bs = len(list(fixed_z.values())[0])
    G_kwargs = copy.deepcopy(dict(G_kwargs))
    G_kwargs["img_size"] = img_size
    nrow = int(math.sqrt(bs))

    def run(net, meta, zs=fixed_z):
        return net(zs, return_aux_img=True, forward_points=forward_points, **meta)[0]

    meta = copy.deepcopy(G_kwargs); meta["h_stddev"] = 0; meta["v_stddev"] = 0
    save_image(run(G, meta), f"{saved_dir}/0Gz.jpg", nrow=nrow, normalize=True, scale_each=True)
    save_image(run(G_ema, meta), f"{saved_dir}/0Gz_ema.jpg", nrow=nrow, normalize=True, scale_each=True)
    meta["psi"] = 0.7
    save_image(run(G_ema, meta), f"{saved_dir}/0G_trunc_ema.jpg", nrow=nrow, normalize=True, scale_each=True)
    meta = copy.deepcopy(G_kwargs); meta["h_stddev"] = 0; meta["v_stddev"] = 0; meta["h_mean"] = math.pi * 0.5 + 0.5
    save_image(run(G, meta), f"{saved_dir}/0Gz_tilted.jpg", nrow=nrow, normalize=True, scale_each=True)
    save_image(run(G_ema, meta), f"{saved_dir}/0Gz_tilted_ema.jpg", nrow=nrow, normalize=True, scale_each=True)
    bs2 = min(20, bs)
    sub = {k: v[:bs2] for k, v in fixed_z.items()}
    meta = copy.deepcopy(G_kwargs); meta["h_stddev"] = 0; meta["v_stddev"] = 0; meta["h_mean"] = 1.44
    f1 = run(G_ema, meta, sub)
    meta["h_mean"] = 1.70
    f2 = run(G_ema, meta, sub)
    save_image(torch.cat([f1, f2]), f"{saved_dir}/0G_flip_ema.jpg", nrow=max(bs2 // 2, 1), normalize=True, scale_each=True)


def saved_models(model_dict, info_msg, G, G_ema, G_kwargs, fixed_z, img_size, saved_dir):
    """train.py:56-84 without tl2's rotating-directory bookkeeping: checkpoint files + info + sample grids."""
    from .checkpoint import save_models
    save_models(saved_dir, model_dict, info_msg=info_msg)
    save_images(saved_dir, G, G_ema, G_kwargs, fixed_z, img_size)
