"""diagnostic: NeRF-side gradient error of the hierarchical path vs the fp32 oracle at several sizes (gates pinned)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import seeded_generator, pack_bitplane, max_rel
from oracle import cips3d_oracle as orc
from cips3d_amd import ops
KW = dict(fov=12, ray_start=0.88, ray_end=1.12, h_stddev=0.3, v_stddev=0.155)
d = torch.device("cuda:0")

def run(b, img, S, hier, noise, aux, seed=5):
    g = torch.Generator().manual_seed(seed)
    n = img * img; E = 2 * S if hier else S
    zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
    rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g), phi=torch.randn(b, 1, generator=g),
                noise_c=torch.randn(b, n, S, 1, generator=g), u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, E, 1, generator=g))
    nimg = 2 * b if aux else b
    G0 = torch.randn(nimg, 3, img, img, generator=g) / (nimg * 3 * n)
    Gc = seeded_generator(1234)
    tape = orc.GateTape()
    with orc.gate_tape(tape):
        ref = orc.generator_forward(dict(Gc.named_parameters()), zs, rand, img, KW["fov"], KW["ray_start"], KW["ray_end"], S, KW["h_stddev"],
                                    KW["v_stddev"], hier, nerf_noise=noise, return_aux_img=aux, keep=True)
    (ref["imgs"] * G0).sum().backward()
    Gd = seeded_generator(1234, device=d)
    fz = ref["fine_z"].detach().reshape(b * n, S) if hier else None
    with ops.gate_debug(pin=[pack_bitplane(t) for t in tape.rec]), ops.resample_debug(pin=[fz] if hier else None):
        imgs, _ = Gd({k: v.to(d) for k, v in zs.items()}, img_size=img, num_steps=S, hierarchical_sample=hier, sample_dist="gaussian",
                     nerf_noise=noise, return_aux_img=aux, grad_points=None, forward_points=None,
                     rand_override={k: v.to(d) for k, v in rand.items()}, **KW)
    (imgs * G0.to(d)).sum().backward()
    torch.cuda.synchronize()
    errs = {}
    for (k, p), (_, q) in zip(Gd.named_parameters(), Gc.named_parameters()):
        if q.grad is not None and k.startswith(("siren", "aux")):
            errs[k] = float((p.grad.cpu().double() - q.grad.double()).norm() / q.grad.double().norm().clamp_min(1e-300))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f"b={b} img={img} S={S} hier={hier} noise={noise} aux={aux}: imgs {max_rel(imgs, ref['imgs']):.2e}; worst siren grads: " +
          ", ".join(f"{k.replace('siren.', '')} {v:.1e}" for k, v in top), flush=True)

CFGS = [(2, 128, 12, True, 0.1, True), (2, 128, 24, False, 0.1, True), (2, 128, 12, True, 0.1, False), (2, 128, 12, True, 0.0, True)]
print("SIREN forward mode:", ops.SIREN_FWD_MODE, " INR mode:", ops.INR_MODE)
for cfg in CFGS:
    run(*cfg)
