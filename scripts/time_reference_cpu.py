"""Build container only (/root/reference exists here, not on the GPU box): wall time of the UNMODIFIED reference generator
(imported through oracle/ref_shim.py) for G forward + backward at the bench geometry (r64, S = 24 flat, b = 4), next to the
oracle's time on the same cores — SURVEY §8(d) asks for the reference's CPU path beside the GPU number; bench.py's
cpu_baseline times the oracle (kind "port") because the reference does not travel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import ref_shim
ref_shim.install()
import yaml
from tl2.proj.fvcore import build_model
from exp.cips3d.models import generator as ref_gen       # noqa: F401
from oracle import cips3d_oracle as orc
from conftest import seeded_generator

cfg = yaml.safe_load(open("/root/reference/exp/cips3d/configs/ffhq_exp.yaml"))
torch.manual_seed(1234)
G = build_model(cfg["G_cfg_3D2D"], device="cpu")
b, img, S = 4, 64, 24
kw = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=S, h_stddev=0.3, v_stddev=0.155, hierarchical_sample=False,
          psi=1., sample_dist="gaussian")

def ref_once():
    zs = G.get_zs(b)
    G.zero_grad()
    t0 = time.time()
    imgs, _ = G(zs, img_size=img, nerf_noise=0., return_aux_img=False, grad_points=None, forward_points=None, **kw)
    imgs.backward(torch.ones_like(imgs) / imgs.numel())
    return time.time() - t0

Go = seeded_generator(1234)
sd = dict(Go.named_parameters())
g = torch.Generator().manual_seed(1)
n = img * img

def orc_once():
    zs = {"z_nerf": torch.randn(b, 256, generator=g), "z_inr": torch.randn(b, 512, generator=g)}
    rand = dict(jitter=torch.rand(b, n, S, 1, generator=g), theta=torch.randn(b, 1, generator=g), phi=torch.randn(b, 1, generator=g),
                noise_c=torch.randn(b, n, S, 1, generator=g), u=torch.rand(b * n, S, generator=g), noise_f=torch.randn(b, n, S, 1, generator=g))
    Go.zero_grad()
    t0 = time.time()
    out = orc.generator_forward(sd, zs, rand, img, 12, 0.88, 1.12, S, 0.3, 0.155, False)
    out["imgs"].backward(torch.ones_like(out["imgs"]) / out["imgs"].numel())
    return time.time() - t0

res = {}
for name, f in (("reference (unmodified, shimmed imports)", ref_once), ("oracle (port)", orc_once)):
    f()
    ts = sorted(f() for _ in range(3))
    res["reference" if f is ref_once else "oracle"] = {"value": round(b / ts[1], 4), "min": round(b / ts[2], 4), "max": round(b / ts[0], 4)}
    print(f"{name}: G fwd+bwd r{img} S={S} b={b} on {torch.get_num_threads()} threads: median {ts[1]:.2f} s = {b / ts[1]:.3f} img/s "
          f"(min {ts[0]:.2f} s, max {ts[2]:.2f} s)")

# tracked record bench.py copies into cpu_baseline.reference (the GPU box has no /root/reference: this is the only place the
# reference's own CPU path can be timed; the cores are the build container's, stated in the record)
import json
rec = {"unit": "img/s", "cores": torch.get_num_threads(), "where": "build container (not the GPU box's host)",
       "sample": f"unmodified reference G fwd+bwd (exp/cips3d/models/generator.py via oracle/ref_shim.py), r{img}, S={S} flat, b={b}, "
                 f"median of 3 timed iterations after one warm-up",
       "value": res["reference"]["value"], "min": res["reference"]["min"], "max": res["reference"]["max"],
       "oracle_same_cores": res["oracle"], "script": "scripts/time_reference_cpu.py"}
json.dump(rec, open(os.path.join(ROOT, "profiles", "reference_cpu.json"), "w"), indent=1)
print(json.dumps(rec))
