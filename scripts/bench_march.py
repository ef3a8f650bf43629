"""Timing of the ray-march front end at C2 (r64, S=24 flat, b=32): fused cips_march_fwd_x3 vs the five-kernel path,
no_grad and training forward + backward of the NeRF part only (upstream gradient on pixels_fea); and composite_fwd alone
on the hierarchical layout (S=12+12)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import G_CFG
from cips3d_amd import ops
from cips3d_amd.generator import GeneratorNerfINR


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
    b, img, S = 32, 64, 24
    n = img * img
    style = {k: torch.randn(b, 128, device=d) for k in G.siren.style_dim_dict}
    xg = torch.linspace(-1, 1, img, device=d); yg = torch.linspace(1, -1, img, device=d); zg = torch.linspace(0.88, 1.12, S, device=d)
    zc = -1.0 / float(torch.tan(torch.tensor(3.14159265 * 12 / 360)))
    c2w = torch.eye(4, device=d).repeat(b, 1, 1); c2w[:, 2, 3] = 1.0
    jit = torch.rand(b, n, S, device=d)
    up = torch.randn(b, n, 32, device=d)
    geom = (b, img, img, S, zc, 0.0, 0, 0, True)
    res = {}

    def fused_ng():
        with torch.no_grad():
            G.siren.march(style, geom[:-1] + (False,), xg, yg, zg, c2w, jit, None)

    def unfused_ng():
        with torch.no_grad():
            pts, z, dirs = ops.rays_fwd(xg, yg, zg, zc, c2w, jit, b, img, img, S)
            f, s = G.siren.evaluate(pts.view(b, n * S, 3), style)
            ops.CompositeFunction.apply(f.view(b * n, S, 32), s.view(b * n, S), z.view(b * n, S), None, None, None, None, 0.0, 0, 0)

    def fused_tr():
        for p in G.siren.parameters():
            p.grad = None
        fea, _ = G.siren.march(style, geom, xg, yg, zg, c2w, jit, None)
        fea.backward(up)

    def unfused_tr():
        for p in G.siren.parameters():
            p.grad = None
        with torch.no_grad():
            pts, z, dirs = ops.rays_fwd(xg, yg, zg, zc, c2w, jit, b, img, img, S)
        f, s = G.siren.evaluate(pts.view(b, n * S, 3), style)
        fea = ops.CompositeFunction.apply(f.view(b * n, S, 32), s.view(b * n, S), z.view(b * n, S), None, None, None, None, 0.0, 0, 0)[0]
        fea.backward(up.view(b * n, 32))

    res["fused_nograd_us"] = timeit(fused_ng)
    res["unfused_nograd_us"] = timeit(unfused_ng)
    res["fused_train_us"] = timeit(fused_tr)
    res["unfused_train_us"] = timeit(unfused_tr)
    rays = b * n
    res["fused_nograd_bytes_per_ray_algorithmic"] = 4 * S + 132
    # composite alone, hierarchical layout
    S2 = 12
    fc, ff = torch.randn(rays, S2, 32, device=d), torch.randn(rays, S2, 32, device=d)
    sc, sf = torch.randn(rays, S2, device=d), torch.randn(rays, S2, device=d)
    zc_, zf = torch.rand(rays, S2, device=d), torch.rand(rays, S2, device=d)
    t = timeit(lambda: ops.CompositeFunction.apply(fc, sc, zc_, ff, sf, zf, None, 0.0, 0, 0))
    byt = rays * (140 * 24 + 132)
    res["composite_fwd_hier_us"] = t
    res["composite_fwd_hier_TBps"] = byt / t / 1e6
    fl = torch.randn(rays, 24, 32, device=d); sl = torch.randn(rays, 24, device=d); zl = torch.rand(rays, 24, device=d).sort(-1)[0]
    t = timeit(lambda: ops.CompositeFunction.apply(fl, sl, zl, None, None, None, None, 0.0, 0, 0))
    res["composite_fwd_flat_us"] = t
    res["composite_fwd_flat_TBps"] = byt / t / 1e6
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
