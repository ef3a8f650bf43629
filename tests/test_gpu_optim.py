"""GPU: the fused clip + Adam + EMA step (cips3d_amd/optim.py, csrc/optim.hip) against the reference's own recipe —
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(betas, weight_decay=0) + EMA.update
(exp/cips3d/scripts/train.py:420-491, exp/comm/comm_model_utils.py:97-118) — over several steps, including
parameters without a gradient, tensors spanning several chunks, clipping active and inactive."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("betas,max_norm", [((0.0, 0.999), 10.0), ((0.0, 0.999), 0.05), ((0.9, 0.99), None)])
def test_fused_clip_adam_ema_matches_torch(betas, max_norm):
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    shapes = [(7, 5), (300000,), (3,), (512, 512), (1,), (65536,), (4,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, generator=g).to(d)) for s in shapes]
    p_ref = mk()
    p_fus = [torch.nn.Parameter(p.detach().clone()) for p in p_ref]
    e_ref = [p.detach().clone() for p in p_ref]
    e_fus = [p.detach().clone() for p in p_ref]
    opt = torch.optim.Adam([{"params": p_ref}], lr=2e-3, betas=betas, weight_decay=0)
    fus = FusedClipAdamEMA(p_fus, lr=2e-3, betas=betas, max_norm=max_norm, ema_params=e_fus, ema_decay=0.999)
    for it in range(4):
        grads = [torch.randn(*s, generator=g).to(d) * (0.3 + it) for s in shapes]
        for k, (a, b) in enumerate(zip(p_ref, p_fus)):
            if k == 6 or (k == 2 and it % 2):          # a parameter that never / sometimes gets a gradient
                a.grad = b.grad = None
            else:
                a.grad, b.grad = grads[k].clone(), grads[k].clone()
        if max_norm:
            n_ref = torch.nn.utils.clip_grad_norm_(p_ref, max_norm)
        else:
            n_ref = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in p_ref if p.grad is not None]))
        opt.step()
        with torch.no_grad():
            for e, p in zip(e_ref, p_ref):
                e.copy_(e * 0.999 + p * (1 - 0.999))
        n_fus = fus.step()
        torch.cuda.synchronize()
        assert abs(float(n_fus) - float(n_ref)) <= 1e-5 * float(n_ref)
        for k, (a, b, ea, eb) in enumerate(zip(p_ref, p_fus, e_ref, e_fus)):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, k, float((a - b).abs().max()))
            assert torch.allclose(ea, eb, rtol=2e-6, atol=1e-7), (it, k)
            if max_norm and a.grad is not None:       # the clipped gradient is visible in .grad like clip_grad_norm_'s
                assert torch.allclose(a.grad, b.grad, rtol=2e-6, atol=1e-9), (it, k)
    sd = fus.state_dict()
    ref_sd = opt.state_dict()
    for i in range(len(shapes)):
        if i in ref_sd["state"]:
            assert torch.allclose(sd["state"][i]["exp_avg_sq"], ref_sd["state"][i]["exp_avg_sq"], rtol=2e-5, atol=1e-12)   # the clip coefficient (norm in double here, fp32 norm-of-norms in torch) enters squared


def test_fused_optimizer_on_generator_parameters():
    """all 172-key generator parameters through one fused step: finite, changed, EMA moved"""
    from conftest import G_CFG
    from cips3d_amd.generator import GeneratorNerfINR
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    G = GeneratorNerfINR(**G_CFG, device=d).to(d); G.device = d
    G_ema = copy.deepcopy(G)
    opt = FusedClipAdamEMA(G.parameters(), lr=2e-4, betas=(0.0, 0.999), max_norm=10.0, ema_params=G_ema.parameters())
    kw = dict(fov=12, ray_start=0.88, ray_end=1.12, num_steps=4, h_stddev=0.3, v_stddev=0.155,
              hierarchical_sample=True, psi=1., sample_dist="gaussian")
    imgs, _ = G(G.get_zs(2), img_size=8, nerf_noise=0.1, return_aux_img=True, grad_points=None, forward_points=None, **kw)
    imgs.square().mean().backward()
    before = [p.detach().clone() for p in G.parameters()]
    norm = opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(norm).all() and float(norm) > 0
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(before, G.parameters()))
    assert changed > 100
    assert all(torch.isfinite(p).all() for p in G.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(G_ema.parameters(), before))


def test_fused_step_queued_without_host_sync():
    """Six steps enqueued back to back with NO synchronisation between them and freshly allocated gradient tensors
    every step (what zero_grad(set_to_none=True) + backward does): the parameter table is re-uploaded every step
    while earlier steps are still queued.  A long kernel in front keeps the stream behind the host, so a staging
    buffer that were reused before its copy ran would hand an earlier step the next step's gradient pointers."""
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 33), (200000,), (5,), (512, 512)]
    p_ref = [torch.nn.Parameter(torch.randn(*s, generator=g).to(d)) for s in shapes]
    p_fus = [torch.nn.Parameter(p.detach().clone()) for p in p_ref]
    opt = torch.optim.Adam([{"params": p_ref}], lr=1e-3, betas=(0.5, 0.99), weight_decay=0)
    fus = FusedClipAdamEMA(p_fus, lr=1e-3, betas=(0.5, 0.99), max_norm=None)
    all_grads = [[torch.randn(*s, generator=g).to(d) * (1 + it) for s in shapes] for it in range(6)]
    for it in range(6):
        for a, gr in zip(p_ref, all_grads[it]):
            a.grad = gr.clone()
        opt.step()
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device=d)
    keep = []
    for _ in range(6):
        big = big @ big * 1e-4                     # ~1 ms each: the stream falls behind the host
    for it in range(6):
        for b, gr in zip(p_fus, all_grads[it]):
            b.grad = gr.clone() + 0                # a fresh allocation per step
            keep.append(b.grad)
        fus.step()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(p_ref, p_fus)):
        assert torch.allclose(a, b, rtol=5e-6, atol=1e-7), (k, float((a - b).abs().max()))
    assert fus.steps == [6, 6, 6, 6]


def test_fused_step_replayed_as_hipgraph_advances_bias_correction():
    """The step captured once in a hipGraph and replayed: the Adam step counts live on the device and advance on every
    replay (with host-side counts the bias corrections would stay at the captured step)."""
    from cips3d_amd.optim import FusedClipAdamEMA
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(10)
    shapes = [(300, 7), (70000,)]
    p_ref = [torch.nn.Parameter(torch.randn(*s, generator=g).to(d)) for s in shapes]
    p_fus = [torch.nn.Parameter(p.detach().clone()) for p in p_ref]
    for p in p_fus:
        p.grad = torch.zeros_like(p)              # static gradient buffers, as a captured training step has them
    opt = torch.optim.Adam([{"params": p_ref}], lr=1e-2, betas=(0.9, 0.999), weight_decay=0)
    fus = FusedClipAdamEMA(p_fus, lr=1e-2, betas=(0.9, 0.999), max_norm=None)
    grads = [[torch.randn(*s, generator=g).to(d) for s in shapes] for _ in range(4)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fus.step()                                # warm-up outside the graph (step 1 with zero gradients)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in p_ref:
        p.grad = torch.zeros_like(p)
    opt.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fus.step()
    for p in p_ref:                               # the capture itself does not execute
        pass
    for it in range(4):
        for a, b, gr in zip(p_ref, p_fus, grads[it]):
            a.grad = gr.clone()
            b.grad.copy_(gr)
        opt.step()
        graph.replay()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(p_ref, p_fus)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (k, float((a - b).abs().max()))
    # an EAGER step with freshly allocated gradients in between (it uploads another tensor table through the staging
    # ring), then a replay: the captured upload reads a pinned table of its own, so the replay still steps through the
    # static gradient buffers it was captured with (round-2 advisor: a ring slot would have been overwritten)
    static = [b.grad for b in p_fus]
    g2 = [torch.randn(*s_, generator=g).to(d) for s_ in shapes]
    for a, b, gr in zip(p_ref, p_fus, g2):
        a.grad = gr.clone()
        b.grad = gr.clone() + 0
    opt.step(); fus.step()
    g3 = [torch.randn(*s_, generator=g).to(d) for s_ in shapes]
    for a, b, st, gr in zip(p_ref, p_fus, static, g3):
        a.grad = gr.clone()
        st.copy_(gr)
        b.grad = st
    opt.step(); graph.replay()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(p_ref, p_fus)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (k, float((a - b).abs().max()))
    assert fus.steps == [7, 7]
