"""hipGraph capture of a training / inference step built from this package's operators.

A CIPS-3D generator step at the headline configuration is ~570 kernel launches through ctypes; replaying them as one hipGraph
removes the host from the loop (C2: 16.15 ms eager -> 15.9 ms).  The reference has no counterpart (its `train.py` is eager
PyTorch); this is the MI355X-side replacement for "a tracing compiler": the step is an ordinary Python closure, captured once.

What a closure may do inside a capture (all of it is exercised by bench.py and tests/test_gpu_graph.py):
  * draw latents / noise with torch's CUDA generator (Philox offsets are graph-safe: every replay draws fresh numbers);
  * call `G(...)`, `D(...)`, `.backward()`, `torch.autograd.grad(..., create_graph=True)`, the fused optimizer
    (`cips3d_amd.optim`) — the generator's side stream for the INR mapping network forks from and joins the capturing stream;
  * read tensors that the caller refreshes between replays with `tensor.copy_()` (static buffers: real images, labels).
What it may not do: keep the only reference to a tensor of an earlier call alive across calls (e.g. `out['y'] = y`: the next call
frees the previous tensor in the middle of the capture — copy into a preallocated buffer instead), synchronise with the host (`.item()`, `.cpu()`, printing a tensor), allocate with a different size per
call, or issue collectives (keep the gradient all-reduce outside: `cips3d_amd.distributed.GradAllReducer` after `replay()`).
While a stream captures, the discriminator's weight-plane cache and the generator's ray-grid cache are bypassed; a caller that
replays a graph containing an optimizer step and then runs D eagerly calls `cips3d_amd.discriminator.invalidate_weight_cache(D)`.
"""
import torch

__all__ = ["CapturedStep", "capture"]


class CapturedStep:
    """`step = CapturedStep(fn)`; `step()` replays.  `fn` takes no arguments and returns nothing that is needed on the host;
    tensors it assigns to attributes of outer objects (e.g. `.grad`) live in the graph's memory pool and are overwritten by
    every replay.

    warmup: eager calls of `fn` on a side stream before the capture (allocator and per-shape caches settle; autograd's
            accumulate-grad nodes exist).  Gradients are set to None before the capture so that the captured backward WRITES
            `.grad` instead of accumulating into a tensor outside the pool.
    thread_local: capture error mode "thread_local" (needed when another thread of the process may call the HIP runtime during
            the capture, e.g. RCCL's watchdog after `init_process_group`); default: chosen from torch.distributed's state.
    On any capture error the exception propagates; nothing is left capturing (torch.cuda.graph ends the capture)."""

    def __init__(self, fn, warmup=2, thread_local=None, params=None):
        if not torch.cuda.is_available():
            raise RuntimeError("cips3d_amd.graph: hipGraph capture needs a GPU (there is no CPU path)")
        self.fn = fn
        if thread_local is None:
            thread_local = torch.distributed.is_available() and torch.distributed.is_initialized()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 0)):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if params is not None:
            for p in params:
                p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local" if thread_local else "global"):
            fn()

    def replay(self):
        self.graph.replay()

    __call__ = replay


def capture(fn, warmup=2, thread_local=None, params=None):
    """-> CapturedStep(fn, ...)"""
    return CapturedStep(fn, warmup=warmup, thread_local=thread_local, params=params)
