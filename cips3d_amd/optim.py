"""Training-step tail on the HIP path: gradient-norm clip + Adam + EMA over all tensors of one optimiser in two
launches (csrc/optim.hip).  Mirrors what exp/cips3d/scripts/train.py:420-491 does with
torch.nn.utils.clip_grad_norm_, torch.optim.Adam(betas=global_cfg.betas, weight_decay=0) and
comm_model_utils.EMA.update (exp/comm/comm_model_utils.py:97-118) — same update rule, same state
(`exp_avg`, `exp_avg_sq`, step count), exportable as a torch.optim.Adam state_dict."""
import ctypes as C

import torch

from . import _lib
from ._lib import check


class _OptTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("ema", C.c_void_p), ("n", C.c_longlong), ("step", C.c_longlong)]


class FusedClipAdamEMA:
    """opt = FusedClipAdamEMA(G.parameters(), lr, betas, max_norm=grad_clip, ema_params=G_ema.parameters())
    ... loss.backward(); total_norm = opt.step()       # clip -> Adam -> EMA, 2 kernel launches

    Parameters whose .grad is None are left untouched by Adam (like torch.optim.Adam) but still averaged into the
    EMA copy.  `ema_start_itr` reproduces EMA(start_itr=...): before it the EMA copy is not updated."""

    def __init__(self, params, lr, betas=(0.0, 0.999), eps=1e-8, max_norm=None, ema_params=None, ema_decay=0.999,
                 ema_start_itr=0, capture_slots=4):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedClipAdamEMA needs parameters on the GPU (no CPU fallback)")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError("fp32 contiguous parameters on one device expected")
        self.ema = [p for p in ema_params] if ema_params is not None else None
        if self.ema is not None:
            if len(self.ema) != len(self.params) or any(e.shape != p.shape for e, p in zip(self.ema, self.params)):
                raise ValueError("ema_params must mirror params")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.max_norm = float(max_norm) if max_norm else 0.0
        self.ema_decay, self.ema_start_itr = float(ema_decay), int(ema_start_itr)
        # per-parameter step counts (torch.optim.Adam's state) live on the DEVICE and are advanced by the kernel: no
        # host-written value goes stale between steps or inside a captured hipGraph of the step
        self._steps_dev = torch.zeros(len(self.params), dtype=torch.int64, device=dev)
        self.device = dev
        total = sum(p.numel() for p in self.params)
        self._m = torch.zeros(total, device=dev)
        self._v = torch.zeros(total, device=dev)
        self.exp_avg, self.exp_avg_sq, o = [], [], 0
        for p in self.params:
            n = p.numel()
            self.exp_avg.append(self._m[o:o + n].view_as(p)); self.exp_avg_sq.append(self._v[o:o + n].view_as(p))
            o += n
        lib = _lib.load()
        chunk = lib.cips_opt_chunk()
        ct, co = [], []
        for i, p in enumerate(self.params):
            for off in range(0, max(p.numel(), 1), chunk):
                ct.append(i); co.append(off)
        self.nchunks = len(ct)
        self._chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self._chunk_off = torch.tensor(co, dtype=torch.int64, device=dev)
        self._partial = torch.empty(self.nchunks, dtype=torch.float64, device=dev)
        self._norm = torch.zeros(1, device=dev)
        self._table_host = (_OptTensor * len(self.params))()
        self._table_dev = torch.empty(C.sizeof(self._table_host), dtype=torch.uint8, device=dev)
        # pinned staging ring for the table upload: a slot is rewritten only after the asynchronous copy that read it
        # has completed (its event), so a host that runs ahead of the stream cannot overwrite a table in flight
        self._ring = [[torch.empty(C.sizeof(self._table_host), dtype=torch.uint8).pin_memory(), None] for _ in range(4)]
        self._ring_pos = 0
        self._captured = []          # pinned tables read by captured uploads (kept alive for the graphs' lifetime)
        self._cap_free = [torch.empty(C.sizeof(self._table_host), dtype=torch.uint8).pin_memory() for _ in range(int(capture_slots))]
        self._uploaded = None        # bytes of the table the device currently holds

    @property
    def steps(self):
        """per-parameter Adam step counts (host copy; synchronises)"""
        return [int(v) for v in self._steps_dev.tolist()]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, itr=None):
        """Returns the pre-clip total gradient norm as a 1-element device tensor (no host sync).

        The kernels write parameters and EMA copies through raw pointers; the eager call bumps their autograd version
        counters so that operands memoised per parameter version (the discriminator's weight planes) are rebuilt.  A
        REPLAY of a captured step runs no Python: call cips3d_amd.discriminator.invalidate_weight_cache(module) after
        replaying a graph that contains this step."""
        lib = _lib.load()
        do_ema = self.ema is not None and (itr is None or itr >= self.ema_start_itr)
        for i, p in enumerate(self.params):
            t = self._table_host[i]
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = p.grad = g.float().contiguous()
            t.param, t.grad = p.data_ptr(), (g.data_ptr() if g is not None else None)
            t.exp_avg, t.exp_avg_sq = self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
            t.ema = self.ema[i].data_ptr() if do_ema else None
            t.n = p.numel()
            t.step = 0                       # unused: the step counts are device-side (steps_dev)
        raw = bytes(self._table_host)
        capturing = torch.cuda.is_current_stream_capturing()
        if raw != self._uploaded or capturing:   # pointers change when autograd reallocates gradients; often they do not
            if capturing:
                # a captured upload is re-executed by every replay: it reads a pinned buffer of its own that nothing
                # rewrites afterwards (a ring slot would be overwritten by later eager uploads, and a replay would then
                # upload whatever pointers the slot holds at that time)
                if not self._cap_free:
                    raise RuntimeError("FusedClipAdamEMA: more captured steps with distinct tensor tables than capture slots "
                                       f"({len(self._captured)}); construct with more `capture_slots`")
                buf = self._cap_free.pop()       # pinned memory cannot be allocated while a stream is capturing
                self._captured.append(buf)
                C.memmove(buf.data_ptr(), C.addressof(self._table_host), C.sizeof(self._table_host))
                self._table_dev.copy_(buf, non_blocking=True)
                self._uploaded = None            # a replay may restore this table at any time: never skip an eager upload
            else:
                slot = self._ring[self._ring_pos]
                self._ring_pos = (self._ring_pos + 1) % len(self._ring)
                if slot[1] is not None:
                    slot[1].synchronize()        # the copy that last read this slot (4 uploads ago) — long done
                C.memmove(slot[0].data_ptr(), C.addressof(self._table_host), C.sizeof(self._table_host))
                self._table_dev.copy_(slot[0], non_blocking=True)
                slot[1] = torch.cuda.Event()
                slot[1].record()
                self._uploaded = raw if not self._captured else None
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(lib.cips_opt_step(C.c_void_p(self._table_dev.data_ptr()), C.c_void_p(self._chunk_tensor.data_ptr()),
                                C.c_void_p(self._chunk_off.data_ptr()), self.nchunks,
                                C.c_void_p(self._partial.data_ptr()), C.c_void_p(self._norm.data_ptr()),
                                self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps,
                                self.ema_decay, 1, C.c_void_p(self._steps_dev.data_ptr()), st), "cips_opt_step")
        # the kernel wrote the parameters (and the EMA copies) through raw pointers: tell autograd, so that anything
        # derived from a parameter and memoised on its version (the discriminator's weight planes) is rebuilt
        for p in self.params:
            torch.autograd.graph.increment_version(p)
        if do_ema:
            for e in self.ema:
                torch.autograd.graph.increment_version(e)
        return self._norm

    def state_dict(self):
        """torch.optim.Adam-compatible state (per-parameter step / exp_avg / exp_avg_sq)."""
        steps = self.steps
        state = {i: dict(step=torch.tensor(float(steps[i])), exp_avg=self.exp_avg[i].clone(),
                         exp_avg_sq=self.exp_avg_sq[i].clone()) for i in range(len(self.params)) if steps[i] > 0}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                     params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        for i, st in sd["state"].items():
            self.exp_avg[int(i)].copy_(st["exp_avg"]); self.exp_avg_sq[int(i)].copy_(st["exp_avg_sq"])
            self._steps_dev[int(i)] = int(float(st["step"]))
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])
