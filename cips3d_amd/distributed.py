"""Data-parallel gradient exchange for the batch-sharded hot path.

The reference wraps G and D in DistributedDataParallel(find_unused_parameters=True)
(exp/cips3d/scripts/train.py:235-236).  The only collective is the parameter-gradient mean;
here it is one flat-bucket all-reduce per bucket over RCCL/xGMI (backend "nccl" on ROCm, "gloo"
in the CPU tests), restricted to the parameters that actually received a gradient this step
(~0.2 % of G — SinStyleMod.norm.*, to_rgbs 4/8/16 — never do; generator.py:444-445, :1139).

`GradAllReducer` is the steady-state form: which parameters take part is agreed between the ranks
once (a tiny MAX all-reduce of a presence vector, the same exchange DDP's find_unused_parameters
does per step) and reused while this rank's own presence pattern stays the same, so a training
step issues, per bucket, one concatenation, one all-reduce and one multi-tensor copy-back — no
host synchronisation and no per-parameter launches (the G step is ~21 ms on an MI355X; 170
per-tensor copies plus a `.tolist()` would cost > 5 % of it)."""
import contextlib
import weakref

import torch
import torch.distributed as dist

_null = contextlib.nullcontext


class GradAllReducer:
    """Average `.grad` over the process group, in place.  Every rank must construct it with the
    same parameter list.  A parameter whose grad is None on this rank receives the others' mean
    if any rank has one, and stays None if no rank has (find_unused_parameters semantics).

    Contract (that of DDP's static_graph): the set of parameters with a gradient may differ from
    rank to rank, but when it changes it changes on every rank in the same step — it is a function
    of the step's configuration (aux image on/off, frozen NeRF, which loss), not of the data.  Each
    rank re-plans when ITS pattern changes, and the re-plan is a collective.  The contract is
    checked, cheaply: every bucket carries one extra element, a checksum of the plan it was packed
    with; the reduced value is compared on the host one call later (through an event that has
    long completed by then), so diverging ranks raise instead of silently mixing gradients.

    Parameters are NOT filtered by `requires_grad` (train.py:335-336, 441-442 toggle it on G and D
    every step): what takes part is decided per call by which gradients exist."""

    def __init__(self, params, bucket_mb=64.0, group=None, overlap=False):
        """overlap=True: once a plan exists, a bucket's all-reduce is issued from a post-accumulate-grad hook as soon as
        the last of its locally expected gradients has arrived (buckets are formed in REVERSE parameter order — roughly
        the order in which backward produces gradients — and issued strictly in bucket order, like DDP's reducer, so
        every rank issues the same sequence of collectives); on a GPU the concatenation, the collective and the copy-back
        run on a side stream that the caller's stream joins in __call__.  __call__ then issues whatever has not been
        issued, waits, and handles a changed presence pattern.  One backward per __call__ (no gradient accumulation
        across several backward passes), eager autograd only (hooks do not run inside a replayed hipGraph)."""
        self.params = list(params)
        if not self.params:
            raise ValueError("GradAllReducer: empty parameter list")
        self.group = group
        self.limit = int(bucket_mb * 1024 * 1024)
        self.overlap = bool(overlap)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._bucket_of = {}     # id(param) -> bucket index of the cached plan
        self._expected = []      # per bucket: how many of its parameters THIS rank produces a gradient for
        self._arrived = []       # per bucket: how many of those have arrived in the current backward
        self._next = 0           # next bucket to issue (buckets are issued in order)
        self._inflight = []      # (flat, work, grads, sig_view) of issued buckets
        self._dirty = False      # a gradient arrived that the plan does not expect: the pattern changed
        self._side = None        # side stream (GPU)
        self._seen = set()       # indices of the parameters whose gradient arrived in the current backward (hooks)
        self._early = 0          # buckets issued from hooks during the current backward
        self.last_launched_early = 0   # ... during the backward that the last __call__ closed (statistics / tests)
        if self.overlap:
            # torch refuses a hook on a tensor that does not require a gradient, and train.py toggles requires_grad on
            # G and D every step (a reducer may be built while a sub-net is frozen, or for a frozen-NeRF generator):
            # register with the flag raised for the moment; a parameter that stays frozen simply never fires its hook
            # (its bucket is issued by __call__'s forced pass)
            for p in self.params:
                frozen = not p.requires_grad
                if frozen:
                    p.requires_grad_(True)
                p.register_post_accumulate_grad_hook(self._on_grad)
                if frozen:
                    p.requires_grad_(False)
        self._grad_refs = None   # weak references to the gradient tensors left behind by the last call (see __call__)
        self._local = None       # this rank's presence pattern the cached plan was built for
        self._union = None       # the agreed pattern (union over ranks): what `.grad is not None` looks like after a call
        self._buckets = None     # list of lists of parameters (the union over ranks, bucketed)
        self._sig = 0.0
        self._pending = None     # (event, pinned host tensor, expected) of the previous call's plan check

    def _plan(self, local):
        dev = self.params[0].device
        present = torch.tensor([1.0 if f else 0.0 for f in local], device=dev)
        dist.all_reduce(present, op=dist.ReduceOp.MAX, group=self.group)
        flags = [f > 0 for f in present.tolist()]
        used = [p for p, f in zip(self.params, flags) if f]
        if self.overlap:
            used = used[::-1]                    # backward produces the last layers' gradients first
        buckets, cur, size = [], [], 0
        for p in used:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.limit:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        self._local, self._union, self._buckets = local, tuple(flags), buckets
        self._bucket_of = {id(p): b for b, bucket in enumerate(buckets) for p in bucket}
        self._expected = [sum(1 for p in bucket if local[self._index[id(p)]]) for bucket in buckets]
        self._reset_step()
        # plan checksum: exactly representable in fp32 so that the mean over identical ranks is exact
        self._sig = float(sum((i + 1) * 7 for i, f in enumerate(flags) if f) % 65521)

    def _sig_for(self, dtype):
        """the checksum as carried inside a bucket of `dtype`: half-precision buckets (fp16 / bf16 gradients) hold integers
        exactly only up to 2048 / 256, so the value is folded below that; fp32 / fp64 carry it as is"""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if dtype == torch.float16:
            return float(int(self._sig) % max(1, 2048 // world))       # every partial sum over the ranks stays an exact integer
        if dtype == torch.bfloat16:
            return float(int(self._sig) % max(1, 256 // world))
        return self._sig

    def _reset_step(self):
        self._arrived = [0] * len(self._buckets or [])
        self._next = 0
        self._inflight = []
        self._dirty = False
        self._seen = set()
        self._early = 0

    def _on_grad(self, p):
        """post-accumulate-grad hook (overlap mode): count the gradient in; issue every bucket that is complete"""
        i = self._index[id(p)]
        if i in self._seen:                  # a second backward before __call__: not supported in overlap mode
            self._dirty = True
        self._seen.add(i)
        if self._buckets is None or self._dirty or not dist.is_initialized():
            return
        b = self._bucket_of.get(id(p))
        if b is None or not self._local[i]:
            self._dirty = True               # a gradient the plan does not expect from this rank: handled in __call__
            return
        self._arrived[b] += 1
        before = self._next
        self._issue_ready()
        self._early += self._next - before

    def _issue_ready(self, force=False):
        nb = len(self._buckets)
        while self._next < nb and (force or self._arrived[self._next] >= self._expected[self._next]):
            self._issue(self._next)
            self._next += 1

    def _issue(self, b):
        bucket = self._buckets[b]
        for p in bucket:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        grads = [p.grad for p in bucket]
        dev = grads[0].device
        if dev.type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)        # the gradients of this bucket are complete on the producing stream
                sig = torch.full((1,), self._sig_for(grads[0].dtype), device=dev, dtype=grads[0].dtype)
                flat = torch.cat([g.reshape(-1) for g in grads] + [sig])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            sig = torch.full((1,), self._sig_for(grads[0].dtype), device=dev, dtype=grads[0].dtype)
            flat = torch.cat([g.reshape(-1) for g in grads] + [sig])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((flat, work, grads))

    def _drain(self, world):
        """wait for the issued buckets, divide, copy back; returns (bytes, checksum views, expected checksums)"""
        nbytes, sigs, wants = 0, [], []
        dev = self.params[0].device
        ctx = torch.cuda.stream(self._side) if (dev.type == "cuda" and self._side is not None) else _null()
        with ctx:
            for flat, work, grads in self._inflight:
                work.wait()
                flat.div_(world)
                views, off = [], 0
                for g in grads:
                    n = g.numel()
                    views.append(flat[off:off + n].view_as(g))
                    off += n
                torch._foreach_copy_(grads, views)
                sigs.append(flat[off:off + 1])
                wants.append(self._sig_for(flat.dtype))
                nbytes += (flat.numel() - 1) * flat.element_size()
        if dev.type == "cuda" and self._side is not None:
            torch.cuda.current_stream(dev).wait_stream(self._side)
        self._inflight = []
        return nbytes, sigs, wants

    def _check_pending(self, block=False):
        if self._pending is None:
            return
        ev, host, want = self._pending
        if ev is not None:
            if not block and not ev.query():
                return
            ev.synchronize()
        self._pending = None
        if any(abs(float(v) - w) > 1e-3 for v, w in zip(host.tolist(), want)):
            raise RuntimeError("GradAllReducer: ranks reduced with different bucket plans — the set of parameters with "
                               "a gradient changed on some ranks only (see the class contract)")

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized():
            return 0
        world = dist.get_world_size(self.group)
        if world == 1:
            return 0
        self._check_pending()
        if self.overlap and self._buckets is not None:
            return self._finish_overlapped(world)
        local = tuple(p.grad is not None for p in self.params)
        # a change of the local pattern (another loss, a frozen sub-net) re-plans; see the contract above.  A caller
        # that did not reset the gradients to None (zero_grad(set_to_none=False), gradient accumulation) shows the
        # agreed union pattern on every rank: same plan, no re-plan.  That shortcut is taken only when the gradient
        # tensors ARE the ones the last call left behind (then every rank sees the union, so the decision is the same
        # everywhere); a rank whose freshly produced pattern merely happens to equal the old union re-plans like its
        # peers whose pattern changed — the re-plan is a collective, all ranks or none must enter it.
        kept = self._grad_refs is not None and local == self._union and all(
            (not f) or (r is not None and r() is p.grad) for p, f, r in zip(self.params, self._union, self._grad_refs))
        if local != self._local and not kept:
            self._check_pending(block=True)
            self._plan(local)
        nbytes = 0
        sigs, wants = [], []
        for bucket in self._buckets:
            for p in bucket:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            grads = [p.grad for p in bucket]
            sig = torch.full((1,), self._sig_for(grads[0].dtype), device=grads[0].device, dtype=grads[0].dtype)
            flat = torch.cat([g.reshape(-1) for g in grads] + [sig])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(world)
            nbytes += (flat.numel() - 1) * flat.element_size()
            views, off = [], 0
            for g in grads:
                n = g.numel()
                views.append(flat[off:off + n].view_as(g))
                off += n
            torch._foreach_copy_(grads, views)
            sigs.append(flat[off:off + 1])
            wants.append(self._sig_for(flat.dtype))
        self._record_sigs(sigs, wants)
        self._grad_refs = [None if p.grad is None else weakref.ref(p.grad) for p in self.params]
        if self.overlap:
            self._reset_step()
        return nbytes

    def _record_sigs(self, sigs, want):
        """sigs: one reduced checksum element per bucket; want: the value each must hold"""
        if not sigs:
            return
        got = torch.cat(sigs).float()
        if got.is_cuda:
            host = torch.empty(got.shape, dtype=torch.float32).pin_memory()
            host.copy_(got, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (ev, host, want)
        else:
            self._pending = (None, got.clone(), want)

    def _finish_overlapped(self, world):
        """overlap mode, plan present: issue what the hooks did not, wait, copy back; then deal with a changed pattern"""
        old_union = self._union
        # the presence pattern of THIS backward = the gradients whose hooks fired (buckets issued early have already filled
        # the gradients this rank does not produce with zeros, so `.grad is not None` no longer tells)
        local = tuple(i in self._seen for i in range(len(self.params)))
        self.last_launched_early = self._early
        self._issue_ready(force=True)             # same bucket order on every rank, whatever arrived
        nbytes, sigs, wants = self._drain(world)
        self._record_sigs(sigs, wants)
        changed = local != self._local
        if changed:
            # (contract: every rank sees a change in the same step)  Parameters of the old plan are reduced; agree on the
            # new union and reduce the parameters that were not part of the old one
            self._check_pending(block=True)
            self._plan(local)
            extra = [p for p, was, now in zip(self.params, old_union, self._union) if now and not was]
            if extra:
                for p in extra:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                grads = [p.grad for p in extra]
                flat = torch.cat([g.reshape(-1) for g in grads])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.div_(world)
                views, off = [], 0
                for g in grads:
                    n = g.numel()
                    views.append(flat[off:off + n].view_as(g))
                    off += n
                torch._foreach_copy_(grads, views)
                nbytes += flat.numel() * flat.element_size()
        self._reset_step()
        return nbytes


def allreduce_grads(params, bucket_mb=64.0, group=None):
    """One-off form of GradAllReducer (plans, reduces, forgets): returns the bytes reduced."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    return GradAllReducer(params, bucket_mb=bucket_mb, group=group)()
