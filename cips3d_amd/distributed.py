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
import torch
import torch.distributed as dist


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

    def __init__(self, params, bucket_mb=64.0, group=None):
        self.params = list(params)
        if not self.params:
            raise ValueError("GradAllReducer: empty parameter list")
        self.group = group
        self.limit = int(bucket_mb * 1024 * 1024)
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
        # plan checksum: exactly representable in fp32 so that the mean over identical ranks is exact
        self._sig = float(sum((i + 1) * 7 for i, f in enumerate(flags) if f) % 65521)

    def _check_pending(self, block=False):
        if self._pending is None:
            return
        ev, host, want = self._pending
        if ev is not None:
            if not block and not ev.query():
                return
            ev.synchronize()
        self._pending = None
        if any(abs(float(v) - want) > 1e-3 for v in host.tolist()):
            raise RuntimeError("GradAllReducer: ranks reduced with different bucket plans — the set of parameters with "
                               "a gradient changed on some ranks only (see the class contract)")

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized():
            return 0
        world = dist.get_world_size(self.group)
        if world == 1:
            return 0
        self._check_pending()
        local = tuple(p.grad is not None for p in self.params)
        # a change of the local pattern (another loss, a frozen sub-net) re-plans; see the contract above.  A caller
        # that did not reset the gradients to None (zero_grad(set_to_none=False), gradient accumulation) shows the
        # agreed union pattern on every rank: same plan, no re-plan.
        if local != self._local and local != self._union:
            self._check_pending(block=True)
            self._plan(local)
        nbytes = 0
        sigs = []
        for bucket in self._buckets:
            for p in bucket:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            grads = [p.grad for p in bucket]
            sig = torch.full((1,), self._sig, device=grads[0].device, dtype=grads[0].dtype)
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
        if sigs:
            got = torch.cat(sigs).float()
            if got.is_cuda:
                host = torch.empty(got.shape, dtype=torch.float32).pin_memory()
                host.copy_(got, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (ev, host, self._sig)
            else:
                self._pending = (None, got, self._sig)
        return nbytes


def allreduce_grads(params, bucket_mb=64.0, group=None):
    """One-off form of GradAllReducer (plans, reduces, forgets): returns the bytes reduced."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    return GradAllReducer(params, bucket_mb=bucket_mb, group=group)()
