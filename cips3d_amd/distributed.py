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
    rank re-plans when ITS pattern changes, and the re-plan is a collective."""

    def __init__(self, params, bucket_mb=64.0, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.limit = int(bucket_mb * 1024 * 1024)
        self._local = None       # this rank's presence pattern the cached plan was built for
        self._buckets = None     # list of lists of parameters (the union over ranks, bucketed)

    def _plan(self, local):
        dev = self.params[0].device
        present = torch.tensor([1.0 if f else 0.0 for f in local], device=dev)
        dist.all_reduce(present, op=dist.ReduceOp.MAX, group=self.group)
        used = [p for p, f in zip(self.params, present.tolist()) if f > 0]
        buckets, cur, size = [], [], 0
        for p in used:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.limit:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        self._local, self._buckets = local, buckets

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized() or not self.params:
            return 0
        world = dist.get_world_size(self.group)
        if world == 1:
            return 0
        local = tuple(p.grad is not None for p in self.params)
        # a change of the local pattern (another loss, a frozen sub-net) re-plans; see the contract above
        if local != self._local:
            self._plan(local)
        nbytes = 0
        for bucket in self._buckets:
            for p in bucket:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            grads = [p.grad for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(world)
            nbytes += flat.numel() * flat.element_size()
            views, off = [], 0
            for g in grads:
                n = g.numel()
                views.append(flat[off:off + n].view_as(g))
                off += n
            torch._foreach_copy_(grads, views)
        # parameters that were None here but got the others' mean now carry a grad: the local pattern for the next
        # call is computed from p.grad again, after the caller's zero_grad / set-to-None, so nothing to fix up
        return nbytes


def allreduce_grads(params, bucket_mb=64.0, group=None):
    """One-off form of GradAllReducer (plans, reduces, forgets): returns the bytes reduced."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    return GradAllReducer(params, bucket_mb=bucket_mb, group=group)()
