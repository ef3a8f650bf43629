"""Data-parallel gradient exchange for the batch-sharded hot path.

The reference wraps G and D in DistributedDataParallel(find_unused_parameters=True)
(exp/cips3d/scripts/train.py:235-236).  The only collective is the parameter-gradient mean;
here it is one flat-bucket all-reduce per bucket over RCCL/xGMI (backend "nccl" on ROCm, "gloo"
in the CPU tests), restricted to the parameters that actually received a gradient this step
(~0.2 % of G — SinStyleMod.norm.*, to_rgbs 4/8/16 — never do; generator.py:444-445, :1139)."""
import torch
import torch.distributed as dist


def allreduce_grads(params, bucket_mb=64.0, group=None):
    """Average .grad of `params` over the process group, in place.  Every rank must pass the same
    parameter list; parameters whose grad is None on this rank contribute zeros only if some
    rank has a gradient for them (decided by a tiny presence all-reduce)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    world = dist.get_world_size(group)
    dev = params[0].device
    present = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=dev)
    dist.all_reduce(present, op=dist.ReduceOp.MAX, group=group)
    used = [p for p, f in zip(params, present.tolist()) if f > 0]
    nbytes = 0
    bucket, bsize = [], 0
    limit = int(bucket_mb * 1024 * 1024)

    def flush():
        nonlocal bucket, bsize, nbytes
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        nbytes += flat.numel() * flat.element_size()
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        bucket, bsize = [], 0

    for p in used:
        bucket.append(p)
        bsize += p.numel() * p.element_size()
        if bsize >= limit:
            flush()
    flush()
    return nbytes
