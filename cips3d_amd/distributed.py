"""Data-parallel gradient exchange for the batch-sharded hot path.

The reference wraps G and D in DistributedDataParallel(find_unused_parameters=True)
(exp/cips3d/scripts/train.py:235-236).  The only collective is the parameter-gradient mean;
here it is one flat-bucket all-reduce per bucket over RCCL/xGMI (backend "nccl" on ROCm, "gloo"
in the CPU tests), restricted to the parameters that actually received a gradient this step
(~0.2 % of G — SinStyleMod.norm.*, to_rgbs 4/8/16 — never do; generator.py:444-445, :1139).

`GradAllReducer` is the steady-state form: which parameters take part is agreed between the ranks
on EVERY call by a tiny MAX all-reduce of a presence vector (the exchange DDP's
find_unused_parameters does per step); the bucket plan is rebuilt — locally, from the agreed
union, so identically on every rank — only when that union or this rank's own pattern changed.
On a GPU the presence exchange is issued and read back on its own control stream, so the main
stream is never synchronised; in the classic (non-overlapped) form the host is ahead of the device
when `__call__` runs and the exchange completes under the step's own kernels.  (Collectives of one
communicator are serialised by ProcessGroupNCCL: in the overlapped form the exchange is issued
after the bucket reductions and the host's read therefore waits for them — that form trades host
run-ahead for the overlap.)  A training step then issues, per bucket, one concatenation, one
all-reduce and one multi-tensor copy-back — no per-parameter launches (the G step is ~16 ms on an
MI355X; 170 per-tensor copies would cost > 5 % of it)."""
import contextlib

import torch
import torch.distributed as dist

_null = contextlib.nullcontext


class GradAllReducer:
    """Average `.grad` over the process group, in place.  Every rank must construct it with the
    same parameter list.  A parameter whose grad is None on this rank receives the others' mean
    if any rank has one, and stays None if no rank has (find_unused_parameters semantics).

    The set of parameters with a gradient may differ from rank to rank and may change on any subset
    of the ranks in any step (round 4: the re-plan decision is no longer rank-local — every rank
    enters the same presence exchange on every call and derives the same plan from its result, so
    no rank can issue a bucket SUM against a peer's re-plan collective).  Defence in depth: every
    bucket carries one extra element, a checksum of the plan it was packed with; the reduced value
    is compared on the host one call later (through an event that has long completed by then), so
    ranks that were constructed with different parameter lists raise instead of mixing gradients.

    Parameters are NOT filtered by `requires_grad` (train.py:335-336, 441-442 toggle it on G and D
    every step): what takes part is decided per call by which gradients exist."""

    def __init__(self, params, bucket_mb=64.0, group=None, overlap=False, single_rank_exchange=False):
        """single_rank_exchange=True: a group of ONE rank still runs the whole exchange (presence MAX on the control stream,
        bucket concatenation, all-reduce, copy-back, plan checksum) instead of returning early — the way to execute this
        code over real RCCL streams on a one-GPU box (bench.py --rccl; the result is the identity).

        overlap=True: once a plan exists, a bucket's all-reduce is issued from a post-accumulate-grad hook as soon as
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
        self.single_rank_exchange = bool(single_rank_exchange)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._bucket_of = {}     # id(param) -> bucket index of the cached plan
        self._expected = []      # per bucket: how many of its parameters THIS rank produces a gradient for
        self._arrived = []       # per bucket: how many of those have arrived in the current backward
        self._next = 0           # next bucket to issue (buckets are issued in order)
        self._inflight = []      # (flat, work, grads, sig_view) of issued buckets
        self._dirty = False      # a gradient arrived that the plan does not expect: the pattern changed
        self._side = None        # side stream (GPU)
        self._seen = set()       # indices of the parameters whose gradient arrived in the current backward (hooks)
        self._late = {}          # index -> this rank's gradient of a parameter whose bucket had already been issued
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
        self._ctl = None         # control stream of the presence exchange (GPU)
        self._streams = []       # per bucket: the streams its gradients were produced on in the current backward (hooks)
        self._local = None       # this rank's presence pattern the cached plan was built for
        self._union = None       # the agreed pattern (union over ranks): what `.grad is not None` looks like after a call
        self._buckets = None     # list of lists of parameters (the union over ranks, bucketed)
        self._sig = 0.0
        self._pending = None     # (event, pinned host tensor, expected) of the previous call's plan check

    def _agree(self, local, late=()):
        """the union over ranks of the presence patterns: one MAX all-reduce of len(params) ints, entered by EVERY rank on
        EVERY call -> (union, late union).  A parameter in `late` (overlap mode: its gradient arrived after its bucket had
        been issued) is sent as 2, so the MAX also agrees on the set of parameters that need the fix-up reduction.  GPU:
        issued and read back on a control stream; the host's wait ends when the collective does (which, in overlap mode,
        is behind the bucket reductions of the same communicator)."""
        dev = self.params[0].device
        vals = [2 if i in late else (1 if f else 0) for i, f in enumerate(local)]
        if dev.type == "cuda":
            if self._ctl is None:
                self._ctl = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._ctl):
                present = torch.tensor(vals, dtype=torch.int32, device=dev)
                dist.all_reduce(present, op=dist.ReduceOp.MAX, group=self.group)
                flags = present.tolist()                     # synchronises the control stream only
        else:
            present = torch.tensor(vals, dtype=torch.int32)
            dist.all_reduce(present, op=dist.ReduceOp.MAX, group=self.group)
            flags = present.tolist()
        return tuple(f > 0 for f in flags), tuple(i for i, f in enumerate(flags) if f > 1)

    def _plan(self, local, flags):
        """bucket plan for the agreed union `flags` — a pure function of (flags, this rank's `local`): no collective"""
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
        self._streams = [set() for _ in (self._buckets or [])]
        self._next = 0
        self._inflight = []
        self._dirty = False
        self._seen = set()
        self._early = 0
        self._late = {}

    def _on_grad(self, p):
        """post-accumulate-grad hook (overlap mode): count the gradient in; issue every bucket that is complete"""
        i = self._index[id(p)]
        planned = self._buckets is not None and dist.is_initialized()
        if i in self._seen and planned:
            # a second backward before __call__ (gradient accumulation, train.py:466's batch_split loop): the buckets of this
            # step went out with the first backward's gradients, so what arrives now cannot be told from a LATE gradient and
            # the result would be mean(g1) + mean(g1 + g2) (measured 40 % off).  Refuse instead of mis-reducing.
            raise RuntimeError("GradAllReducer(overlap=True): a parameter's gradient arrived twice before the reduction — "
                               "gradient accumulation over several backward passes needs overlap=False")
        self._seen.add(i)
        if not planned:
            return
        b = self._bucket_of.get(id(p))
        if b is not None and b < self._next and p.grad is not None:
            # LATE: the bucket of this parameter is already on the wire — the plan did not expect a gradient from this rank
            # (a peer produced it last step, this rank did not), so the bucket went out with zeros in its place and the
            # copy-back will overwrite what autograd accumulates now.  Keep this rank's contribution (zero fill + accumulate
            # = exactly it); __call__ agrees on the late set with the other ranks and adds the missing mean (ADVICE r4:
            # before, the contribution was silently lost on every rank, and the plan checksum could not see it).
            self._late[i] = p.grad.detach().clone()
            self._dirty = True
            return
        if self._dirty:
            return
        if b is None or not self._local[i]:
            self._dirty = True               # a gradient the plan does not expect from this rank: handled in __call__
            return
        self._arrived[b] += 1
        if p.grad is not None and p.grad.is_cuda:
            # autograd runs a node on its forward's stream (the INR mapping network's is a side stream, generator.py):
            # the gradients of one bucket may come from several streams, and they only join when backward() returns
            self._streams[b].add(torch.cuda.current_stream(p.grad.device))
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
        # what goes on the wire: this rank's gradient, or a PRIVATE zero vector for a parameter it has none for.  (Round 5
        # bound zeros to p.grad here: autograd then accumulated a late gradient into the very tensor the side stream's
        # concatenation was still reading — ADVICE r5, write-after-read across streams — and under create_graph=True
        # AccumulateGrad rebinds p.grad, which left the copy-back writing an orphan.)  The copy-back looks p.grad up again.
        dev = bucket[0].device
        dtype = next((p.grad.dtype for p in bucket if p.grad is not None), bucket[0].dtype)
        grads = [p.grad if p.grad is not None else None for p in bucket]
        if dev.type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            # every stream a gradient of this bucket was produced on (one event on the last hook's stream let
            # the concatenation read a gradient still being written on another stream)
            producers = set(self._streams[b]) if b < len(self._streams) else set()
            producers.add(torch.cuda.current_stream(dev))
            evs = []
            for st in producers:
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            with torch.cuda.stream(self._side):
                for ev in evs:
                    self._side.wait_event(ev)    # the gradients of this bucket are complete on their producing streams
                flat = self._pack(bucket, grads, dtype, dev)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat = self._pack(bucket, grads, dtype, dev)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((flat, work, bucket))

    def _pack(self, bucket, grads, dtype, dev):
        sig = torch.full((1,), self._sig_for(dtype), device=dev, dtype=dtype)
        parts = [g.reshape(-1) if g is not None else torch.zeros(p.numel(), dtype=dtype, device=dev) for p, g in zip(bucket, grads)]
        return torch.cat(parts + [sig])

    def _drain(self, world):
        """wait for the issued buckets, divide, copy back; returns (bytes, checksum views, expected checksums)"""
        nbytes, sigs, wants = 0, [], []
        dev = self.params[0].device
        ctx = torch.cuda.stream(self._side) if (dev.type == "cuda" and self._side is not None) else _null()
        with ctx:
            for flat, work, bucket in self._inflight:
                work.wait()
                flat.div_(world)
                dst, views, off = [], [], 0
                for p in bucket:
                    n = p.numel()
                    v = flat[off:off + n].view_as(p)
                    off += n
                    if p.grad is None:               # no gradient on this rank: it receives the others' mean
                        p.grad = v.clone()
                    else:                            # the tensor p.grad is bound to NOW (autograd may have rebound it)
                        dst.append(p.grad)
                        views.append(v)
                if dst:
                    torch._foreach_copy_(dst, views)
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
            raise RuntimeError("GradAllReducer: ranks reduced with different bucket plans — the reducers were constructed with "
                               "different parameter lists, or the ranks are not calling them in step")

    def __call__(self):
        if not dist.is_available() or not dist.is_initialized():
            return 0
        world = dist.get_world_size(self.group)
        if world == 1 and not self.single_rank_exchange:
            return 0
        self._check_pending()
        if self.overlap and self._buckets is not None:
            return self._finish_overlapped(world)
        local = tuple(p.grad is not None for p in self.params)
        # every rank enters the presence exchange, every call; the plan follows from its result alone.  (A caller that
        # leaves gradients in place — zero_grad(set_to_none=False) — shows the previous union on every rank: same plan.)
        union, _ = self._agree(local)
        if union != self._union or local != self._local:
            self._check_pending(block=True)
            self._plan(local, union)
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
        # the presence exchange comes AFTER the forced pass: the number of buckets the hooks issued early differs from
        # rank to rank, the sequence "all buckets of the old plan, then the exchange" does not
        late_mine = dict(self._late)
        union, late = self._agree(local, late=late_mine)
        dev = self.params[0].device
        if dev.type == "cuda" and self._side is not None:
            # backward() has returned, so the caller's stream has joined every stream a gradient was produced on — including
            # the accumulation of a LATE gradient, which no bucket event covers.  The copy-back below must come after it.
            self._side.wait_stream(torch.cuda.current_stream(dev))
        nbytes, sigs, wants = self._drain(world)
        self._record_sigs(sigs, wants)
        if late:
            # the same index set on every rank (an agreed value): the contributions that missed their buckets, zeros from
            # the ranks that were not late, summed and added to the means the buckets delivered
            ps = [self.params[i] for i in late]
            parts = [late_mine[i].reshape(-1) if i in late_mine else torch.zeros(p.numel(), dtype=p.grad.dtype, device=p.grad.device)
                     for i, p in zip(late, ps)]
            flat = torch.cat(parts)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad.add_(flat[off:off + n].view_as(p.grad))
                off += n
            nbytes += flat.numel() * flat.element_size()
        if union != old_union or local != self._local:
            # the parameters of the old plan are reduced; the ones new to the union (the same set on every rank: both
            # unions are agreed values) are reduced now, and the next backward runs under the new plan
            self._check_pending(block=True)
            self._plan(local, union)
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
