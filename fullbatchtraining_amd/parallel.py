"""Data-parallel sharding of the full-batch gradient over the GPUs of one node (one process per GPU, RCCL over xGMI).

Reference behaviour being replaced: ``DistributedSampler`` + per-rank running mean pre-divided by the world size + one
flat ``all_reduce`` (``fullbatch/training/utils.py:31-41``, ``training.py:168,179-180``).  That scheme is *not* the exact
mean and changes chunk composition; here the 1-process semantics are kept:

  * chunk k is always samples [k*chunk, (k+1)*chunk); rank r owns a contiguous range of chunks (``ShardPlan``)
  * each rank folds its chunks into a local running mean, scales it by K_r/K, and ONE reduce-scatter(SUM) yields the
    exact global mean, sharded;  the global clip norm needs one 4-byte all-reduce;  clip + Nesterov-SGD run on the
    local shard with sharded momentum (ZeRO-1 style);  ONE all-gather republishes the updated parameters
  * BN running statistics: the sequential EMA is linear in the per-chunk batch statistics, so rank-local EMA
    contributions are recombined exactly (``combine_running_stats``)
  * per-chunk stats (loss, #correct, squared norms) are all-gathered (K floats each) so every rank logs full stats

The collective wiring is independent of the compute backend: ``ShardOps`` supplies scale / squared-norm / update
callables (HIP kernels in the product, CPU stand-ins in the gloo tests).
"""
import os

import torch
import torch.distributed as dist


def group_size(n_chunks, chunk_group, cap=None):
    """Chunks per batched launch for a rank that owns ``n_chunks``: equal-sized groups (at most one chunk of difference), as many as
    brings the size closest to ``chunk_group`` -- 390 chunks run as 4 x 98, a rank's 49 chunks (8 GPUs) as one group of 49.  The size
    stays below 1.5 x chunk_group and never exceeds ``cap`` (the largest group whose activation tensors stay below 2^31 bytes, the
    range of the 32-bit buffer offsets of the fast kernels: ``engine.max_group``)."""
    if n_chunks <= 0:
        return max(1, min(chunk_group, cap or chunk_group))
    g = max(1, min(chunk_group, n_chunks))
    n_groups = max(1, int(n_chunks / g + 0.5))
    g = -(-n_chunks // n_groups)
    if cap is not None and g > cap:
        n_groups = -(-n_chunks // max(1, cap))
        g = -(-n_chunks // n_groups)
    return g


class ShardPlan:
    """Contiguous chunk ranges: the first ``K % world`` ranks own one chunk more."""

    def __init__(self, n_chunks, world, rank):
        q, r = divmod(n_chunks, world)
        self.n_chunks, self.world, self.rank = n_chunks, world, rank
        self.counts = [q + (1 if i < r else 0) for i in range(world)]
        self.firsts = [sum(self.counts[:i]) for i in range(world)]
        self.first, self.count = self.firsts[rank], self.counts[rank]


def _reduce_scatter_sum(out_shard, full, group=None):
    if dist.get_backend(group) == "gloo":      # gloo has no reduce_scatter: all-reduce and keep the local shard
        dist.all_reduce(full, group=group)
        n = out_shard.numel()
        out_shard.copy_(full[dist.get_rank(group) * n:(dist.get_rank(group) + 1) * n])
        if os.environ.get("FB_EXCHANGE_POISON", "1") != "0":
            full.fill_(float("nan"))            # RCCL leaves ``full`` un-reduced: nothing may read it again (see BucketExchange.start)
    else:
        dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=group)


class ShardOps:
    """Compute callbacks used by ``reduce_scatter_update_all_gather``."""

    def __init__(self, scale, sqnorm, update):
        self.scale, self.sqnorm, self.update = scale, sqnorm, update


class BucketExchange:
    """The gradient exchange of one step in buckets of the flat arena (``bounds`` = [0, b, P]: early parameters, late parameters).

    Every bucket is reduce-scattered on its own (rank r owns the r-th part of EACH bucket), so a bucket whose gradients are complete
    can leave while the backward pass still works on the other: ``start(i)`` (scale by K_r/K + asynchronous reduce-scatter) may be
    called early -- from the stream on which bucket i's running mean was completed (``Engine.full_gradient(late_bucket=...)``) -- and
    ``finish()`` starts whatever has not left yet, waits, all-reduces the squared norm, runs the shard-local clip + update on the
    rank's range of every bucket and all-gathers the updated parameters bucket by bucket.
    Total traffic is that of ONE reduce-scatter + ONE all-gather of the arena, cut in two messages each.
    On gloo (CPU tests, shared-device development runs) ``start`` is a BLOCKING all-reduce of the bucket in place: the ranks rendezvous
    inside the backward pass, so that path checks ordering and arithmetic, never overlap; and the ranges of ``avg`` a rank does not own
    hold the reduced values there but the rank's un-reduced local values on RCCL -- they are undefined until ``gather_sharded_state``."""

    def __init__(self, avg, theta, plan, ops, bounds, group=None):
        self.avg, self.theta, self.plan, self.ops, self.group = avg, theta, plan, ops, group
        self.bounds = list(bounds)
        for lo, hi in zip(self.bounds, self.bounds[1:]):
            assert (hi - lo) % plan.world == 0 and lo % 4 == 0, "bucket bounds must be multiples of lcm(4, world)"
        self.pending = {}
        # FB_EXCHANGE_TIMING=1: device timestamps around every bucket's reduce-scatter (on the stream it was started from) and at the point
        # where the main stream comes to wait for it -- how much of the exchange the backward pass hides (``timing_summary``)
        self.timed = os.environ.get("FB_EXCHANGE_TIMING") == "1" and avg.is_cuda
        self.stamps = {}

    def timing_summary(self):
        """After a device synchronisation: per bucket the duration of its reduce-scatter, how long before the main stream needed the result
        it had started (``lead_ms``) and the part of it that ran hidden under the backward pass (``overlap_frac``)."""
        out = []
        for i, (t0, t1, tm) in sorted(self.stamps.items()):
            rs, lead = t0.elapsed_time(t1), t0.elapsed_time(tm)
            out.append(dict(bucket=i, elements=self.bounds[i + 1] - self.bounds[i], reduce_scatter_ms=round(rs, 3), lead_ms=round(lead, 3),
                            exposed_ms=round(max(0.0, rs - max(lead, 0.0)), 3), overlap_frac=round(min(max(lead, 0.0), rs) / rs, 4) if rs > 0 else None))
        return out

    def ranges(self):
        """This rank's (lo, n) range of every bucket."""
        out = []
        for lo, hi in zip(self.bounds, self.bounds[1:]):
            n = (hi - lo) // self.plan.world
            out.append((lo + self.plan.rank * n, n))
        return out

    def start(self, i):
        lo, hi = self.bounds[i], self.bounds[i + 1]
        n = (hi - lo) // self.plan.world
        part = self.avg[lo:hi]
        self.ops.scale(part, self.plan.count / self.plan.n_chunks)
        shard = torch.empty(n, device=part.device, dtype=part.dtype)
        t0 = t1 = None
        if self.timed:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        if dist.get_backend(self.group) == "gloo":          # no reduce_scatter in gloo: all-reduce, keep the local part
            dist.all_reduce(part, group=self.group)
            shard.copy_(part[self.plan.rank * n:(self.plan.rank + 1) * n])
            work = None
        else:
            work = dist.reduce_scatter_tensor(shard, part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.timed:
            if work is not None:
                work.wait()                                 # (the stream this bucket was started from waits: nothing else is queued on it)
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self.stamps[i] = [t0, t1, None]
        poison = os.environ.get("FB_EXCHANGE_POISON")
        if poison == "1" or (poison is None and work is None):
            # Once the bucket has left, nothing may read this rank's LOCAL values of it again (finish() consumes ``shard``; the ranges of ``avg``
            # a rank does not own are undefined until gather_sharded_state()).  Overwrite them with NaN behind the collective, on the stream it
            # was started from: any later consumer of the stale slice poisons the step.  FB_EXCHANGE_POISON=1: a test switch on RCCL; on the
            # gloo stand-in (tests and shared-device development runs only) it is the DEFAULT -- gloo's all-reduce leaves the reduced values in
            # every range, which would let a read of a range the rank does not own pass every gloo test and fail only on real RCCL.
            if work is not None:
                work.wait()
            part.fill_(float("nan"))
        self.pending[i] = (shard, work, torch.cuda.current_stream().record_event() if part.is_cuda else None)

    def finish(self):
        n_buckets = len(self.bounds) - 1
        tm = None
        if self.timed:                                      # here the main stream turns to the exchange (buckets started below hide nothing)
            tm = torch.cuda.Event(enable_timing=True)
            tm.record()
        for i in reversed(range(n_buckets)):               # late buckets first: the order every rank issues its collectives in, whether or
            if i not in self.pending:                      # not it started a bucket early
                self.start(i)
        gnorm2 = None
        for st in self.stamps.values():
            st[2] = tm
        for i, (lo_r, n) in enumerate(self.ranges()):
            shard, work, ev = self.pending[i]
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)  # the stream the bucket was started on (scale + gloo copy)
            if work is not None:
                work.wait()                                 # current stream waits for the collective
            if shard.is_cuda:
                shard.record_stream(torch.cuda.current_stream())     # allocated under the side stream, read here
            self.avg[lo_r:lo_r + n].copy_(shard)
            part2 = self.ops.sqnorm(self.avg[lo_r:lo_r + n]).reshape(1).clone()
            gnorm2 = part2 if gnorm2 is None else gnorm2 + part2
        dist.all_reduce(gnorm2, group=self.group)
        for lo_r, n in self.ranges():
            self.ops.update(lo_r, n, gnorm2)
        for i, (lo_r, n) in enumerate(self.ranges()):
            lo, hi = self.bounds[i], self.bounds[i + 1]
            dist.all_gather_into_tensor(self.theta[lo:hi], self.theta[lo_r:lo_r + n].clone(), group=self.group)
        self.pending = {}
        return gnorm2[0]


def reduce_scatter_update_all_gather(avg, theta, plan, ops, group=None):
    """avg: this rank's local running mean (flat, numel divisible by world); theta: replicated parameters (flat).

    On return ``theta`` holds the updated parameters on every rank, ``avg[lo:hi]`` the (clipped) global mean shard.
    Returns the global squared gradient norm (0-d tensor on avg's device)."""
    world, rank = plan.world, plan.rank
    P = avg.numel()
    assert P % world == 0, "arena must be padded to a multiple of the world size"
    n = P // world
    lo = rank * n
    ops.scale(avg, plan.count / plan.n_chunks)
    shard = torch.empty(n, device=avg.device, dtype=avg.dtype)
    _reduce_scatter_sum(shard, avg, group)
    avg[lo:lo + n].copy_(shard)
    gnorm2 = ops.sqnorm(avg[lo:lo + n]).reshape(1).clone()
    dist.all_reduce(gnorm2, group=group)
    ops.update(lo, n, gnorm2)
    new_shard = theta[lo:lo + n].clone()
    dist.all_gather_into_tensor(theta, new_shard, group=group)
    return gnorm2[0]


def all_gather_chunk_stats(local, plan, group=None):
    """local: [count_r] (or [R, count_r]: R statistics in ONE collective) per-chunk values of this rank -> [K] (or [R, K]) in chunk
    order on every rank."""
    flat = local.dim() == 1
    rows = local.reshape(1, -1) if flat else local
    width = max(plan.counts)
    padded = torch.zeros(rows.shape[0], width, device=rows.device, dtype=rows.dtype)
    padded[:, :rows.shape[1]] = rows
    out = torch.empty(plan.world * padded.numel(), device=rows.device, dtype=rows.dtype)
    dist.all_gather_into_tensor(out, padded.reshape(-1), group=group)
    out = out.view(plan.world, rows.shape[0], width)
    res = torch.cat([out[r, :, :plan.counts[r]] for r in range(plan.world)], dim=1)
    return res[0] if flat else res


def combine_running_stats(r0, r_local, plan, updates_per_chunk, momentum=0.1, group=None):
    """Exact recombination of rank-local sequential EMAs.

    r_local = keep^{n_r} r0 + S_r with n_r = updates of rank r; the 1-process result is
    keep^{N} r0 + sum_r keep^{(updates after rank r)} S_r.  r0/r_local: [2, ch] (mean row, var row)."""
    keep = 1.0 - momentum
    n_r = [c * updates_per_chunk for c in plan.counts]
    s_local = r_local - (keep ** n_r[plan.rank]) * r0
    gathered = torch.empty(plan.world * s_local.numel(), device=s_local.device, dtype=s_local.dtype)
    dist.all_gather_into_tensor(gathered, s_local.reshape(-1).contiguous(), group=group)
    gathered = gathered.view(plan.world, -1)
    # out = keep^N r0 + sum_r keep^(updates after rank r) S_r : one weighted sum over the rank axis (weights evaluated in float64 on the host;
    # two elementwise launches instead of 2 x world)
    weights = torch.tensor([keep ** sum(n_r[r + 1:]) for r in range(plan.world)], dtype=s_local.dtype, device=s_local.device)
    return (keep ** sum(n_r)) * r0 + (weights[:, None] * gathered).sum(0).view_as(r0)


def bn_updates_per_chunk(hyp):
    """BN running-statistic updates per chunk in the main loop: the base pass plus the finite-difference passes (which run when
    either regulariser strength is non-zero: reference modules.py:150-152)."""
    fd = hyp.grad_reg.block_strength != 0 or hyp.grad_reg.acc_strength != 0
    return 1 + (0 if not fd else (2 if hyp.grad_reg.implementation == "central-differences" else 1))


def reduce_pre_pass(trainer):
    """``grad_reg.acc_strength`` with several ranks: the pre-pass mean over the local chunks (``engine.pre``) becomes the global mean
    (x K_r/K, one all-reduce) before the first chunk is regularised; its BN updates (one per chunk, all before the main loop's)
    are recombined on their own, and the main loop's recombination starts from the result."""
    from .lib import call
    eng, plan = trainer.engine, trainer.shard
    call("fb_mt_scale", eng.pre.data_ptr(), eng.pre.numel(), float(plan.count / plan.n_chunks))
    dist.all_reduce(eng.pre)
    combined = combine_running_stats(trainer._running0, torch.stack([eng.running_mean, eng.running_var]), plan, 1)
    eng.running_mean.copy_(combined[0])
    eng.running_var.copy_(combined[1])
    eng.num_batches_tracked += plan.n_chunks - plan.count
    trainer._running0 = combined.clone()


def replicated_reduce(trainer, loss_k, correct_k, sq_k):
    """The exchange for the options that need the WHOLE averaged gradient on every rank (SAM's ascent step, the norm bias, the
    L-infinity clip, per-tensor weight decay): local running mean x K_r/K -> ONE all-reduce(SUM) = the exact global mean, replicated;
    BN running statistics recombined, per-chunk statistics gathered.  The caller then continues exactly like the 1-process step (the
    update is replicated: 45 MB per pass, nothing next to the gradient evaluation)."""
    from .lib import call
    eng, hyp, plan = trainer.engine, trainer.cfg.hyp, trainer.shard
    call("fb_mt_scale", eng.avg.data_ptr(), eng.avg.numel(), float(plan.count / plan.n_chunks))
    dist.all_reduce(eng.avg)
    passes = bn_updates_per_chunk(hyp)
    r_local = torch.stack([eng.running_mean, eng.running_var])
    combined = combine_running_stats(trainer._running0, r_local, plan, passes)
    eng.running_mean.copy_(combined[0])
    eng.running_var.copy_(combined[1])
    eng.num_batches_tracked += (plan.n_chunks - plan.count) * passes
    gathered = all_gather_chunk_stats(torch.stack([loss_k, correct_k, sq_k]), plan)
    return gathered[0], gathered[1], gathered[2]


def shard_ops(trainer, lr, weight_decay=None):
    """HIP kernels as ShardOps for ``FullBatchTrainer``.  ``weight_decay`` overrides hyp.optim's (LARS / LARC step without it)."""
    from .lib import call
    eng, hyp = trainer.engine, trainer.cfg.hyp
    o = hyp.optim
    first = eng.first_step

    def scale(t, a):
        call("fb_mt_scale", t.data_ptr(), t.numel(), float(a))

    def sqnorm(t):
        call("fb_mt_norms2", t.data_ptr(), None, t.numel(), eng.norms2.data_ptr(), eng.mt_ws.data_ptr())
        return eng.norms2[0]

    def update(lo, n, gnorm2):
        eng.norms2[0:1].copy_(gnorm2)
        eng.first_step = first                   # one optimizer step, several arena ranges
        eng.sgd_step(lr, o.weight_decay if weight_decay is None else weight_decay, o.momentum, o.dampening, o.nesterov, hyp.grad_clip, lo=lo, n=n)

    return ShardOps(scale, sqnorm, update)


def exchange_bounds(trainer):
    """[0, b, P]: b = start of the last stage's parameters rounded UP to the shard granule (the late bucket then holds only gradients
    that are complete once the backward pass has left the last stage)."""
    eng, world = trainer.engine, trainer.shard.world
    granule = 4 * world // __import__("math").gcd(4, world)
    b = (eng.plan.late_offset + granule - 1) // granule * granule
    return [0, b, eng.plan.P]


def sharded_update(trainer, loss_k, correct_k, sq_k, lr, weight_decay=None, exchange=None):
    """Product wiring of the above for ``FullBatchTrainer`` (HIP kernels as ShardOps).  ``weight_decay`` overrides hyp.optim's (the
    LARS / LARC wrappers step without it).  ``exchange``: the step's ``BucketExchange`` when the closure has already started its late
    bucket."""
    from .lib import call
    eng, hyp, plan = trainer.engine, trainer.cfg.hyp, trainer.shard
    P = eng.plan.P
    # parameter norm of the (replicated) pre-update parameters for the stats
    call("fb_mt_norms2", eng.theta.data_ptr(), None, P, eng.norms2.data_ptr(), eng.mt_ws.data_ptr())
    pnorm2 = eng.norms2[0].clone()
    if exchange is None:
        exchange = BucketExchange(eng.avg, eng.theta, plan, shard_ops(trainer, lr, weight_decay), exchange_bounds(trainer))
    gnorm2 = exchange.finish()
    eng.norms2[0] = gnorm2
    eng.norms2[1] = pnorm2
    # BN running statistics: recombine the rank-local EMAs
    passes = bn_updates_per_chunk(hyp)
    r_local = torch.stack([eng.running_mean, eng.running_var])
    combined = combine_running_stats(trainer._running0, r_local, plan, passes)
    eng.running_mean.copy_(combined[0])
    eng.running_var.copy_(combined[1])
    eng.num_batches_tracked += (plan.n_chunks - plan.count) * passes
    gathered = all_gather_chunk_stats(torch.stack([loss_k, correct_k, sq_k]), plan)        # one collective for the three statistics
    return gathered[0], gathered[1], gathered[2]


def gather_sharded_state(trainer, group=None):
    """After ``sharded_update`` a rank holds only ITS shard of the momentum and of the clipped averaged gradient.  Before a
    checkpoint is written (rank 0 saves the whole ``momentum_buffer`` list, reference training/utils.py:43-49) or ``p.grad`` is
    exposed (closure contract), the shards are all-gathered into the full arenas.  Collective: every rank calls it."""
    eng, plan = trainer.engine, trainer.shard
    bounds = exchange_bounds(trainer)
    for lo, hi in zip(bounds, bounds[1:]):          # the shard layout of BucketExchange: rank r owns the r-th part of every bucket
        n = (hi - lo) // plan.world
        for arena in (eng.mom, eng.avg):
            dist.all_gather_into_tensor(arena[lo:hi], arena[lo + plan.rank * n:lo + (plan.rank + 1) * n].clone(), group=group)
