"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference distributes with TF1 asynchronous parameter servers over gRPC
(/root/reference/distribute_training.py:174-196,226-234): every worker samples the SAME batches
(same unseeded rng) and pushes stale gradients.  Here the path is synchronous and sharded the way
the reference's own sampler already partitions a batch (base/Base.cpp:85-92): the batch belongs to
`workThreads` virtual threads, each with its own rng stream and output slice; rank g of G owns the
threads [g*W/G, (g+1)*W/G).  The union over ranks is bit-identical to the single-process batch, each
rank differentiates its slice with the GLOBAL mean denominator, and the dense summed-gradient
accumulators are all-reduced (SUM) before every replica applies the identical update.

Collective choice: the accumulators are dense [rows, dim] tables (the deduplicated IndexedSlices
sum), so one all-reduce per table per step; for FB15k-237-sized tables that is ~12 MB per step.
"""


def thread_range(rank, world_size, work_threads):
    """Half-open range of virtual sampler threads owned by `rank`."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    if work_threads % world_size != 0:
        raise ValueError("workThreads (%d) must be a multiple of the number of ranks (%d): "
                         "use Config.set_work_threads" % (work_threads, world_size))
    per = work_threads // world_size
    return rank * per, (rank + 1) * per


def slice_positions(batch_size, work_threads, thread_lo, thread_hi):
    """Pure-Python mirror of kge_slice_positions (Base.cpp:85-92): (first_position, count)."""
    def one(i):
        if batch_size % work_threads == 0:
            per = batch_size // work_threads
            return i * per, (i + 1) * per
        per = batch_size // work_threads + 1
        return min(i * per, batch_size), min((i + 1) * per, batch_size)
    if thread_lo >= thread_hi:
        return 0, 0
    lo = one(thread_lo)[0]
    hi = one(thread_hi - 1)[1]
    return lo, hi - lo


def allreduce_gradients(tensors, group=None):
    """SUM all-reduce of the per-table gradient accumulators (and the loss scalar).  Launched
    asynchronously so the tables overlap on the wire; returns when all are complete."""
    import torch.distributed as dist
    import os
    if len(tensors) > 1 and tensors[0].is_cuda and dist.get_backend(group) == "nccl" and os.environ.get("KGE_NO_COALESCE") != "1":
        # one RCCL group call: the tables and the loss scalar travel in a single fused launch instead of one
        # latency-bound collective each
        try:
            with dist._coalescing_manager(group=group, device=tensors[0].device, async_ops=True) as cm:
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            cm.wait()
            return
        except (AttributeError, TypeError, RuntimeError):
            pass
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors]
    for w in works:
        w.wait()


def max_slice_positions(lib, batch_size, world_size, work_threads):
    """Largest number of batch positions any rank owns (Base.cpp:85-92 slices grouped by rank)."""
    import ctypes
    best = 0
    for g in range(world_size):
        lo, hi = thread_range(g, world_size, work_threads)
        first = ctypes.c_int64(0)
        best = max(best, int(lib.kge_slice_positions(batch_size, lo, hi, ctypes.byref(first))))
    return max(best, 1)


def allgather_records(rec, dst, rec_all, dst_all, process_group=None):
    """Sparse gradient exchange: every rank receives every rank's int8 sign records and their destination rows
    (rank-major order; integer sums downstream make the result independent of that order)."""
    import torch
    import torch.distributed as dist
    for src, out in ((rec, rec_all), (dst, dst_all)):
        if src.is_cuda and dist.get_backend(process_group) == "gloo":
            # gloo has no device all-gather: stage through the host (test rigs only; RCCL gathers in HBM)
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, src.cpu(), group=process_group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, src, group=process_group)
