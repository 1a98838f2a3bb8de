"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference distributes with TF1 asynchronous parameter servers over gRPC
(/root/reference/distribute_training.py:174-196,226-234): the variables are SHARDED over the ps tasks
(`replica_device_setter`, :193-196), every worker samples the SAME batches (same unseeded rng) and pushes
stale gradients.  Here the path is synchronous and every per-rank stage is O(1/N) of the global work:

* the batch is split the way the reference's own sampler already partitions it (base/Base.cpp:85-92):
  `workThreads` virtual threads, each with its own rng stream and output slice; rank g of G owns the
  threads [g*W/G, (g+1)*W/G).  The union over ranks is bit-identical to the single-process batch and each
  rank differentiates its slice with the GLOBAL mean denominator;
* dense tables (FB15k-237 / WN18RR sized): the summed-gradient image (int32 sign counts for TransE, fp32
  accumulators otherwise) is REDUCE-SCATTERED, rank g applies the optimizer to its chunk of the flat
  parameter buffer only (owner computes: no replicated Adam sweep), and the updated chunks are
  ALL-GATHERED.  Replicas are bit-identical by construction (one owner computes every element);
* tables too large to replicate (config #5: 50 M x 512 = 102 GB): the entity table itself is sharded by
  row range, rows and int8 gradient records travel by ALL-TO-ALL to / from their owners
  (Config._sharded_step, csrc/shard.hip); only the small relation table is replicated and its integer
  count image all-reduced.

The helpers below are thin wrappers over torch.distributed that also serve the `gloo` test rigs (gloo has no
device collectives for these ops: CUDA tensors are staged through the host there, and only there).
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


def max_slice_positions(lib, batch_size, world_size, work_threads):
    """Largest number of batch positions any rank owns (Base.cpp:85-92 slices grouped by rank)."""
    import ctypes
    best = 0
    for g in range(world_size):
        lo, hi = thread_range(g, world_size, work_threads)
        first = ctypes.c_int64(0)
        best = max(best, int(lib.kge_slice_positions(batch_size, lo, hi, ctypes.byref(first))))
    return max(best, 1)


def chunk_size(total, world_size, align=1):
    """Elements (or rows) per rank so that world_size chunks cover `total`, each a multiple of `align`."""
    per = -(-int(total) // int(world_size))
    return -(-per // align) * align


def _host_staged(tensor, group):
    import torch.distributed as dist
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_sum(tensors, group=None):
    """SUM all-reduce of a few small tensors (the loss scalar, the relation count image), launched together."""
    import torch.distributed as dist
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors]
    for w in works:
        w.wait()


def reduce_scatter_sum(out, inp, group=None, async_op=False):
    """out[chunk] = sum over ranks of inp[rank*chunk : (rank+1)*chunk]; inp has world_size * out.numel() elements.
    async_op=True: returns the collective's Work (wait() makes the CURRENT STREAM wait, not the host), or None where the
    exchange had to be staged through the host and is already complete (gloo test rigs)."""
    import torch
    import torch.distributed as dist
    if inp.numel() != out.numel() * dist.get_world_size(group):
        raise ValueError("reduce_scatter_sum: input must hold world_size equal chunks")
    if _host_staged(inp, group):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(h_out, inp.cpu(), op=dist.ReduceOp.SUM, group=group)
        out.copy_(h_out)
        return None
    work = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else None


def all_gather_chunks(out, inp, group=None, async_op=False):
    """out = concatenation over ranks of inp (equal chunks).  `inp` may be this rank's slice of `out` (in place).
    async_op: as reduce_scatter_sum."""
    import torch
    import torch.distributed as dist
    if out.numel() != inp.numel() * dist.get_world_size(group):
        raise ValueError("all_gather_chunks: output must hold world_size equal chunks")
    if _host_staged(inp, group):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, inp.cpu(), group=group)
        out.copy_(h_out)
        return None
    if dist.get_backend(group) == "gloo":
        dist.all_gather_into_tensor(out, inp.clone(), group=group)   # gloo copies chunk by chunk: keep input and output apart
        return None
    work = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    return work if async_op else None


def exchange_counts_max(send_counts, group=None):
    """send_counts: int32 tensor [world] (how many rows this rank sends to each peer).  Returns the two HOST lists (send, recv)
    -- the one synchronisation point of a variable-size exchange -- and the LARGEST per-peer row count anywhere in the group
    (every rank's own maximum travels in the same all-to-all as a second column): the number all ranks must agree on before
    they decide how to cut an oversized payload (all_to_all_rows)."""
    import torch
    import torch.distributed as dist
    staged = _host_staged(send_counts, group)
    src = send_counts.cpu() if staged else send_counts
    pair = torch.stack([src, src.max().expand_as(src)], dim=1).contiguous()      # [world, 2]: (rows for peer p, this rank's maximum)
    got = torch.empty_like(pair)
    dist.all_to_all_single(got, pair, group=group)
    h_send, h_got = src.cpu().tolist(), got.cpu()
    return h_send, h_got[:, 0].tolist(), int(h_got[:, 1].max())


def exchange_counts(send_counts, group=None):
    """(send, recv) of exchange_counts_max."""
    send, recv, _ = exchange_counts_max(send_counts, group)
    return send, recv


_A2A_MAX_BYTES = 1 << 30      # per-peer message size above which all_to_all_rows slices the payload (see there)


def all_to_all_rows(out, inp, recv_counts, send_counts, group=None, max_rows=None):
    """Variable-size all-to-all along dim 0: rows [sum(send[:p]), sum(send[:p+1])) of `inp` go to peer p; `out` receives
    sum(recv_counts) rows, grouped by source rank.  Row payloads of any width / dtype.  max_rows: the largest per-peer row count in
    the whole group (exchange_counts_max) -- with more than one rank every rank must pass it, so that all of them cut an oversized
    payload into the same number of slices."""
    import torch
    import torch.distributed as dist
    n_out, n_in = int(sum(recv_counts)), int(sum(send_counts))
    if out.shape[0] < n_out or inp.shape[0] < n_in:
        raise ValueError("all_to_all_rows: buffers smaller than the split sizes")
    o, i = out[:n_out], inp[:n_in]
    if _host_staged(inp, group):
        h_out = torch.empty(o.shape, dtype=o.dtype)
        dist.all_to_all_single(h_out, i.cpu().contiguous(), list(recv_counts), list(send_counts), group=group)
        o.copy_(h_out)
    else:
        # Messages of 2 GB and more come back half-copied from all_to_all_single on this stack (torch 2.10 / RCCL 2.26, seen with a
        # one-rank group whose whole payload is a self-send: tests/test_gpu_dp.py pins it): beyond 1 GiB per peer the payload goes
        # in column slices, each exchanged on its own.
        row_bytes = (inp[0].numel() if inp.shape[0] else (out[0].numel() if out.shape[0] else 0)) * inp.element_size()
        worst = (int(max_rows) if max_rows is not None else max([0] + [int(c) for c in send_counts] + [int(c) for c in recv_counts])) * row_bytes
        parts = -(-worst // _A2A_MAX_BYTES) if worst else 1
        if parts <= 1 or i.dim() != 2:
            dist.all_to_all_single(o, i, list(recv_counts), list(send_counts), group=group)
        else:
            width = i.shape[1]
            step = -(-width // parts)
            for c0 in range(0, width, step):
                c1 = min(c0 + step, width)
                part = torch.empty((n_out, c1 - c0), dtype=o.dtype, device=o.device)
                dist.all_to_all_single(part, i[:, c0:c1].contiguous(), list(recv_counts), list(send_counts), group=group)
                o[:, c0:c1] = part
    return o


class StreamRccl:
    """RCCL collectives enqueued on the CALLER'S stream (the one the engine's kernels run on), through a communicator of our own.
    torch.distributed's NCCL backend runs every collective on a stream of the process group and orders it against the current
    stream with events: for the dense data-parallel step (kernels -> reduce-scatter -> kernel -> all-gather -> kernels) that is
    four stream hand-offs per step, measured as ~45 us of idle GPU on a 0.28 ms step with a one-rank group.  Enqueued in order on
    one stream the hand-offs disappear.  The library is the librccl.so that torch itself loaded (no second RCCL in the process);
    the communicator is bootstrapped by broadcasting ncclUniqueId through the torch process group.  "nccl" backend only."""

    def __init__(self, group=None):
        import ctypes
        import os
        import torch
        import torch.distributed as dist
        if dist.get_backend(group) != "nccl":
            raise RuntimeError("StreamRccl needs the nccl (RCCL) backend")
        self.lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

        class UniqueId(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_char * 128)]
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().tolist())
        ctypes.memmove(ctypes.byref(uid), raw, 128)
        self.comm = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        self.lib.ncclReduceScatter.argtypes = [vp, vp, sz, ci, ci, vp, vp]
        self.lib.ncclAllGather.argtypes = [vp, vp, sz, ci, vp, vp]
        self.lib.ncclGetErrorString.restype = ctypes.c_char_p
        self._dt = {torch.int32: 2, torch.float32: 7}      # ncclInt32, ncclFloat32 (rccl.h ncclDataType_t)

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.ncclGetErrorString(rc)
            raise RuntimeError("%s failed: %s" % (what, msg.decode() if msg else rc))

    def reduce_scatter_sum(self, out, inp, stream):
        if inp.numel() != out.numel() * self.world or out.dtype != inp.dtype:
            raise ValueError("StreamRccl.reduce_scatter_sum: input must hold world_size equal chunks of the output's type")
        self._check(self.lib.ncclReduceScatter(inp.data_ptr(), out.data_ptr(), out.numel(), self._dt[out.dtype], 0, self.comm, stream),
                    "ncclReduceScatter")

    def all_gather_chunks(self, out, inp, stream):
        """`inp` may be this rank's slice of `out` (NCCL's in-place form: sendbuff == recvbuff + rank * count)."""
        if out.numel() != inp.numel() * self.world or out.dtype != inp.dtype:
            raise ValueError("StreamRccl.all_gather_chunks: output must hold world_size equal chunks of the input's type")
        self._check(self.lib.ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), self._dt[inp.dtype], self.comm, stream), "ncclAllGather")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
