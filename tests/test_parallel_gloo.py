"""The N>1 path on CPU: world_size-2 `gloo` processes.

What is exercised without a GPU: the rank -> virtual-thread -> batch-slice partition (the library's
host function and its Python mirror), the gradient all-reduce helper, and the data-parallel
arithmetic itself -- each rank differentiates ITS slice of the same global batch with the GLOBAL
mean denominator (oracle, test infrastructure) and the SUM all-reduce must reproduce the
single-process gradient and loss."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _worker(rank, world, port, kg_dir, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from openkeonspark_amd.parallel import thread_range, slice_positions, allreduce_gradients
    W, B, n, D = 8, 203, 3, 32   # 203 % 8 != 0: ragged slices
    kg = oracle.KG(kg_dir, work_threads=W, bern=1)
    bh, bt, br, _ = kg.sampling(B, n, 0)      # every rank draws the same global batch (same seeds)
    lo, hi = thread_range(rank, world, W)
    first, cnt = slice_positions(B, W, lo, hi)
    # this rank's slice in the [positives | negatives round k] layout
    idx = np.concatenate([np.arange(first, first + cnt) + k * B for k in range(1 + n)])
    model = oracle.Model("transe", kg.entTotal, kg.relTotal, D, seed=1)
    loss, g = model.grad(bh[idx], bt[idx], br[idx], cnt, n, denom=B * n)
    tensors = [torch.from_numpy(g[k]) for k in model.names] + [torch.tensor([loss], dtype=torch.float32)]
    allreduce_gradients(tensors)
    if rank == 0:
        full_loss, full_g = model.grad(bh, bt, br, B, n)
        np.savez(os.path.join(out_dir, "res.npz"), loss=tensors[-1].numpy(), full_loss=full_loss,
                 **{"g_" + k: tensors[i].numpy() for i, k in enumerate(model.names)},
                 **{"f_" + k: full_g[k] for k in model.names})
    counts = torch.tensor([cnt]); dist.all_reduce(counts)
    assert int(counts) == B
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_single_process(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.start_processes(_worker, args=(2, port, os.path.join(GOLDEN, "kg_small"), str(tmp_path)), nprocs=2,
                       join=True, start_method="spawn")
    z = np.load(str(tmp_path / "res.npz"))
    assert abs(float(z["loss"][0]) - float(z["full_loss"])) <= 1e-5 * abs(float(z["full_loss"]))
    for k in ("ent_embeddings", "rel_embeddings"):
        a, b = z["g_" + k], z["f_" + k]
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


def test_thread_range_requires_divisibility():
    from openkeonspark_amd.parallel import thread_range
    assert thread_range(1, 2, 8) == (4, 8)
    assert thread_range(7, 8, 8) == (7, 8)
    with pytest.raises(ValueError):
        thread_range(0, 3, 8)


def _records_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openkeonspark_amd import _lib
    from openkeonspark_amd.parallel import allgather_records, max_slice_positions, thread_range, slice_positions
    lib = _lib.load()
    W, B, dw = 6, 100, 4            # 100 % 6 != 0: ragged slices, ranks own 51 and 49 positions
    lib.setWorkThreads(W)
    m = max_slice_positions(lib, B, world, W)
    lo, hi = thread_range(rank, world, W)
    first, cnt = slice_positions(B, W, lo, hi)
    assert cnt <= m
    rec = torch.full((m, dw), -1, dtype=torch.int32)
    dst = torch.full((m,), -1, dtype=torch.int32)
    dst[:cnt] = torch.arange(first, first + cnt, dtype=torch.int32)   # record i of the global batch
    rec[:cnt] = dst[:cnt, None] * 10 + torch.arange(dw, dtype=torch.int32)
    rec_all = torch.empty((m * world, dw), dtype=torch.int32)
    dst_all = torch.empty(m * world, dtype=torch.int32)
    allgather_records(rec, dst, rec_all, dst_all)
    np.savez(os.path.join(out_dir, "rec%d.npz" % rank), rec=rec_all.numpy(), dst=dst_all.numpy(), m=m)
    dist.destroy_process_group()


def test_record_allgather_covers_every_position_once(tmp_path):
    """The sparse exchange: equal-sized padded record blocks, every global batch position exactly once."""
    port = 31500 + os.getpid() % 2000
    mp.start_processes(_records_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    z0, z1 = np.load(str(tmp_path / "rec0.npz")), np.load(str(tmp_path / "rec1.npz"))
    assert int(z0["m"]) == 51
    assert np.array_equal(z0["rec"], z1["rec"]) and np.array_equal(z0["dst"], z1["dst"])
    live = z0["dst"] >= 0
    assert sorted(z0["dst"][live].tolist()) == list(range(100))
    assert np.array_equal(z0["rec"][live], z0["dst"][live][:, None] * 10 + np.arange(4))
