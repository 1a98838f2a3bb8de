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
    from openkeonspark_amd.parallel import thread_range, slice_positions, allreduce_sum as allreduce_gradients
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


def _collectives_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openkeonspark_amd import parallel as par
    # reduce-scatter of an int32 image: rank g ends with the column sums of ITS chunk
    chunk = 6
    img = (torch.arange(world * chunk, dtype=torch.int32) + 1) * (rank + 1)
    own = torch.zeros(chunk, dtype=torch.int32)
    par.reduce_scatter_sum(own, img)
    want = (torch.arange(world * chunk, dtype=torch.int32) + 1)[rank * chunk:(rank + 1) * chunk] * sum(range(1, world + 1))
    assert torch.equal(own, want)
    # all-gather in place: every rank's chunk of the flat buffer reaches every rank
    flat = torch.full((world * chunk,), -1.0)
    flat[rank * chunk:(rank + 1) * chunk] = torch.arange(chunk, dtype=torch.float32) + 100 * rank
    par.all_gather_chunks(flat, flat[rank * chunk:(rank + 1) * chunk])
    assert torch.equal(flat, torch.cat([torch.arange(chunk, dtype=torch.float32) + 100 * g for g in range(world)]))
    # variable-size all-to-all of rows (ragged, including empty sends): rank r sends (r + p) % 3 rows to peer p
    send = [(rank + p) % 3 for p in range(world)]
    s_counts, r_counts, g_max = par.exchange_counts_max(torch.tensor(send, dtype=torch.int32))
    assert s_counts == send and r_counts == [(p + rank) % 3 for p in range(world)]
    assert g_max == max((r + p) % 3 for r in range(world) for p in range(world))          # the same number on every rank
    rows = torch.tensor([[rank, p, i] for p in range(world) for i in range(send[p])], dtype=torch.int32).reshape(-1, 3)
    out = torch.full((sum(r_counts) + 2, 3), -7, dtype=torch.int32)
    got = par.all_to_all_rows(out, rows, r_counts, s_counts)
    want = torch.tensor([[p, rank, i] for p in range(world) for i in range(r_counts[p])], dtype=torch.int32).reshape(-1, 3)
    assert torch.equal(got, want) and int(out[sum(r_counts):].max()) == -7
    # the same exchange with the per-peer message limit forced below one row block: the payload goes in column slices
    # (all_to_all_rows' guard against the 2 GB fault of all_to_all_single, tests/test_gpu_dp.py) and must arrive identical
    wide = torch.stack([rows[:, 0] * 1000 + rows[:, 1] * 10 + rows[:, 2] + c for c in range(7)], dim=1).to(torch.float32).reshape(-1, 7)
    out_w = torch.full((sum(r_counts) + 1, 7), -7.0)
    limit, par._A2A_MAX_BYTES = par._A2A_MAX_BYTES, 8
    try:
        got_w = par.all_to_all_rows(out_w, wide, r_counts, s_counts, max_rows=g_max)
    finally:
        par._A2A_MAX_BYTES = limit
    want_w = torch.stack([want[:, 0] * 1000 + want[:, 1] * 10 + want[:, 2] + c for c in range(7)], dim=1).to(torch.float32).reshape(-1, 7)
    assert torch.equal(got_w, want_w) and float(out_w[sum(r_counts):].max()) == -7.0
    assert par.chunk_size(10, 4) == 3 and par.chunk_size(10, 4, 4) == 4 and par.chunk_size(8, 4, 1) == 2
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_helpers_on_gloo(tmp_path, world):
    """reduce-scatter / all-gather (dense owner-computes update) and the variable-size all-to-all (sharded sparse path)."""
    port = 31500 + os.getpid() % 2000 + world
    mp.start_processes(_collectives_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    assert all(os.path.exists(str(tmp_path / ("ok%d" % r))) for r in range(world))


def test_bench_self_launch_reports_a_failing_rank(tmp_path):
    """bench.py --gpus 2 from a bare shell starts its ranks itself (bench.launch_ranks; the reference's launcher:
    /root/reference/main_spark.py:340).  Without a GPU every rank fails at its first device call: the parent must exit non-zero
    and print no result line -- a failed N-rank job never looks like a measurement."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check (on a GPU box tests/test_gpu_dp.py runs the real thing)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    # --gpus must divide the 8 virtual sampler threads: refused before anything is started
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode != 0 and b"must divide" in p.stderr
