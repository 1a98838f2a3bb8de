"""Data-parallel path end to end on real kernels: two ranks (sharing the one GPU of the test box,
`gloo` so that no second device is needed) against the single-process run of the same global batch.
The driver's 8-GPU run uses the same code with backend "nccl" (RCCL)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, model_name, opt, sparse=False, prefetch=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.set_in_path(os.path.join(GOLDEN, "kg_small"))
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(48); con.set_nbatches(10)  # B = 600
    con.set_ent_neg_rate(3); con.set_alpha(0.02); con.set_opt_method(opt)
    con.sparse_rows = sparse
    con.prefetch_sampling = prefetch   # (data-parallel default: on; then the rng states run one batch ahead)
    con.counts_min_records = 0
    con.init()
    con.set_model_and_session(getattr(pkg, model_name))
    assert con.sparse_rows == sparse
    if world > 1:
        con.init_distributed()
    losses = [con.train_step() for _ in range(4)]
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "w%d_r%d.npz" % (world, rank)), losses=np.array(losses),
             states=con.get_stream_states(), **con.get_parameters())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("model_name,opt", [("TransE", "SGD"), ("TransE", "Adam"), ("TransH", "SGD")])
def test_two_ranks_equal_single_process(tmp_path, model_name, opt):
    import torch.multiprocessing as mp
    port = 29600 + os.getpid() % 1000
    mp.start_processes(_worker, args=(1, port, str(tmp_path), model_name, opt), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(2, port + 1, str(tmp_path), model_name, opt), nprocs=2, join=True, start_method="spawn")
    one = np.load(str(tmp_path / "w1_r0.npz"))
    r0 = np.load(str(tmp_path / "w2_r0.npz"))
    r1 = np.load(str(tmp_path / "w2_r1.npz"))
    assert np.array_equal(r0["states"], one["states"]) and np.array_equal(r1["states"], one["states"])
    assert np.allclose(r0["losses"], one["losses"], rtol=2e-5, atol=0)
    for k in one.files:
        if k in ("losses", "states"):
            continue
        assert np.array_equal(r0[k], r1[k]), k  # replicas apply the identical all-reduced update
        tol = 2e-5 if opt == "SGD" else 2e-4
        assert np.abs(r0[k] - one[k]).max() <= tol * np.abs(one[k]).max(), k


def test_two_ranks_sparse_record_exchange(tmp_path):
    """Sparse-row mode: the ranks all-gather their int8 records; integer sums make the replicas' tables equal
    to the single-process tables bit for bit."""
    import torch.multiprocessing as mp
    port = 29700 + os.getpid() % 1000
    mp.start_processes(_worker, args=(1, port, str(tmp_path), "TransE", "SGD", True), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(2, port + 1, str(tmp_path), "TransE", "SGD", True), nprocs=2, join=True, start_method="spawn")
    one = np.load(str(tmp_path / "w1_r0.npz"))
    r0 = np.load(str(tmp_path / "w2_r0.npz"))
    r1 = np.load(str(tmp_path / "w2_r1.npz"))
    assert np.array_equal(r0["states"], one["states"]) and np.array_equal(r1["states"], one["states"])
    assert np.allclose(r0["losses"], one["losses"], rtol=2e-5, atol=0)
    for k in one.files:
        if k in ("losses", "states"):
            continue
        assert np.array_equal(r0[k], r1[k]), k
        assert np.array_equal(r0[k], one[k]), k


def test_two_ranks_with_prefetched_sampling(tmp_path):
    """The data-parallel default draws batch i+1 during step i's all-reduce: same tables as without."""
    import torch.multiprocessing as mp
    port = 29800 + os.getpid() % 1000
    mp.start_processes(_worker, args=(1, port, str(tmp_path), "TransE", "Adam", False, False), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(2, port + 1, str(tmp_path), "TransE", "Adam", False, None), nprocs=2, join=True, start_method="spawn")
    one = np.load(str(tmp_path / "w1_r0.npz"))
    r0 = np.load(str(tmp_path / "w2_r0.npz"))
    r1 = np.load(str(tmp_path / "w2_r1.npz"))
    assert np.allclose(r0["losses"], one["losses"], rtol=2e-5, atol=0)
    for k in one.files:
        if k in ("losses", "states"):
            continue
        assert np.array_equal(r0[k], r1[k]), k
        assert np.array_equal(r0[k], one[k]), k    # integer counts: bit-identical to the single-process run
