"""Data-parallel path end to end on real kernels: two ranks (sharing the one GPU of the test box,
`gloo` so that no second device is needed) against the single-process run of the same global batch.
The driver's 8-GPU run uses the same code with backend "nccl" (RCCL)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_dir, model_name, opt, sparse=False, prefetch=False, nbatches=10, pieces=0, rccl_one_rank=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif rccl_one_rank:     # a ONE-rank RCCL group: the collectives of the data-parallel step on device memory, as the 8-GPU run issues them
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    import openkeonspark_amd as pkg
    if sparse:   # the sharded step emits against a per-step row cache: norms from the gathered rows in both runs
        pkg._lib.lib().kge_set_option(b"inv_table_max_bytes", 0)
    if os.environ.get("KGE_TEST_FORCE_PAIR_PATH") == "1":   # TransH / TransD: every step through the pair-count path
        pkg._lib.lib().kge_set_option(b"float_records_min", 0)
        pkg._lib.lib().kge_set_option(b"pair_counts_min_neg", 1)
    con = pkg.Config()
    con.set_in_path(os.path.join(GOLDEN, "kg_small"))
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(48); con.set_nbatches(nbatches)  # B = 600 by default
    con.set_ent_neg_rate(3); con.set_alpha(0.02); con.set_opt_method(opt)
    con.sparse_rows = sparse
    con.prefetch_sampling = prefetch   # (data-parallel default: on; then the rng states run one batch ahead)
    con.counts_min_records = 0
    if pieces:
        con.dp_pieces = pieces      # the flat buffers in `pieces` segments, each exchanged and updated on its own (Config._dp_exchange)
    con.init()
    con.set_model_and_session(getattr(pkg, model_name))
    assert (con.sparse_rows or con.sparse_inplace) == sparse      # (TransH / TransD: the row-wise SGD in place from float records)
    if rccl_one_rank:
        con.force_data_parallel = True
    if world > 1 or rccl_one_rank:
        con.init_distributed()
        if pieces and not sparse:
            assert con._pieces == pieces
    if rccl_one_rank:
        assert con._dp and dist.get_backend() == "nccl" and (sparse or hasattr(con, "_flat_p"))
        if not sparse and not pieces:       # one piece: the collectives run on the engine's own stream (parallel.StreamRccl)
            assert con._stream_rccl() is not None
    losses = [con.train_step() for _ in range(4)]
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "w%d_r%d%s.npz" % (world, rank, "_rccl" if rccl_one_rank else "")), losses=np.array(losses),
             states=con.get_stream_states(), **con.get_parameters())
    if world > 1 or rccl_one_rank:
        dist.barrier()
        dist.destroy_process_group()


def _run_worlds(tmp_path, worlds, *args):
    import torch.multiprocessing as mp
    port = 29600 + os.getpid() % 1000
    for i, w in enumerate(worlds):
        mp.start_processes(_worker, args=(w, port + i, str(tmp_path)) + args, nprocs=w, join=True, start_method="spawn")
    return {w: [np.load(str(tmp_path / ("w%d_r%d.npz" % (w, r)))) for r in range(w)] for w in worlds}


@pytest.mark.parametrize("model_name,opt,world", [("TransE", "SGD", 2), ("TransE", "Adam", 2), ("TransE", "Adam", 4), ("TransH", "SGD", 2),
                                                  ("TransD", "Adam", 4), ("TransR", "SGD", 2)])
def test_ranks_equal_single_process(tmp_path, model_name, opt, world):
    _ranks_equal_single_process(tmp_path, model_name, opt, world, 0)


@pytest.mark.parametrize("model_name,opt,world", [("TransE", "Adam", 2), ("TransH", "SGD", 2), ("TransD", "Adam", 4)])
def test_ranks_equal_single_process_in_two_pieces(tmp_path, model_name, opt, world):
    """The same with the flat buffers cut into TWO pieces (what tables from 64 MB on get): each piece is reduce-scattered,
    updated by its owners and all-gathered on its own, the loss riding in the LAST piece's spare tail slot."""
    _ranks_equal_single_process(tmp_path, model_name, opt, world, 2)


def _ranks_equal_single_process(tmp_path, model_name, opt, world, pieces):
    """Reduce-scatter of the gradient image, optimizer on the owned chunk, all-gather of the parameters: every replica
    holds the identical tables (one owner computes each element) and they equal the single-process run -- bit for bit
    for TransE (integer counts), to fp32 summation order for the fp32-accumulator models."""
    res = _run_worlds(tmp_path, [1, world], model_name, opt, False, False, 10, pieces)
    one = res[1][0]
    for r in res[world]:
        assert np.array_equal(r["states"], one["states"])
        assert np.allclose(r["losses"], one["losses"], rtol=2e-5, atol=0)
        assert np.array_equal(r["losses"], res[world][0]["losses"])      # the loss reaches every rank through the all-gather: same bits
    for k in one.files:
        if k in ("losses", "states"):
            continue
        for r in res[world][1:]:
            assert np.array_equal(res[world][0][k], r[k]), k
        if model_name == "TransE":
            assert np.array_equal(res[world][0][k], one[k]), k
        else:
            tol = 2e-5 if opt == "SGD" else 2e-4
            assert np.abs(res[world][0][k] - one[k]).max() <= tol * np.abs(one[k]).max(), k


@pytest.mark.parametrize("model_name,opt", [("TransH", "SGD"), ("TransD", "Adam")])
def test_ranks_equal_single_process_on_the_pair_count_path(tmp_path, model_name, opt, monkeypatch):
    """The same equality with every rank's forward/backward going through the pair-count path (csrc/pairs.hip) on its slice of
    the batch: the dense gradient image it fills is what the reduce-scatter exchanges."""
    monkeypatch.setenv("KGE_TEST_FORCE_PAIR_PATH", "1")
    res = _run_worlds(tmp_path, [1, 2], model_name, opt)
    one = res[1][0]
    for r in res[2]:
        assert np.array_equal(r["states"], one["states"])
        assert np.allclose(r["losses"], one["losses"], rtol=2e-5, atol=0)
    for k in one.files:
        if k in ("losses", "states"):
            continue
        assert np.array_equal(res[2][0][k], res[2][1][k]), k
        tol = 2e-5 if opt == "SGD" else 2e-4
        assert np.abs(res[2][0][k] - one[k]).max() <= tol * np.abs(one[k]).max(), k


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharded_sparse_table(tmp_path, world):
    """Sparse-row mode on N ranks: the entity table is SHARDED by row range, rows and int8 records travel by all-to-all
    to / from their owners, the relation counts are all-reduced.  Integer sums and one per-row update formula: the union
    of the shards equals the single-process tables bit for bit, the replicated relation table too."""
    res = _run_worlds(tmp_path, [1, world], "TransE", "SGD", True)
    one = res[1][0]
    for r in res[world]:
        assert np.array_equal(r["states"], one["states"])
        assert np.allclose(r["losses"], one["losses"], rtol=2e-5, atol=0)
        for k in one.files:
            if k not in ("losses", "states"):
                assert np.array_equal(r[k], one[k]), k


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharded_lazy_adam(tmp_path, world):
    """opt_method "LazyAdam" (touched rows only; opt-in, non-parity against the reference's dense TF1 Adam) on N ranks: the entity
    rows AND their two moment rows are sharded by row range (owner computes), the relation table and its moments stay replicated,
    and a relation no rank had a record for keeps its row and moments.  Integer sums and one per-row update function: the union of
    the shards after four steps equals the single-process table bit for bit (a wrong moment shard would show from the second step
    on).  Reference for the optimizer choice: distribute_training.py:95-98."""
    res = _run_worlds(tmp_path, [1, world], "TransE", "LazyAdam", True)
    one = res[1][0]
    for r in res[world]:
        assert np.array_equal(r["states"], one["states"])
        assert np.allclose(r["losses"], one["losses"], rtol=2e-5, atol=0)
        for k in one.files:
            if k not in ("losses", "states"):
                assert np.array_equal(r[k], one[k]), k


@pytest.mark.parametrize("model_name,world", [("TransH", 2), ("TransH", 4), ("TransD", 2)])
def test_ranks_row_wise_sgd_from_gathered_records(tmp_path, model_name, world):
    """sparse_rows with TransH / TransD on N ranks (Config._records_step): every rank turns its slice of the batch into float
    gradient records, the records are all-gathered and every rank adds -lr * the per-row sums of ALL of them to its replica.
    Every rank reduces the same records in the same order: the replicas are bit-identical; against the single-process step
    (kge_forward_backward_sgd_rows, the same records in batch order) the per-row sums differ in fp32 order only.  A rank whose
    slice is shorter than the others' fills the rest of its slice with keyless records (B = 600 over 8 virtual threads: equal
    slices at 2 and 4 ranks; the empty-slice case is test_rank_with_an_empty_slice's)."""
    res = _run_worlds(tmp_path, [1, world], model_name, "SGD", True)
    one = res[1][0]
    for r in res[world]:
        assert np.array_equal(r["states"], one["states"])
        assert np.allclose(r["losses"], one["losses"], rtol=2e-5, atol=0)
        assert np.array_equal(r["losses"], res[world][0]["losses"])
    for k in one.files:
        if k in ("losses", "states"):
            continue
        for r in res[world][1:]:
            assert np.array_equal(res[world][0][k], r[k]), k
        assert np.abs(res[world][0][k] - one[k]).max() <= 2e-5 * np.abs(one[k]).max(), k


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


def _lp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import json
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.set_in_path(os.path.join(GOLDEN, "kg_small"))
    con.set_work_threads(2); con.set_dimension(32); con.set_test_link_prediction(True)
    con.init()
    torch.manual_seed(0)
    con.set_model_and_session(pkg.TransH)
    for t in con._tables:
        t.mul_(3.0)
    if world > 1:
        con.init_distributed()
    met = con.link_prediction_distributed(test_head=True)
    if world == 1:
        _, ref = con.link_prediction(test_head=True)
        assert ref.keys() == met.keys()
        for k in ref:
            assert abs(ref[k] - met[k]) <= 1e-12 * max(1.0, abs(ref[k])), k
    json.dump(met, open(os.path.join(out_dir, "lp_w%d_r%d.json" % (world, rank)), "w"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_link_prediction_split_over_ranks(tmp_path):
    """Static range split of the test set over ranks + all-reduced accumulators (distribute_training.py:430-441,
    main_spark.py:430-448): every rank reports the single-process metrics."""
    import json
    import torch.multiprocessing as mp
    port = 29900 + os.getpid() % 1000
    mp.start_processes(_lp_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_lp_worker, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    one = json.load(open(str(tmp_path / "lp_w1_r0.json")))
    for r in (0, 1):
        two = json.load(open(str(tmp_path / ("lp_w2_r%d.json" % r))))
        assert one.keys() == two.keys() and len(one) == 40
        for k in one:
            assert abs(one[k] - two[k]) <= 1e-12 * max(1.0, abs(one[k])), k
    assert one["r_filter_rank"] >= 1.0 and one["l_rank"] >= one["l_filter_rank"]


@pytest.mark.parametrize("model_name,sparse", [("TransE", False), ("TransE", True), ("TransH", True)])
def test_rank_with_an_empty_slice(tmp_path, model_name, sparse):
    """B = 3 positions over 8 virtual threads: rank 1 (threads 4..7) owns NO position of any batch.  It must still
    advance the rng streams, join the exchange and apply the same update (ragged / empty inputs of the reference's
    slice rule, Base.cpp:85-92).  TransH with sparse rows: its whole slice of the gathered records is keyless."""
    import torch.multiprocessing as mp
    port = 30100 + os.getpid() % 1000
    args1 = (1, port, str(tmp_path), model_name, "SGD", sparse, False, 2000)
    args2 = (2, port + 1, str(tmp_path), model_name, "SGD", sparse, False, 2000)
    mp.start_processes(_worker, args=args1, nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=args2, nprocs=2, join=True, start_method="spawn")
    one = np.load(str(tmp_path / "w1_r0.npz"))
    r0 = np.load(str(tmp_path / "w2_r0.npz"))
    r1 = np.load(str(tmp_path / "w2_r1.npz"))
    assert np.array_equal(r0["states"], one["states"]) and np.array_equal(r1["states"], one["states"])
    assert np.allclose(r0["losses"], one["losses"], rtol=2e-5, atol=0)
    for k in one.files:
        if k in ("losses", "states"):
            continue
        assert np.array_equal(r0[k], r1[k]), k
        if model_name == "TransE":
            assert np.array_equal(r0[k], one[k]), k
        else:      # (rank 0 holds every record, in the single-process order: the same sums)
            assert np.abs(r0[k] - one[k]).max() <= 2e-5 * np.abs(one[k]).max(), k


def _driver_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "KGE_SINGLE_DEVICE": "1", "KGE_DIST_BACKEND": "gloo",
                       "KGE_COUNTS_MIN_RECORDS": "0"})
    import torch.distributed as dist
    from openkeonspark_amd import distribute_training as dt
    out = os.path.join(out_dir, "run_w%d" % world)
    args = ["--input_path", os.path.join(GOLDEN, "kg_small"), "--output_path", out, "--embedding_dimension", "32",
            "--n_mini_batches", "4", "--ent_neg_rate", "2", "--alpha", "0.01", "--optimizer", "SGD", "--bern_flag", "1",
            "--train_times", "3", "--work_threads", "4"]
    con = dt.main_fun(dt.parse_args(args))
    np.savez(os.path.join(out_dir, "drv_w%d_r%d.npz" % (world, rank)), step=con.global_step, **con.get_parameters())
    # (main_fun made and destroyed its own process group; the second call makes another one: on a port of its own -- a rendezvous
    # on the port the first group's store is still closing was refused once in a soak run, "Connection reset by peer")
    os.environ["MASTER_PORT"] = str(port + 500)
    met = dt.main_fun(dt.parse_args(args + ["--mode", "test"]))
    import json
    json.dump(met, open(os.path.join(out_dir, "drvlp_w%d_r%d.json" % (world, rank)), "w"))
    if world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def test_cli_driver_under_two_ranks(tmp_path):
    """distribute_training.main_fun launched as two torchrun-style ranks: same parameters as one process, one checkpoint
    per epoch written by rank 0 only, and the test mode's link prediction split over the ranks."""
    import json
    import torch.multiprocessing as mp
    port = 30300 + os.getpid() % 1000
    mp.start_processes(_driver_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_driver_worker, args=(2, port + 1, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    one = np.load(str(tmp_path / "drv_w1_r0.npz"))
    r0 = np.load(str(tmp_path / "drv_w2_r0.npz"))
    r1 = np.load(str(tmp_path / "drv_w2_r1.npz"))
    assert int(one["step"]) == int(r0["step"]) == int(r1["step"]) == 12
    for k in one.files:
        assert np.array_equal(r0[k], r1[k]), k
        assert np.array_equal(r0[k], one[k]), k
    ck1 = sorted(f for f in os.listdir(str(tmp_path / "run_w1")) if f.startswith("model.ckpt"))
    ck2 = sorted(f for f in os.listdir(str(tmp_path / "run_w2")) if f.startswith("model.ckpt"))
    assert ck1 == ck2 and len(ck1) >= 1
    m1 = json.load(open(str(tmp_path / "drvlp_w1_r0.json")))
    m2 = json.load(open(str(tmp_path / "drvlp_w2_r1.json")))
    assert m1.keys() == m2.keys()
    for k in m1:
        assert abs(m1[k] - m2[k]) <= 1e-12 * max(1.0, abs(m1[k])), k


def _resume_worker(rank, world, port, out_dir, run, train_times, sparse=False):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank), "KGE_SINGLE_DEVICE": "1", "KGE_DIST_BACKEND": "gloo",
                       "KGE_COUNTS_MIN_RECORDS": "0"})
    import torch.distributed as dist
    from openkeonspark_amd import distribute_training as dt
    args = ["--input_path", os.path.join(GOLDEN, "kg_small"), "--output_path", os.path.join(out_dir, run),
            "--embedding_dimension", "32", "--n_mini_batches", "5", "--ent_neg_rate", "3", "--alpha", "0.01",
            "--optimizer", "Adam", "--bern_flag", "1", "--train_times", str(train_times)]
    if sparse:      # the table-sharded sparse mode: SGD, per-rank shard files beside the checkpoint
        args[args.index("--optimizer") + 1] = "SGD"
        args += ["--sparse_rows", "1"]
        from openkeonspark_amd import _lib
        _lib.lib().kge_set_option(b"inv_table_max_bytes", 0)
    if sparse:      # gathering the sharded table afterwards is a collective: keep the process group beyond main_fun
        dist.init_process_group("gloo", rank=rank, world_size=world)
    con = dt.main_fun(dt.parse_args(args))
    assert con.prefetch_sampling                      # the data-parallel default: batch i+1 is drawn during step i
    assert (not sparse) or (con.sparse_rows and con._tables[0].shape[0] == 500)
    np.savez(os.path.join(out_dir, "%s_t%d_r%d.npz" % (run, train_times, rank)), step=con.global_step, **con.get_parameters())
    if sparse:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_resume_equals_uninterrupted_run(tmp_path):
    """Checkpoint / resume under data parallelism with sampling one step ahead: the checkpoint stores the rng states the
    NEXT batch starts from (the prefetched batch is rewound), rank 0 reads it and broadcasts it, and the resumed run ends
    with the tables of the uninterrupted one, bit for bit."""
    import torch.multiprocessing as mp
    port = 30500 + os.getpid() % 1000
    for i, (run, times) in enumerate((("full", 4), ("split", 2), ("split", 2))):
        mp.start_processes(_resume_worker, args=(2, port + i, str(tmp_path), run, times), nprocs=2, join=True, start_method="spawn")
        if run == "split" and i == 1:
            os.rename(str(tmp_path / "split_t2_r0.npz"), str(tmp_path / "split_first_r0.npz"))
    full = np.load(str(tmp_path / "full_t4_r0.npz"))
    first = np.load(str(tmp_path / "split_first_r0.npz"))
    second = np.load(str(tmp_path / "split_t2_r0.npz"))
    assert int(first["step"]) == 10 and int(second["step"]) == 20 and int(full["step"]) == 20
    for k in full.files:
        assert np.array_equal(second[k], full[k]), k


def _config5_worker(rank, world, port, out_dir):
    """configs[4]'s regime at a reduced entity count: 200 k entities x dim 512 (0.4 GB table), B = 50 000, n = 1, sparse rows."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import openkeonspark_amd as pkg
    pkg._lib.lib().kge_set_option(b"inv_table_max_bytes", 0)
    rng = np.random.default_rng(5)
    E, R, n_tr = 200_000, 500, 1_000_000
    h = rng.integers(0, E, n_tr); t = rng.integers(0, E, n_tr); r = rng.integers(0, R, n_tr)
    con = pkg.Config()
    con.prefetch_sampling = False
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(512); con.set_ent_neg_rate(1); con.set_alpha(0.5)
    con.set_opt_method("SGD"); con.set_nbatches(20)
    con.sparse_rows = True
    con.init_from_arrays(E, R, h, t, r)
    con.set_model_and_session(pkg.TransE)
    if world > 1:
        con.init_distributed()
        assert con._tables[0].shape[0] == E // world          # the entity table is sharded, not replicated
    losses = [con.train_step() for _ in range(3)]
    ent = con.get_parameters_by_name("ent_embeddings")        # collective: gathers the shards
    rel = con.get_parameters_by_name("rel_embeddings")
    torch.cuda.synchronize()
    if rank == 0:
        # a digest instead of the 0.4 GB table: row sums in float64 are order-free and see every element
        np.savez(os.path.join(out_dir, "c5_w%d.npz" % world), losses=np.array(losses), ent_rowsum=ent.astype(np.float64).sum(1),
                 ent_sample=ent[::997].copy(), rel=rel, states=con.get_stream_states())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_config5_regime_two_ranks_sharded_equals_one_rank(tmp_path):
    """The table-sharded path at configs[4]'s shape (dim 512, uniform popularity, one negative): 2 ranks, each holding half of the
    entity rows and exchanging rows / int8 records by all-to-all, end with the single-process tables bit for bit (the one-rank
    step itself is checked against the oracle at this size in tests/test_gpu_configs.py)."""
    import torch.multiprocessing as mp
    port = 30700 + os.getpid() % 1000
    for i, w in enumerate((1, 2)):
        mp.start_processes(_config5_worker, args=(w, port + i, str(tmp_path)), nprocs=w, join=True, start_method="spawn")
    one, two = np.load(str(tmp_path / "c5_w1.npz")), np.load(str(tmp_path / "c5_w2.npz"))
    assert np.array_equal(one["states"], two["states"])
    assert np.allclose(one["losses"], two["losses"], rtol=2e-5, atol=0)
    assert np.array_equal(one["ent_rowsum"], two["ent_rowsum"]) and np.array_equal(one["ent_sample"], two["ent_sample"])
    assert np.array_equal(one["rel"], two["rel"])


def test_two_rank_sharded_checkpoint_and_resume(tmp_path):
    """The table-sharded sparse mode under the driver: every rank writes its rows of the entity table beside the checkpoint
    (model.ckpt-<step>.shard<g>of<N>.npz), a resumed 2-rank run reads them back, and ends with the tables of the uninterrupted
    run bit for bit."""
    import torch.multiprocessing as mp
    port = 30900 + os.getpid() % 1000
    for i, (run, times) in enumerate((("full", 4), ("split", 2), ("split", 2))):
        mp.start_processes(_resume_worker, args=(2, port + i, str(tmp_path), run, times, True), nprocs=2, join=True, start_method="spawn")
        if run == "split" and i == 1:
            os.rename(str(tmp_path / "split_t2_r0.npz"), str(tmp_path / "split_first_r0.npz"))
            parts = sorted(f for f in os.listdir(str(tmp_path / "split")) if ".shard" in f and "ckpt-10." in f)
            assert parts == ["model.ckpt-10.shard0of2.npz", "model.ckpt-10.shard1of2.npz"]
    full = np.load(str(tmp_path / "full_t4_r0.npz"))
    second = np.load(str(tmp_path / "split_t2_r0.npz"))
    assert int(second["step"]) == 20 and int(full["step"]) == 20
    for k in full.files:
        assert np.array_equal(second[k], full[k]), k


def _oracle_worker(rank, world, port, out_dir, model_name, opt):
    """Two ranks take ONE step; rank 0 also saves the tables the step started from and the global batch (host copies of both
    ranks' slices are reassembled by the test from the oracle sampler, which the device sampler equals bit for bit)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.set_in_path(os.path.join(GOLDEN, "kg_small"))
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(48); con.set_nbatches(10)
    con.set_ent_neg_rate(3); con.set_alpha(0.02); con.set_opt_method(opt)
    con.prefetch_sampling = False
    con.counts_min_records = 0
    con.init()
    con.set_model_and_session(getattr(pkg, model_name))
    con.init_distributed()
    before = con.get_parameters()
    states = con.get_stream_states()
    loss = con.train_step()
    torch.cuda.synchronize()
    after = con.get_parameters()
    if rank == 0:
        np.savez(os.path.join(out_dir, "oracle_case.npz"), loss=loss, states=states, batch=con.batch_size,
                 **{"before_" + k: v for k, v in before.items()}, **{"after_" + k: v for k, v in after.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model_name,opt", [("TransE", "SGD"), ("TransE", "Adam"), ("TransH", "SGD")])
def test_two_rank_step_against_the_oracle_full_batch_step(tmp_path, model_name, opt):
    """A 2-rank step compared DIRECTLY with the oracle's step on the full global batch (not with the 1-rank engine): the ranks'
    slices are the virtual sampler threads [0,4) and [4,8) of the reference's own partition (Base.cpp:85-92), each rank
    differentiates its slice with the GLOBAL denominator, the images are summed -- so the result must be the single-process
    reference step on the whole batch (TransE.py:26-51, distribute_training.py:95-101): loss to 2e-5, the SGD update to 1e-5 of
    its largest element, every element of the Adam update explained by a gradient within 1e-5 (tests/parity_util.py)."""
    import torch.multiprocessing as mp
    from oracle import oracle
    port = 29800 + os.getpid() % 1000
    mp.start_processes(_oracle_worker, args=(2, port, str(tmp_path), model_name, opt), nprocs=2, join=True, start_method="spawn")
    z = np.load(str(tmp_path / "oracle_case.npz"))
    kg_dir = os.path.join(GOLDEN, "kg_small")
    kg = oracle.KG(kg_dir, work_threads=8, bern=1)
    kg.set_stream_states(z["states"])
    B, n = int(z["batch"]), 3
    bh, bt, br, _ = kg.sampling(B, n, 0)
    before = {k[len("before_"):]: z[k] for k in z.files if k.startswith("before_")}
    after = {k[len("after_"):]: z[k] for k in z.files if k.startswith("after_")}
    orc = oracle.Model(model_name.lower(), kg.entTotal, kg.relTotal, 48, 48, margin=1.0, params=before)
    loss_o = orc.sgd_step(bh, bt, br, B, n, 0.02) if opt == "SGD" else orc.adam_step(bh, bt, br, B, n, 0.02)
    assert abs(float(z["loss"]) - loss_o) <= 2e-5 * abs(loss_o), (float(z["loss"]), loss_o)
    g_o = None
    if opt == "Adam":      # the gradient the oracle's Adam step saw, for the per-element explanation below
        ref = oracle.Model(model_name.lower(), kg.entTotal, kg.relTotal, 48, 48, margin=1.0, params=before)
        _, g_o = ref.grad(bh, bt, br, B, n)
    for k in before:
        du_o = orc.params[k].astype(np.float64) - before[k]
        du_g = after[k].astype(np.float64) - before[k]
        if opt == "SGD":
            quantum = np.abs(before[k]).max() * 2.0 ** -23
            assert np.abs(du_g - du_o).max() <= 1e-5 * np.abs(du_o).max() + quantum, (k, np.abs(du_g - du_o).max(), np.abs(du_o).max())
        else:
            # a first Adam step moves every element by ~lr_t * sign(g): where g nearly cancels, a difference inside the 1e-5
            # gradient tolerance is a visible fraction of the step -- each element must be one such a gradient can produce
            from parity_util import adam_update_explained
            zero = np.zeros_like(before[k])
            rep = adam_update_explained(before[k], zero, zero, g_o[k], du_g, du_o, float(oracle.adam_lr_t(0.02, 0.9, 0.999, 1)))
            assert rep["unexplained"].size == 0, (k, rep["unexplained"][:8], rep["worst_steps"])


@pytest.mark.parametrize("model_name,opt,sparse,pieces", [("TransE", "Adam", False, 2), ("TransE", "SGD", False, 0), ("TransH", "SGD", False, 2),
                                                          ("TransR", "SGD", False, 0), ("TransE", "SGD", True, 0), ("TransE", "LazyAdam", True, 0),
                                                          ("TransH", "SGD", True, 0), ("TransD", "SGD", True, 0)])
def test_one_rank_rccl_group_runs_the_data_parallel_step(tmp_path, model_name, opt, sparse, pieces):
    """The box has one GPU, so RCCL cannot run with two ranks here -- but it can run with ONE: `force_data_parallel` sends the
    step of a one-rank "nccl" process group through the whole exchange (asynchronous reduce-scatter of the count / gradient image
    on device memory, optimizer on the owned share, in-place all-gather, the loss riding in the buffers; for the sharded table
    the all-to-all exchanges).  Every collective is then a copy, so tables, losses and rng states must equal the plain
    single-process run: bit for bit for TransE (integer counts), to the order of the fp32 atomics for the other models (two
    plain runs of those differ by as much).  What this pins that the
    `gloo` tests cannot: stream ordering between the engine's kernels and RCCL's own stream, int32 / fp32 reduce-scatter and the
    in-place all-gather on device buffers."""
    import torch.multiprocessing as mp
    port = 29700 + os.getpid() % 1000
    prefetch = not sparse
    mp.start_processes(_worker, args=(1, port, str(tmp_path), model_name, opt, sparse, prefetch, 10, 0), nprocs=1, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(1, port + 1, str(tmp_path), model_name, opt, sparse, prefetch, 10, pieces, True), nprocs=1, join=True,
                       start_method="spawn")
    plain, rccl = np.load(str(tmp_path / "w1_r0.npz")), np.load(str(tmp_path / "w1_r0_rccl.npz"))
    assert sorted(plain.files) == sorted(rccl.files)
    for k in plain.files:
        if model_name == "TransE" or k == "states":
            np.testing.assert_array_equal(plain[k], rccl[k], err_msg=k)
        else:
            np.testing.assert_allclose(rccl[k], plain[k], rtol=1e-5, atol=1e-7, err_msg=k)


def _big_a2a_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from openkeonspark_amd import parallel
    rows, D = 1_100_000, 512                                   # 2.25 GB of row payload
    inp = (torch.arange(rows * D, device="cuda", dtype=torch.float32) % 1000003).view(rows, D)
    raw = torch.zeros_like(inp)
    dist.all_to_all_single(raw, inp, [rows], [rows])
    torch.cuda.synchronize()
    raw_bad = int((raw != inp).any(dim=1).sum())
    del raw
    out = torch.zeros_like(inp)
    got = parallel.all_to_all_rows(out, inp, [rows], [rows])
    torch.cuda.synchronize()
    ours_bad = int((got != inp).any(dim=1).sum())
    np.savez(os.path.join(out_dir, "a2a.npz"), raw_bad=raw_bad, ours_bad=ours_bad)
    dist.destroy_process_group()


def test_row_exchange_of_more_than_two_gigabytes_arrives_whole(tmp_path):
    """`all_to_all_single` on this stack returns a 2 GB+ message half-copied (found when the table-sharded step was run at BASELINE
    config #5's per-GPU batch on a one-rank group: loss 5.47 instead of 1.007).  parallel.all_to_all_rows slices such payloads; this
    pins both the slicing and -- informationally -- whether the library still shows the fault."""
    import torch.multiprocessing as mp
    mp.start_processes(_big_a2a_worker, args=(29800 + os.getpid() % 1000, str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    z = np.load(str(tmp_path / "a2a.npz"))
    from conftest import parity_report
    parity_report("all_to_all_single_2GB_rows_corrupted_by_the_library", rows=int(z["raw_bad"]))
    assert int(z["ours_bad"]) == 0


def test_bench_starts_its_own_ranks_from_a_bare_shell():
    """`python3 bench.py --gpus 2 ...` from an environment with no launcher variables: bench.py itself starts the two ranks
    (torch.distributed.run children, before anything in the parent touches the GPU), the ranks run the data-parallel step, and
    the parent relays rank 0's single JSON line and exits 0.  On this one-GPU box the ranks share device 0 and use gloo
    (KGE_BENCH_BACKEND / KGE_BENCH_SINGLE_DEVICE); the driver's 8-GPU run takes the same path with RCCL.  Replaces what the
    reference's launcher does with TFCluster.run (/root/reference/main_spark.py:340)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(KGE_BENCH_BACKEND="gloo", KGE_BENCH_SINGLE_DEVICE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["rccl_ranks"] == 2
    assert d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak"
    # (rank 0 owns virtual threads 0..3 of 8: slices of ceil(B / 8) positions each, Base.cpp:85-92)
    assert d["config"]["parallelism"] == "dp2" and abs(d["config"]["global_batch"] - 2 * d["config"]["per_gpu_batch"]) <= 8
    assert np.isfinite(d["config"]["final_loss"]) and 0.0 < d["config"]["final_loss"] < 2.0
    assert d["value"] > 0 and "roofline" in d
