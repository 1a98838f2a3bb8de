"""Sparse-row TransE path (stage-level C ABI: emit records -> compact per-row counts -> row SGD) against the
dense count image and the oracle.  Integer sums: the two engine paths must agree bit for bit."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_config(kg, dim, n_neg, sparse, nbatches=4, threads=4, seed=7, fused=True):
    import torch
    import openkeonspark_amd as ok
    con = ok.Config()
    con.set_in_path(os.path.join(GOLD, kg) + "/")
    con.set_work_threads(threads)
    con.set_nbatches(nbatches)
    con.set_alpha(0.05)
    con.set_margin(1.0)
    con.set_bern(1)
    con.set_dimension(dim)
    con.set_ent_neg_rate(n_neg)
    con.set_rel_neg_rate(0)
    con.set_opt_method("SGD")
    con.sparse_rows = sparse
    con.sparse_fused = fused
    con.counts_min_records = 0
    con.init()
    torch.manual_seed(seed)
    con.set_model_and_session(ok.TransE)
    return con


@pytest.mark.parametrize("dim,n_neg", [(16, 1), (50, 3), (200, 25), (512, 2)])
@pytest.mark.parametrize("inv_table", [True, False])
@pytest.mark.parametrize("fused", [True, False])
def test_sparse_equals_dense_counts(dim, n_neg, inv_table, fused):
    import torch
    import openkeonspark_amd as ok
    from openkeonspark_amd import _lib
    lib = _lib.load()
    _lib.check(lib.kge_set_option(b"libc_rand_restart", 1), lib)
    _lib.check(lib.kge_set_option(b"inv_table_max_bytes", (256 << 20) if inv_table else 0), lib)
    try:
        dense = make_config("kg_small", dim, n_neg, sparse=False)
        assert dense.use_counts and not dense.sparse_rows
        p0 = [t.clone() for t in dense._tables]
        losses_d = [dense.train_step() for _ in range(6)]
        _lib.check(lib.kge_set_option(b"libc_rand_restart", 1), lib)
        sparse = make_config("kg_small", dim, n_neg, sparse=True, fused=fused)
        assert sparse.sparse_rows and sparse._grads == []
        for t, q in zip(sparse._tables, p0):
            t.copy_(q)
        losses_s = [sparse.train_step() for _ in range(6)]
    finally:
        _lib.check(lib.kge_set_option(b"inv_table_max_bytes", 256 << 20), lib)
    # Same integer counts, same per-row formula; widths that are a multiple of 4 sum the row's dot products in the
    # vectorised kernels' lane order, so they may differ from the dense apply kernel in the last bit per step.
    if dim % 4:
        assert losses_d == losses_s
        for a, b in zip(dense._tables, sparse._tables):
            assert torch.equal(a, b)
    else:
        assert np.allclose(losses_d, losses_s, rtol=1e-6, atol=0)
        for a, b, q in zip(dense._tables, sparse._tables, p0):
            step = (a - q).abs().max()
            assert float((a - b).abs().max()) <= 1e-5 * float(step)
    rows, counts = sparse.sparse_row_gradients()
    rows = rows.cpu().numpy()
    assert len(rows) > 0 and np.all(np.diff(rows) > 0) and rows.max() < sparse.entTotal + sparse.relTotal


def test_sparse_row_counts_match_dense_image():
    """The compact image is the dense image restricted to its touched rows."""
    import torch
    from openkeonspark_amd import _lib
    lib = _lib.load()
    _lib.check(lib.kge_set_option(b"libc_rand_restart", 1), lib)
    dense = make_config("kg_tiny", 64, 4, sparse=False, nbatches=2, threads=2)
    dev, n_pos = dense.sample_device()
    stride = max(dense._n_local, 1)
    denom = dense.batch_size * 4
    dense.forward_counts(dev, n_pos, stride, denom)
    image = dense._counts.clone()
    _lib.check(lib.kge_set_option(b"libc_rand_restart", 1), lib)
    sparse = make_config("kg_tiny", 64, 4, sparse=True, nbatches=2, threads=2, fused=False)
    for t, q in zip(sparse._tables, dense._tables):
        t.copy_(q)
    sparse.set_alpha(0.0)   # keep the tables: only the reduction is under test
    sparse.train_step()
    rows, counts = sparse.sparse_row_gradients()
    touched = torch.zeros(image.shape[0], dtype=torch.bool, device=image.device)
    touched[rows.long()] = True
    assert torch.equal(image[rows.long()], counts)
    assert int(image[~touched].abs().sum()) == 0


def test_sparse_rejects_unshaped_host_batch():
    import openkeonspark_amd as ok
    con = make_config("kg_tiny", 32, 1, sparse=True, nbatches=2, threads=1)
    n = 8
    h = np.arange(n, dtype=np.int64) % con.entTotal
    t = (np.arange(n, dtype=np.int64) + 3) % con.entTotal
    r = np.zeros(n, dtype=np.int64)
    # negatives that change head AND tail: not single-slot corruptions
    bh = np.concatenate([h, (h + 1) % con.entTotal])
    bt = np.concatenate([t, (t + 1) % con.entTotal])
    br = np.concatenate([r, r])
    with pytest.raises(ok.KgeError):
        con.train_step(bh, bt, br, None)


def test_sparse_large_table_properties():
    """Config #5's regime scaled to fit a test (2 M entities x 512 = 4 GB table, initialised in HBM): size-independent
    properties of one sparse step.  Every scored triple adds +g to its head row and -g to its tail row, so the integer
    counts over the ENTITY rows sum to zero column by column; the row list is strictly increasing; only listed rows move."""
    import torch
    import openkeonspark_amd as ok
    E, R, D, n_tr = 2_000_000, 500, 512, 400_000
    rng = np.random.default_rng(9)
    h = rng.integers(0, E, n_tr); t = rng.integers(0, E, n_tr); r = rng.integers(0, R, n_tr)
    con = ok.Config()
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(D); con.set_ent_neg_rate(2); con.set_rel_neg_rate(0)
    con.set_alpha(0.01); con.set_opt_method("SGD"); con.set_nbatches(8)        # B = 50 000
    con.sparse_rows = True
    con.sparse_fused = False          # the complete compact count image is inspected below
    con.init_from_arrays(E, R, h, t, r)
    con.set_model_and_session(ok.TransE)
    assert con._grads == [] and not hasattr(con, "_counts")                     # no dense image of any kind
    ent = con._tables[0]
    assert ent.shape == (E, D) and abs(float(ent[:1000].std()) - np.sqrt(2.6 / (E + D)) * 0.88) < 2e-5
    probe = torch.arange(0, E, 997, device=ent.device)
    before = ent[probe].clone()
    loss = con.train_step()
    assert 0.5 < loss < 1.5
    rows, counts = con.sparse_row_gradients()
    rows_h = rows.cpu().numpy()
    assert np.all(np.diff(rows_h) > 0) and rows_h.min() >= 0 and rows_h.max() < E + R
    is_ent = rows < E
    assert int(counts[is_ent].sum(dim=0).abs().max()) == 0                     # head/tail contributions cancel exactly
    assert int(counts[~is_ent].abs().sum()) > 0
    assert int(counts.abs().max()) <= con.batch_size * 3                        # bounded by the records of a hub relation
    touched = torch.zeros(E, dtype=torch.bool, device=ent.device)
    touched[rows[is_ent].long()] = True
    moved = (ent[probe] != before).any(dim=1)
    assert not bool((moved & ~touched[probe]).any())                            # untouched rows are bit-identical
    assert bool(moved.any()) or not bool(touched[probe].any())


@pytest.mark.parametrize("dim,n_neg", [(16, 1), (200, 25), (512, 2)])
def test_fused_reduce_apply_equals_two_stage(dim, n_neg):
    """kge_transe_reduce_apply_records_sgd == kge_transe_reduce_records + kge_transe_apply_rows_sgd, bit for bit (which
    rows are chunk-interior depends on the record count, i.e. on the number of ranks: the bits must not)."""
    import torch
    from openkeonspark_amd import _lib
    lib = _lib.load()
    runs = []
    for fused in (True, False):
        _lib.check(lib.kge_set_option(b"libc_rand_restart", 1), lib)
        con = make_config("kg_small", dim, n_neg, sparse=True, fused=fused)
        losses = [con.train_step() for _ in range(5)]
        runs.append((losses, [t.clone() for t in con._tables]))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("model,opt,n_neg", [("TransH", "Adam", 1), ("TransE", "Adam", 1), ("TransR", "SGD", 1)])
def test_sparse_rows_request_that_cannot_be_honoured_raises(model, opt, n_neg):
    """sparse_rows=True needs a touched-rows-only update: TransE / TransH / TransD with SGD (or TransE with the labelled LazyAdam).
    TF1's Adam moves every row and TransR's matrices have no such path: those must fail loudly instead of silently allocating dense
    gradient / count / Adam images."""
    import openkeonspark_amd as ok
    con = ok.Config()
    con.set_in_path(os.path.join(GOLD, "kg_tiny") + "/")
    con.set_work_threads(2); con.set_nbatches(2); con.set_dimension(16); con.set_ent_neg_rate(n_neg); con.set_opt_method(opt)
    con.sparse_rows = True
    con.init()
    with pytest.raises(ok.KgeError):
        con.set_model_and_session(getattr(ok, model))


@pytest.mark.parametrize("model,n_neg", [("TransH", 1), ("TransD", 2), ("TransE", 64)])
def test_sparse_rows_request_off_the_count_path_takes_the_in_place_row_update(model, n_neg):
    """TransH / TransD -- and TransE with more negatives than the int8 records hold -- honour sparse_rows=True through
    kge_forward_backward_sgd_rows: no gradient tables are allocated and sampled steps train (loss finite and falling)."""
    import openkeonspark_amd as ok
    con = ok.Config()
    con.set_in_path(os.path.join(GOLD, "kg_tiny") + "/")
    con.set_work_threads(2); con.set_nbatches(2); con.set_dimension(16); con.set_ent_neg_rate(n_neg); con.set_opt_method("SGD")
    con.set_alpha(0.05)
    con.sparse_rows = True
    con.init()
    con.set_model_and_session(getattr(ok, model))
    assert con.sparse_inplace and not con.sparse_rows and con._grads == []
    losses = [con.train_step() for _ in range(30)]
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])


def test_shard_rows_drawn_directly_equal_the_rows_of_the_whole_table():
    """A rank of the table-sharded mode draws ONLY its rows (Model.xavier_normal_device with row_range, through one scratch
    block) -- they must be the rows the single-process table holds, or the union of the shards is not that table.
    600 000 x 512 floats: beyond LARGE_TABLE_ELEMS, several generator blocks, a shard that straddles block boundaries."""
    import torch
    from openkeonspark_amd.Model import xavier_normal_device, LARGE_TABLE_ELEMS
    shape = (600000, 512)
    assert shape[0] * shape[1] > LARGE_TABLE_ELEMS
    full = xavier_normal_device(shape, "cuda", 3)
    for lo, hi, chunk in ((0, 150000, 150000), (150000, 450000, 300000), (450000, 600000, 150016)):
        part = xavier_normal_device(shape, "cuda", 3, row_range=(lo, hi), out_rows=chunk)
        assert part.shape == (chunk, 512)
        assert torch.equal(part[:hi - lo], full[lo:hi])
        assert not part[hi - lo:].any()
