"""The fused single-process TransE step (kge_transe_train_step_counts: 2-bit negative records, segmented sum and optimizer in
one kernel -- csrc/transe_counts.hip segapply_kernel) against the two-call form it replaces (kge_transe_forward_counts: int8
records, segmented sum into the count image; kge_transe_apply_counts_tables), which the data-parallel step keeps.

Both forms sum the same integer signs and push them through ONE per-row update function, so the results must be equal BIT FOR
BIT -- tables, Adam moments, loss, the emit kernel's 1/|row| table (through the following steps) -- whatever the embedding width
(team shapes 16x4, 32x4, 64x4, 64x8), optimizer, chunk capacity (rows longer than the capacity go through the count image in
pieces) and batch kind (device-sampled; hand-made with negatives that are not single-slot corruptions: exact fp32 residuals).
The arithmetic itself is checked against the oracle by test_gpu_models.py / test_gpu_configs.py, which now run this path.
Reference semantics: TransE.py:26-51, distribute_training.py:95-101."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_models import rand_batch

pytestmark = pytest.mark.gpu


def engine(path, dim, nbatches, n, opt, fused, cap=0, arrays=None, streams=None):
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.lib.kge_set_option(b"counts_fused_cap", cap)
    con.fused_counts = fused
    con.counts_min_records = 0
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(dim); con.set_nbatches(nbatches)
    con.set_ent_neg_rate(n); con.set_alpha(0.01 if opt == "SGD" else 0.001); con.set_margin(1.0); con.set_opt_method(opt)
    if arrays is None:
        con.set_in_path(path)
        con.init()
    else:
        con.init_from_arrays(*arrays)
    con.set_model_and_session(pkg.TransE)
    assert con.use_counts and not con.sparse_rows
    # every engine of a test starts from the SAME rng streams: randReset (Random.h:8-13) continues the process-wide libc sequence,
    # so a second init() in one process would otherwise draw other batches
    if streams is not None:
        con.lib.kge_set_stream_states(streams.ctypes.data, 8)
    return con


def state(con):
    import torch
    torch.cuda.synchronize()
    out = {k: v.copy() for k, v in con.get_parameters().items()}
    if con._adam:
        for i, k in enumerate(con.trainModel.table_names):
            out["m/" + k] = con._adam_m[i].cpu().numpy()
            out["v/" + k] = con._adam_v[i].cpu().numpy()
    out["counts"] = con._counts.cpu().numpy()
    return out


@pytest.mark.parametrize("dim", [200, 64, 100, 512, 16])
@pytest.mark.parametrize("opt", ["Adam", "SGD"])
@pytest.mark.parametrize("cap", [0, 7])
def test_fused_step_equals_the_two_call_step_on_sampled_batches(dim, opt, cap):
    """Six device-sampled steps on the small golden graph (500 entities, 6 000 triples): B = 600 x 5 negatives -- ~8 records per
    entity row, hundreds per relation row, so capacity 7 sends many entity rows through the image in pieces as well."""
    path = os.path.join(GOLDEN, "kg_small")
    a = engine(path, dim, 10, 5, opt, fused=True, cap=cap)
    s0 = a.get_stream_states()
    losses_a = [a.train_step() for _ in range(6)]
    sa = state(a)
    a.lib.kge_set_option(b"counts_fused_cap", 0)
    b = engine(path, dim, 10, 5, opt, fused=False, streams=s0)
    losses_b = [b.train_step() for _ in range(6)]
    sb = state(b)
    assert losses_a == losses_b
    for k in sb:
        assert np.array_equal(sa[k], sb[k]), (k, float(np.abs(sa[k].astype(np.float64) - sb[k]).max()))
    assert not sa["counts"].any()                    # the image is left all-zero
    assert a.get_stream_states().tolist() == b.get_stream_states().tolist()


def test_fused_step_equals_the_two_call_step_at_bench_size(fb_dir):
    """BASELINE configs[1] as bench.py runs it (B = 34 014 x 25 negatives, dim 200, TF1 Adam): five steps, bit for bit."""
    streams = []

    def run(fused):
        import openkeonspark_amd as pkg
        con = pkg.Config()
        con.fused_counts = fused
        con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(1); con.set_dimension(200); con.set_nbatches(8)
        con.set_ent_neg_rate(25); con.set_alpha(0.001); con.set_margin(1.0); con.set_opt_method("Adam")
        con.init()
        con.set_model_and_session(pkg.TransE)
        if streams:
            con.lib.kge_set_stream_states(streams[0].ctypes.data, 8)
        else:
            streams.append(con.get_stream_states())
        losses = [con.train_step() for _ in range(5)]
        return losses, state(con)
    la, sa = run(True)
    lb, sb = run(False)
    assert la == lb
    for k in sb:
        assert np.array_equal(sa[k], sb[k]), k


@pytest.mark.parametrize("opt", ["Adam", "SGD"])
def test_fused_step_with_hand_made_batches_and_fp32_residuals(opt):
    """Fed batches in which 30 % of the negatives are arbitrary triples (not single-slot corruptions): those groups are
    differentiated in fp32 into the residual tables, which the fused kernel adds to the rows it updates exactly as the apply
    kernel does; rows without any record but with a residual are left to the apply kernel.  Equal to the two-call form up to the
    ORDER of the fp32 atomic adds into the residual tables (the only non-integer accumulation): SGD within 1e-6 of the largest
    update; Adam, whose step lr_t m / (sqrt(v) + eps) turns a last-bit difference of a nearly cancelling gradient element into a
    visible fraction of a step (tests/parity_util.py), 99.9 % of the elements within 1e-4 of the largest update and none further
    than the steps taken allow."""
    rng = np.random.default_rng(5)
    E, R, D, B, n = 300, 11, 100, 256, 4
    hh = np.arange(40) % E
    arrays = (E, R, hh, (hh + 1) % E, hh % R)
    a = engine(None, D, 1, n, opt, fused=True, arrays=arrays)
    b = engine(None, D, 1, n, opt, fused=False, arrays=arrays)
    b.set_parameters(a.get_parameters())
    start = a.get_parameters()
    for step in range(4):
        bh, bt, br = rand_batch(rng, E, R, B, n, 0, foreign=0.3)
        la = a.train_step(bh, bt, br, None)
        lb = b.train_step(bh, bt, br, None)
        assert abs(la - lb) <= 1e-6 * abs(lb)
    sa, sb = state(a), state(b)
    for k in start:
        scale = np.abs(sb[k] - start[k]).max()
        diff = np.abs(sa[k].astype(np.float64) - sb[k])
        if opt == "SGD":
            assert diff.max() <= 1e-6 * scale + np.abs(sb[k]).max() * 2.0 ** -23, k
        else:
            assert np.quantile(diff, 0.999) <= 1e-4 * scale and diff.max() <= 2.1 * 4 * 0.001, (k, float(diff.max()), float(scale))
    assert not sa["counts"].any() and not any(g.any() for g in a.get_gradients().values())
