"""Many training steps inside one persistent launch (csrc/persist.hip, Config.train_steps) against the same steps as
separate launches and against the CPU oracle: the loop body of distribute_training.py:267-283 at the reference's own batch
sizes (Config.py:189-210).  The batches must be bit-identical (same rng streams afterwards); gradients go through fp32
atomics in both forms, whose order is not reproducible, so tables agree to summation-order rounding, not bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, parity_report
from oracle import oracle

pytestmark = pytest.mark.gpu


def engine(path, model, dim, nbatches, n, alpha, opt="SGD", bern=0, nr=0):
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.prefetch_sampling = False
    con.set_in_path(path); con.set_work_threads(8); con.set_bern(bern); con.set_dimension(dim)
    con.set_nbatches(nbatches); con.set_ent_neg_rate(n); con.set_rel_neg_rate(nr); con.set_alpha(alpha); con.set_margin(1.0)
    con.set_opt_method(opt)
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    return con


def update_err(got, want, start, tol):
    """(rows whose accumulated update differs by more than tol * the table's largest update, worst relative difference over
    the OTHER rows, total rows)"""
    outside = total = 0
    worst = 0.0
    for k in want:
        du_w = want[k].astype(np.float64) - start[k]
        du_g = got[k].astype(np.float64) - start[k]
        rel = (np.abs(du_g - du_w) / (np.abs(du_w).max() + 1e-30)).reshape(du_w.shape[0], -1).max(1)
        bad = rel > tol
        outside += int(bad.sum()); total += len(bad)
        if (~bad).any():
            worst = max(worst, float(rel[~bad].max()))
    return outside, worst, total


@pytest.mark.parametrize("model,graph,dim,nbatches,n,nr,opt", [
    ("TransE", "fb", 100, 0, 1, 0, "SGD"),        # configs[0]: auto batch 2 721
    ("TransH", "wn", 200, 0, 1, 0, "SGD"),        # configs[2]: auto batch 8 683, relation hub copies folded in the sweep
    ("TransD", "fb", 64, 0, 2, 1, "Adam"),
    ("TransE", "small", 50, 7, 3, 1, "Adam"),     # ragged slices (857 positives over 8 threads), dim % 4 != 0
    ("TransH", "small", 24, 10, 2, 0, "SGD"),
])
def test_persistent_steps_equal_separate_launches(fb_dir, wn_dir, model, graph, dim, nbatches, n, nr, opt):
    from openkeonspark_amd import _lib
    path = {"fb": fb_dir, "wn": wn_dir, "small": os.path.join(GOLDEN, "kg_small")}[graph]
    S, alpha = 12, 0.01
    runs = []
    for persistent in (False, True):
        _lib.lib().kge_set_option(b"libc_rand_restart", 1)
        con = engine(path, model, dim, nbatches, n, alpha, opt, bern=1, nr=nr)
        assert con.persistent_supported()
        start = con.get_parameters()
        losses = con.train_steps(S, persistent=persistent)
        runs.append((losses, con.get_parameters(), con.get_stream_states(), con.global_step))
        for g in con.get_gradients().values():
            assert not g.any()                                   # accumulators (and hub copies) end re-zeroed
    (l0, p0, s0, g0), (l1, p1, s1, g1) = runs
    assert g0 == g1 == S and s0.tolist() == s1.tolist()          # the same batches were drawn
    # fp32 atomics add in a different order in every run of EITHER form, so two runs differ by rounding; wherever an element of
    # e = h^ + r^ - t^ or a hinge sits within that rounding of zero the next step's gradient row flips and the difference is
    # carried on: most rows agree to rounding, a few (reported, bounded) carry such a flip
    tol = 2e-5 if opt == "SGD" else 2e-3                         # Adam: m / (sqrt(v) + eps) amplifies rounding where v ~ 0
    outside, worst, total = update_err(p1, p0, start, tol)
    parity_report("persistent_vs_launches[%s-%s-%s]" % (model, graph, opt), steps=S, loss_relerr=float(np.abs(l1 / l0 - 1).max()),
                  rows_outside=outside, of_rows=total, worst_other_rows=worst, tol=tol)
    # (the losses of the first steps agree to rounding; with Adam a carried flip moves the later ones by up to a few 1e-4 --
    # seen: 1.7e-4 at step 10 of TransD / Adam -- so the trajectory is held to 1e-3 there and the first three steps to 1e-5)
    assert np.allclose(l0[:3], l1[:3], rtol=1e-5, atol=0), np.abs(l0[:3] / l1[:3] - 1).max()
    assert np.allclose(l0, l1, rtol=1e-4 if opt == "SGD" else 1e-3, atol=0), np.abs(l0 / l1 - 1).max()
    assert outside <= max(3, 0.01 * total), (outside, total)


@pytest.mark.parametrize("model,graph,dim", [("transe", "fb", 100), ("transh", "wn", 200)])
def test_persistent_steps_match_oracle(fb_dir, wn_dir, model, graph, dim):
    """configs[0] and configs[2] at their auto batch through the persistent launch against the oracle.  Each launch (one
    step, so that the oracle can restart from the engine's tables: two fp32 trajectories of this loss drift apart at every
    element of e within rounding of zero) must give the oracle's loss and one-step update to 1e-5; rows outside it have to be
    rows of a group with such an element (tests/torch_ref.py::near_kink_rows).  Multi-step launches are pinned to the
    one-launch-per-stage path by test_persistent_steps_equal_separate_launches."""
    from torch_ref import near_kink_rows
    from test_gpu_configs import tie_group_rows
    path = fb_dir if graph == "fb" else wn_dir
    alpha, n = 0.01, 1
    con = engine(path, {"transe": "TransE", "transh": "TransH"}[model], dim, 0, n, alpha)
    kg = oracle.KG(path, work_threads=8, bern=0)
    kg.set_stream_states(con.get_stream_states())
    B = con.batch_size
    orc = oracle.Model(model, con.entTotal, con.relTotal, dim, dim, margin=1.0, params=con.get_parameters())
    outside = kink_elems = tie_groups = 0
    for launch in range(3):
        start = con.get_parameters()
        orc.params = {k: v.copy() for k, v in start.items()}
        bh, bt, br, _ = kg.sampling(B, n, 0)
        hm = orc.hinge_margins(bh, bt, br, B, n)     # at the launch's starting tables (sgd_step moves orc.params)
        want = orc.sgd_step(bh, bt, br, B, n, alpha)
        got = con.train_steps(1, persistent=True)
        assert con.get_stream_states().tolist() == kg.stream_states().tolist()
        assert abs(got[0] - want) <= 1e-5 * abs(want), (got, want)
        after = con.get_parameters()
        kink = None
        for k in orc.params:
            du_o = orc.params[k].astype(np.float64) - start[k]
            du_g = after[k].astype(np.float64) - start[k]
            quantum = np.abs(start[k]).max() * 2.0 ** -23
            bad = np.nonzero((np.abs(du_g - du_o) > 1e-5 * np.abs(du_o).max() + quantum).reshape(du_o.shape[0], -1).any(1))[0]
            if len(bad):
                if kink is None:
                    kink, n_el = near_kink_rows(model, start, bh, bt, br, B, n, dim, dim, tol=3e-7)
                    kink_elems += n_el
                    tied, n_tied = tie_group_rows(hm, start, bh, bt, br, B, n)   # or of a group whose hinge sits on its switch point
                    tie_groups += n_tied
                    for kk in kink:
                        kink[kk] |= tied[kk]
                outside += len(bad)
                assert not set(bad.tolist()) - kink[k], (k, sorted(set(bad.tolist()) - kink[k])[:10])
    parity_report("persistent_vs_oracle[%s]" % model, launches=3, batch=B, update_rows_outside_1e5=outside,
                  elements_of_e_within_3e7_of_zero=kink_elems, groups_with_hinge_within_tie_tol=tie_groups)
    assert outside <= 6 * (kink_elems + tie_groups)
