"""The BASELINE.json workloads on their OWN graphs and batch sizes, through the C ABI, against the CPU oracle.

configs[2]  WN18RR-shaped TransH dim 200     (TransH.py:12-69)   auto batch 8 683 (atomic path + hub copies) and
                                                                   B = 28 945 (float records + segmented sum)
configs[3]  FB15k-237-shaped TransR 200x200  (TransR.py:16-75)   auto batch 2 721 and B = 34 014, tilings chosen
                                                                   automatically (no transr_v1 forcing)
configs[4]  50 M-entity TransE dim 512       (TransE.py:11-51)   sparse-row step at a reduced entity count against the
                                                                   oracle, and size-independent properties at full size
Batch rule: Config.py:189-210.  Each test first checks the device sampler bit for bit against the oracle's
batch (Base.cpp:74-172), then loss, summed gradients and the SGD parameters (distribute_training.py:98-101) to the
1e-5 relative tolerance of BASELINE.json's north_star.  Rows that fall outside it are COUNTED, reported, bounded and
each one EXPLAINED: it must belong to a group in which an element of e = h^ + r^ - t^ is within fp32 rounding of zero
(d|e|/de jumps there), which is a property of the loss, not of either implementation."""
import os
import numpy as np
import pytest

from conftest import parity_report
from oracle import oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-5
# |e| below this: two correct fp32 evaluations of e = h^ + r^ - t^ may disagree on its sign.  Each x^ = x * rsqrt(sum x^2) carries
# the rounding of a D-term sum of squares whose order differs between implementations (a lane-strided fma chain + butterfly here,
# a sequential loop in the oracle: ~sqrt(D) * 2^-24 ~ 1e-6 relative at D = 200 in the worst case, 1e-7 typically, times |x^| <= 1),
# plus half an ulp for each of the three normalised values and the two additions (ulp(0.5) = 6e-8): differences of 1e-7 are
# typical, 5e-7 occurs (observed on the bench batch: a flipped element with |e| = 5.2e-7 in fp64).  1e-6 covers it; of the 177 M
# elements of the bench batch about 1 100 lie below it.
KINK_TOL = 1e-6
KINK_TOL_TRANSR = 1e-6   # (there h^, t^ additionally come out of a 200-term projection summed in different orders: MFMA tiles vs a scalar loop)


# |p - n + margin| below this: the two L1 scores may put the hinge on either side.  p and n are sums of D = 200 terms |e_i| of
# normalised vectors, 10..20 in value, where one fp32 ulp is 1.9e-6; the engine adds them lane-strided + butterfly, the oracle
# sequentially, and the normalised inputs already differ in the last bit: differences of several ulps of the partial sums occur
# (observed on the bench batch: a flipped hinge 1.7e-5 from its switch point).  5e-5 covers it; of the 850 350 hinges of a bench
# step about 20 lie inside the band.
TIE_TOL = 5e-5


def tie_group_rows(hm, params, bh, bt, br, B, N, tol=TIE_TOL):
    """{table: rows} of the groups with a hinge within `tol` of its switch point (hm = the oracle's hinge margins [B, N] at the
    step's STARTING parameters): a flipped hinge switches the gradient rows of the whole (positive, negative) pair on or off,
    which reaches every row of the group."""
    hm = np.abs(hm)
    groups = np.nonzero((hm < tol).any(1))[0]
    rows = {k: set() for k in params}
    idx = (groups[:, None] + B * np.arange(N + 1)[None, :]).ravel() if len(groups) else np.zeros(0, np.int64)
    ents = set(np.asarray(bh)[idx].tolist()) | set(np.asarray(bt)[idx].tolist())
    rels = set(np.asarray(br)[idx].tolist())
    for k in rows:
        rows[k] = set(ents) if k in ("ent_embeddings", "ent_transfer") else set(rels)
    return rows, len(groups)


def closest_switch_points(hm, model, params, bh, bt, br, B, N, dims, table, rows):
    """diagnostics for an unexplained row: the smallest |hinge| and smallest |e| over the groups that touch it"""
    from torch_ref import near_kink_rows
    out = []
    hm = np.abs(hm)
    bh, bt, br = np.asarray(bh), np.asarray(bt), np.asarray(br)
    for row in rows:
        use = (br == row) if table in ("rel_embeddings", "normal_vectors", "rel_transfer", "transfer_matrix") else ((bh == row) | (bt == row))
        groups = np.unique(np.nonzero(use)[0] % B)
        idx = (groups[:, None] + B * np.arange(N + 1)[None, :]).ravel()
        tol = 1e-4
        while tol > 1e-9 and near_kink_rows(model, params, bh[idx], bt[idx], br[idx], len(groups), N, dims[0], dims[1], tol=tol)[1] > 0:
            tol /= 2
        out.append(dict(row=int(row), groups=len(groups), min_abs_hinge=float(hm[groups].min()), no_e_below=tol))
    return out


def check_sampled_batch(con, kg, B, n):
    """Device sampler == oracle sampler for the next batch (both advance); returns device batch + host arrays."""
    dev, n_pos = con.sample_device()
    bh, bt, br, _ = kg.sampling(B, n, 0)
    host = dev.cpu().numpy()
    assert n_pos == B
    assert np.array_equal(host[0], bh) and np.array_equal(host[1], bt) and np.array_equal(host[2], br)
    return dev, bh, bt, br


def run_steps(con, kg, orc, B, n, alpha, steps, name, model, dims, max_outside_rows=24, sampler_shaped=False):
    """`steps` SGD steps on device-sampled batches.  Every step the oracle starts from the ENGINE's current tables, so
    each comparison is of one forward/backward/update on identical inputs.  Rows outside 1e-5 must be explained: they
    have to be rows of a group in which some element of e = h^ + r^ - t^ lies within fp32 rounding of zero (fp64
    evaluation, tests/torch_ref.py::near_kink_rows), or of a group with a hinge p - n + margin within TIE_TOL of its switch
    point (the two scores are sums of D terms added in different orders) -- the two places where two correct fp32
    evaluations of this loss legitimately differ by more than rounding."""
    import torch
    from torch_ref import near_kink_rows
    worst = dict(loss=0.0, grad=0.0, grad_rows=0, update_rows=0, ties=0, kink_elems=0, tie_groups=0)
    for step in range(steps):
        start = con.get_parameters()
        orc.params = {k: v.copy() for k, v in start.items()}
        dev, bh, bt, br = check_sampled_batch(con, kg, B, n)
        hm = orc.hinge_margins(bh, bt, br, B, n)            # at the step's starting parameters (orc.params moves below)
        worst["ties"] += int((np.abs(hm) < 2e-6).sum())
        loss_o, g_o = orc.grad(bh, bt, br, B, n)
        con.forward_backward(dev, B, B, B * n, sampler_shaped=sampler_shaped)      # (True: what train_step passes for a device-sampled batch)
        torch.cuda.synchronize()
        loss_g = float(con._loss.item())
        worst["loss"] = max(worst["loss"], abs(loss_g - loss_o) / abs(loss_o))
        g_g = con.get_gradients()
        con.apply_gradients()
        orc.apply_sgd(g_o, alpha)
        got = con.get_parameters()
        kink = None
        for k in g_o:
            scale = np.abs(g_o[k]).max() + 1e-30
            diff = np.abs(g_g[k].astype(np.float64) - g_o[k])
            bad = np.nonzero((diff > RTOL * scale).reshape(diff.shape[0], -1).any(1))[0]
            du_o = orc.params[k].astype(np.float64) - start[k]
            du_g = got[k].astype(np.float64) - start[k]
            quantum = np.abs(start[k]).max() * 2.0 ** -23           # p - lr*g is rounded at the parameter's magnitude
            bad_u = np.nonzero((np.abs(du_g - du_o) > RTOL * np.abs(du_o).max() + quantum).reshape(diff.shape[0], -1).any(1))[0]
            worst["grad_rows"] += len(bad); worst["update_rows"] += len(bad_u)
            clean = np.ones(diff.shape[0], bool); clean[bad] = False
            if clean.any():
                worst["grad"] = max(worst["grad"], float(diff[clean].max() / scale))
            if len(bad) or len(bad_u):
                if kink is None:
                    kink, n_el = near_kink_rows(model, start, bh, bt, br, B, n, dims[0], dims[1],
                                                tol=KINK_TOL_TRANSR if model == "transr" else KINK_TOL)
                    worst["kink_elems"] += n_el
                    tied, n_tied = tie_group_rows(hm, start, bh, bt, br, B, n)
                    worst["tie_groups"] += n_tied
                    for kk in kink:
                        kink[kk] |= tied[kk]
                unexplained = (set(bad.tolist()) | set(bad_u.tolist())) - kink[k]
                if unexplained and os.environ.get("KGE_DUMP_UNEXPLAINED"):   # the groups touching the first such row, for offline study
                    r0 = sorted(unexplained)[0]
                    hh, tt, rr = np.asarray(bh), np.asarray(bt), np.asarray(br)
                    gs = np.unique(np.nonzero((hh == r0) | (tt == r0))[0] % B)
                    idx = (gs[:, None] + B * np.arange(n + 1)[None, :]).ravel()
                    ents = np.unique(np.concatenate([hh[idx], tt[idx]])); rels = np.unique(rr[idx])
                    np.savez(os.path.join(os.environ["KGE_DUMP_UNEXPLAINED"], "unexplained.npz"), row=r0, groups=gs, bh=hh[idx], bt=tt[idx], br=rr[idx],
                             ents=ents, rels=rels, **{"ent_" + kk: start[kk][ents] for kk in start if start[kk].shape[0] == con.entTotal},
                             **{"rel_" + kk: start[kk][rels] for kk in start if start[kk].shape[0] == con.relTotal},
                             g_engine=g_g[k][ents], g_oracle=g_o[k][ents], hinge=hm[gs])
                if unexplained:   # diagnostics on stdout (pytest shows it with the failure)
                    print("UNEXPLAINED", closest_switch_points(hm, model, start, bh, bt, br, B, n, dims, k, sorted(unexplained)[:3]))
                    print("UNEXPLAINED", [dict(row=r, grad_diff_over_scale=float(diff[r].max() / scale),
                                               engine_norm=float(np.linalg.norm(g_g[k][r])), oracle_norm=float(np.linalg.norm(g_o[k][r])))
                                          for r in sorted(unexplained)[:3]])
                assert not unexplained, (name, step, k, sorted(unexplained)[:10],
                                         "rows outside 1e-5 with no |e| < KINK_TOL and no hinge within TIE_TOL in their group")
    assert con.get_stream_states().tolist() == kg.stream_states().tolist()
    parity_report(name, batch=B, steps=steps, loss_relerr=worst["loss"], grad_relerr_other_rows=worst["grad"],
                  grad_rows_outside_1e5=worst["grad_rows"], update_rows_outside_1e5=worst["update_rows"],
                  elements_of_e_within_tol_of_zero=worst["kink_elems"], tol=KINK_TOL, near_tie_hinges=worst["ties"],
                  groups_with_hinge_within_tie_tol=worst["tie_groups"], tie_tol=TIE_TOL)
    assert worst["loss"] <= RTOL, worst
    assert worst["grad"] <= RTOL, worst
    # every outside row was explained above; bound their number too (a flipped element reaches the <= 6 rows of its group)
    assert worst["grad_rows"] <= 6 * (worst["kink_elems"] + (2 + n) * worst["tie_groups"]) and worst["grad_rows"] <= max_outside_rows, worst


def engine(path, model, dim, nbatches, n, alpha, bern=0):
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.prefetch_sampling = False
    con.set_in_path(path); con.set_work_threads(8); con.set_bern(bern); con.set_dimension(dim)
    con.set_nbatches(nbatches); con.set_ent_neg_rate(n); con.set_alpha(alpha); con.set_margin(1.0); con.set_opt_method("SGD")
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    return con


# ------------------------------------------------------------------------------------------------------------------
# configs[2]: WN18RR-shaped TransH dim 200
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbatches,B,path", [(0, 8683, "fp32 atomics + hub copies"), (3, 28945, "float records")])
def test_config3_wn18rr_transh(wn_dir, nbatches, B, path):
    from openkeonspark_amd import _lib
    n, alpha = 1, 0.01
    con = engine(wn_dir, "TransH", 200, nbatches, n, alpha)
    assert (con.entTotal, con.relTotal, con.batch_size) == (40943, 11, B)
    # which accumulation runs is decided by the number of gradient rows of a step (include/kge_mi355.h "float_records_min")
    assert (B * (4 + n) >= (1 << 16)) == (path == "float records")
    kg = oracle.KG(wn_dir, work_threads=8, bern=0)
    kg.set_stream_states(con.get_stream_states())
    orc = oracle.Model("transh", con.entTotal, con.relTotal, 200, 200, margin=1.0, params=con.get_parameters())
    run_steps(con, kg, orc, B, n, alpha, steps=3, name="config3 WN18RR TransH D=200 B=%d (%s)" % (B, path), model="transh", dims=(200, 200))
    _lib.raise_if_error(con.lib)


# ------------------------------------------------------------------------------------------------------------------
# the projecting models with many negatives: the pair-count path (csrc/pairs.hip; chosen by the engine from 5 (TransH) / 3 (TransD)
# negatives and 65 536 entity-side rows per step on)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,graph,nbatches,B", [("TransH", "wn", 8, 10854), ("TransD", "fb", 32, 8503)])
def test_pair_count_path_many_negatives(wn_dir, fb_dir, model, graph, nbatches, B):
    from openkeonspark_amd import _lib
    n, alpha = 25, 0.01
    d = wn_dir if graph == "wn" else fb_dir
    con = engine(d, model, 200, nbatches, n, alpha, bern=1)
    assert con.batch_size == B and B * (2 + n) >= (1 << 16)
    kg = oracle.KG(d, work_threads=8, bern=1)
    kg.set_stream_states(con.get_stream_states())
    orc = oracle.Model(model.lower(), con.entTotal, con.relTotal, 200, 200, margin=1.0, params=con.get_parameters())
    run_steps(con, kg, orc, B, n, alpha, steps=2, name="%s D=200 n=25 B=%d on the %s-shaped graph (pair-count path)" % (model, B, graph),
              model=model.lower(), dims=(200, 200), max_outside_rows=64)   # 26 scored triples per group: a flip reaches more rows
    _lib.raise_if_error(con.lib)


# ------------------------------------------------------------------------------------------------------------------
# configs[3]: FB15k-237-shaped TransR 200 x 200, tilings chosen by the engine
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["three kernels", "group layout", "group layout, separate vector stage"])
@pytest.mark.parametrize("nbatches,B", [(0, 2721), (8, 34014)])
def test_config4_fb15k237_transr(fb_dir, nbatches, B, path):
    """configs[3] at the reference's auto batch and at B = 34 014, device-sampled batches passed as such (what train_step does):
    "three kernels" = the separate project / vector stage / dgrad launches (jobs sorted by relation); "group layout" = groups
    sorted by relation, a group's rows side by side in one 16-row sub-tile, the vector stage inside the projection's epilogue
    (the default from ~64 rows per relation on; forced here at both sizes), or as its own launch.  All against the oracle
    (TransR.py:16-75)."""
    from openkeonspark_amd import _lib
    n, alpha = 1, 0.01
    _lib.lib().kge_set_option(b"transr_groups", 2 if path.startswith("group") else 0)
    _lib.lib().kge_set_option(b"transr_fuse_vec", 0 if path.endswith("separate vector stage") else 1)
    try:
        con = engine(fb_dir, "TransR", 200, nbatches, n, alpha)
        assert (con.entTotal, con.relTotal, con.batch_size) == (14541, 237, B)
        kg = oracle.KG(fb_dir, work_threads=8, bern=0)
        kg.set_stream_states(con.get_stream_states())
        orc = oracle.Model("transr", con.entTotal, con.relTotal, 200, 200, margin=1.0, params=con.get_parameters())
        # B = 34 014: relations with >= 256 rows take the all-output-tiles wgrad, the skewed rest the 32-row tiles (transr.hip)
        run_steps(con, kg, orc, B, n, alpha, steps=2, name="config4 FB15k-237 TransR 200x200 B=%d (%s)" % (B, path), model="transr", dims=(200, 200),
                  sampler_shaped=True)
    finally:
        _lib.lib().kge_set_option(b"transr_groups", 1)
        _lib.lib().kge_set_option(b"transr_fuse_vec", 1)


# ------------------------------------------------------------------------------------------------------------------
# configs[1]: the HEADLINE workload at its bench size, through the exact kernel chain bench.py times
# ------------------------------------------------------------------------------------------------------------------
def bench_engine(fb_dir):
    """bench.py's engine, setting for setting (bench.py make_engine): FB15k-237-shaped graph, TransE dim 200, TF1 Adam 0.001,
    25 negatives (bern), nbatches 8 -> B = 34 014, default counts_min_records, sampling prefetched behind the emit kernel."""
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(1); con.set_dimension(200); con.set_nbatches(8)
    con.set_ent_neg_rate(25); con.set_rel_neg_rate(0); con.set_alpha(0.001); con.set_margin(1.0); con.set_opt_method("Adam")
    con.init()
    con.set_model_and_session(pkg.TransE)
    assert (con.entTotal, con.relTotal, con.batch_size) == (14541, 237, 34014)
    assert con.use_counts and con.prefetch_sampling and not con.sparse_rows
    assert con.batch_size * (3 + 25) >= con.counts_min_records        # the sign-count pipeline, not the fused atomic kernel
    return con


def test_config2_fb15k237_transe_adam_n25_bench_size(fb_dir):
    """BASELINE configs[1] as bench.py runs it (B = 34 014 x 25 negatives, dim 200, TF1-semantics Adam, the next batch's sampler
    riding in the step): three `con.train_step()` calls -- the exact kernel chain bench.py times -- each against ONE oracle Adam
    step restarted from the engine's own tables and Adam slots (TransE.py:26-51, distribute_training.py:95-101).  Per step:
      * the batch the engine trained on is the oracle sampler's batch bit for bit; loss to 1e-5;
      * the gradient -- read back from Adam's first moment, m1 = b1 m0 + (1 - b1) g -- to 1e-5 of its largest element on EVERY
        row.  A row outside that is not excused by set membership: its difference must lie, element by element, inside what the
        switch points OF ITS OWN SLOTS can produce (parity_util.transe_row_radius: a sign taken the other way at an |e| < KINK_TOL
        element, a hinge taken the other way within TIE_TOL -- the gradient is linear in both).  Such rows are counted
        (`rows_excused`, bounded at 100 per 3 steps) next to the size of the set round 3 excused (`rows_in_kink_set`).  The
        interval is centred on the fp64 evaluation of the formula for that row, not on the fp32 oracle: the oracle adds a hub
        row's thousands of contributions one by one in fp32 (error reported: `worst_oracle_fp32_error_on_excused_rows`);
      * the second moment on every row, against the gradient that row was checked to have (the oracle's; for an excused row the
        engine's own, which the interval check has just bounded);
      * every element of the parameter update must be one that a gradient within 1e-5 of that checked gradient produces through
        Adam (parity_util.adam_update_explained) -- no row is skipped."""
    import torch
    from parity_util import check_transe_adam_step, new_adam_step_totals
    con = bench_engine(fb_dir)
    B, n, alpha, b1, b2, eps = 34014, 25, 0.001, 0.9, 0.999, 1e-8
    names = con.trainModel.table_names
    kg = oracle.KG(fb_dir, work_threads=8, bern=1)
    kg.set_stream_states(con.get_stream_states())
    orc = oracle.Model("transe", con.entTotal, con.relTotal, 200, 200, margin=1.0, params=con.get_parameters())
    tot = new_adam_step_totals()
    for step in range(3):
        p0 = con.get_parameters()
        m0 = {k: con._adam_m[i].cpu().numpy() for i, k in enumerate(names)}
        v0 = {k: con._adam_v[i].cpu().numpy() for i, k in enumerate(names)}
        orc.params = {k: v.copy() for k, v in p0.items()}
        orc.adam_m = {k: v.copy() for k, v in m0.items()}
        orc.adam_v = {k: v.copy() for k, v in v0.items()}
        orc.step = con.global_step
        bh, bt, br, _ = kg.sampling(B, n, 0)
        hm = orc.hinge_margins(bh, bt, br, B, n)
        loss_o, g_o = orc.grad(bh, bt, br, B, n, nthreads=8)
        lr_t = oracle.adam_lr_t(alpha, b1, b2, con.global_step + 1)
        orc.apply_adam(g_o, alpha, b1, b2, eps)
        loss_g = con.train_step()
        used = con._dev_batch[con._slot ^ 1].cpu().numpy()       # (the other slot already holds the prefetched next batch)
        assert np.array_equal(used[0], bh) and np.array_equal(used[1], bt) and np.array_equal(used[2], br), step
        tot["loss"] = max(tot["loss"], abs(loss_g - loss_o) / abs(loss_o))
        assert abs(loss_g - loss_o) <= RTOL * abs(loss_o), (step, loss_g, loss_o)
        p1 = con.get_parameters()
        m1 = {k: con._adam_m[i].cpu().numpy() for i, k in enumerate(names)}
        v1 = {k: con._adam_v[i].cpu().numpy() for i, k in enumerate(names)}
        check_transe_adam_step(tot, step, p0, m0, v0, p1, m1, v1, g_o, orc.params, bh, bt, br, B, n, hm, float(lr_t), b1, b2, eps,
                               RTOL, KINK_TOL, TIE_TOL)
    assert con.global_step == 3 and orc.step == 3
    parity_report("config2 FB15k-237 TransE D=200 Adam n=25 B=34014 (bench chain, prefetch on)", steps=3, loss_relerr=tot["loss"],
                  grad_relerr_fully_checked_rows=tot["grad"], rows_excused=tot["rows_excused"], rows_in_kink_set=tot["rows_in_kink_set"],
                  rows_fully_checked=tot["rows_fully_checked"], rows_excused_needing_a_switch_point=tot["rows_excused_needing_a_switch_point"],
                  worst_excused_diff_over_its_radius=tot["worst_excused_over_radius"],
                  worst_oracle_minus_fp64_on_excused_rows_incl_its_own_switch_choices=tot["worst_oracle_fp32_error_on_excused_rows"],
                  worst_engine_error_vs_fp64_beyond_radius_on_excused_rows=tot["worst_engine_error_vs_fp64_on_excused_rows"],
                  elements_of_e_within_tol_of_zero=tot["kink_elems"], hinges_within_tie_tol=tot["tie_hinges"], kink_tol=KINK_TOL, tie_tol=TIE_TOL,
                  adam_elements_beyond_1e3_of_a_step_all_explained=tot["amplified"], worst_in_steps=tot["worst_steps"],
                  worst_adam_gain=tot["worst_gain"], v_relerr=tot["v"])
    assert tot["grad"] <= RTOL and tot["rows_excused"] <= 100, tot


def test_config2_loss_trajectory_20_steps(fb_dir):
    """The same engine for 20 steps (prefetch on) against the oracle in two forms, on the oracle sampler's batches:
    (a) RESTARTED: every step the oracle starts from the engine's tables and Adam slots -- each step's loss within 2e-5
        (observed 0: same hinge sum), for all 20 steps;
    (b) INDEPENDENT: a second oracle model runs its own 20 Adam steps from the shared initial tables.  The two runs drift:
        in the first steps Adam moves an element by ~lr_t * sign(g) whatever |g| is, so a kink flip in a near-cancelling gradient
        element moves that parameter by up to 2 * 0.001, and the differences compound (observed on MI355X: 1e-7, 0, 1e-7, 4e-6,
        1e-5 ... 5e-5 at step 20).  Bounded at 2e-4 and reported; (a) is the per-step parity statement."""
    con = bench_engine(fb_dir)
    B, n, alpha = 34014, 25, 0.001
    names = con.trainModel.table_names
    kg = oracle.KG(fb_dir, work_threads=8, bern=1)
    kg.set_stream_states(con.get_stream_states())
    start = con.get_parameters()
    restarted = oracle.Model("transe", con.entTotal, con.relTotal, 200, 200, margin=1.0, params=start)
    independent = oracle.Model("transe", con.entTotal, con.relTotal, 200, 200, margin=1.0, params=start)
    got, want_r, want_i = np.zeros(20), np.zeros(20), np.zeros(20)
    for step in range(20):
        restarted.params = con.get_parameters()
        restarted.adam_m = {k: con._adam_m[i].cpu().numpy() for i, k in enumerate(names)}
        restarted.adam_v = {k: con._adam_v[i].cpu().numpy() for i, k in enumerate(names)}
        restarted.step = con.global_step
        bh, bt, br, _ = kg.sampling(B, n, 0)
        want_r[step] = restarted.loss(bh, bt, br, B, n)
        want_i[step] = independent.adam_step(bh, bt, br, B, n, alpha, nthreads=8)
        got[step] = con.train_step()
    rel_r = np.abs(got - want_r) / np.abs(want_r)
    rel_i = np.abs(got - want_i) / np.abs(want_i)
    parity_report("config2 loss trajectory, 20 steps", worst_relerr_restarted_oracle=float(rel_r.max()),
                  worst_relerr_independent_oracle_run=float(rel_i.max()), drift_by_step=[float("%.2e" % x) for x in rel_i],
                  first_loss=float(got[0]), last_loss=float(got[-1]))
    assert (rel_r <= 2e-5).all(), rel_r.tolist()
    assert (rel_i <= 2e-4).all(), rel_i.tolist()
    assert got[-1] < got[0]          # it trains
    # the rng streams: the engine has drawn one batch more than it trained on (the prefetched one)
    kg.sampling(B, n, 0)
    assert con.get_stream_states().tolist() == kg.stream_states().tolist()


def test_config1_fb15k237_transe_auto_batch(fb_dir):
    """configs[0]: TransE dim 100, SGD, 1 negative, the reference's auto batch 2 721 (fused fp32-atomic kernel)."""
    n, alpha = 1, 0.01
    con = engine(fb_dir, "TransE", 100, 0, n, alpha)
    assert con.batch_size == 2721 and con.nbatches == 100
    kg = oracle.KG(fb_dir, work_threads=8, bern=0)
    kg.set_stream_states(con.get_stream_states())
    orc = oracle.Model("transe", con.entTotal, con.relTotal, 100, 100, margin=1.0, params=con.get_parameters())
    run_steps(con, kg, orc, 2721, n, alpha, steps=3, name="config1 FB15k-237 TransE D=100 B=2721", model="transe", dims=(100, 100))


# ------------------------------------------------------------------------------------------------------------------
# configs[4]: sparse-row TransE dim 512 against the oracle at a reduced entity count
# ------------------------------------------------------------------------------------------------------------------
def config5_graph(E=200_000, R=500, triples=1_000_000, seed=5):
    rng = np.random.default_rng(seed)
    return E, R, rng.integers(0, E, triples), rng.integers(0, E, triples), rng.integers(0, R, triples)


def test_config5_sparse_step_matches_oracle():
    """200 k entities x dim 512 (0.4 GB table), B = 50 000, n = 1, SGD with lr = 1 so that p_before - p_after IS the
    applied gradient: int8 records -> radix sort -> fused segmented sum + row SGD, against the oracle's dense gradient."""
    import openkeonspark_amd as pkg
    E, R, h, t, r = config5_graph()
    D, n = 512, 1
    con = pkg.Config()
    con.prefetch_sampling = False
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(D); con.set_ent_neg_rate(n); con.set_alpha(1.0)
    con.set_opt_method("SGD"); con.set_nbatches(20)
    con.sparse_rows = True
    con.init_from_arrays(E, R, h, t, r)
    con.set_model_and_session(pkg.TransE)
    B = con.batch_size
    assert B == 50_000 and con.sparse_rows and con._grads == []
    kg = oracle.KG(arrays=(E, R, h, t, r, 0), work_threads=8, bern=1)
    kg.set_stream_states(con.get_stream_states())
    params = con.get_parameters()
    orc = oracle.Model("transe", E, R, D, D, margin=1.0, params=params)
    from parity_util import transe_switch_points, transe_row_radius, transe_row_grad_fp64, switch_point_rows
    rows_excused = rows_in_kink_set = kink_elems = tie_hinges = 0
    worst_over_radius = 0.0
    for step in range(2):
        before = con.get_parameters()
        states = con.get_stream_states()
        dev, bh, bt, br = check_sampled_batch(con, kg, B, n)
        con.lib.kge_set_stream_states(states.ctypes.data, 8)        # rewind: train_step draws the same batch again
        orc.params = {k: v.copy() for k, v in before.items()}
        hm = orc.hinge_margins(bh, bt, br, B, n)
        loss_o, g_o = orc.grad(bh, bt, br, B, n, nthreads=8)
        loss_g = con.train_step()
        assert abs(loss_g - loss_o) <= RTOL * abs(loss_o), (loss_g, loss_o)
        after = con.get_parameters()
        kinks, ties, w_max = transe_switch_points(before, bh, bt, br, B, n, hm, KINK_TOL, TIE_TOL, chunk=20_000)
        kink_elems += len(kinks); tie_hinges += len(ties)
        in_set = switch_point_rows(kinks, ties, bh, bt, br, B, n)
        for k in g_o:
            g_g = before[k].astype(np.float64) - after[k].astype(np.float64)
            quantum = np.abs(before[k]).max() * 2.0 ** -23
            scale = np.abs(g_o[k]).max()
            diff = np.abs(g_g - g_o[k])
            bad_rows = np.nonzero((diff > RTOL * scale + quantum).any(1))[0]
            rows_excused += len(bad_rows); rows_in_kink_set += len(in_set[k])
            # a row outside 1e-5 is not excused by belonging to a set: it is re-judged against the fp64 evaluation of the formula for
            # that row and must lie, element by element, within 1e-5 of it plus what the switch points of ITS OWN slots can produce (a
            # sign taken the other way where |e| < KINK_TOL, a hinge within TIE_TOL)
            for row in bad_rows.tolist():
                rad = transe_row_radius(before, bh, bt, br, B, n, k, row, kinks, ties, w_max)
                d64 = np.abs(g_g[row] - transe_row_grad_fp64(before, bh, bt, br, B, n, k, row, hm))
                over = d64 - (rad + RTOL * scale + quantum)
                assert (over <= 0).all(), (step, k, row, float(over.max()), "outside 1e-5 by more than its own switch points allow")
                worst_over_radius = max(worst_over_radius, float(((d64 - RTOL * scale - quantum) / np.where(rad > 0, rad, np.inf)).max()))
    parity_report("config5 sparse rows E=200k D=512 B=50000", rows_excused=rows_excused, rows_in_kink_set=rows_in_kink_set,
                  rows_fully_checked=2 * (E + R) - rows_excused, worst_excused_diff_over_its_radius=worst_over_radius,
                  elements_of_e_within_tol_of_zero=kink_elems, tol=KINK_TOL, hinges_within_tie_tol=tie_hinges, tie_tol=TIE_TOL)
    assert rows_excused <= 60, (rows_excused, kink_elems, tie_hinges)


def test_config5_full_size_properties():
    """configs[4] at its full size on one GPU -- 50 M entities x dim 512 (102 GB table in HBM), 500 M training triples
    indexed on the device, B = 1 050 420 (nbatches 476), n = 1 -- through properties that need no oracle: head and tail
    contributions cancel column by column over the entity rows, the touched-row list is strictly increasing, untouched
    rows keep their bits, the loss of a random-init model sits near the margin.  Needs ~135 GB of free HBM and ~30 GB of
    host memory for the triple arrays; skipped otherwise."""
    import torch
    import openkeonspark_amd as pkg
    free, _ = torch.cuda.mem_get_info()
    try:
        import psutil
        host_free = psutil.virtual_memory().available
    except Exception:
        host_free = 0
    if free < 150 * (1 << 30) or host_free < 40 * (1 << 30):
        pytest.skip("needs 150 GB of free HBM and 40 GB of host memory (have %.0f / %.0f GB)" % (free / 2**30, host_free / 2**30))
    E, R, D, n_tr = 50_000_000, 1000, 512, 500_000_000
    rng = np.random.default_rng(5)
    h = rng.integers(0, E, n_tr, dtype=np.int64); t = rng.integers(0, E, n_tr, dtype=np.int64); r = rng.integers(0, R, n_tr, dtype=np.int64)
    con = pkg.Config()
    con.set_work_threads(8); con.set_bern(1); con.set_dimension(D); con.set_ent_neg_rate(1); con.set_rel_neg_rate(0)
    con.set_alpha(0.01); con.set_opt_method("SGD"); con.set_nbatches(476)
    con.sparse_fused = False          # the complete compact count image is inspected below
    con.init_from_arrays(E, R, h, t, r)
    del h, t, r
    con.set_model_and_session(pkg.TransE)
    assert con.sparse_rows and con.batch_size == 1_050_420 and con._grads == []
    ent = con._tables[0]
    probe = torch.arange(0, E, 9973, device=ent.device)
    before = ent[probe].clone()
    loss = con.train_step()
    assert 0.9 < loss < 1.3, loss
    rows, counts = con.sparse_row_gradients()
    assert bool((rows[1:] > rows[:-1]).all()) and int(rows.min()) >= 0 and int(rows.max()) < E + R
    is_ent = rows < E
    assert int(counts[is_ent].sum(dim=0).abs().max()) == 0
    assert int(counts[~is_ent].abs().sum()) > 0
    touched = torch.zeros(E, dtype=torch.bool, device=ent.device)
    touched[rows[is_ent].long()] = True
    moved = (ent[probe] != before).any(dim=1)
    assert not bool((moved & ~touched[probe]).any())
    assert bool(moved.any())
    parity_report("config5 full size 50M x 512 / 500M triples", loss=loss, touched_rows=int(rows.numel()),
                  hbm_GB=torch.cuda.max_memory_allocated() / 1e9)
    con.sparse_fused = True
    loss2 = con.train_step()            # the production (fused reduce+apply) step runs too
    assert 0.9 < loss2 < 1.3
