"""HIP forward/backward + optimiser kernels through the C ABI versus the CPU oracle.

Tolerance: 1e-5 relative (BASELINE.json north_star) measured on each table's gradient / parameter
scale; the loss to 1e-5 relative.  The oracle itself is checked against fp64 autograd in
tests/test_oracle_models.py."""
import numpy as np
import pytest

from conftest import parity_report
from oracle import oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-5
# carve-outs, each reported through parity_report with the count actually seen (bounds = the counts observed on MI355X
# when they were last revised; a regression inside a bound still shows in gpurun_out/parity_counts.jsonl)
SIGN_FLIP_ROWS_BOUND = 2     # rows of a table whose gradient carries one sign flip of an element within rounding of zero (observed: <= 2)
EXACT_COUNT_ROWS_BOUND = 0   # rows of the integer count image touched by a hinge within rounding of its switch point (observed: 0)


def relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def seed_of(*key):
    """Deterministic seed (Python's hash() of a tuple with strings changes from process to process)."""
    import zlib
    return zlib.crc32(repr(key).encode())


def batch_without_ties(orc, make, B, N, tries=20):
    """A batch none of whose hinges is within 1e-4 of zero.  max(x,0) switches a whole gradient row on
    x >= 0; when x is within fp32 rounding of 0 the oracle and the kernels may disagree on the switch
    (seen once in ~40 random batches), which is a property of the loss, not an error of either."""
    for _ in range(tries):
        bh, bt, br = make()
        if np.abs(orc.hinge_margins(bh, bt, br, B, N)).min() > 1e-4:
            return bh, bt, br
    raise AssertionError("no tie-free batch found")


def rand_batch(rng, E, R, B, n, nr, foreign=0.0, distinct=False):
    """distinct=True: a corrupted slot always differs from the positive's (as the reference's filtered
    sampler guarantees); otherwise a negative may coincide with its positive by chance."""
    N = n + nr
    h = np.zeros(B * (1 + N), np.int64); t = h.copy(); r = h.copy()
    h[:B] = rng.integers(0, E, B); t[:B] = rng.integers(0, E, B); r[:B] = rng.integers(0, R, B)
    for k in range(N):
        s = slice(B * (k + 1), B * (k + 2))
        h[s], t[s], r[s] = h[:B], t[:B], r[:B]
        if k < n:
            side = rng.random(B) < 0.5
            new = rng.integers(0, E, B)
            if distinct:
                old = np.where(side, h[:B], t[:B])
                new = (old + 1 + rng.integers(0, E - 1, B)) % E
            h[s] = np.where(side, new, h[:B]); t[s] = np.where(side, t[:B], new)
        else:
            r[s] = rng.integers(0, R, B)
            if distinct:
                r[s] = (r[:B] + 1 + rng.integers(0, R - 1, B)) % R
    if foreign > 0:  # arbitrary negatives: any slot combination may differ, or none
        m = rng.random(B * N) < foreign
        idx = np.nonzero(m)[0] + B
        h[idx] = rng.integers(0, E, len(idx)); t[idx] = rng.integers(0, E, len(idx)); r[idx] = rng.integers(0, R, len(idx))
        same = idx[: len(idx) // 4]
        h[same], t[same], r[same] = h[(same - B) % B], t[(same - B) % B], r[(same - B) % B]
    return h, t, r


def make_engine(model, E, R, D, n, nr, margin=1.0, opt="SGD", alpha=0.01, params=None, Dr=None, use_counts=True):
    """A Config over a dummy dataset (the model ops only need the totals) with given parameters."""
    from openkeonspark_amd.Config import Config
    import openkeonspark_amd as pkg
    con = Config()
    con.use_counts = use_counts
    con.counts_min_records = 0   # small test batches must still exercise the count pipeline
    con.set_ent_neg_rate(n); con.set_rel_neg_rate(nr); con.set_margin(margin)
    con.set_opt_method(opt); con.set_alpha(alpha)
    if Dr is None:
        con.set_dimension(D)
    else:
        con.set_ent_dimension(D); con.set_rel_dimension(Dr); con.hidden_size = D
    hh = np.arange(40) % E
    con.init_from_arrays(E, R, hh, (hh + 1) % E, hh % R)
    con.set_model_and_session(getattr(pkg, {"transe": "TransE", "transh": "TransH", "transd": "TransD", "transr": "TransR"}[model]))
    if params is not None:
        con.set_parameters(params)
    return con


CASES = [("transe", 300, 11, 16), ("transe", 300, 11, 100), ("transe", 200, 7, 200), ("transe", 100, 5, 512),
         ("transe", 64, 5, 50), ("transe", 64, 5, 7),
         ("transh", 300, 11, 24), ("transh", 200, 7, 200), ("transh", 64, 5, 100),
         ("transd", 300, 11, 20), ("transd", 200, 7, 200), ("transd", 64, 5, 100)]

TRANSR_CASES = [(120, 9, 12, 8), (150, 7, 200, 200), (90, 5, 64, 100), (60, 4, 33, 50)]


@pytest.fixture(params=["v3-bf16x3", "v2-16x16x4", "v2-wgrad-all-tiles", "v2-wgrad-32x32x2", "v1-32x32x2"])
def transr_tiles(request):
    """The MFMA tilings of the TransR projections: 128-row tiles with the fp32 products formed as six bf16 term products of an
    exact three-term split (the default for dims <= 208, multiples of 4), the same tiles on the fp32 MFMA (16x16x4; its
    all-output-tiles wgrad is only chosen for well-filled buckets, so it is forced here) and 32x32x2 / 32-row tiles."""
    from openkeonspark_amd import _lib
    L = _lib.lib()
    L.kge_set_option(b"transr_bf16x3", 1 if request.param == "v3-bf16x3" else 0)
    L.kge_set_option(b"transr_v1", {"v1-32x32x2": 1, "v2-wgrad-all-tiles": 2, "v2-wgrad-32x32x2": 3, "v2-16x16x4": 0, "v3-bf16x3": 0}[request.param])
    yield request.param
    L.kge_set_option(b"transr_v1", 0)
    L.kge_set_option(b"transr_bf16x3", 1)


@pytest.mark.parametrize("E,R,De,Dr", TRANSR_CASES + [(80, 3, 208, 16), (70, 6, 100, 208)])
@pytest.mark.parametrize("n,nr,foreign", [(1, 0, 0.0), (5, 0, 0.0), (2, 1, 0.0), (3, 1, 0.3), (4, 0, 0.3)])
def test_transr_forward_backward_matches_oracle(E, R, De, Dr, n, nr, foreign, transr_tiles):
    """Relation-bucketed fp32-MFMA TransR (projection, dgrad, wgrad) against the oracle."""
    import torch
    rng = np.random.default_rng(seed_of(De, Dr, n, nr))
    B = 301
    params = oracle.init_params(oracle.TRANSR, E, R, De, Dr, seed=4)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model("transr", E, R, De, Dr, margin=0.9, negative_rel=nr, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, nr, foreign), B, n + nr)
    loss_o, g_o = orc.grad(bh, bt, br, B, n + nr)
    con = make_engine("transr", E, R, De, n, nr, margin=0.9, params=params, Dr=Dr)
    dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
    con.forward_backward(dev, B, B, B * (n + nr))
    torch.cuda.synchronize()
    loss_g = float(con._loss.item())
    g_g = con.get_gradients()
    assert abs(loss_g - loss_o) <= RTOL * abs(loss_o), (loss_g, loss_o)
    check_gradients_with_kinks("transr", params, bh, bt, br, B, n + nr, (De, Dr), nr, orc, g_g, g_o)


@pytest.fixture(params=["atomic", "records-bucket", "records-sort", "pairs", "pairs-sort"])
def grad_path(request):
    """The accumulations of the gradient rows: memory-side fp32 atomics; float records ordered by the two-level counting
    sort / by rocPRIM's radix sort and summed by segments; and, for TransH / TransD at widths that are multiples of 4 up to
    256, int8 sign records keyed by (entity, relation) with the backward applied once per pair (the default on large steps;
    other models and widths fall through to the float records), their keys ordered by the counting sort or (key spaces
    beyond 4.2 M pairs; forced here) by rocPRIM's radix sort."""
    from openkeonspark_amd import _lib
    L = _lib.lib()
    L.kge_set_option(b"float_records", 0 if request.param == "atomic" else 1)
    L.kge_set_option(b"float_records_min", 0)
    L.kge_set_option(b"counts_force_sort", 1 if request.param in ("records-sort", "pairs-sort") else 0)
    L.kge_set_option(b"pair_counts", 1 if request.param in ("pairs", "pairs-sort") else 0)
    L.kge_set_option(b"pair_counts_min_neg", 1)
    yield request.param
    L.kge_set_option(b"float_records", 1)
    L.kge_set_option(b"float_records_min", 1 << 16)
    L.kge_set_option(b"counts_force_sort", 0)
    L.kge_set_option(b"pair_counts", 1)
    L.kge_set_option(b"pair_counts_min_neg", 0)


@pytest.mark.parametrize("model,E,R,D", CASES)
@pytest.mark.parametrize("n,nr,foreign", [(1, 0, 0.0), (5, 0, 0.0), (2, 1, 0.0), (3, 1, 0.3)])
def test_forward_backward_matches_oracle(model, E, R, D, n, nr, foreign, grad_path):
    import torch
    rng = np.random.default_rng(seed_of(model, D, n, nr))
    B = 257
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=3)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model(model, E, R, D, D, margin=0.9, negative_rel=nr, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, nr, foreign), B, n + nr)
    loss_o, g_o = orc.grad(bh, bt, br, B, n + nr)
    con = make_engine(model, E, R, D, n, nr, margin=0.9, params=params)
    dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
    con.forward_backward(dev, B, B, B * (n + nr))
    torch.cuda.synchronize()
    loss_g = float(con._loss.item())
    g_g = con.get_gradients()
    assert abs(loss_g - loss_o) <= RTOL * abs(loss_o), (loss_g, loss_o)
    check_gradients_with_kinks(model, params, bh, bt, br, B, n + nr, D, nr, orc, g_g, g_o)


def check_gradients_with_kinks(model, params, bh, bt, br, B, N, D, nr, orc, g_g, g_o):
    """Every row within RTOL of the oracle's -- except rows reached by a KINK: an element of e = h^ + r^ - t^ within 1e-7 of zero
    (fp64) gets sign +1 from one correct fp32 evaluation and -1 or 0 from another (batch_without_ties keeps the hinges away from
    their switch points, the kinks it cannot avoid: seen with |e| = 1.5e-9).  Such rows are not waved through: for TransE the
    difference must lie inside what the kink elements of the row's own slots allow (parity_util.transe_row_radius); for the
    projecting models, whose rows see a kink through the projection, it is bounded by the flips' total weight."""
    from parity_util import transe_switch_points, transe_row_radius
    from torch_ref import near_kink_rows
    bad = {k: np.nonzero((np.abs(g_g[k] - g_o[k]) > RTOL * (np.abs(g_o[k]).max() + 1e-30)).reshape(g_o[k].shape[0], -1).any(1))[0] for k in g_o}
    if not any(len(v) for v in bad.values()):
        return
    De, Dr = D if isinstance(D, tuple) else (D, D)
    kink, n_el = near_kink_rows(model, params, bh, bt, br, B, N, De, Dr, tol=1e-7, negative_rel=nr)
    assert n_el > 0, ("gradient rows outside 1e-5 and no element of e near zero", {k: v[:5].tolist() for k, v in bad.items()})
    parity_report("forward_backward kink rows", model=model, dim=list(D) if isinstance(D, tuple) else D, kink_elements=n_el, rows_outside={k: len(v) for k, v in bad.items()})
    if model == "transe" and nr == 0:
        hm = orc.hinge_margins(bh, bt, br, B, N)
        kinks, ties, w_max = transe_switch_points(params, bh, bt, br, B, N, hm, 1e-7, 0.0)
        for k in g_o:
            scale = np.abs(g_o[k]).max()
            for row in bad[k].tolist():
                rad = transe_row_radius(params, bh, bt, br, B, N, k, row, kinks, ties, w_max)
                assert (np.abs(g_g[k][row].astype(np.float64) - g_o[k][row]) <= rad + RTOL * scale).all(), (k, row)
        return
    unit = 1.0 / (B * N)
    for k in g_o:
        assert set(bad[k].tolist()) <= kink[k], (k, sorted(set(bad[k].tolist()) - kink[k])[:5])
        assert len(bad[k]) <= 8 * n_el
        # one flipped sign moves dL/d(normalised vector) by at most 2 (1 + N) / (B N) per element; the backward through normalise and
        # projection does not amplify it beyond a few 1 / |row|
        if len(bad[k]):    # (the smallest norm among the vector tables: a matrix row takes the flip through an entity row)
            min_norm = min(np.sqrt((np.asarray(params[t], dtype=np.float64) ** 2).sum(1)).min() for t in ("ent_embeddings", "rel_embeddings"))
            assert np.abs(g_g[k][bad[k]] - g_o[k][bad[k]]).max() <= n_el * 8 * (1 + N) * unit / max(min_norm, 1e-6), k


def adam_step_explained(con, orc, model, bh, bt, br, B, n, alpha, dims, step_index, tag):
    """ONE Adam step on a fed batch, the oracle restarted from the engine's own tables and Adam slots (so the comparison
    is of one forward / backward / update on identical inputs, distribute_training.py:95-101).  Loss to 2e-5; every
    element of the parameter update must be one that a gradient within 1e-5 of the oracle's can produce through Adam's
    lr_t * m / (sqrt(v) + eps) (tests/parity_util.adam_update_explained); rows of a group with an element of e within
    rounding of zero or a hinge at its switch point -- where the GRADIENT itself legitimately differs -- are set aside and
    counted.  Returns the report counts."""
    from parity_util import adam_update_explained
    from torch_ref import near_kink_rows
    names = con.trainModel.table_names
    p0 = con.get_parameters()
    m0 = {k: con._adam_m[i].cpu().numpy() for i, k in enumerate(names)}
    v0 = {k: con._adam_v[i].cpu().numpy() for i, k in enumerate(names)}
    orc.params = {k: v.copy() for k, v in p0.items()}
    orc.adam_m = {k: v.copy() for k, v in m0.items()}
    orc.adam_v = {k: v.copy() for k, v in v0.items()}
    orc.step = step_index
    hm = np.abs(orc.hinge_margins(bh, bt, br, B, n))
    loss_o, g_o = orc.grad(bh, bt, br, B, n)
    lr_t = float(oracle.adam_lr_t(alpha, 0.9, 0.999, step_index + 1))
    orc.apply_adam(g_o, alpha)
    loss_g = con.train_step(bh, bt, br, None)
    assert abs(loss_g - loss_o) <= 2e-5 * abs(loss_o), (tag, step_index, loss_g, loss_o)
    p1 = con.get_parameters()
    kink, n_el = near_kink_rows(model, p0, bh, bt, br, B, n, dims[0], dims[1], tol=1e-6)
    groups = np.nonzero((hm < 1e-5).any(1))[0]
    if len(groups):
        idx = (groups[:, None] + B * np.arange(n + 1)[None, :]).ravel()
        ents = set(np.asarray(bh)[idx].tolist()) | set(np.asarray(bt)[idx].tolist()); rels = set(np.asarray(br)[idx].tolist())
        for k in kink:
            kink[k] |= ents if k in ("ent_embeddings", "ent_transfer") else rels
    out = dict(amplified=0, worst_steps=0.0, worst_gain=0.0, kink_elems=n_el, tie_groups=len(groups), set_aside_rows=0)
    for k in names:
        W = p0[k].shape[1]
        rep = adam_update_explained(p0[k], m0[k], v0[k], g_o[k], p1[k].astype(np.float64) - p0[k],
                                    orc.params[k].astype(np.float64) - p0[k], lr_t, grad_rtol=RTOL, skip_rows=kink[k])
        assert rep["unexplained"].size == 0, (tag, step_index, k, [(int(j // W), int(j % W)) for j in rep["unexplained"][:8]],
                                               "Adam update elements that no gradient within 1e-5 of the oracle's explains")
        out["amplified"] += rep["amplified"]; out["set_aside_rows"] += len(kink[k])
        out["worst_steps"] = max(out["worst_steps"], rep["worst_steps"]); out["worst_gain"] = max(out["worst_gain"], rep["worst_gain"])
    return out


@pytest.mark.parametrize("model", ["transe", "transh", "transd", "transr"])
@pytest.mark.parametrize("opt", ["SGD", "Adam"])
def test_training_steps_match_oracle(model, opt, grad_path):
    """Several optimiser steps on fed batches: parameters, Adam slots and losses track the oracle."""
    rng = np.random.default_rng(11)
    E, R, D, B, n = 120, 6, 64, 128, 3
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=5)
    alpha = 0.05 if opt == "SGD" else 0.01
    orc = oracle.Model(model, E, R, D, D, margin=1.0, params=params)
    con = make_engine(model, E, R, D, n, 0, margin=1.0, opt=opt, alpha=alpha, params=params)
    tot = dict(amplified=0, worst_steps=0.0, worst_gain=0.0, kink_elems=0, tie_groups=0, set_aside_rows=0)
    for step in range(5):
        bh, bt, br = rand_batch(rng, E, R, B, n, 0)
        if opt == "Adam":
            rep = adam_step_explained(con, orc, model, bh, bt, br, B, n, alpha, (D, D), step, "%s-%s" % (model, grad_path))
            for kk in rep:
                tot[kk] = max(tot[kk], rep[kk]) if kk.startswith("worst") else tot[kk] + rep[kk]
            continue
        lo = orc.sgd_step(bh, bt, br, B, n, alpha)
        lg = con.train_step(bh, bt, br, None)
        assert abs(lg - lo) <= 2e-5 * abs(lo), (step, lg, lo)
        got = con.get_parameters()
        for k in orc.params:
            # compare the accumulated UPDATE (the parameters' own magnitude would hide it)
            du_o = orc.params[k].astype(np.float64) - params[k]
            du_g = got[k].astype(np.float64) - params[k]
            assert np.abs(du_g - du_o).max() <= 1e-4 * np.abs(du_o).max(), (step, k)
    if opt == "Adam":
        parity_report("training_steps_match_oracle[%s-Adam-%s]" % (model, grad_path),
                      adam_elements_beyond_1e3_of_a_step_all_explained=tot["amplified"], worst_in_steps=tot["worst_steps"],
                      worst_adam_gain=tot["worst_gain"], rows_set_aside_as_kink_or_tie=tot["set_aside_rows"],
                      elements_of_e_within_1e6_of_zero=tot["kink_elems"], groups_with_hinge_within_1e5=tot["tie_groups"])
    assert con.global_step == 5
    for g in con.get_gradients().values():
        assert not g.any()  # accumulators are re-zeroed by the update kernels


@pytest.fixture(params=["bucket", "sort"])
def reducer(request):
    """Both reductions of the sign-count records: LDS buckets (small tables) and sort + segmented sum."""
    from openkeonspark_amd import _lib
    L = _lib.lib()
    L.kge_set_option(b"counts_force_sort", 1 if request.param == "sort" else 0)
    yield request.param
    L.kge_set_option(b"counts_force_sort", 0)


@pytest.mark.parametrize("D", [16, 50, 100, 200, 512])
@pytest.mark.parametrize("n,nr,foreign", [(1, 0, 0.0), (25, 0, 0.0), (3, 2, 0.0), (4, 1, 0.3)])
def test_transe_sign_count_gradient_matches_oracle(D, n, nr, foreign, reducer):
    """TransE integer sign-count path (int8 records -> sort -> segmented sum -> per-row normalise
    backward): the gradient it applies, read back as p_before - p_after of an SGD step with lr = 1,
    against the oracle's dense gradient.  Few rows / many records per row: runs that span chunks,
    hub rows, and non sampler-shaped negatives (residual path)."""
    rng = np.random.default_rng(seed_of(D, n, nr))
    E, R, B = (97, 5, 700) if D != 100 else (1500, 9, 700)   # 1500 rows: several rows per bucket
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=6)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model("transe", E, R, D, D, margin=0.8, negative_rel=nr, params=params)
    con = make_engine("transe", E, R, D, n, nr, margin=0.8, opt="SGD", alpha=1.0, params=params)
    assert con.use_counts
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, nr, foreign), B, n + nr)
    loss_o, g_o = orc.grad(bh, bt, br, B, n + nr)
    loss_g = con.train_step(bh, bt, br, None)
    assert abs(loss_g - loss_o) <= RTOL * abs(loss_o)
    got = con.get_parameters()
    unit = 1.0 / (B * (n + nr))
    flips = {}
    for k in g_o:
        g_g = params[k].astype(np.float64) - got[k].astype(np.float64)
        # p - 1.0*g is rounded to fp32 at the parameter's magnitude: allow that quantum on top of 1e-5
        quantum = np.abs(params[k]).max() * 2.0 ** -23
        diff = np.abs(g_g - g_o[k])
        bad = diff > RTOL * np.abs(g_o[k]).max() + quantum
        # d|e|/de jumps at e = 0: an element of h^+r^-t^ within fp32 rounding of zero (a few per million)
        # gets sign +1 from one evaluation order and -1 from another (the kernel forms e with one fma, the
        # oracle with add/sub).  One such element changes the integer count of ONE entry of a row by 2,
        # i.e. that row's gradient by at most 2*unit/|row| (through the entry itself and, much less,
        # through the normalise-backward's dot product).  A few rows may carry such a flip; every other
        # row must agree to 1e-5.
        min_norm = np.sqrt((params[k].astype(np.float64) ** 2).sum(1)).min()
        bad_rows = np.nonzero(bad.any(1))[0]
        flips[k] = len(bad_rows)
        assert len(bad_rows) <= SIGN_FLIP_ROWS_BOUND, (k, len(bad_rows))
        assert diff.max() <= 2.05 * unit / min_norm + RTOL * np.abs(g_o[k]).max() + quantum, (k, diff.max())
    parity_report("transe_sign_count_gradient[D=%d n=%d nr=%d foreign=%.1f %s]" % (D, n, nr, foreign, reducer),
                  rows_with_a_sign_flip=flips, bound_rows=SIGN_FLIP_ROWS_BOUND)
    assert not con._counts.any().item()
    for g in con.get_gradients().values():
        assert not g.any()


def numpy_sign_counts(params, bh, bt, br, B, N, margin):
    """Integer sign-count gradient of the TransE loss w.r.t. the NORMALISED rows, straight from the
    definition (TransE.py:11-15,48-51): rows [0,E) entities, [E,E+R) relations."""
    ent, rel = params["ent_embeddings"], params["rel_embeddings"]
    E, D = ent.shape
    l2 = lambda x: x / np.sqrt(np.maximum((x * x).sum(-1, keepdims=True), 1e-12)).astype(np.float32)
    en, rn = l2(ent), l2(rel)
    S = np.zeros((E + rel.shape[0], D), np.int64)
    e_p = en[bh[:B]] + rn[br[:B]] - en[bt[:B]]
    p = np.abs(e_p).sum(-1)
    for k in range(N):
        sl = slice(B * (k + 1), B * (k + 2))
        e_k = en[bh[sl]] + rn[br[sl]] - en[bt[sl]]
        active = (p - np.abs(e_k).sum(-1) + np.float32(margin)) >= 0
        sp, sk = np.sign(e_p).astype(np.int64), np.sign(e_k).astype(np.int64)
        for b in np.nonzero(active)[0]:
            S[bh[b]] += sp[b]; S[E + br[b]] += sp[b]; S[bt[b]] -= sp[b]
            S[bh[sl][b]] -= sk[b]; S[E + br[sl][b]] -= sk[b]; S[bt[sl][b]] += sk[b]
    return S


@pytest.mark.parametrize("D,n", [(16, 25), (50, 25), (16, 40), (100, 40), (200, 63), (64, 3)])
def test_transe_sign_counts_are_the_exact_integer_sums(D, n, reducer):
    """The int32 counts before the per-row finalisation equal the definition's integer sums EXACTLY
    (multi-round id prefetch when n exceeds the team width, int8 saturation margin at n = 63)."""
    import torch
    rng = np.random.default_rng(D * 100 + n)
    E, R, B = 61, 4, 300
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=2)
    bh, bt, br = rand_batch(rng, E, R, B, n, 0, distinct=True)
    con = make_engine("transe", E, R, D, n, 0, margin=1.0, params=params)
    dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
    con.forward_counts(dev, B, B, B * n)
    got = con._counts.cpu().numpy().astype(np.int64)
    want = numpy_sign_counts(params, bh, bt, br, B, n, 1.0)
    bad_rows = np.nonzero((got != want).any(1))[0]
    # a hinge within rounding of zero may legitimately flip: allow a couple of rows, not a pattern
    parity_report("transe_sign_counts_exact[D=%d n=%d %s]" % (D, n, reducer), rows_differing=len(bad_rows),
                  largest_count_difference=int(np.abs(got - want).max()), bound_rows=EXACT_COUNT_ROWS_BOUND)
    assert len(bad_rows) <= EXACT_COUNT_ROWS_BOUND, (len(bad_rows), bad_rows[:10], np.abs(got - want).max())
    con._counts.zero_()


@pytest.mark.parametrize("D,n", [(100, 1), (200, 25)])
@pytest.mark.parametrize("opt", ["SGD", "Adam"])
def test_transe_sign_count_training_tracks_oracle(D, n, opt):
    """Several steps.  SGD: the comparison is on the accumulated UPDATE (p_k - p_0), not on the parameters, whose
    magnitude would hide it.  Adam is scale-free in g, so elements whose gradient nearly cancels amplify rounding
    differences to a fraction of one step: each step is checked element by element against what a gradient inside the
    1e-5 tolerance can produce (adam_step_explained) -- no element count is waived."""
    rng = np.random.default_rng(23)
    E, R, B = 300, 7, 512
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=7)
    alpha = 0.05 if opt == "SGD" else 0.01
    orc = oracle.Model("transe", E, R, D, D, margin=1.0, params=params)
    con = make_engine("transe", E, R, D, n, 0, margin=1.0, opt=opt, alpha=alpha, params=params)
    if opt == "Adam":
        tot = dict(amplified=0, worst_steps=0.0, worst_gain=0.0, kink_elems=0, tie_groups=0, set_aside_rows=0)
        for step in range(4):
            bh, bt, br = rand_batch(rng, E, R, B, n, 0)
            rep = adam_step_explained(con, orc, "transe", bh, bt, br, B, n, alpha, (D, D), step, "sign-count D=%d n=%d" % (D, n))
            for kk in rep:
                tot[kk] = max(tot[kk], rep[kk]) if kk.startswith("worst") else tot[kk] + rep[kk]
        parity_report("transe_sign_count_training[D=%d n=%d Adam]" % (D, n),
                      adam_elements_beyond_1e3_of_a_step_all_explained=tot["amplified"], worst_in_steps=tot["worst_steps"],
                      worst_adam_gain=tot["worst_gain"], rows_set_aside_as_kink_or_tie=tot["set_aside_rows"],
                      elements_of_e_within_1e6_of_zero=tot["kink_elems"], groups_with_hinge_within_1e5=tot["tie_groups"])
        return
    for step in range(4):
        bh, bt, br = rand_batch(rng, E, R, B, n, 0)
        lo = orc.sgd_step(bh, bt, br, B, n, alpha)
        lg = con.train_step(bh, bt, br, None)
        assert abs(lg - lo) <= 2e-5 * abs(lo), (step, lg, lo)
    got = con.get_parameters()
    for k in orc.params:
        du_o = orc.params[k].astype(np.float64) - params[k]
        du_g = got[k].astype(np.float64) - params[k]
        assert np.abs(du_g - du_o).max() <= 1e-4 * np.abs(du_o).max(), k


def test_transe_generic_path_still_matches_oracle():
    """use_counts=False keeps the fp32-atomic path (the one data-parallel H/D/R also use) for TransE."""
    rng = np.random.default_rng(17)
    E, R, D, B, n = 120, 6, 64, 128, 3
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=5)
    orc = oracle.Model("transe", E, R, D, D, params=params)
    con = make_engine("transe", E, R, D, n, 0, opt="SGD", alpha=0.05, params=params, use_counts=False)
    assert not con.use_counts
    for step in range(3):
        bh, bt, br = rand_batch(rng, E, R, B, n, 0)
        lo = orc.sgd_step(bh, bt, br, B, n, 0.05)
        lg = con.train_step(bh, bt, br, None)
        assert abs(lg - lo) <= 2e-5 * abs(lo)
    for k in orc.params:
        assert relerr(con.get_parameters()[k], orc.params[k]) < 2e-5


def test_adam_kernel_bitwise_on_identical_gradient():
    """The Adam sweep alone (same summed gradient in, TF1 op order) is bit-identical to the oracle."""
    import ctypes
    import torch
    from openkeonspark_amd import _lib
    rng = np.random.default_rng(2)
    n = 1000 * 36 + 3
    p = rng.standard_normal(n).astype(np.float32); m = (rng.standard_normal(n) * 0.1).astype(np.float32)
    v = (rng.random(n) * 0.01).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    g[rng.random(n) < 0.7] = 0
    lr_t = oracle.adam_lr_t(0.001, 0.9, 0.999, 7)
    po, mo, vo = p.copy(), m.copy(), v.copy()
    oracle.lib().orc_adam_apply_dense(po.ctypes.data, mo.ctypes.data, vo.ctypes.data, g.ctypes.data, n,
                                      ctypes.c_float(lr_t), ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8))
    L = _lib.lib()
    tp, tm, tv, tg = (torch.from_numpy(x.copy()).cuda() for x in (p, m, v, g))
    _lib.check(L.kge_adam_update(tp.data_ptr(), tm.data_ptr(), tv.data_ptr(), tg.data_ptr(), n, float(lr_t), 0.9, 0.999, 1e-8, None))
    torch.cuda.synchronize()
    assert np.array_equal(tm.cpu().numpy(), mo) and np.array_equal(tv.cpu().numpy(), vo)
    assert np.abs(tp.cpu().numpy() - po).max() <= 1e-7  # division/sqrt rounding mode of the two ISAs
    assert not tg.cpu().numpy().any()


def test_sampled_training_matches_oracle_end_to_end(fb_dir):
    """Device sampler + fused step vs oracle sampler + oracle step on the FB15k-237-shaped graph
    (config #1 shape at a reduced batch): same batches bit for bit, same losses / parameters."""
    from openkeonspark_amd.Config import Config
    from openkeonspark_amd.TransE import TransE
    D, n = 100, 1
    con = Config()
    con.prefetch_sampling = False   # so that the rng states after k steps are those after k batches
    con.set_in_path(fb_dir); con.set_work_threads(8); con.set_dimension(D); con.set_nbatches(400)
    con.set_ent_neg_rate(n); con.set_alpha(0.01); con.set_margin(1.0)
    con.init()
    con.set_model_and_session(TransE)
    kg = oracle.KG(fb_dir, work_threads=8, bern=0)
    kg.set_stream_states(con.get_stream_states())
    orc = oracle.Model("transe", con.entTotal, con.relTotal, D, D, margin=1.0, params=con.get_parameters())
    B = con.batch_size
    for step in range(3):
        bh, bt, br, _ = kg.sampling(B, n, 0)
        lo = orc.sgd_step(bh, bt, br, B, n, 0.01)
        lg = con.train_step()
        assert abs(lg - lo) <= 2e-5 * abs(lo), (step, lg, lo)
    got = con.get_parameters()
    for k in orc.params:
        assert relerr(got[k], orc.params[k]) < 2e-5
    assert con.get_stream_states().tolist() == kg.stream_states().tolist()


@pytest.mark.parametrize("D,force_sort", [(64, 0), (50, 0), (64, 1)])
def test_prefetched_sampling_is_bit_identical(fb_dir, D, force_sort):
    """Drawing batch i+1 on a side stream while step i finishes changes nothing: same losses, same
    parameters, bit for bit (the sampler never reads the parameters).  D = 50 takes the scalar emit kernel (widths that
    are not multiples of 4), whose launch must record the event the side stream waits for just as the vectorised one does:
    with sync=False the host runs ahead, and an unordered sampler would overwrite batch slots that queued kernels still read.
    On the sign-count path the next batch's sampler is ARMED and rides in the step's bucket-scatter launch
    (kge_sampling_attach); force_sort = 1 takes the reduction without a scatter kernel, where the armed sampler must be
    launched on its own by kge_sampling_flush."""
    from openkeonspark_amd.Config import Config
    from openkeonspark_amd.TransE import TransE
    from openkeonspark_amd import _lib
    _lib.lib().kge_set_option(b"counts_force_sort", force_sort)
    runs = []
    for prefetch in (False, True):
        con = Config()
        con.prefetch_sampling = prefetch
        con.counts_min_records = 0   # exact count pipeline: reproducible bit for bit
        con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(1); con.set_dimension(D); con.set_nbatches(100)
        con.set_ent_neg_rate(4); con.set_alpha(0.01); con.set_opt_method("Adam")
        con.init()
        seeds = np.array(oracle.libc_rand_sequence(8), dtype=np.uint64)
        con.lib.kge_set_stream_states(seeds.ctypes.data, 8)
        con.set_model_and_session(TransE)
        import torch
        dev_losses = [con.train_step(sync=False).clone() for _ in range(12)]   # no host sync between steps: the host runs ahead
        losses = [float(x) for x in torch.stack([l.reshape(()) for l in dev_losses]).cpu().numpy()]
        runs.append((losses, con.get_parameters(), con.get_stream_states(before_prefetch=True)))
    _lib.lib().kge_set_option(b"counts_force_sort", 0)
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        assert np.array_equal(runs[0][1][k], runs[1][1][k])
    assert runs[0][2].tolist() == runs[1][2].tolist()      # rewound by the one batch drawn ahead: the same rng states


def test_transr_sampler_riding_in_the_relation_scatter_draws_the_same_batches(fb_dir):
    """TransR on one GPU: with prefetch_sampling (the default there) the next batch's sampler is armed before the step and rides in
    its relation-scatter launch.  Same batches in the same order: the rng states (rewound by the batch drawn ahead) and the
    losses equal those of sampling at the start of every step; the tables agree to fp32-atomic summation order."""
    from openkeonspark_amd.Config import Config
    from openkeonspark_amd.TransR import TransR
    runs = []
    for prefetch in (False, True):
        con = Config()
        con.prefetch_sampling = prefetch
        con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(0); con.set_dimension(48); con.set_nbatches(50)
        con.set_ent_neg_rate(1); con.set_alpha(0.01); con.set_opt_method("SGD")
        con.init()
        seeds = np.array(oracle.libc_rand_sequence(8), dtype=np.uint64)
        con.lib.kge_set_stream_states(seeds.ctypes.data, 8)
        con.set_model_and_session(TransR)
        losses = [con.train_step() for _ in range(5)]
        runs.append((losses, con.get_parameters(), con.get_stream_states(before_prefetch=True)))
    assert np.allclose(runs[0][0], runs[1][0], rtol=2e-6, atol=0)
    assert runs[0][2].tolist() == runs[1][2].tolist()
    for k in runs[0][1]:
        assert np.abs(runs[0][1][k] - runs[1][1][k]).max() <= 1e-5 * np.abs(runs[0][1][k]).max(), k


@pytest.mark.parametrize("D,opt", [(200, "Adam"), (100, "SGD"), (64, "Adam"), (200, "SGD")])
def test_inverse_norm_table_carried_across_steps_is_bit_identical(fb_dir, D, opt):
    """The vectorised emit kernel reads 1/|row| from a per-row table.  The full-table apply kernel refreshes the entry of every
    row it rewrites, so the pre-pass over the tables runs only when the table is stale (first step, after set_parameters).
    Carrying the table (default) must give the same bits as recomputing it in front of every step (option inv_carry = 0),
    including across a write of the tables from outside (set_parameters in mid-run rescales every entity row: a stale
    table would normalise with the old norms) and for SGD, where untouched rows keep their old entry."""
    from openkeonspark_amd.Config import Config
    from openkeonspark_amd.TransE import TransE
    from openkeonspark_amd import _lib
    L = _lib.lib()
    runs = []
    try:
        for carry in (0, 1):
            L.kge_set_option(b"inv_carry", carry)
            con = Config()
            con.prefetch_sampling = False
            con.counts_min_records = 0
            con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(1); con.set_dimension(D); con.set_nbatches(40)
            con.set_ent_neg_rate(5); con.set_alpha(0.01); con.set_opt_method(opt)
            con.init()
            seeds = np.array(oracle.libc_rand_sequence(8), dtype=np.uint64)
            con.lib.kge_set_stream_states(seeds.ctypes.data, 8)
            con.set_model_and_session(TransE)
            losses = []
            for step in range(7):
                losses.append(con.train_step())
                if step == 2:
                    p = con.get_parameters()
                    p["ent_embeddings"] = (p["ent_embeddings"] * np.float32(1.5)).astype(np.float32)
                    con.set_parameters(p)
            runs.append((losses, con.get_parameters()))
    finally:
        L.kge_set_option(b"inv_carry", 1)
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert np.array_equal(runs[0][1][k], runs[1][1][k]), k


@pytest.mark.parametrize("model", ["transe", "transh", "transd", "transr"])
def test_predict_matches_oracle(model):
    rng = np.random.default_rng(3)
    E, R, D = 90, 4, 100
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=8)
    con = make_engine(model, E, R, D, 1, 0, params=params)
    orc = oracle.Model(model, E, R, D, D, params=params)
    h = rng.integers(0, E, 333); t = rng.integers(0, E, 333); r = rng.integers(0, R, 333)
    if model == "transr":
        r[0] = 2  # TransR.py:83: every triple is projected with predict_r[0]'s matrix
    got = con.test_step(h, t, r)
    want = orc.predict(h, t, r)
    assert np.allclose(got, want, rtol=1e-5, atol=0)


def test_count_image_invariants_at_bench_size(fb_dir):
    """BASELINE configs[1] at the bench's full size (B = 34 014 positives x 25 negatives, dim 200): properties that
    need no oracle.  Each scored triple adds +g to its head row and -g to its tail row, so the integer count image
    sums to zero over the ENTITY rows, column by column; the count of a relation row is bounded by its records; the
    step is reproducible bit for bit; and the sampled batch keeps the reference's layout (negative k of positive b at
    B(k+1)+b shares the relation and exactly one entity with it)."""
    import torch
    from openkeonspark_amd.Config import Config
    from openkeonspark_amd.TransE import TransE
    images, batches = [], []
    for _ in range(2):
        con = Config()
        con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(1); con.set_dimension(200); con.set_nbatches(8)
        con.set_ent_neg_rate(25); con.set_alpha(0.001); con.set_opt_method("Adam")
        con.init()
        seeds = np.array(oracle.libc_rand_sequence(8), dtype=np.uint64)
        con.lib.kge_set_stream_states(seeds.ctypes.data, 8)
        con.set_model_and_session(TransE)
        B, n = con.batch_size, 25
        assert B == 34014
        dev, n_pos = con.sample_device()
        con.forward_counts(dev, n_pos, B, B * n, sampler_shaped=True)
        images.append(con._counts.clone())
        batches.append(dev.clone())
        loss = float(con._loss.item())
        assert 0.5 < loss < 1.5
    img, bat = images[0], batches[0]
    assert torch.equal(images[0], images[1]) and torch.equal(batches[0], batches[1])
    E = con.entTotal
    assert int(img[:E].sum(dim=0).abs().max()) == 0
    assert int(img[E:].abs().sum()) > 0 and int(img.abs().max()) <= B * (1 + n)
    h, t, r = bat[0].view(1 + n, B), bat[1].view(1 + n, B), bat[2].view(1 + n, B)
    assert bool((r[1:] == r[:1]).all())
    same_h, same_t = h[1:] == h[:1], t[1:] == t[:1]
    assert bool((same_h ^ same_t).all())          # exactly one side corrupted, and never to the same entity


@pytest.mark.parametrize("model", ["transh", "transd"])
@pytest.mark.parametrize("copies", [1, 0])
def test_atomic_path_hub_copies_match_oracle(model, copies):
    """Few relations, many groups per step: the atomic path spreads the relation-side rows over copies and folds them
    (same-address atomics serialise).  Same gradients as the oracle either way; accumulators end re-zeroed."""
    import torch
    from openkeonspark_amd import _lib
    L = _lib.lib()
    E, R, D, B, n = 400, 2, 72, 1500, 2
    rng = np.random.default_rng(seed_of(model, "hub"))
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=6)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model(model, E, R, D, D, margin=0.9, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, 0.05), B, n)
    loss_o, g_o = orc.grad(bh, bt, br, B, n)
    L.kge_set_option(b"float_records", 0)
    L.kge_set_option(b"hub_copies", copies)
    try:
        con = make_engine(model, E, R, D, n, 0, margin=0.9, params=params)
        dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
        for _ in range(2):                       # twice: the copies must come back zeroed
            for g in con._grads:
                g.zero_()
            con.forward_backward(dev, B, B, B * n)
            torch.cuda.synchronize()
            g_g = con.get_gradients()
            assert abs(float(con._loss.item()) - loss_o) <= RTOL * abs(loss_o)
            for k in g_o:
                assert relerr(g_g[k], g_o[k]) < RTOL, (k, relerr(g_g[k], g_o[k]))
    finally:
        L.kge_set_option(b"float_records", 1)
        L.kge_set_option(b"hub_copies", 1)


@pytest.mark.parametrize("model", ["transe", "transh", "transd", "transr"])
@pytest.mark.parametrize("B,n", [(1, 1), (1, 7), (3, 2)])
def test_tiny_batches_match_oracle(model, B, n, grad_path):
    """One positive, a handful of triples: every tile / chunk / bucket is mostly padding."""
    import torch
    E, R, D = 37, 3, 36
    rng = np.random.default_rng(seed_of(model, B, n, "tiny"))
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=8)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model(model, E, R, D, D, margin=2.0, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, distinct=True), B, n)
    loss_o, g_o = orc.grad(bh, bt, br, B, n)
    con = make_engine(model, E, R, D, n, 0, margin=2.0, params=params)
    dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
    con.forward_backward(dev, B, B, B * n)
    torch.cuda.synchronize()
    assert abs(float(con._loss.item()) - loss_o) <= RTOL * abs(loss_o) + 1e-12
    g_g = con.get_gradients()
    for k in g_o:
        assert np.abs(g_g[k] - g_o[k]).max() <= RTOL * (np.abs(g_o[k]).max() + 1e-30), k
    if model == "transe":      # and the integer path on the same tiny batch
        for g in con._grads:
            g.zero_()
        con.forward_counts(dev, B, B, B * n)
        con.set_opt_method("SGD")
        before = con.get_parameters()
        con.apply_counts(B * n)
        after = con.get_parameters()
        for k in g_o:
            upd = (before[k] - after[k]) / con.alpha
            assert np.abs(upd - g_o[k]).max() <= 1e-4 * (np.abs(g_o[k]).max() + 1e-30), k


def test_out_of_range_ids_are_rejected_on_the_host():
    from openkeonspark_amd import KgeError
    con = make_engine("transe", 20, 3, 16, 1, 0)
    ok = np.array([0, 1]), np.array([2, 3]), np.array([0, 0])
    con.test_step(*ok)
    for bad in ((np.array([0, 20]), ok[1], ok[2]), (ok[0], np.array([2, -1]), ok[2]), (ok[0], ok[1], np.array([0, 3]))):
        with pytest.raises(KgeError):
            con.test_step(*bad)
        with pytest.raises(KgeError):
            con.train_step(bad[0], bad[1], bad[2], None)


@pytest.mark.parametrize("D", [16, 200])
def test_negative_zero_has_sign_zero(D, reducer):
    """tf.sign(-0.0) is 0 (TransE.py:15 abs -> its gradient sign(e)).  The vectorised emit kernel reads the sign from the
    bit pattern of e, where -0.0 would read as -1: rows are built so that e = h^ + r^ - t^ is EXACTLY -0.0 in whole columns
    ((-0) + (-0) - (+0) for the positive, fma(f, -0, -0) for corrupted heads / tails / relation vectors) and exactly +0.0 in
    others.  The integer counts must equal the definition's (numpy: sign(+-0) = 0) with no tolerance in those columns."""
    import torch
    rng = np.random.default_rng(D)
    E, R, B, n = 40, 4, 256, 6
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=12)
    ent, rel = params["ent_embeddings"], params["rel_embeddings"]
    ent[0::2, 0] = -0.0; ent[1::2, 0] = 0.0          # column 0: even entities -0, odd entities +0
    rel[:, 0] = -0.0
    ent[:, 1] = 0.0; rel[:, 1] = 0.0                 # column 1: all +0 -> e = +0
    ent[:, 2] = -0.0; rel[:, 2] = 0.0                # column 2: (-0) + (+0) - (-0) = +0
    ent[:, 3] = -0.0; rel[:, 3] = -0.0               # column 3: (-0) + (-0) - (-0) = +0 (IEEE: -0 + +0)
    bh, bt, br = rand_batch(rng, E, R, B, n, 2, distinct=True)
    con = make_engine("transe", E, R, D, n, 2, margin=1.0, params=params)
    assert np.signbit(con.get_parameters()["ent_embeddings"][0, 0])          # the negative zeros reached the device
    dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
    con.forward_counts(dev, B, B, B * (n + 2))
    got = con._counts.cpu().numpy().astype(np.int64)
    want = numpy_sign_counts(params, bh, bt, br, B, n + 2, 1.0)
    assert int(np.abs(want[:, 4:]).sum()) > 0
    assert not got[:, :4].any(), np.nonzero(got[:, :4])
    bad_rows = np.nonzero((got != want).any(1))[0]
    parity_report("negative_zero_sign[D=%d %s]" % (D, reducer), rows_differing=len(bad_rows), bound_rows=EXACT_COUNT_ROWS_BOUND)
    assert len(bad_rows) <= EXACT_COUNT_ROWS_BOUND
    con._counts.zero_()


@pytest.mark.gpu
@pytest.mark.parametrize("model,Dr", [("transh", None), ("transd", None), ("transr", 64)])
def test_sampled_entry_equals_general_entry(model, Dr):
    """kge_forward_backward_sampled (the caller vouches that the batch is sampler-shaped: include/kge_mi355.h) against
    kge_forward_backward on such a batch: same loss and gradients, for the paths that skip their exact pass (the TransH /
    TransD pair-count path, TransR's lean vector stage), and both against the oracle."""
    import torch
    from openkeonspark_amd import _lib
    L = _lib.lib()
    E, R, D, B, n = 300, 9, 72, 700, 5
    rng = np.random.default_rng(seed_of(model, "sampled-entry"))
    dr = D if Dr is None else Dr
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, dr, seed=12)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model(model, E, R, D, dr, margin=1.5, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, distinct=True), B, n)
    loss_o, g_o = orc.grad(bh, bt, br, B, n)
    L.kge_set_option(b"float_records_min", 0)
    L.kge_set_option(b"pair_counts_min_neg", 1)
    try:
        con = make_engine(model, E, R, D, n, 0, margin=1.5, params=params, Dr=Dr)
        dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
        got = {}
        for sampled in (False, True, True):          # twice: buffers the sampled entry leaves behind must be reusable
            for g in con._grads:
                g.zero_()
            con.forward_backward(dev, B, B, B * n, sampler_shaped=sampled)
            torch.cuda.synchronize()
            got[sampled] = (float(con._loss.item()), con.get_gradients())
            assert abs(got[sampled][0] - loss_o) <= RTOL * abs(loss_o)
            for k in g_o:
                assert relerr(got[sampled][1][k], g_o[k]) < RTOL, (sampled, k, relerr(got[sampled][1][k], g_o[k]))
        assert abs(got[True][0] - got[False][0]) <= 1e-6 * abs(got[False][0])
        for k in g_o:
            assert relerr(got[True][1][k], got[False][1][k]) < 1e-6, k
    finally:
        L.kge_set_option(b"float_records_min", 1 << 16)
        L.kge_set_option(b"pair_counts_min_neg", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["transh", "transd"])
@pytest.mark.parametrize("E,R,D,B,n", [(50, 1, 4, 300, 4), (120, 3, 256, 200, 17), (90, 5, 64, 150, 63), (300, 7, 132, 257, 16)])
def test_pair_count_path_shapes(model, E, R, D, B, n):
    """Edges of the pair-count path (csrc/pairs.hip): one float4 chunk per row and four, a single relation, 16 / 17 / 63
    negatives (one full round of a 16-lane team, a second round with one negative, the maximum; int8 sums of 2n = 126),
    widths that leave the last chunk partly empty.  Arbitrary negatives in 5 % of the slots go to the exact pass."""
    import torch
    from openkeonspark_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(seed_of(model, E, R, D, B, n, "pair-shapes"))
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=14)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model(model, E, R, D, D, margin=1.0, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, 0.05), B, n)
    loss_o, g_o = orc.grad(bh, bt, br, B, n)
    L.kge_set_option(b"float_records_min", 0)
    L.kge_set_option(b"pair_counts_min_neg", 1)
    try:
        con = make_engine(model, E, R, D, n, 0, params=params)
        dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()
        con.forward_backward(dev, B, B, B * n)
        torch.cuda.synchronize()
        assert abs(float(con._loss.item()) - loss_o) <= RTOL * abs(loss_o)
        g_g = con.get_gradients()
        for k in g_o:
            assert relerr(g_g[k], g_o[k]) < RTOL, (k, relerr(g_g[k], g_o[k]))
    finally:
        L.kge_set_option(b"float_records_min", 1 << 16)
        L.kge_set_option(b"pair_counts_min_neg", 0)


def test_lazy_adam_moves_the_touched_rows_only():
    """opt_method "LazyAdam" -- opt-in and NON-PARITY: the reference trains with TF1's AdamOptimizer, which moves every row of
    every table each step (distribute_training.py:95-101; the dense path above is the parity path).  The lazy rule
    (tf.contrib.opt.LazyAdamOptimizer) applies the same element formula to the rows a step touches and leaves all other rows
    and their moments alone.  Checked here against that rule written out in numpy on the oracle's gradient: rows outside the
    touched set keep p, m, v bit for bit; touched rows move by what a gradient within 1e-5 of the oracle's produces."""
    from parity_util import adam_step_fp64, adam_update_explained
    from torch_ref import near_kink_rows
    rng = np.random.default_rng(21)
    E, R, D, B, n, alpha = 400, 9, 64, 96, 3, 0.01
    params = oracle.init_params(oracle.MODEL_IDS["transe"], E, R, D, D, seed=8)
    orc = oracle.Model("transe", E, R, D, D, margin=1.0, params=params)
    con = make_engine("transe", E, R, D, n, 0, margin=1.0, opt="LazyAdam", alpha=alpha, params=params)
    assert con.sparse_rows and con._lazy_adam and not con._adam
    names = con.trainModel.table_names
    moved_rows = 0
    for step in range(4):
        bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, distinct=True), B, n)
        p0 = con.get_parameters()
        m0 = {k: con._adam_m[i].cpu().numpy() for i, k in enumerate(names)}
        v0 = {k: con._adam_v[i].cpu().numpy() for i, k in enumerate(names)}
        orc.params = {k: v.copy() for k, v in p0.items()}
        active = orc.hinge_margins(bh, bt, br, B, n) > 0           # [B, n]
        loss_o, g_o = orc.grad(bh, bt, br, B, n)
        touched = {"ent_embeddings": set(), "rel_embeddings": set()}
        for b in np.nonzero(active.any(1))[0]:
            idx = [b] + [b + B * (k + 1) for k in np.nonzero(active[b])[0]]
            touched["ent_embeddings"] |= set(np.asarray(bh)[idx].tolist()) | set(np.asarray(bt)[idx].tolist())
            touched["rel_embeddings"] |= set(np.asarray(br)[idx].tolist())
        loss_g = con.train_step(bh, bt, br, None)
        assert abs(loss_g - loss_o) <= 2e-5 * abs(loss_o)
        p1 = con.get_parameters()
        m1 = {k: con._adam_m[i].cpu().numpy() for i, k in enumerate(names)}
        v1 = {k: con._adam_v[i].cpu().numpy() for i, k in enumerate(names)}
        lr_t = float(oracle.adam_lr_t(alpha, 0.9, 0.999, step + 1))
        kink, _ = near_kink_rows("transe", p0, bh, bt, br, B, n, D, D, tol=1e-6)
        for k in names:
            T = np.array(sorted(touched[k]), dtype=np.int64)
            rest = np.setdiff1d(np.arange(p0[k].shape[0]), T)
            # untouched rows: nothing moves, not even the moments (this is what TF1's dense Adam would NOT do)
            np.testing.assert_array_equal(p1[k][rest], p0[k][rest], err_msg=k)
            np.testing.assert_array_equal(m1[k][rest], m0[k][rest], err_msg=k)
            np.testing.assert_array_equal(v1[k][rest], v0[k][rest], err_msg=k)
            assert (g_o[k][rest] == 0).all()
            # touched rows: the Adam element rule on the oracle's gradient (zero-gradient elements of a touched row decay too)
            pT, mT, vT, gT = (a[T].astype(np.float64) for a in (p0[k], m0[k], v0[k], g_o[k]))
            du_o = adam_step_fp64(pT, mT, vT, gT, lr_t, 0.9, 0.999, 1e-8)
            skip = {int(np.searchsorted(T, r)) for r in kink[k] if r in touched[k]}
            rep = adam_update_explained(p0[k][T], m0[k][T], v0[k][T], g_o[k][T], p1[k][T].astype(np.float64) - p0[k][T], du_o, lr_t,
                                        skip_rows=skip)
            assert rep["unexplained"].size == 0, (step, k, rep["unexplained"][:8])
            moved_rows += len(T)
    assert moved_rows > 0 and con.global_step == 4
    parity_report("lazy_adam_touched_rows_only", touched_rows_checked=moved_rows, steps=4)


@pytest.mark.parametrize("model", ["transh", "transd", "transe"])
def test_row_wise_sgd_in_place_tracks_oracle(model):
    """sparse_rows=True with TransH / TransD (and TransE off the sign-count path): kge_forward_backward_sgd_rows -- gradient rows
    as float records, summed by destination row and added to the PARAMETER rows as -lr * sum; no gradient tables exist.  Same
    update as GradientDescentOptimizer on the summed gradient (distribute_training.py:99-101) up to fp32 summation order: losses
    and accumulated updates against the oracle's SGD steps, rows the batches never touch bit-identical to their initial values."""
    rng = np.random.default_rng(31)
    E, R, D, B, n, alpha = 300, 7, 64, 160, 3, 0.05
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, D, D, seed=6)
    orc = oracle.Model(model, E, R, D, D, margin=1.0, params=params)
    from openkeonspark_amd.Config import Config
    import openkeonspark_amd as pkg
    con = Config()
    con.use_counts = False          # (TransE: the sign-count path has its own sparse-row mode)
    con.sparse_rows = True
    con.set_ent_neg_rate(n); con.set_rel_neg_rate(0); con.set_margin(1.0); con.set_opt_method("SGD"); con.set_alpha(alpha)
    con.set_dimension(D)
    hh = np.arange(40) % E
    con.init_from_arrays(E, R, hh, (hh + 1) % E, hh % R)
    con.set_model_and_session(getattr(pkg, {"transe": "TransE", "transh": "TransH", "transd": "TransD"}[model]))
    con.set_parameters(params)
    assert con.sparse_inplace and not con.sparse_rows and con._grads == []
    seen_e, seen_r = set(), set()
    for step in range(5):
        bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, distinct=True), B, n)
        seen_e |= set(np.asarray(bh).tolist()) | set(np.asarray(bt).tolist()); seen_r |= set(np.asarray(br).tolist())
        lo = orc.sgd_step(bh, bt, br, B, n, alpha)
        lg = con.train_step(bh, bt, br, None)
        assert abs(lg - lo) <= 2e-5 * abs(lo), (step, lg, lo)
        got = con.get_parameters()
        for k in orc.params:
            du_o = orc.params[k].astype(np.float64) - params[k]
            du_g = got[k].astype(np.float64) - params[k]
            # a row takes one add per run of its records (relation-side rows: one per virtual copy and chunk), each rounding the
            # PARAMETER to its own ulp -- the dense path rounds it once per step: a few ulps of p on top of 1e-4 of the update
            atol = 16 * float(np.spacing(np.float32(np.abs(params[k]).max())))
            assert np.abs(du_g - du_o).max() <= 1e-4 * np.abs(du_o).max() + atol, (step, k)
    for k in orc.params:
        rows = seen_e if k in ("ent_embeddings", "ent_transfer") else seen_r
        rest = np.setdiff1d(np.arange(params[k].shape[0]), np.array(sorted(rows), dtype=np.int64))
        np.testing.assert_array_equal(got[k][rest], params[k][rest], err_msg=k)
    assert con.global_step == 5
    # a negative equal to its positive is not a single-slot corruption: its exact path adds rows atomically, which the in-place
    # update cannot take -- the step says so instead of training on a silently different batch
    bh, bt, br = rand_batch(rng, E, R, B, n, 0, distinct=True)
    bh[B], bt[B], br[B] = bh[0], bt[0], br[0]
    with pytest.raises(pkg.KgeError, match="single-slot"):
        con.train_step(bh, bt, br, None)
