"""CPU checks of the parity accounting itself (tests/parity_util.py): the explanation of Adam's amplification must accept what a
gradient inside the 1e-5 tolerance can produce and nothing else, and the chunked kink search must agree with the one-shot form."""
import numpy as np

from oracle import oracle
from parity_util import adam_step_fp64, adam_update_explained, kink_rows_chunked


def _state(seed, n=4000):
    rng = np.random.default_rng(seed)
    p0 = rng.standard_normal(n).astype(np.float32)
    m0 = (rng.standard_normal(n) * 1e-3).astype(np.float32)
    v0 = (rng.random(n) * 1e-6).astype(np.float32)
    g = (rng.standard_normal(n) * 1e-3).astype(np.float32)
    g[:50] *= 1e-6                      # nearly cancelling elements ...
    m0[:50] = 0.0; v0[:50] = 0.0        # ... of rows with no history (first steps): where Adam amplifies
    g[50:60] = 0.0                      # untouched elements
    return p0, m0, v0, g


def test_adam_explanation_accepts_gradients_inside_the_tolerance_and_rejects_others():
    lr_t = float(oracle.adam_lr_t(0.001, 0.9, 0.999, 3))
    p0, m0, v0, g = _state(1)
    rng = np.random.default_rng(2)
    du_o = adam_step_fp64(p0.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64), g.astype(np.float64), lr_t, 0.9, 0.999, 1e-8)
    d = 1e-5 * np.abs(g).max()
    g_in = g.astype(np.float64) + rng.uniform(-0.9, 0.9, g.shape) * d        # an engine whose gradient is inside the tolerance
    du_in = adam_step_fp64(p0.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64), g_in, lr_t, 0.9, 0.999, 1e-8)
    rep = adam_update_explained(p0, m0, v0, g, du_in, du_o, lr_t)
    assert rep["unexplained"].size == 0
    assert rep["amplified"] > 0 and rep["worst_gain"] > 10      # the near-cancelling elements DID move by a visible fraction of a step
    # every amplified element is one whose step interval is itself wider than 1e-3 of a step: the amplification, not an error
    assert (rep["width_steps"][rep["amplified_mask"]] > 1e-3 * 0.99).all()
    # an engine whose gradient is off by 10x the tolerance where the step is still sensitive to g (no history, |g| ~ 10 d, i.e.
    # sqrt(v) comparable to eps) is NOT explained; elements whose step has saturated at ~lr_t * sign(g) would not show it
    big = np.arange(60, 70)
    g[big] = (10 * d * np.where(np.arange(10) % 2 == 0, 1.0, -1.0)).astype(np.float32)
    m0[big] = 0.0; v0[big] = 0.0
    du_o = adam_step_fp64(p0.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64), g.astype(np.float64), lr_t, 0.9, 0.999, 1e-8)
    g_out = g.astype(np.float64).copy()
    g_out[big] *= 2.0
    du_out = adam_step_fp64(p0.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64), g_out, lr_t, 0.9, 0.999, 1e-8)
    rep = adam_update_explained(p0, m0, v0, g, du_out, du_o, lr_t)
    assert set(rep["unexplained"].tolist()) == set(big.tolist())
    # ... unless those rows are declared kink rows
    rep = adam_update_explained(p0.reshape(-1, 1), m0.reshape(-1, 1), v0.reshape(-1, 1), g.reshape(-1, 1), du_out.reshape(-1, 1),
                                du_o.reshape(-1, 1), lr_t, skip_rows=set(big.tolist()))
    assert rep["unexplained"].size == 0


def test_adam_fp64_restatement_follows_the_oracle_sweep():
    """adam_step_fp64 is the fp64 form of orc_adam_apply_dense (oracle/kge_oracle.c), untouched elements included."""
    import ctypes
    lr_t = oracle.adam_lr_t(0.001, 0.9, 0.999, 5)
    p0, m0, v0, g = _state(3)
    p, m, v = p0.copy(), m0.copy(), v0.copy()
    oracle.lib().orc_adam_apply_dense(p.ctypes.data, m.ctypes.data, v.ctypes.data, g.ctypes.data, g.size, ctypes.c_float(lr_t),
                                      ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8))
    du = adam_step_fp64(p0.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64), g.astype(np.float64), float(lr_t), 0.9, 0.999, 1e-8)
    assert np.abs((p.astype(np.float64) - p0) - du).max() <= 2e-6 * np.abs(du).max() + np.abs(p0).max() * 2.0 ** -23


def test_chunked_kink_search_equals_the_one_shot_form():
    from torch_ref import near_kink_rows
    rng = np.random.default_rng(4)
    E, R, D, B, N = 50, 4, 8, 300, 3
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=1)
    bh = rng.integers(0, E, B * (1 + N)); bt = rng.integers(0, E, B * (1 + N)); br = np.tile(rng.integers(0, R, B), 1 + N)
    tol = 2e-3      # loose on purpose: a few hundred hits on this tiny case
    want, n_want = near_kink_rows("transe", params, bh, bt, br, B, N, D, D, tol=tol)
    got, n_got = kink_rows_chunked(params, bh, bt, br, B, N, tol, chunk=97)
    assert n_got == n_want and n_got > 0
    assert got["ent_embeddings"] == want["ent_embeddings"] and got["rel_embeddings"] == want["rel_embeddings"]


def _kink_case():
    """A tiny TransE problem with two crafted switch points: element 0 of e is exactly zero in triple 2 (positive of group 2: h^_0 =
    t^_0, r^_0 = 0), and the hinge of (group 1, negative 0) sits 5e-6 above its switch point."""
    rng = np.random.default_rng(7)
    E, R, D, B, N = 12, 3, 8, 4, 2
    params = oracle.init_params(oracle.TRANSE, E, R, D, D, seed=3)
    ent, rel = params["ent_embeddings"], params["rel_embeddings"]
    bh = rng.integers(0, E, B * (1 + N)); bt = rng.integers(0, E, B * (1 + N)); br = np.tile(rng.integers(0, R, B), 1 + N)
    bh[2], bt[2] = 1, 2                               # (group 2: its hinges stay active after the margin is moved below)
    bh[B + 1], bt[B + 1], bh[2 * B + 1], bt[2 * B + 1] = 5, 6, 7, 8      # the two negatives of group 1 are different triples
    ent[2] = ent[1][::-1].copy()                      # same norm
    ent[2][0] = ent[1][0]                             # ... and the same element 0 (swap keeps the norm: put ent[1][0]'s old partner back)
    ent[2][D - 1] = ent[1][D - 1]
    ent[2][1:D - 1] = ent[1][1:D - 1][::-1]
    rel[br[2]][0] = 0.0
    return params, bh, bt, br, E, R, D, B, N


def test_switch_point_radius_covers_a_kink_flip_and_a_tie_flip_and_nothing_else():
    """parity_util.transe_row_radius: the engine's row may differ from the oracle's by what its OWN switch points can do, element
    by element -- a sign taken the other way at a kink, a hinge taken the other way at a tie -- and by nothing more."""
    from parity_util import transe_switch_points, transe_row_radius
    import torch
    from torch_ref import loss_and_grads
    params, bh, bt, br, E, R, D, B, N = _kink_case()
    hm1 = oracle.Model("transe", E, R, D, D, margin=1.0, params=params).hinge_margins(bh, bt, br, B, N)
    # move the margin so that one hinge of another group is 5e-6 ABOVE zero for the oracle (active) while group 2 keeps an active hinge
    for tb, tk in [(b, k) for b in (0, 1, 3) for k in range(N)]:
        margin = 1.0 - float(hm1[tb, tk]) + 5e-6
        m = oracle.Model("transe", E, R, D, D, margin=margin, params=params)
        hm = m.hinge_margins(bh, bt, br, B, N)
        if (hm[2] > 0.1).any() and (np.abs(hm) < 5e-5).sum() == 1:
            break
    assert 0 < hm[tb, tk] < 1e-5 and (hm[2] > 0.1).any()
    _, g_o = m.grad(bh, bt, br, B, N)
    kinks, ties, w_max = transe_switch_points(params, bh, bt, br, B, N, hm, kink_tol=1e-6, tie_tol=5e-5)
    assert [2, 0] in kinks.tolist() and ties.tolist() == [[tb, tk]]
    # ... and an "engine" that resolves both the other way: element 0 of triple 0 nudged to a definite sign, the hinge 5e-6 BELOW zero
    p2 = {k: v.astype(np.float64).copy() for k, v in params.items()}
    p2["rel_embeddings"][br[2]][0] = 1e-9
    _, g_e = loss_and_grads("transe", p2, bh, bt, br, B, N, margin - 1e-5, D, D)
    scale = {k: np.abs(g_o[k]).max() for k in g_o}
    seen_excused = 0
    for k in g_o:
        diff = np.abs(g_e[k] - g_o[k])
        bad = np.nonzero((diff > 1e-5 * scale[k]).any(1))[0]
        for row in range(g_o[k].shape[0]):
            rad = transe_row_radius(params, bh, bt, br, B, N, k, row, kinks, ties, w_max)
            assert (diff[row] <= rad + 1e-5 * scale[k]).all(), (k, row)          # every row, excused or not, is inside its interval
            if row in bad:
                seen_excused += 1
                assert rad.max() > 1e-5 * scale[k]
            # a row with no switch point in its slots has radius exactly zero: nothing is excused there
            touches_kink = any((bh[j] == row or bt[j] == row) if k == "ent_embeddings" else br[j] == row for j in kinks[:, 0])
            touches_tie = any(((bh[j] == row or bt[j] == row) if k == "ent_embeddings" else br[j] == row)
                              for b, kk in ties for j in (b, B * (kk + 1) + b))
            if not (touches_kink or touches_tie):
                assert rad.max() == 0.0 and row not in bad
    assert seen_excused >= 3          # the flips did move rows beyond the tolerance: the case exercises the carve-out
    # the bound is quantitative: twice the flip's effect on a kink row is rejected
    row = int(bh[2])
    rad = transe_row_radius(params, bh, bt, br, B, N, "ent_embeddings", row, kinks, ties, w_max)
    wrong = g_o["ent_embeddings"][row] + 2.5 * (g_e["ent_embeddings"][row] - g_o["ent_embeddings"][row])
    assert (np.abs(wrong - g_o["ent_embeddings"][row]) > rad + 1e-5 * scale["ent_embeddings"]).any()


def _simulated_engine_step(params, m0, v0, bh, bt, br, B, N, margin, flip_elems, flip_hinges, lr_t):
    """An 'engine' for the checker: fp64 autograd of the TransE loss (torch_ref's graph) in which chosen elements of e and chosen
    hinges are pushed to the other side of their switch point, followed by TF1 Adam in fp64 rounded to fp32.  Independent of
    parity_util.transe_row_radius: autograd does the backward."""
    import torch
    from torch_ref import l2n
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    h, t, r = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (bh, bt, br))
    e = l2n(P["ent_embeddings"][h]) + l2n(P["rel_embeddings"][r]) - l2n(P["ent_embeddings"][t])
    shift = torch.zeros_like(e)
    for j, i in flip_elems:
        v = float(e[j, i].detach())
        shift[j, i] = -2.0 * v if v != 0.0 else 1e-12          # the other sign (an exact zero: a definite one)
    s = (e + shift).abs().sum(-1)
    hm = s[:B].view(B, 1) - s[B:].view(N, B).t() + margin
    mshift = torch.zeros_like(hm)
    for b, k in flip_hinges:
        v = float(hm[b, k].detach())
        mshift[b, k] = -2.0 * v if v != 0.0 else -1e-12
    loss = torch.clamp(hm + mshift, min=0).mean()
    loss.backward()
    out = {}
    for k in params:
        g = P[k].grad.numpy()
        b1f, b2f, omb1, omb2, epsf = (float(np.float32(0.9)), float(np.float32(0.999)), float(np.float32(1) - np.float32(0.9)),
                                      float(np.float32(1) - np.float32(0.999)), float(np.float32(1e-8)))
        m1 = b1f * m0[k].astype(np.float64) + np.where(g != 0, omb1 * g, 0.0)
        v1 = b2f * v0[k].astype(np.float64) + np.where(g != 0, omb2 * g * g, 0.0)
        p1 = params[k].astype(np.float64) - float(np.float32(lr_t)) * m1 / (np.sqrt(v1) + epsf)
        out[k] = (p1.astype(np.float32), m1.astype(np.float32), v1.astype(np.float32))
    return out


def test_step_checker_accepts_flips_at_switch_points_and_rejects_a_flip_elsewhere(fb_dir):
    """parity_util.check_transe_adam_step on a real batch of the FB15k-237-shaped graph (B = 3 000 x 25 negatives, dim 200): an
    engine that takes EVERY kink element and EVERY tied hinge of the batch the other way passes -- its deviating rows are counted
    as excused, each inside its own interval, and v / the update are still checked on them -- while an engine that flips one
    element that is no switch point (|e| ~ 1e-3 in an active triple) fails."""
    import pytest
    from parity_util import check_transe_adam_step, new_adam_step_totals, transe_switch_points
    B, N, D = 3000, 25, 200
    kg = oracle.KG(fb_dir, work_threads=8, bern=1)
    params = oracle.init_params(oracle.TRANSE, kg.entTotal, kg.relTotal, D, D, seed=0)
    orc = oracle.Model("transe", kg.entTotal, kg.relTotal, D, D, margin=1.0, params=params)
    rng = np.random.default_rng(11)
    m0 = {k: (rng.standard_normal(v.shape) * 1e-5).astype(np.float32) for k, v in params.items()}
    v0 = {k: (rng.random(v.shape) * 1e-10).astype(np.float32) for k, v in params.items()}
    orc.adam_m = {k: v.copy() for k, v in m0.items()}; orc.adam_v = {k: v.copy() for k, v in v0.items()}
    orc.step = 2
    bh, bt, br, _ = kg.sampling(B, N, 0)
    hm = orc.hinge_margins(bh, bt, br, B, N)
    _, g_o = orc.grad(bh, bt, br, B, N, nthreads=4)
    lr_t = float(oracle.adam_lr_t(0.001, 0.9, 0.999, 3))
    orc.apply_adam(g_o, 0.001)
    KINK, TIE = 1e-6, 5e-5
    kinks, ties, _ = transe_switch_points(params, bh, bt, br, B, N, hm, KINK, TIE)
    assert len(kinks) >= 20                                  # (the bench batch has ~1 100 of them; this one in proportion)
    eng = _simulated_engine_step(params, m0, v0, bh, bt, br, B, N, 1.0, kinks.tolist(), ties.tolist(), lr_t)
    tot = new_adam_step_totals()
    check_transe_adam_step(tot, 0, params, m0, v0, {k: eng[k][0] for k in eng}, {k: eng[k][1] for k in eng}, {k: eng[k][2] for k in eng},
                           g_o, orc.params, bh, bt, br, B, N, hm, lr_t, 0.9, 0.999, 1e-8, 1e-5, KINK, TIE)
    assert tot["rows_excused"] >= 5 and tot["rows_excused"] < tot["rows_in_kink_set"] and tot["grad"] <= 1e-5
    assert 0 < tot["worst_excused_over_radius"] <= 1.0
    # ... and a flip that is NOT at a switch point: an element with |e| ~ 1e-3 of a triple whose hinge is clearly active
    en = params["ent_embeddings"].astype(np.float64); rn = params["rel_embeddings"].astype(np.float64)
    en /= np.sqrt((en * en).sum(-1, keepdims=True)); rn /= np.sqrt((rn * rn).sum(-1, keepdims=True))
    active = np.nonzero((hm > 0.1).any(1))[0]
    e_pos = np.abs(en[bh[active]] + rn[br[active]] - en[bt[active]])
    cand = np.argwhere((e_pos > 5e-4) & (e_pos < 2e-3))
    j, i = int(active[cand[0][0]]), int(cand[0][1])
    eng = _simulated_engine_step(params, m0, v0, bh, bt, br, B, N, 1.0, [(j, i)], [], lr_t)
    with pytest.raises(AssertionError):
        check_transe_adam_step(new_adam_step_totals(), 0, params, m0, v0, {k: eng[k][0] for k in eng}, {k: eng[k][1] for k in eng},
                               {k: eng[k][2] for k in eng}, g_o, orc.params, bh, bt, br, B, N, hm, lr_t, 0.9, 0.999, 1e-8, 1e-5, KINK, TIE)
