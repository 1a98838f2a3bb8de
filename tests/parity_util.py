"""Shared accounting for the two places where two correct fp32 evaluations of the margin-ranking loss differ by more than
rounding, and for what TF1's Adam does to such differences.

(1) the kink of |e| (TransE.py:15): an element of e = h^ + r^ - t^ within rounding of zero gets sign +1 from one summation order
    and -1 (or 0) from another, which moves the gradient rows of its whole group          -> kink_rows_chunked (TransE, any size)
(2) Adam (distribute_training.py:95-96, TF1 _apply_sparse_shared) is scale-free in the gradient: the step of an element is
    lr_t * m / (sqrt(v) + eps) with m, v built from g, so a gradient element that nearly cancels turns a difference far inside
    the 1e-5 gradient tolerance into a visible fraction of one step                         -> adam_update_explained

(3) CONSTRUCTIVE form of (1) and of the hinge ties: a row outside the tolerance is not excused by belonging to a set; its
    gradient is linear in the sign choices of its kink elements and in the on/off state of its tied hinges, so the largest
    difference those switch points can make is computed element by element and the engine's row must lie within it
                                                                                            -> transe_switch_points / transe_row_radius

None of the helpers loosens a tolerance: each says, element by element, whether a difference is one that the stated gradient
tolerance (or a named switch point of the loss) can produce, and the tests fail on any element that is not."""
import numpy as np


def _l2n64(x):
    x = np.asarray(x, dtype=np.float64)
    return x / np.sqrt(np.maximum((x * x).sum(-1, keepdims=True), 1e-12))


def transe_switch_points(params, bh, bt, br, B, N, hm, kink_tol, tie_tol, chunk=100_000):
    """The switch points of one TransE step (TransE.py:11-15,44-51) at the given parameters, evaluated in fp64:

    kinks : int64 [n, 2] (scored triple j, element i) with |e_ji| < kink_tol, e = h^ + r^ - t^ -- d|e|/de may be taken as
            -1, 0 or +1 there by a correct fp32 evaluation;
    ties  : int64 [n, 2] (group b, negative k) with |p_b - n_bk + margin| < tie_tol (hm = the oracle's hinge margins [B, N]) --
            the hinge may be taken as active or inactive.
    Also returns w_max [B (1 + N)]: the largest |dLoss/dscore_j| * B N any resolution of the ties gives triple j (a positive
    counts its possibly-active hinges, a negative is 0 or 1), in units of 1 / (B N)."""
    en, rn = _l2n64(params["ent_embeddings"]), _l2n64(params["rel_embeddings"])
    bh, bt, br = (np.asarray(x, dtype=np.int64) for x in (bh, bt, br))
    n_tr = B * (1 + N)
    kinks = []
    for lo in range(0, n_tr, chunk):
        sl = slice(lo, min(lo + chunk, n_tr))
        jj, ii = np.nonzero(np.abs(en[bh[sl]] + rn[br[sl]] - en[bt[sl]]) < kink_tol)
        kinks.append(np.stack([jj + lo, ii], 1))
    kinks = np.concatenate(kinks) if kinks else np.zeros((0, 2), np.int64)
    hm = np.asarray(hm, dtype=np.float64)
    tied = np.abs(hm) < tie_tol
    maybe = (hm >= 0) | tied                                   # TF's maximum routes the tie to the hinge: active <=> p - n + margin >= 0
    w_max = np.concatenate([maybe.sum(1).astype(np.float64), maybe.T.reshape(-1).astype(np.float64)])   # triple B(k+1)+b <-> (b, k)
    ties = np.stack(np.nonzero(tied), 1).astype(np.int64)
    return kinks.astype(np.int64), ties, w_max


def transe_row_radius(params, bh, bt, br, B, N, table, row, kinks, ties, w_max, denom=None):
    """Element-wise bound on how far two correct evaluations of dLoss/d(table[row]) can lie apart because of the step's switch
    points (transe_switch_points).  The row's gradient is  g = (S - x^ <x^, S>) / |x|  with  S = sum over the slots (triple j, role
    c = +1 head / relation, -1 tail) that address the row of  c * w_j * sign(e_j)  -- linear in every sign and in every w_j:
      * a kink element (j, i) of such a slot can move S_i by up to 2 |w_j| (from -1 to +1), |w_j| <= w_max[j] / denom;
      * a tied hinge (b, k) moves w of the positive b by 1 / denom and w of the negative (b, k) by the same amount the other way:
        S moves by (sum of c * sign(e) over the positive's slots at this row - the same over the negative's) / denom.
    The bound is the sum of the absolute effects, pushed through the (linear) normalise-backward.  -> float64 [D]"""
    denom = float(B * N if denom is None else denom)
    x = np.asarray(params[table][row], dtype=np.float64)
    nrm = np.sqrt(max(float((x * x).sum()), 1e-12))
    xh = x / nrm
    D = x.shape[0]
    bh, bt, br = (np.asarray(v, dtype=np.int64) for v in (bh, bt, br))
    ent = table == "ent_embeddings"

    def roles(j):           # signed multiplicity with which triple(s) j address this row
        if ent:
            return (bh[j] == row).astype(np.float64) - (bt[j] == row).astype(np.float64), \
                   (bh[j] == row).astype(np.float64) + (bt[j] == row).astype(np.float64)
        c = (br[j] == row).astype(np.float64)
        return c, c

    # kink budget per element: K_i = sum over kink elements (j, i) in slots of this row of 2 w_max_j / denom (per slot: a triple with
    # the row as head AND tail counts twice)
    K = np.zeros(D)
    if len(kinks):
        _, mult = roles(kinks[:, 0])
        np.add.at(K, kinks[:, 1], 2.0 * mult * w_max[kinks[:, 0]] / denom)
    # |P e_i| summed with weights K_i, P = I - x^ x^T:  element m gets K_m (1 - x^_m^2) + |x^_m| sum_{i != m} K_i |x^_i|
    A = float((K * np.abs(xh)).sum())
    radius = K * (1.0 - xh * xh) + np.abs(xh) * (A - K * np.abs(xh))
    if len(ties):
        ent_t, rel_t = params["ent_embeddings"], params["rel_embeddings"]
        for b, k in ties:
            jp, jn = int(b), int(B * (k + 1) + b)
            v = np.zeros(D)
            for j, sgn in ((jp, 1.0), (jn, -1.0)):
                c, _ = roles(np.array([j]))
                if c[0] != 0.0:      # (only the three rows of this triple are normalised: the tables may be large)
                    v += sgn * c[0] * np.sign(_l2n64(ent_t[bh[j]]) + _l2n64(rel_t[br[j]]) - _l2n64(ent_t[bt[j]]))
            if v.any():
                radius += np.abs(v - xh * float((xh * v).sum())) / denom
    return radius / nrm


def kink_rows_chunked(params, bh, bt, br, B, N, tol, chunk=100_000):
    """TransE form of torch_ref.near_kink_rows evaluated in chunks of triples (the bench batch holds 884 364 scored triples x
    200 elements: 1.4 GB per fp64 temporary if done at once).  -> ({table: set(rows)}, number of |e| < tol elements)."""
    ent = np.asarray(params["ent_embeddings"], dtype=np.float64)
    rel = np.asarray(params["rel_embeddings"], dtype=np.float64)
    l2n = lambda x: x / np.sqrt(np.maximum((x * x).sum(-1, keepdims=True), 1e-12))
    en, rn = l2n(ent), l2n(rel)
    bh, bt, br = (np.asarray(x, dtype=np.int64) for x in (bh, bt, br))
    n_tr = B * (1 + N)
    near_group = np.zeros(B, bool)
    n_el = 0
    for lo in range(0, n_tr, chunk):
        sl = slice(lo, min(lo + chunk, n_tr))
        near = np.abs(en[bh[sl]] + rn[br[sl]] - en[bt[sl]]) < tol
        n_el += int(near.sum())
        hit = np.nonzero(near.any(-1))[0] + lo
        near_group[hit % B] = True                      # triple j belongs to positive j % B (Base.cpp:109-139)
    groups = np.nonzero(near_group)[0]
    idx = (groups[:, None] + B * np.arange(N + 1)[None, :]).ravel() if len(groups) else np.zeros(0, np.int64)
    rows = {"ent_embeddings": set(bh[idx].tolist()) | set(bt[idx].tolist()), "rel_embeddings": set(br[idx].tolist())}
    return rows, n_el


def _f32_constants(beta1, beta2, eps):
    """The constants as the fp32 sweeps hold them (oracle/kge_oracle.c orc_adam_apply_dense, csrc apply kernels):
    beta as float, 1 - beta formed IN fp32 -- `1.0f - 0.999f` is 1.3e-5 (relative) away from 0.001, which through
    1/sqrt(v) is 6e-6 of a step: far more than fp32 rounding of the sweep itself."""
    f = np.float32
    b1, b2 = f(beta1), f(beta2)
    return float(b1), float(b2), float(f(1) - b1), float(f(1) - b2), float(f(eps))


def adam_step_fp64(p0, m0, v0, g, lr_t, beta1, beta2, eps):
    """The update p1 - p0 of orc_adam_apply_dense (oracle/kge_oracle.c) evaluated in fp64 on the fp32 constants, elementwise,
    for a gradient array g."""
    b1, b2, omb1, omb2, eps = _f32_constants(beta1, beta2, eps)
    touched = g != 0
    m = b1 * m0 + np.where(touched, omb1 * g, 0.0)
    v = b2 * v0 + np.where(touched, omb2 * g * g, 0.0)
    return -(float(np.float32(lr_t)) * m) / (np.sqrt(v) + eps)


def adam_update_explained(p0, m0, v0, g_o, du_engine, du_oracle, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_rtol=1e-5,
                          skip_rows=(), grad_scale=None, grad_atol=None):
    """Is every element of the engine's Adam update one that a gradient within `grad_rtol` of the oracle's can produce?

    p0, m0, v0 : the state BOTH sides started the step from (fp32 arrays, one table);  g_o : the oracle's summed gradient;
    du_* = p1 - p0 of each side (fp64).  For each element the update is evaluated in fp64 over g in [g_o - d, g_o + d] with
    d = grad_rtol * max|g_o| (the parity tolerance on the gradient itself) -- on a 17-point grid, at the interior extremum and, where
    the interval contains it, at 0 (the step is +-lr_t/sqrt(1-beta2)-like on either side of a cancelling element) -- and the
    engine's element must lie inside [min, max] of those, widened by 2e-5 of the largest step (fp32 rounding of the step, grid
    resolution at an interior extremum) and one ulp of |p0| (rounding of p0 - step).
    -> dict(unexplained=int array of flat indices, amplified=number of elements further than 1e-3 of the largest step from the
            oracle's, worst_steps=largest |du_engine - du_oracle| in units of the largest step, worst_gain=largest factor by
            which an element's step interval exceeds d scaled to step units -- the amplification the docstrings talk about)."""
    p0, m0, v0, g = (np.asarray(x, dtype=np.float64) for x in (p0, m0, v0, g_o))
    # grad_scale: the largest gradient element the tolerance is relative to (default: of g_o itself); grad_atol: a further absolute
    # allowance per element (scalar or array shaped like g: e.g. the fp32 storage quantum of a gradient read back from Adam's m)
    d = grad_rtol * (np.abs(g).max() if grad_scale is None else float(grad_scale))
    if grad_atol is not None:
        d = d + np.asarray(grad_atol, dtype=np.float64)
    # the step is not monotone in g (with m0 and g of opposite signs it has an interior extremum near g* = (1-b1) b2 v0 /
    # (b1 m0 (1-b2))): a 17-point grid over the interval plus g* itself where it falls inside
    b1f, b2f, omb1, omb2, _ = _f32_constants(beta1, beta2, eps)
    with np.errstate(divide="ignore", invalid="ignore"):
        g_star = omb1 * b2f * v0 / (b1f * m0 * omb2)
    g_star = np.where(np.isfinite(g_star), np.clip(g_star, g - d, g + d), g)
    grid = [g + t * d for t in np.linspace(-1.0, 1.0, 17)] + [g_star]
    lo = hi = None
    for gg in grid:
        s = adam_step_fp64(p0, m0, v0, gg, lr_t, beta1, beta2, eps)
        lo = s if lo is None else np.minimum(lo, s)
        hi = s if hi is None else np.maximum(hi, s)
    crosses = np.abs(g) <= d                               # the interval contains an exactly cancelling gradient: untouched-row form
    if crosses.any():
        b1f, b2f, _, _, epsf = _f32_constants(beta1, beta2, eps)
        s0 = -(float(np.float32(lr_t)) * b1f * m0) / (np.sqrt(b2f * v0) + epsf)
        lo = np.where(crosses, np.minimum(lo, s0), lo)
        hi = np.where(crosses, np.maximum(hi, s0), hi)
    step = max(np.abs(du_oracle).max(), 1e-30)
    slack = 2e-5 * step + np.abs(p0) * 2.0 ** -23        # (grid resolution at an interior extremum: ~2e-6 of a step observed)
    ok = (du_engine >= lo - slack) & (du_engine <= hi + slack)
    if len(skip_rows):
        ok[np.asarray(sorted(skip_rows), dtype=np.int64)] = True
    diff = np.abs(du_engine - du_oracle)
    amplified = diff > 1e-3 * step
    width_steps = (hi - lo) / step
    return dict(unexplained=np.flatnonzero(~ok), amplified=int(amplified.sum()), worst_steps=float(diff.max() / step),
                worst_gain=float((width_steps / (2 * grad_rtol)).max()), amplified_mask=amplified, width_steps=width_steps)


def transe_row_grad_fp64(params, bh, bt, br, B, N, table, row, hm, denom=None, chunk=50_000):
    """dLoss/d(table[row]) of one TransE step evaluated in fp64 from the formula itself (TransE.py:11-15,44-51): S = sum over the
    slots (triple j, role c) that address the row of c * w_j * sign(e_j), with e in fp64 and the hinge states the oracle's (hm >= 0:
    TF's maximum routes a tie to the hinge), then g = (S - x^ <x^, S>) / |x|.  This is what both fp32 evaluations approximate; a hub
    row that thousands of slots address carries an fp32 accumulation error in the ORACLE (its gradient rows are added one by one)
    that the engine's integer sign sums do not have, so a row that disagrees with the oracle is judged against this value."""
    denom = float(B * N if denom is None else denom)
    bh, bt, br = (np.asarray(v, dtype=np.int64) for v in (bh, bt, br))
    ent_t, rel_t = params["ent_embeddings"], params["rel_embeddings"]
    x = np.asarray(params[table][row], dtype=np.float64)
    nrm = np.sqrt(max(float((x * x).sum()), 1e-12))
    xh = x / nrm
    act = np.asarray(hm) >= 0                                            # [B, N]
    w_all = np.concatenate([act.sum(1).astype(np.float64), -act.T.reshape(-1).astype(np.float64)])   # triple B(k+1)+b <-> (b, k)
    if table == "ent_embeddings":
        slots = np.nonzero((bh == row) | (bt == row))[0]
    else:
        slots = np.nonzero(br == row)[0]
    S = np.zeros_like(x)
    for lo in range(0, len(slots), chunk):
        j = slots[lo:lo + chunk]
        j = j[w_all[j] != 0]
        if not len(j):
            continue
        sg = np.sign(_l2n64(ent_t[bh[j]]) + _l2n64(rel_t[br[j]]) - _l2n64(ent_t[bt[j]]))
        if table == "ent_embeddings":
            c = (bh[j] == row).astype(np.float64) - (bt[j] == row).astype(np.float64)
        else:
            c = np.ones(len(j))
        S += ((c * w_all[j])[:, None] * sg).sum(0)
    S /= denom
    return (S - xh * float((xh * S).sum())) / nrm


def switch_point_rows(kinks, ties, bh, bt, br, B, N):
    """{table: rows} of every group that holds a switch point -- the set round 3 excused wholesale; now only REPORTED
    (rows_in_kink_set), to show how much smaller the set of rows that actually deviate is."""
    groups = np.unique(np.concatenate([kinks[:, 0] % B, ties[:, 0]])) if (len(kinks) or len(ties)) else np.zeros(0, np.int64)
    idx = (groups[:, None] + B * np.arange(N + 1)[None, :]).ravel() if len(groups) else np.zeros(0, np.int64)
    return {"ent_embeddings": set(np.asarray(bh)[idx].tolist()) | set(np.asarray(bt)[idx].tolist()),
            "rel_embeddings": set(np.asarray(br)[idx].tolist())}


def new_adam_step_totals():
    return dict(rows_excused=0, rows_excused_needing_a_switch_point=0, rows_in_kink_set=0, rows_fully_checked=0, kink_elems=0, tie_hinges=0,
                amplified=0, worst_steps=0.0, worst_gain=0.0, loss=0.0, grad=0.0, v=0.0, worst_excused_over_radius=0.0,
                worst_oracle_fp32_error_on_excused_rows=0.0, worst_engine_error_vs_fp64_on_excused_rows=0.0)


def check_transe_adam_step(tot, step, p0, m0, v0, p1, m1, v1, g_o, p1_oracle, bh, bt, br, B, n, hm, lr_t, b1, b2, eps, rtol, kink_tol,
                           tie_tol):
    """One TransE + TF1-Adam step of an ENGINE (state p0, m0, v0 -> p1, m1, v1; dicts of fp32 arrays by table name) against the
    oracle's summed gradient g_o and updated tables p1_oracle for the same batch and starting state.  Asserts, for BOTH tables:
      gradient  read back from the first moment, m1 = b1 m0 + (1 - b1) g, within rtol of the largest element on EVERY row; a row
                outside that is re-judged against the fp64 evaluation of the formula for that row (transe_row_grad_fp64: the
                oracle adds a hub row's thousands of fp32 contributions one by one, the engine sums integers) and must lie,
                element by element, within rtol of THAT plus what the switch points of its own slots allow (transe_row_radius);
      v1        on every row, against the gradient the row was checked to have (the oracle's; an excused row: the engine's own);
      p1 - p0   every element one that a gradient within rtol of that checked gradient produces (adam_update_explained, no skips).
    Accumulates counts into `tot` (new_adam_step_totals)."""
    b1f, b2f, omb1, omb2, _ = _f32_constants(b1, b2, eps)
    kinks, ties, w_max = transe_switch_points(p0, bh, bt, br, B, n, hm, kink_tol, tie_tol)
    tot["kink_elems"] += len(kinks); tot["tie_hinges"] += len(ties)
    in_set = switch_point_rows(kinks, ties, bh, bt, br, B, n)
    for k in g_o:
        m1k, v1k = m1[k].astype(np.float64), v1[k].astype(np.float64)
        scale = np.abs(g_o[k]).max()
        g_eng = (m1k - b1f * m0[k].astype(np.float64)) / omb1
        quantum = 4 * 2.0 ** -24 * np.abs(m1k).max() / omb1                      # m1 is stored in fp32
        diff = np.abs(g_eng - g_o[k])
        bad = np.nonzero((diff > rtol * scale + quantum).any(1))[0]
        for row in bad.tolist():
            rad = transe_row_radius(p0, bh, bt, br, B, n, k, row, kinks, ties, w_max)
            g64 = transe_row_grad_fp64(p0, bh, bt, br, B, n, k, row, hm)
            d64 = np.abs(g_eng[row] - g64)
            over = d64 - (rad + rtol * scale + quantum)
            if (over > 0).any():
                e = int(np.argmax(over))
                print("UNEXPLAINED", dict(step=step, table=k, row=row, element=e, engine_minus_fp64=float(d64[e]), engine_minus_oracle=float(diff[row][e]),
                                          oracle_minus_fp64=float(abs(g_o[k][row][e] - g64[e])), radius=float(rad[e]),
                                          tol=float(rtol * scale + quantum), in_kink_set=row in in_set[k],
                                          row_norm_engine=float(np.linalg.norm(g_eng[row])), row_norm_oracle=float(np.linalg.norm(g_o[k][row]))))
            assert (over <= 0).all(), (step, k, row, "gradient row outside the tolerance of its fp64 value by more than the switch points of its own slots allow")
            beyond = d64 - rtol * scale - quantum
            if (beyond > 0).any():           # within rtol of the fp64 value only with the help of a switch point
                tot["rows_excused_needing_a_switch_point"] += 1
                frac = np.where(rad > 0, beyond / np.where(rad > 0, rad, 1.0), 0.0)
                tot["worst_excused_over_radius"] = max(tot["worst_excused_over_radius"], float(frac.max()))
            tot["worst_oracle_fp32_error_on_excused_rows"] = max(tot["worst_oracle_fp32_error_on_excused_rows"],
                                                                 float(np.abs(g_o[k][row] - g64).max() / scale))
            tot["worst_engine_error_vs_fp64_on_excused_rows"] = max(tot["worst_engine_error_vs_fp64_on_excused_rows"],
                                                                    float(np.minimum(d64, np.maximum(d64 - rad, 0)).max() / scale))
        tot["rows_excused"] += len(bad)
        tot["rows_in_kink_set"] += len(in_set[k])
        tot["rows_fully_checked"] += diff.shape[0] - len(bad)
        clean = np.ones(diff.shape[0], bool); clean[bad] = False
        # (net of the fp32 storage quantum of the first moment the gradient is read back from: the same allowance as in `bad`)
        tot["grad"] = max(tot["grad"], float(np.maximum(diff[clean] - quantum, 0.0).max() / scale))
        g_chk = g_o[k].astype(np.float64).copy()
        g_chk[bad] = g_eng[bad]
        atol = np.zeros_like(g_chk); atol[bad] = quantum
        # second moment on EVERY row: v1 = b2 v0 + (1 - b2) g^2, so a gradient within d moves it by (1 - b2)(2|g| d + d^2)
        d = rtol * scale + atol
        v_exp = b2f * v0[k].astype(np.float64) + np.where(g_chk != 0, omb2 * g_chk * g_chk, 0.0)
        dv = np.abs(v1k - v_exp)
        allow = omb2 * (2 * np.abs(g_chk) * d + d * d) + 4 * 2.0 ** -24 * np.abs(v1k).max()
        assert (dv <= allow).all(), (step, k, float((dv - allow).max()))
        tot["v"] = max(tot["v"], float(dv.max() / max(np.abs(v1k).max(), 1e-30)))
        rep = adam_update_explained(p0[k], m0[k], v0[k], g_chk, p1[k].astype(np.float64) - p0[k], p1_oracle[k].astype(np.float64) - p0[k],
                                    lr_t, b1, b2, eps, grad_rtol=rtol, grad_scale=scale, grad_atol=atol)
        D = p0[k].shape[1]
        assert rep["unexplained"].size == 0, (step, k, [(int(j // D), int(j % D)) for j in rep["unexplained"][:8]])
        tot["amplified"] += rep["amplified"]
        tot["worst_steps"] = max(tot["worst_steps"], rep["worst_steps"]); tot["worst_gain"] = max(tot["worst_gain"], rep["worst_gain"])
