"""Shared accounting for the two places where two correct fp32 evaluations of the margin-ranking loss differ by more than
rounding, and for what TF1's Adam does to such differences.

(1) the kink of |e| (TransE.py:15): an element of e = h^ + r^ - t^ within rounding of zero gets sign +1 from one summation order
    and -1 (or 0) from another, which moves the gradient rows of its whole group          -> kink_rows_chunked (TransE, any size)
(2) Adam (distribute_training.py:95-96, TF1 _apply_sparse_shared) is scale-free in the gradient: the step of an element is
    lr_t * m / (sqrt(v) + eps) with m, v built from g, so a gradient element that nearly cancels turns a difference far inside
    the 1e-5 gradient tolerance into a visible fraction of one step                         -> adam_update_explained

Neither helper loosens a tolerance: each says, element by element, whether a difference is one that the stated gradient
tolerance can produce, and the tests fail on any element that is not."""
import numpy as np


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
                          skip_rows=()):
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
    d = grad_rtol * np.abs(g).max()
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
