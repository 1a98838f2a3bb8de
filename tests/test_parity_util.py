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
