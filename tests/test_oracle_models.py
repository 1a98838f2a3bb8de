"""The C oracle's model arithmetic (fp32, hand-written backward) against fp64 torch.autograd.

The oracle is 'parity unpinned' versus TensorFlow (TF1 is an absent, un-vendored
dependency and the reference holds no tests); this file pins it to an independent autodiff of the
same graph instead.  Tolerance 1e-5 relative (BASELINE.json north_star) on the gradient scale.
"""
import numpy as np
import pytest

from oracle import oracle
import torch_ref


def rand_batch(rng, E, R, B, n, nr):
    """A sampler-shaped batch: every negative shares r and one entity with its positive."""
    N = n + nr
    h = np.zeros(B * (1 + N), np.int64); t = h.copy(); r = h.copy()
    h[:B] = rng.integers(0, E, B); t[:B] = rng.integers(0, E, B); r[:B] = rng.integers(0, R, B)
    for k in range(N):
        s = slice(B * (k + 1), B * (k + 2))
        h[s], t[s], r[s] = h[:B], t[:B], r[:B]
        if k < n:
            side = rng.random(B) < 0.5
            new = rng.integers(0, E, B)
            h[s] = np.where(side, new, h[:B]); t[s] = np.where(side, t[:B], new)
        else:
            r[s] = rng.integers(0, R, B)
    return h, t, r


def relerr(a, b):
    scale = np.abs(b).max() + 1e-30
    return np.abs(a - b).max() / scale


CASES = [("transe", 50, 7, 16, 16), ("transe", 40, 5, 100, 100), ("transh", 50, 7, 24, 24),
         ("transd", 50, 7, 20, 20), ("transr", 30, 5, 12, 8), ("transr", 30, 5, 16, 16)]


@pytest.mark.parametrize("model,E,R,De,Dr", CASES)
@pytest.mark.parametrize("n,nr", [(1, 0), (3, 0), (2, 1)])
def test_grad_matches_autograd(model, E, R, De, Dr, n, nr):
    if model == "transr" and nr > 0 and (n + nr) > 1:
        pytest.skip("reference graph itself is ill-shaped here (TransR.py:62 reshape), see torch_ref")
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((model, De, n, nr)).encode()))
    B = 37
    params = oracle.init_params(oracle.MODEL_IDS[model], E, R, De, Dr, seed=3)
    for k in params:  # larger values than xavier so that margins are active and inactive
        params[k] = (params[k] * 3).astype(np.float32)
    bh, bt, br = rand_batch(rng, E, R, B, n, nr)
    m = oracle.Model(model, E, R, De, Dr, margin=0.7, negative_rel=nr, params=params)
    loss, g = m.grad(bh, bt, br, B, n + nr)
    loss64, g64 = torch_ref.loss_and_grads(model, params, bh, bt, br, B, n + nr, 0.7, De, Dr, nr)
    assert abs(loss - loss64) <= 1e-5 * abs(loss64)
    assert abs(m.loss(bh, bt, br, B, n + nr) - loss64) <= 1e-5 * abs(loss64)
    for k in g:
        assert relerr(g[k], g64[k]) < 1e-5, k


def test_threads_do_not_change_the_sum_beyond_rounding():
    rng = np.random.default_rng(5)
    E, R, D, B, n = 200, 9, 32, 257, 4
    bh, bt, br = rand_batch(rng, E, R, B, n, 0)
    m = oracle.Model("transe", E, R, D, seed=1)
    l1, g1 = m.grad(bh, bt, br, B, n, nthreads=1)
    l4, g4 = m.grad(bh, bt, br, B, n, nthreads=4)
    assert l1 == l4
    for k in g1:
        assert relerr(g4[k], g1[k]) < 1e-6


def test_sgd_sequential_equals_dense_up_to_rounding():
    rng = np.random.default_rng(6)
    E, R, D, B, n = 60, 5, 16, 64, 2
    bh, bt, br = rand_batch(rng, E, R, B, n, 0)
    a = oracle.Model("transe", E, R, D, seed=2)
    b = oracle.Model("transe", E, R, D, seed=2)
    la = a.sgd_step(bh, bt, br, B, n, 0.01, sequential=True)
    lb = b.sgd_step(bh, bt, br, B, n, 0.01, sequential=False)
    assert la == lb
    for k in a.params:
        assert np.abs(a.params[k] - b.params[k]).max() < 1e-6
        assert not np.array_equal(a.params[k], oracle.init_params(0, E, R, D, D, seed=2)[k])


def test_adam_step_matches_torch_dense_formula():
    """TF1 sparse Adam == dense Adam on the summed gradient: every row decays and moves."""
    rng = np.random.default_rng(7)
    E, R, D, B, n = 40, 4, 8, 16, 1
    bh, bt, br = rand_batch(rng, E, R, B, n, 0)
    m = oracle.Model("transe", E, R, D, seed=4)
    p0 = {k: v.copy() for k, v in m.params.items()}
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    mm = {k: np.zeros_like(v, dtype=np.float64) for k, v in p0.items()}
    vv = {k: np.zeros_like(v, dtype=np.float64) for k, v in p0.items()}
    p64 = {k: v.astype(np.float64) for k, v in p0.items()}
    for step in range(1, 4):
        _, g = torch_ref.loss_and_grads("transe", {k: v.astype(np.float32) for k, v in p64.items()},
                                        bh, bt, br, B, n, 1.0, D, D)
        lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        for k in p64:
            mm[k] = b1 * mm[k] + (1 - b1) * g[k]
            vv[k] = b2 * vv[k] + (1 - b2) * g[k] ** 2
            p64[k] = p64[k] - lr_t * mm[k] / (np.sqrt(vv[k]) + eps)
        m.adam_step(bh, bt, br, B, n, lr, b1, b2, eps)
    untouched = np.setdiff1d(np.arange(E), np.concatenate([bh, bt]))
    assert len(untouched) > 0
    for k in p64:
        assert relerr(m.params[k], p64[k]) < 2e-5
    # rows never touched keep m=v=0 and therefore do not move; touched rows keep moving afterwards
    assert np.array_equal(m.params["ent_embeddings"][untouched], p0["ent_embeddings"][untouched])


def test_predict_reductions():
    # TransE predicts reduce_mean over the dimension (TransE.py:58), the others reduce_sum
    E, R, D = 20, 3, 8
    h = np.array([1, 2, 3]); t = np.array([4, 5, 6]); r = np.array([0, 0, 0])
    for model in ("transe", "transh", "transd", "transr"):
        m = oracle.Model(model, E, R, D, seed=9)
        s = m.predict(h, t, r)
        P = {k: __import__("torch").tensor(v, dtype=__import__("torch").float64) for k, v in m.params.items()}
        import torch
        if model == "transe":
            ref = torch_ref.calc(P["ent_embeddings"][h], P["ent_embeddings"][t], P["rel_embeddings"][r]).mean(-1)
            assert np.allclose(s, ref.numpy(), rtol=1e-5)
        assert s.shape == (3,) and np.all(s > 0)
