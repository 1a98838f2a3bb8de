"""Link-prediction ranker on the device: the Base.so-compatible testHead/testTail (host score vectors)
against the reference's golden outputs, and the device-native kge_link_prediction (scores never leave
the GPU) against oracle predict + oracle ranker.  Integer results: bit-exact."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, parity_report
from oracle import oracle

pytestmark = pytest.mark.gpu


def make_config(kg, model="TransE", dim=32):
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.set_in_path(os.path.join(GOLDEN, kg))
    con.set_work_threads(1)
    con.set_dimension(dim)
    con.set_test_link_prediction(True)
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    return con


@pytest.mark.parametrize("kg", ["kg_tiny", "kg_small"])
def test_testhead_testtail_abi_matches_reference(kg):
    z = np.load(os.path.join(GOLDEN, "lp_%s.npz" % kg))
    con = make_config(kg)
    L = con.lib
    assert [L.getTestTotal(), L.getValidTotal(), L.getTripleTotal()] == z["totals"].tolist()
    E = con.entTotal
    for i in range(len(z["out"])):
        # getTailBatch / getHeadBatch exactly as distribute_training.py:467,532 call them
        h = np.zeros(E, np.int64); t = np.zeros(E, np.int64); r = np.zeros(E, np.int64)
        L.getTailBatch(i, h.ctypes.data, t.ctypes.data, r.ctypes.data)
        assert (h == z["triples"][i, 0]).all() and (t == np.arange(E)).all() and (r == z["triples"][i, 2]).all()
        L.getHeadBatch(i, h.ctypes.data, t.ctypes.data, r.ctypes.data)
        assert (h == np.arange(E)).all() and (t == z["triples"][i, 1]).all()
        for side, fn in ((0, L.testHead), (1, L.testTail)):
            sc = np.ascontiguousarray(z["scores"][i, side])
            got = list(fn(i, sc.ctypes.data).contents)
            assert got == z["out"][i, side].tolist(), (i, side)


@pytest.mark.parametrize("model", ["TransE", "TransH", "TransD", "TransR"])
def test_device_link_prediction_matches_oracle(model):
    kg = "kg_small"
    con = make_config(kg, model, dim=24)
    ev = oracle.Eval(os.path.join(GOLDEN, kg))
    params = con.get_parameters()
    orc = oracle.Model(model.lower(), con.entTotal, con.relTotal, 24, 24, params=params)
    n = 12
    out, metrics = con.link_prediction(first=3, count=n, test_head=True)
    E = con.entTotal
    ar = np.arange(E)
    mismatched = 0
    for k in range(n):
        i = 3 + k
        h, t, r = ev.test_triple(i)
        tail_scores = orc.predict(np.full(E, h), ar, np.full(E, r))
        head_scores = orc.predict(ar, np.full(E, t), np.full(E, r))
        want_t = ev.rank(i, tail_scores, head=False)
        want_h = ev.rank(i, head_scores, head=True)
        # scores are fp32 from two implementations: a candidate within an ulp of the target may flip a count
        for got, want in ((out[k, 0], want_t), (out[k, 1], want_h)):
            if got.tolist() != want.tolist():
                mismatched += 1
                assert np.abs(got[:4] - want[:4]).max() <= 1, (k, got, want)
    parity_report("device_link_prediction_vs_oracle[%s]" % model,
                  eight_vectors_differing_by_one_count=mismatched, of=2 * n, bound=1)
    assert mismatched <= 1
    assert 0.0 <= metrics["r_filter_tot"] <= 1.0 and metrics["r_rank"] >= 1.0 and metrics["l_filter_rank"] >= 1.0
    assert metrics["r_filter_rank"] <= metrics["r_rank"]


@pytest.mark.parametrize("model", ["TransE", "TransH", "TransD", "TransR"])
@pytest.mark.parametrize("test_head", [True, False])
def test_relation_grouped_ranker_equals_generic_predict_path(model, test_head):
    """kge_link_prediction builds one table of projected + normalised candidates per relation and streams it against
    every request of that relation; the older route materialises getHeadBatch / getTailBatch and runs kge_predict.
    Same functions, same operation order: identical 8-vectors for the WHOLE test set (ragged relation groups,
    relations with a single test triple, both sides)."""
    kg = "kg_small"
    con = make_config(kg, model, dim=40)
    import torch
    for t in con._tables:                       # spread the scores: xavier-initialised tables rank almost at random
        t.mul_(3.0)
    L = con.lib
    outs = []
    for v1 in (1, 0):
        L.kge_set_option(b"lp_v1", v1)
        try:
            out, met = con.link_prediction(test_head=test_head)
        finally:
            L.kge_set_option(b"lp_v1", 0)
        outs.append((out, met))
    assert outs[0][0].shape[0] == L.getTestTotal()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    assert outs[0][0][:, 0, 0].max() > 0
    # a sub-range starting inside a relation group
    L.kge_set_option(b"lp_v1", 0)
    sub, _ = con.link_prediction(first=5, count=17, test_head=test_head)
    assert np.array_equal(sub, outs[1][0][5:22])


def test_config_test_and_predict_helpers(capsys):
    """Config.test() (triple classification with validation-fitted thresholds + link prediction) and the predict_*
    helpers of the reference class (Config.py:491-516, 574-663) against their definitions over test_step."""
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.set_in_path(os.path.join(GOLDEN, "kg_small"))
    con.set_work_threads(1); con.set_dimension(16)
    con.set_test_link_prediction(True); con.set_test_triple_classification(True)
    con.init()
    con.set_model_and_session(pkg.TransE)
    for t in con._tables:
        t.mul_(3.0)
    res = con.test()
    assert 0.0 <= res["acc"] <= 1.0 and res["r_filter_rank"] >= 1.0
    # the accuracy is the definition: positives at or below, negatives above the relation's fitted threshold
    pos = con.test_step(con.test_pos_h, con.test_pos_t, con.test_pos_r).reshape(-1)
    neg = con.test_step(con.test_neg_h, con.test_neg_t, con.test_neg_r).reshape(-1)
    # (relations without validation triples have no threshold and are skipped, Test.h:353)
    seen = np.isin(con.test_pos_r, np.unique(con.valid_pos_r))
    want = ((pos <= con.relThresh[con.test_pos_r])[seen].sum() + (neg > con.relThresh[con.test_neg_r])[seen].sum()) / (2.0 * seen.sum())
    assert abs(res["acc"] - want) < 1e-6
    E, R = con.entTotal, con.relTotal
    ar = np.arange(E)
    heads = con.predict_head_entity(5, 2, 7)
    assert heads.tolist() == con.test_step(ar, np.full(E, 5), np.full(E, 2)).reshape(-1).argsort()[:7].tolist()
    tails = con.predict_tail_entity(9, 1, 4)
    assert tails.tolist() == con.test_step(np.full(E, 9), ar, np.full(E, 1)).reshape(-1).argsort()[:4].tolist()
    rels = con.predict_relation(3, 8, 3)
    assert rels.tolist() == con.test_step(np.full(R, 3), np.full(R, 8), np.arange(R)).reshape(-1).argsort()[:3].tolist()
    s = float(con.test_step([3], [8], [rels[0]])[0])
    assert con.predict_triple(3, 8, int(rels[0]), thresh=s + 1.0) is True
    assert con.predict_triple(3, 8, int(rels[0]), thresh=s - 1.0) is False
    assert con.predict_triple(3, 8, int(rels[0])) in (True, False)
    assert "is correct" in capsys.readouterr().out
