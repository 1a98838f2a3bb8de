"""Device-side index build (index_build.hip) against the host build (kg_index.cpp, itself pinned
to the compiled reference through the sampler fixtures): every array identical, bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_index import index_array

pytestmark = pytest.mark.gpu

ARRAYS = [("pos", np.int32, 4), ("grp", np.int32, 4), ("ht", np.int32, 2), ("tails_hr", np.int32, None),
          ("heads_tr", np.int32, None), ("rels_ht", np.int32, None), ("left_mean", np.float32, None),
          ("right_mean", np.float32, None), ("bern_prob", np.float32, None)]


def snapshot(L):
    out = {name: index_array(L, name, dt, cols) for name, dt, cols in ARRAYS}
    out["totals"] = np.array([L.getEntityTotal(), L.getRelationTotal(), L.getTrainTotal(), L.getTrainTotal_(), L.getBatchTotal()])
    return out


def build(where, E, R, h, t, r, nb=0):
    from openkeonspark_amd.Config import Config
    from openkeonspark_amd import _lib
    con = Config()
    L = con.lib
    _lib.check(L.kge_set_option(b"index_device_min", 0 if where == "device" else -1), L)
    try:
        con.set_work_threads(4)
        con.set_bern(1)
        con.set_nbatches(7)
        con.init_from_arrays(E, R, h, t, r, new_batch_total=nb)
    finally:
        L.kge_set_option(b"index_device_min", 1 << 22)
    return con


def assert_same(a, b):
    for k in a:
        assert a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k], equal_nan=True), k   # relations absent from train have NaN means (Reader.h 0/0)


def read_kg(name):
    d = os.path.join(GOLDEN, name)
    first = lambda f: int(open(os.path.join(d, f)).readline())
    a = np.loadtxt(os.path.join(d, "train2id.txt"), skiprows=1, dtype=np.int64).reshape(-1, 3)
    nb = first("batch2id.txt") if os.path.exists(os.path.join(d, "batch2id.txt")) else 0
    return first("entity2id.txt"), first("relation2id.txt"), a[:, 0], a[:, 1], a[:, 2], nb


@pytest.mark.parametrize("kg_name", ["kg_tiny", "kg_small", "kg_incr"])
def test_device_index_equals_host_index_on_golden_kgs(kg_name):
    E, R, h, t, r, nb = read_kg(kg_name)
    host = snapshot(build("host", E, R, h, t, r, nb).lib)
    dev = snapshot(build("device", E, R, h, t, r, nb).lib)
    assert_same(host, dev)


@pytest.mark.parametrize("E,R,n,seed", [(50, 3, 4000, 1), (5000, 40, 60000, 2), (14541, 237, 272115, 3), (1, 1, 5, 4),
                                        (200000, 1000, 300000, 5)])
def test_device_index_equals_host_index_random(E, R, n, seed):
    """Heavy duplication (first case), hub entities (Zipf draws), a degenerate one-entity KG, a wide id space."""
    from openkeonspark_amd.synthetic import generate_triples
    h, t, r = generate_triples(E, R, n, seed, dup_frac=0.05)
    host = snapshot(build("host", E, R, h, t, r).lib)
    dev = snapshot(build("device", E, R, h, t, r).lib)
    assert host["totals"][2] < host["totals"][3] or n <= 5      # duplicates really were present
    assert_same(host, dev)


def test_sampler_runs_identically_from_the_device_built_index():
    """End to end: batches drawn from the device-built index equal those drawn from the host-built one."""
    E, R, h, t, r, nb = read_kg("kg_small")
    outs = []
    for where in ("host", "device"):
        con = build(where, E, R, h, t, r)
        seeds = np.arange(1, 5, dtype=np.uint64) * np.uint64(2654435761)
        assert con.lib.kge_set_stream_states(seeds.ctypes.data, 4) == 0
        B, n = 300, 5
        tot = B * (1 + n)
        bh = np.zeros(tot, np.int64); bt = np.zeros(tot, np.int64); br = np.zeros(tot, np.int64); by = np.zeros(tot, np.float32)
        res = []
        for _ in range(3):
            con.lib.sampling(bh.ctypes.data, bt.ctypes.data, br.ctypes.data, by.ctypes.data, B, n, 0)
            res.append((bh.copy(), bt.copy(), br.copy()))
        outs.append(res)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_device_index_reports_out_of_range_ids_like_the_host():
    from openkeonspark_amd import KgeError
    h = np.array([0, 1, 2, 9], dtype=np.int64); t = np.array([1, 2, 0, 1], dtype=np.int64); r = np.zeros(4, np.int64)
    msgs = []
    for where in ("host", "device"):
        with pytest.raises(KgeError) as ei:
            build(where, 5, 2, h, t, r)
        msgs.append(str(ei.value))
    assert "line 5" in msgs[0] and msgs[0] == msgs[1]
