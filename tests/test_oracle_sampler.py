"""The CPU oracle's sampler restatement against the reference's golden vectors (bit-exact).

Fixtures come from the reference's own C++ (tests/golden/make_golden.py).  When oracle/_ref/Base.so
is present (build container) the oracle is additionally compared live with it.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle

GRID_SHAPE = [(7, 1, 0), (64, 2, 1), (64, 25, 0), (50, 3, 0)]
CALLS = 3
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "kg_*_W*_bern*.npz")))


def _parse(path):
    name = os.path.basename(path)[:-4]
    kg, w, b = name.rsplit("_", 2)
    return kg, int(w[1:]), int(b[4:])


def test_libc_seed_sequence():
    # Random.h:9-13: unseeded glibc rand(); SURVEY.md A1 lists the first eight
    assert oracle.libc_rand_sequence(8) == [1804289383, 846930886, 1681692777, 1714636915,
                                            1957747793, 424238335, 719885386, 1649760492]


def test_fixture_grid_is_complete():
    assert len(FIXTURES) == 3 * 4 * 2


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_matches_reference_fixture(path):
    kg_name, W, bern = _parse(path)
    z = np.load(path)
    kg = oracle.KG(os.path.join(GOLDEN, kg_name), work_threads=W, bern=bern)
    assert [kg.entTotal, kg.relTotal, kg.trainTotal, kg.trainTotal_, kg.batchTotal] == z["totals"].tolist()
    assert kg.stream_states().tolist() == z["seeds"].tolist()
    # float arrays: bit-exact including the NaN of never-used relations (0/0, Reader.h:174-177)
    assert kg.left_mean().tobytes() == z["left_mean"].tobytes()
    assert kg.right_mean().tobytes() == z["right_mean"].tobytes()
    for which in ("head", "tail", "rel"):
        assert np.array_equal(kg.sorted_copy(which), z["by_" + which])
    for si, (B, n, nr) in enumerate(GRID_SHAPE):
        for c in range(CALLS):
            h, t, r, y = kg.sampling(B, n, nr)
            ref = z["s%d_c%d" % (si, c)]
            assert np.array_equal(h, ref[0]) and np.array_equal(t, ref[1]) and np.array_equal(r, ref[2]), (si, c)
            assert np.array_equal(y, z["y%d_c%d" % (si, c)])
    assert kg.stream_states().tolist() == z["final_states"].tolist()


def test_oracle_matches_fb_digests(fb_dir):
    digests = json.load(open(os.path.join(GOLDEN, "fb_digests.json")))
    for name, d in digests.items():
        if d["B"] * (1 + d["n"] + d["nr"]) * d["calls"] > 4_000_000:
            continue  # the big one is covered on the GPU side
        kg = oracle.KG(fb_dir, work_threads=d["W"], bern=d["bern"])
        assert [kg.entTotal, kg.relTotal, kg.trainTotal, kg.trainTotal_, kg.batchTotal] == d["totals"]
        hsh = hashlib.sha256()
        for c in range(d["calls"]):
            h, t, r, y = kg.sampling(d["B"], d["n"], d["nr"])
            hsh.update(h.tobytes()); hsh.update(t.tobytes()); hsh.update(r.tobytes())
        assert hsh.hexdigest() == d["sha256"], name
        assert [int(x) for x in kg.stream_states()] == d["final_states"]


@pytest.mark.skipif(not os.path.exists(oracle.REF_LIB_PATH), reason="oracle/_ref/Base.so not built")
def test_oracle_matches_live_reference(tmp_path):
    """Live cross-check against the compiled reference on a fresh random graph (child process:
    Base.so holds one dataset per process)."""
    import subprocess
    import sys
    from openkeonspark_amd.synthetic import generate_triples, write_openke_dir
    h, t, r = generate_triples(300, 9, 2500, seed=99, dup_frac=0.02)
    d = write_openke_dir(str(tmp_path / "kg"), 300, 9, h, t, r) + "/"
    code = ("import sys,numpy as np;sys.path.insert(0,%r);from oracle.oracle import ReferenceSampler as R;"
            "s=R(%r,5,1);o=[np.stack(s.sampling(333,4,2)[:3]) for _ in range(4)];np.save(%r,np.stack(o))"
            % (os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0], d, str(tmp_path / "ref.npy")))
    subprocess.check_call([sys.executable, "-c", code], stdout=subprocess.DEVNULL)
    ref = np.load(str(tmp_path / "ref.npy"))
    kg = oracle.KG(d, work_threads=5, bern=1)
    for c in range(4):
        got = np.stack(kg.sampling(333, 4, 2)[:3])
        assert np.array_equal(got, ref[c])
