"""HIP sampler through the C ABI versus the reference's golden vectors and the pinned oracle.
Bit-exact (integer work)."""
import ctypes
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle

pytestmark = pytest.mark.gpu

GRID_SHAPE = [(7, 1, 0), (64, 2, 1), (64, 25, 0), (50, 3, 0)]
CALLS = 3
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "kg_*_W*_bern*.npz")))


def _parse(path):
    name = os.path.basename(path)[:-4]
    kg, w, b = name.rsplit("_", 2)
    return kg, int(w[1:]), int(b[4:])


def make_config(path, W, bern, **kw):
    from openkeonspark_amd.Config import Config
    con = Config()
    con.set_in_path(path)
    con.set_work_threads(W)
    con.set_bern(bern)
    for k, v in kw.items():
        getattr(con, "set_" + k)(v)
    con.init()
    return con


def abi_sampling(con, B, n, nr):
    """Call the Base.so-compatible `sampling` exactly like Config.py:347 does."""
    tot = B * (1 + n + nr)
    h = np.zeros(tot, np.int64); t = np.zeros(tot, np.int64); r = np.zeros(tot, np.int64)
    y = np.zeros(tot, np.float32)
    con.lib.kge_clear_error()
    con.lib.sampling(h.ctypes.data, t.ctypes.data, r.ctypes.data, y.ctypes.data, B, n, nr)
    from openkeonspark_amd import _lib
    _lib.raise_if_error(con.lib)
    return h, t, r, y


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_sampling_abi_matches_reference_fixture(path):
    kg_name, W, bern = _parse(path)
    z = np.load(path)
    con = make_config(os.path.join(GOLDEN, kg_name), W, bern)
    # the library's libc-seed sequence continues across randReset calls in one process, like the
    # reference's; start every case from the fixture's (fresh-process) seeds
    seeds = np.ascontiguousarray(z["seeds"], dtype=np.uint64)
    assert con.lib.kge_set_stream_states(seeds.ctypes.data, W) == 0
    for si, (B, n, nr) in enumerate(GRID_SHAPE):
        for c in range(CALLS):
            h, t, r, y = abi_sampling(con, B, n, nr)
            ref = z["s%d_c%d" % (si, c)]
            assert np.array_equal(h, ref[0]) and np.array_equal(t, ref[1]) and np.array_equal(r, ref[2]), (si, c)
            assert np.array_equal(y, z["y%d_c%d" % (si, c)])
    assert con.get_stream_states().tolist() == z["final_states"].tolist()


def test_first_process_seeds_are_the_unseeded_glibc_sequence():
    # run in a child so that the library's libc generator is fresh (Random.h:9-13)
    import subprocess
    import sys
    code = ("import sys;sys.path.insert(0,%r);import numpy as np;"
            "from openkeonspark_amd.Config import Config;c=Config();c.set_in_path(%r);c.init();"
            "print(c.get_stream_states().tolist())" % (os.path.dirname(os.path.dirname(GOLDEN)), os.path.join(GOLDEN, "kg_tiny")))
    out = subprocess.check_output([sys.executable, "-c", code]).decode().strip().splitlines()[-1]
    assert json.loads(out) == [1804289383, 846930886, 1681692777, 1714636915, 1957747793, 424238335, 719885386, 1649760492]


def test_fb15k237_shaped_streams_match_reference_digests(fb_dir):
    digests = json.load(open(os.path.join(GOLDEN, "fb_digests.json")))
    seeds8 = np.array(oracle.libc_rand_sequence(8), dtype=np.uint64)
    for name, d in digests.items():
        con = make_config(fb_dir, d["W"], d["bern"])
        seeds = np.ascontiguousarray(seeds8[:d["W"]])
        con.lib.kge_set_stream_states(seeds.ctypes.data, d["W"])
        assert [con.entTotal, con.relTotal, con.lib.getTrainTotal(), con.trainTotal, con.bt] == d["totals"]
        hsh = hashlib.sha256()
        for c in range(d["calls"]):
            h, t, r, y = abi_sampling(con, d["B"], d["n"], d["nr"])
            hsh.update(h.tobytes()); hsh.update(t.tobytes()); hsh.update(r.tobytes())
            if c == 0:
                B = d["B"]
                assert [h[:4].tolist(), t[:4].tolist(), r[:4].tolist(), h[B:B + 4].tolist(), t[B:B + 4].tolist()] == d["first"]
        assert hsh.hexdigest() == d["sha256"], name
        assert [int(x) for x in con.get_stream_states()] == d["final_states"]


def test_device_slices_of_ranks_union_to_the_global_batch(fb_dir):
    """kge_sampling_device for thread ranges (what data-parallel ranks call) == slices of the full batch."""
    import torch
    W, B, n, nr = 8, 2721, 3, 1
    full = oracle.KG(fb_dir, work_threads=W, bern=1)
    con = make_config(fb_dir, W, 1)
    seeds = np.ascontiguousarray(full.stream_states())
    for G in (1, 2, 4, 8):
        con.lib.kge_set_stream_states(seeds.ctypes.data, W)
        full.set_stream_states(seeds)
        for call in range(2):
            rh, rt, rr, _ = full.sampling(B, n, nr)
            states_before = con.get_stream_states().copy()
            got = [np.zeros_like(rh) for _ in range(3)]
            for g in range(G):
                con.lib.kge_set_stream_states(states_before.ctypes.data, W)  # every rank starts from the same state
                lo, hi = g * W // G, (g + 1) * W // G
                first = ctypes.c_int64()
                cnt = con.lib.kge_slice_positions(B, lo, hi, ctypes.byref(first))
                buf = torch.zeros((3, max(cnt, 1) * (1 + n + nr)), dtype=torch.int32, device="cuda")
                nl = ctypes.c_int64()
                rc = con.lib.kge_sampling_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), B, n, nr, lo, hi,
                                                 max(cnt, 1), ctypes.byref(nl), None)
                assert rc == 0 and nl.value == cnt
                host = buf.cpu().numpy()
                for k in range(1 + n + nr):
                    for a in range(3):
                        got[a][k * B + first.value:k * B + first.value + cnt] = host[a][k * max(cnt, 1):k * max(cnt, 1) + cnt]
            assert np.array_equal(got[0], rh) and np.array_equal(got[1], rt) and np.array_equal(got[2], rr), (G, call)
            # every rank advanced ALL streams as one reference call does
            assert con.get_stream_states().tolist() == full.stream_states().tolist()


def test_ragged_and_edge_batches(fb_dir):
    """batchSize not divisible by workThreads, workThreads > batchSize (empty slices), B=1."""
    for W, B, n, nr in [(8, 13, 2, 0), (8, 5, 1, 1), (3, 1, 4, 0), (16, 2721, 1, 0)]:
        kg = oracle.KG(fb_dir, work_threads=W, bern=0)
        con = make_config(fb_dir, W, 0)
        seeds = np.ascontiguousarray(kg.stream_states())
        con.lib.kge_set_stream_states(seeds.ctypes.data, W)
        for c in range(3):
            want = kg.sampling(B, n, nr)
            got = abi_sampling(con, B, n, nr)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (W, B, n, nr, c)
        assert con.get_stream_states().tolist() == kg.stream_states().tolist()


def test_many_negatives_and_saturated_groups(tmp_path):
    """Edge cases of the lane mapping and of the division-free modulo: more draws per positive than a wave has lanes
    (1 + 40 + 30 = 71 -> 128 slots), a relation corruption set that leaves exactly ONE candidate (modulo 1), entity
    groups covering all but one entity (modulo 1 again), against the oracle, bit for bit."""
    d = str(tmp_path / "kg_dense")
    os.makedirs(d)
    E, R = 5, 33
    trip = []
    for t in range(1, E):                     # head 0 reaches every other entity through relation 0: tails(0, r0) = E - 1
        trip.append((0, t, 0))
    for r in range(R - 1):                    # (1, 2) is linked by all relations but the last: rels(1, 2) = R - 1
        trip.append((1, 2, r))
    for h in range(2, E):
        trip.append((h, (h + 1) % E, 5))
    open(os.path.join(d, "entity2id.txt"), "w").write("%d\n" % E)
    open(os.path.join(d, "relation2id.txt"), "w").write("%d\n" % R)
    with open(os.path.join(d, "train2id.txt"), "w") as f:
        f.write("%d\n" % len(trip))
        for h, t, r in trip:
            f.write("%d %d %d\n" % (h, t, r))
    for W, bern, B, n, nr in [(2, 1, 40, 40, 30), (3, 0, 17, 63, 1), (1, 1, 9, 1, 0)]:
        kg = oracle.KG(d, work_threads=W, bern=bern)
        con = make_config(d, W, bern)
        seeds = np.ascontiguousarray(kg.stream_states())
        con.lib.kge_set_stream_states(seeds.ctypes.data, W)
        for c in range(3):
            want = kg.sampling(B, n, nr)
            got = abi_sampling(con, B, n, nr)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (W, bern, B, n, nr, c)
        assert con.get_stream_states().tolist() == kg.stream_states().tolist()
