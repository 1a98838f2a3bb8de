"""Triple-classification entry points of the library (host routines, no GPU needed) against the
reference's outputs (tests/golden/tc_*.npz from the compiled reference): type-constrained negatives
drawn with the continuing libc rand() sequence, per-relation thresholds, accuracy, ROC counts."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN
from openkeonspark_amd import _lib
from openkeonspark_amd.Config import Config


@pytest.mark.parametrize("kg", ["kg_tiny", "kg_small"])
def test_classification_abi_matches_reference(kg):
    z = np.load(os.path.join(GOLDEN, "tc_%s.npz" % kg))
    L = _lib.lib()
    L.kge_set_option(b"libc_rand_restart", 1)          # the fixture was made in a fresh process ...
    con = Config()
    con.set_in_path(os.path.join(GOLDEN, kg))
    con.set_work_threads(1)                              # ... whose randReset consumed one libc draw
    con.set_test_link_prediction(True)
    con.init()
    V, T, R = L.getValidTotal(), L.getTestTotal(), con.relTotal
    vp = ctypes.c_void_p
    L.getValidBatch.argtypes = [vp] * 6; L.getTestBatch.argtypes = [vp] * 6
    L.getBestThreshold.argtypes = [vp] * 3
    L.test_triple_classification.argtypes = [vp] * 4   # SURVEY.md 8b caveat: the reference declares 3, passes 4
    L.get_n_interval.argtypes = [ctypes.c_int64, vp, vp]; L.get_n_interval.restype = ctypes.c_int64
    L.get_TPFP.argtypes = [ctypes.c_int64, vp, vp, vp, vp]; L.get_TPFP.restype = ctypes.POINTER(ctypes.c_int64)
    valid = [np.zeros(V, np.int64) for _ in range(6)]
    test = [np.zeros(T, np.int64) for _ in range(6)]
    L.getValidBatch(*[a.ctypes.data for a in valid])
    L.getTestBatch(*[a.ctypes.data for a in test])
    _lib.raise_if_error(L)
    assert np.array_equal(np.stack(valid), z["valid"])
    assert np.array_equal(np.stack(test), z["test"])
    vpos, vneg, tpos, tneg = (np.ascontiguousarray(z[k]) for k in ("vpos", "vneg", "tpos", "tneg"))  # keep them alive
    thresh = np.full(R, -1.0, np.float32)
    L.getBestThreshold(thresh.ctypes.data, vpos.ctypes.data, vneg.ctypes.data)
    assert thresh.tobytes() == z["thresh"].tobytes()
    acc = np.zeros(1, np.float32)
    L.test_triple_classification(thresh.ctypes.data, tpos.ctypes.data, tneg.ctypes.data, acc.ctypes.data)
    assert acc.tobytes() == z["acc"].tobytes()
    for r in range(R):
        assert L.get_n_interval(r, vpos.ctypes.data, vneg.ctypes.data) == z["n_interval"][r]
        key = "tpfp_%d" % r
        ptr = L.get_TPFP(r, vpos.ctypes.data, vneg.ctypes.data, tpos.ctypes.data, tneg.ctypes.data)
        if key in z.files:
            got = [ptr[i] for i in range(len(z[key]))]
            if r in set(z["test"][2].tolist()):
                assert got == z[key].tolist(), r
            else:
                # the reference loops i = testLef[r] .. testRig[r] = -1 .. -1 for a relation without test
                # triples and reads score[-1] (Test.h:432-436): undefined there, all-zero counts here
                assert not any(got)
        else:
            assert not ptr
