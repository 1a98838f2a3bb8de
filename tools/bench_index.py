#!/usr/bin/env python3
"""Index build time, host (kg_index.cpp: std::sort) vs device (index_build.hip: rocPRIM), on a uniform random KG.
usage: bench_index.py TRIPLES [ENTITIES] [RELATIONS]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    n = int(sys.argv[1])
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    rng = np.random.default_rng(5)
    h = rng.integers(0, E, n, dtype=np.int64)
    t = rng.integers(0, E, n, dtype=np.int64)
    r = rng.integers(0, R, n, dtype=np.int64)
    from openkeonspark_amd import _lib
    L = _lib.load()
    out = {"triples": n, "entities": E, "relations": R}
    for where, opt in (("device", 0), ("host", -1)):
        if where == "host" and os.environ.get("SKIP_HOST") == "1":
            continue
        _lib.check(L.kge_set_option(b"index_device_min", opt), L)
        t0 = time.time()
        _lib.check(L.kge_import_train_arrays(E, R, n, h.ctypes.data, t.ctypes.data, r.ctypes.data, 0), L)
        out[where + "_seconds"] = round(time.time() - t0, 3)
        out[where + "_unique"] = int(L.getTrainTotal())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
