#!/usr/bin/env python3
"""Step time of the gradient-accumulation paths against the step size (to place the switch-over thresholds):
TransE sign counts vs fp32 atomics, TransH float records vs fp32 atomics."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(model, dim, n, nbatches, spec, path, steps=200):
    import torch
    import openkeonspark_amd as pkg
    from openkeonspark_amd import _lib
    from openkeonspark_amd.synthetic import make_dataset
    lib = _lib.load()
    lib.kge_set_option(b"float_records", 1 if path == "records" else 0)
    lib.kge_set_option(b"float_records_min", 0)
    con = pkg.Config()
    con.use_counts = path == "counts"
    con.counts_min_records = 0
    con.set_in_path(make_dataset("/tmp/okes_%s" % spec["name"], spec)); con.set_work_threads(8); con.set_bern(1)
    con.set_dimension(dim); con.set_nbatches(nbatches); con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method("SGD")
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    for _ in range(20):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return con.batch_size, dt * 1e6


if __name__ == "__main__":
    from openkeonspark_amd.synthetic import FB15K237, WN18RR
    fb = dict(FB15K237, name="fb15k237_shaped")
    wn = dict(WN18RR, name="wn18rr_shaped")
    import contextlib, io
    for model, dim, n, spec, paths in (("TransE", 100, 1, fb, ("counts", "atomic")), ("TransE", 200, 25, fb, ("counts", "atomic")),
                                       ("TransH", 200, 1, wn, ("records", "atomic")), ("TransD", 200, 1, fb, ("records", "atomic"))):
        for nb in (100, 50, 25, 12, 6, 3):
            row = {"model": model, "dim": dim, "neg": n}
            for path in paths:
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    B, us = one(model, dim, n, nb, spec, path)
                row["batch"] = B
                row[path + "_us"] = round(us, 1)
            slots = {"TransE": 3 + n, "TransH": 4 + n, "TransD": 6 + 2 * n}[model]
            row["records"] = B * slots
            sys.stderr.write(json.dumps(row) + "\n")
