#!/usr/bin/env python3
"""Where does the pair-count path start to pay?  TransH / TransD step time at n = 1..4 negatives with the path forced on / off."""
import json, os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237, WN18RR


def run(model, spec, n, nbatches, min_neg, steps=60, warmup=10):
    d = make_dataset("/tmp/okes_%s" % spec["name"], spec)
    pkg._lib.lib().kge_set_option(b"pair_counts_min_neg", min_neg)
    con = pkg.Config()
    con.set_in_path(d); con.set_work_threads(8); con.set_bern(1); con.set_dimension(200)
    con.set_nbatches(nbatches); con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method("SGD")
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    for _ in range(warmup):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    return con.batch_size, (time.perf_counter() - t0) / steps * 1e3


if __name__ == "__main__":
    wn = dict(WN18RR, name="wn18rr_shaped"); fb = dict(FB15K237, name="fb15k237_shaped")
    for model, spec, nb in (("TransH", wn, 3), ("TransD", fb, 8)):
        for n in (1, 2, 3, 4, 6):
            row = dict(model=model, graph=spec["name"], n=n)
            for label, mn in (("float_records", 64), ("pair_counts", 1)):
                with contextlib.redirect_stdout(io.StringIO()):
                    b, ms = run(model, spec, n, nb, mn)
                row["batch"] = b; row[label + "_ms"] = round(ms, 4)
            print(json.dumps(row), flush=True)
