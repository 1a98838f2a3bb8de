#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not bench lines):
#1 FB15k-237 TransE D=100 SGD n=1 (the reference's auto batch 2721), #3 WN18RR-shaped TransH D=200,
#4 FB15k-237 TransR 200x200, plus TransD.  Prints one JSON line per configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, spec, model, dim, n, opt, nbatches, steps=200, warmup=20, bern=0, persistent=False):
    import torch
    import openkeonspark_amd as pkg
    from openkeonspark_amd.synthetic import make_dataset
    d = make_dataset("/tmp/okes_%s" % spec["name"], spec)
    con = pkg.Config()
    con.set_in_path(d); con.set_work_threads(8); con.set_bern(bern); con.set_dimension(dim)
    con.set_nbatches(nbatches); con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method(opt)
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    if persistent:     # all the timed steps inside ONE persistent launch (csrc/persist.hip)
        con.prefetch_sampling = False
        con.train_steps(warmup, persistent=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        con.train_steps(steps, persistent=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    else:
        for _ in range(warmup):
            con.train_step(sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            con.train_step(sync=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out = {"config": name, "model": model, "dim": dim, "neg": n, "opt": opt, "batch": con.batch_size,
           "positives_per_s": con.batch_size * steps / dt, "ms_per_step": dt / steps * 1e3,
           "loss": float(con._loss.item())}
    if model == "TransR":
        flops = 12.0 * dim * dim * (1 + n) * con.batch_size
        out["mfma_tflops"] = flops / (dt / steps) / 1e12
        out["mfma_frac_of_157TF"] = out["mfma_tflops"] / 157.3
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    from openkeonspark_amd.synthetic import FB15K237, WN18RR
    fb = dict(FB15K237, name="fb15k237_shaped")
    wn = dict(WN18RR, name="wn18rr_shaped")
    sys.stdout.flush()
    devnull = os.open(os.devnull, os.O_WRONLY)
    real = os.dup(1)
    def quiet(f, *a, **k):
        os.dup2(devnull, 1)
        try:
            import io, contextlib
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                f(*a, **k)
            res = buf.getvalue().strip().splitlines()[-1]
        finally:
            import ctypes
            ctypes.CDLL(None).fflush(None)
            os.dup2(real, 1)
        print(res, flush=True)
    quiet(run, "#1 FB15k-237 TransE D=100 SGD n=1 auto-batch", fb, "TransE", 100, 1, "SGD", 0)
    quiet(run, "#1 same, 1000 steps in one persistent launch", fb, "TransE", 100, 1, "SGD", 0, steps=1000, persistent=True)
    quiet(run, "#1b same, nbatches=4 (B=68028)", fb, "TransE", 100, 1, "SGD", 4)
    quiet(run, "#3 WN18RR TransH D=200 n=1 auto-batch", wn, "TransH", 200, 1, "SGD", 0)
    quiet(run, "#3 same, 1000 steps in one persistent launch", wn, "TransH", 200, 1, "SGD", 0, steps=1000, persistent=True)
    quiet(run, "#3b WN18RR TransH D=200 n=25 nbatches=2", wn, "TransH", 200, 25, "SGD", 2)
    quiet(run, "#4 FB15k-237 TransR 200x200 n=1 auto-batch", fb, "TransR", 200, 1, "SGD", 0)
    quiet(run, "#4b FB15k-237 TransR 200x200 n=1 nbatches=8", fb, "TransR", 200, 1, "SGD", 8, steps=50)
    quiet(run, "TransD D=200 n=25 nbatches=8", fb, "TransD", 200, 25, "SGD", 8)
