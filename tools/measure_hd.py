#!/usr/bin/env python3
"""TransH / TransD step times at n = 25 and n = 1 (the projecting models' vector kernel), with the kernel's two builds."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, spec, model, dim, n, nbatches, occ4, steps=100, warmup=10):
    import torch
    import openkeonspark_amd as pkg
    from openkeonspark_amd.synthetic import make_dataset
    d = make_dataset("/tmp/okes_%s" % spec["name"], spec)
    pkg._lib.lib().kge_set_option(b"fb_occ4", occ4)
    con = pkg.Config()
    con.set_in_path(d); con.set_work_threads(8); con.set_bern(1); con.set_dimension(dim)
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
    dt = (time.perf_counter() - t0) / steps
    U = {"TransH": 4 + n, "TransD": 6 + 2 * n}[model]
    alg = 2 * U * dim * 4 * con.batch_size        # read + write each touched row once
    print(json.dumps({"config": name, "occ4": occ4, "batch": con.batch_size, "ms_per_step": dt * 1e3, "positives_per_s": con.batch_size / dt,
                      "algorithmic_GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12}), flush=True)


if __name__ == "__main__":
    from openkeonspark_amd.synthetic import FB15K237, WN18RR
    fb = dict(FB15K237, name="fb15k237_shaped"); wn = dict(WN18RR, name="wn18rr_shaped")
    import contextlib, io
    for occ4 in (1, 0):
        for args in (("WN18RR TransH D=200 n=25 B=43417", wn, "TransH", 200, 25, 2), ("FB15k-237 TransD D=200 n=25 B=34014", fb, "TransD", 200, 25, 8),
                     ("WN18RR TransH D=200 n=1 B=28945", wn, "TransH", 200, 1, 3), ("FB15k-237 TransH D=200 n=25 B=34014", fb, "TransH", 200, 25, 8)):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                run(*args, occ4)
            print(buf.getvalue().strip().splitlines()[-1], flush=True)
