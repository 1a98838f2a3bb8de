#!/usr/bin/env python3
"""Is the headline step bound by the host's launch rate?  Time to ENQUEUE n steps vs time until they have run."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237
d = make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)
out = {}
for prefetch in (False, True):
    con = pkg.Config()
    con.prefetch_sampling = prefetch
    con.set_in_path(d); con.set_work_threads(8); con.set_bern(1); con.set_dimension(200); con.set_nbatches(8)
    con.set_ent_neg_rate(25); con.set_alpha(0.001); con.set_opt_method("Adam"); con.init()
    con.set_model_and_session(pkg.TransE)
    for _ in range(30):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        con.train_step(sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out["prefetch_%s" % prefetch] = dict(enqueue_us_per_step=(t1 - t0) / n * 1e6, total_us_per_step=(t2 - t0) / n * 1e6)
print(json.dumps(out))
