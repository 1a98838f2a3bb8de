#!/usr/bin/env python3
"""Where does a step of the persistent launch spend its time?  Workgroup 0's phase-boundary stamps (option
"persist_trace", 100 MHz clock): sweep, sampling, barrier 1, forward/backward, barrier 2 -- medians over the traced steps."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, spec, model, dim, n, threads, steps=200, touch=0, ahead=1):
    import torch
    import openkeonspark_amd as pkg
    from openkeonspark_amd.synthetic import make_dataset
    d = make_dataset("/tmp/okes_%s" % spec["name"], spec)
    con = pkg.Config(); con.prefetch_sampling = False
    con.set_in_path(d); con.set_work_threads(8); con.set_bern(0); con.set_dimension(dim); con.set_nbatches(0)
    con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method("SGD"); con.init()
    con.set_model_and_session(getattr(pkg, model))
    L = con.lib
    L.kge_set_option(b"persist_threads", threads)
    L.kge_set_option(b"persist_touch", touch)
    L.kge_set_option(b"persist_ahead", ahead)
    con.train_steps(20, persistent=True)
    L.kge_set_option(b"persist_trace", 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    con.train_steps(steps, persistent=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    L.kge_set_option(b"persist_trace", 0)
    raw = np.zeros((256 + 64, 6), np.uint64)
    pkg._lib.check(L.kge_persistent_trace(raw.ctypes.data, 256 + 64), L)
    tr = raw[:steps].astype(np.int64)
    per_block = raw[256:].reshape(-1)[:256].astype(np.int64) / 100.0
    us = lambda a: float(np.median(a)) / 100.0
    body = tr[2:steps - 1]
    nxt = tr[3:steps, 0]
    U = {"TransE": 3 + n, "TransH": 4 + n, "TransD": 6 + 2 * n}[model]
    alg = (2 * U * dim * 4 + 12 * (1 + n)) * con.batch_size      # every touched row read and written once + the ids
    out = {"config": name, "step_algorithmic_bytes": alg, "step_algorithmic_GBps": alg / (dt / steps) / 1e9,
           "step_frac_of_8TBps": alg / (dt / steps) / 8e12, "threads": threads, "touch": touch, "ahead": ahead, "batch": con.batch_size, "us_per_step_wall": dt / steps * 1e6,
           "sweep": us(body[:, 1] - body[:, 0]), "sampling": us(body[:, 2] - body[:, 1]), "barrier1": us(body[:, 3] - body[:, 2]),
           "fwdbwd": us(body[:, 4] - body[:, 3]), "barrier2": us(body[:, 5] - body[:, 4]), "step_by_stamps": us(nxt - body[:, 0]),
           "fwdbwd_per_block_us": {"min": float(per_block.min()), "p50": float(np.median(per_block)), "p90": float(np.percentile(per_block, 90)),
                                   "max": float(per_block.max())}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    from openkeonspark_amd.synthetic import FB15K237, WN18RR
    fb = dict(FB15K237, name="fb15k237_shaped"); wn = dict(WN18RR, name="wn18rr_shaped")
    run("#1 FB15k-237 TransE D=100 n=1 B=2721", fb, "TransE", 100, 1, 512)
    run("#3 WN18RR TransH D=200 n=1 B=8683", wn, "TransH", 200, 1, 512)
