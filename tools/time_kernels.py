#!/usr/bin/env python3
"""Interleaved timing of tuning variants of the TransE sign-count kernels on the bench workload (one process,
HIP events on the launch stream; cdna_hip_programming.md rule 24)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237
con = pkg.Config()
con.set_in_path(make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)); con.set_work_threads(8); con.set_bern(1)
con.set_dimension(200); con.set_nbatches(8); con.set_ent_neg_rate(25); con.set_alpha(0.001); con.set_opt_method("Adam")
con.init(); con.set_model_and_session(pkg.TransE)
for _ in range(5): con.train_step(sync=False)
dev, n_pos = con.sample_device()
B = con.batch_size
def timeit(fn, reps=10):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t)//2] * 1e3, t[0] * 1e3
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]
for rnd in range(3):
    for v in variants:
        con.lib.kge_set_option(b"emit_variant", v)
        med, mn = timeit(lambda: (con.forward_counts(dev, n_pos, n_pos, B * 25), con._counts.zero_()))
        print("round %d variant %d: forward_counts median %.1f us  min %.1f us" % (rnd, v, med, mn), flush=True)
con.lib.kge_set_option(b"emit_variant", 0)
