"""Config #1 (B = 2721): is the separate-launch step host-bound?  Host enqueue vs wall time per step, sampler riding vs in line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237
fb = make_dataset("/tmp/okes_fb15k237_shaped", dict(FB15K237, name="fb15k237_shaped"))
for prefetch in (False, True):
    for clone in (False, True):
        con = pkg.Config()
        con.set_in_path(fb); con.set_work_threads(8); con.set_bern(0); con.set_dimension(100); con.set_nbatches(0)
        con.set_ent_neg_rate(1); con.set_alpha(0.01); con.set_opt_method("SGD")
        con.prefetch_sampling = prefetch
        con.init(); con.set_model_and_session(pkg.TransE)
        for _ in range(200):
            con.train_step(sync=False)
        torch.cuda.synchronize()
        N = 2000
        t0 = time.perf_counter()
        for _ in range(N):
            l = con.train_step(sync=False)
            if clone:
                l = l.clone()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("prefetch(rider) %-5s clone %-5s: host enqueue %.1f us/step, wall %.1f us/step" % (prefetch, clone, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(1000):
    con.train_step(sync=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
