#!/usr/bin/env python3
"""Run N training steps of one configuration (for profiling):  run_one.py MODEL DIM NEG OPT NBATCHES STEPS [wn]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237, WN18RR
model, dim, neg, opt, nb, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
spec = dict(WN18RR, name="wn18rr_shaped") if len(sys.argv) > 7 and sys.argv[7] == "wn" else dict(FB15K237, name="fb15k237_shaped")
con = pkg.Config()
for key, val in os.environ.items():   # KGE_OPT_<engine option>=<int>
    if key.startswith("KGE_OPT_"):
        con.lib.kge_set_option(key[8:].lower().encode(), int(val))
if os.environ.get("KGE_PREFETCH") is not None:    # 0: the sampler as its own kernel in front of every step (for profiling it)
    con.prefetch_sampling = os.environ["KGE_PREFETCH"] != "0"
con.set_in_path(make_dataset("/tmp/okes_%s" % spec["name"], spec)); con.set_work_threads(8); con.set_bern(1)
con.set_dimension(dim); con.set_nbatches(nb); con.set_ent_neg_rate(neg); con.set_alpha(0.001); con.set_opt_method(opt)
con.init()
con.set_model_and_session(getattr(pkg, model))
for _ in range(steps):
    con.train_step(sync=False)
torch.cuda.synchronize()
print("done", float(con._loss.item()))
