#!/usr/bin/env python3
"""One training configuration for rocprofv3: python tools/run_one_config.py MODEL GRAPH(fb|wn) DIM NEG NBATCHES [steps] [opt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237, WN18RR
model, graph, dim, n, nb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
opt = sys.argv[7] if len(sys.argv) > 7 else "SGD"
spec = dict(WN18RR, name="wn18rr_shaped") if graph == "wn" else dict(FB15K237, name="fb15k237_shaped")
d = make_dataset("/tmp/okes_%s" % spec["name"], spec)
con = pkg.Config()
con.set_in_path(d); con.set_work_threads(8); con.set_bern(0); con.set_dimension(dim); con.set_nbatches(nb)
con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method(opt); con.init()
con.set_model_and_session(getattr(pkg, model))
for _ in range(steps):
    con.train_step(sync=False)
torch.cuda.synchronize()
print("done", con.batch_size)
