#!/usr/bin/env python3
"""One configuration of the projecting models for rocprofv3 (kernel trace or one --pmc pass): python tools/run_hd_once.py TransH wn 25 2 [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237, WN18RR
model, graph, n, nb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
spec = dict(WN18RR, name="wn18rr_shaped") if graph == "wn" else dict(FB15K237, name="fb15k237_shaped")
d = make_dataset("/tmp/okes_%s" % spec["name"], spec)
con = pkg.Config()
con.set_in_path(d); con.set_work_threads(8); con.set_bern(1); con.set_dimension(200); con.set_nbatches(nb)
con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method("SGD"); con.init()
con.set_model_and_session(getattr(pkg, model))
for _ in range(steps):
    con.train_step(sync=False)
torch.cuda.synchronize()
print("done", con.batch_size)
