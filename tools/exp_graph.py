#!/usr/bin/env python3
"""Experiment: one SGD training step captured in a hipGraph (torch.cuda.CUDAGraph) versus eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237, WN18RR

def run(model, dim, n, nb, spec):
    con = pkg.Config()
    con.set_in_path(make_dataset("/tmp/okes_%s" % spec["name"], spec)); con.set_work_threads(8); con.set_bern(1)
    con.set_dimension(dim); con.set_nbatches(nb); con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method("SGD")
    con.init()
    con.set_model_and_session(getattr(pkg, model))
    for _ in range(20):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 500 * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 500 * 1e6
    print(model, dim, n, "B=%d" % con.batch_size, "eager %.1f us  graph %.1f us  loss %.4f" % (eager, graph, float(con._loss.item())), flush=True)

if __name__ == "__main__":
    fb = dict(FB15K237, name="fb15k237_shaped"); wn = dict(WN18RR, name="wn18rr_shaped")
    import contextlib, io
    for args in (("TransE", 100, 1, 0, fb), ("TransH", 200, 1, 0, wn), ("TransE", 200, 25, 8, fb), ("TransR", 200, 1, 0, fb)):
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                run(*args)
            sys.stderr.write(buf.getvalue().strip().splitlines()[-1] + "\n")
        except Exception as ex:
            sys.stderr.write("%s failed: %r\n" % (args[0], ex))
