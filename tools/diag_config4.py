"""Diagnostic: where do the TransR gradients of the engine and of the oracle differ on the FB15k-237-shaped graph?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import oracle
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
v1 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fb = make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)
con = pkg.Config(); con.prefetch_sampling = False
con.set_in_path(fb); con.set_work_threads(8); con.set_bern(0); con.set_dimension(200); con.set_nbatches(nb)
con.set_ent_neg_rate(1); con.set_alpha(0.01); con.set_opt_method("SGD"); con.init()
con.lib.kge_set_option(b"transr_v1", v1)
con.set_model_and_session(pkg.TransR)
B = con.batch_size
kg = oracle.KG(fb, work_threads=8, bern=0); kg.set_stream_states(con.get_stream_states())
orc = oracle.Model("transr", con.entTotal, con.relTotal, 200, 200, margin=1.0, params=con.get_parameters())
dev, n_pos = con.sample_device()
bh, bt, br, _ = kg.sampling(B, 1, 0)
loss_o, g_o = orc.grad(bh, bt, br, B, 1)
con.forward_backward(dev, B, B, B); torch.cuda.synchronize()
g_g = con.get_gradients()
print("loss", float(con._loss.item()), loss_o)
hm = orc.hinge_margins(bh, bt, br, B, 1)
print("min |hinge margin|", np.abs(hm).min(), "active", (hm >= 0).sum())
for k in g_o:
    scale = np.abs(g_o[k]).max()
    diff = np.abs(g_g[k].astype(np.float64) - g_o[k])
    bad = np.nonzero((diff > 1e-5 * scale).any(1))[0]
    print(k, "scale", scale, "bad rows", len(bad), "max diff/scale", diff.max() / scale)
    for r in bad[:8]:
        d = diff[r]; e = np.argmax(d)
        print("   row", r, "n_bad_elems", int((d > 1e-5 * scale).sum()), "worst elem", e, "got", g_g[k][r, e], "want", g_o[k][r, e],
              "diff/unit", d[e] * B, "row max", np.abs(g_o[k][r]).max(), "uses in batch", int(((bh == r) | (bt == r)).sum()) if k == "ent_embeddings" else int((br == r).sum()))
