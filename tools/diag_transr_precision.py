#!/usr/bin/env python3
"""TransR gradient error against an fp64 autograd reference (tests/torch_ref.py) for the two tilings of the row GEMMs:
the fp32 MFMA (transr_bf16x3 = 0) and the three-term bf16 split on the bf16 matrix pipe (transr_bf16x3 = 1), and for the
CPU oracle (fp32).  The same batch and parameters for all; rows that a sign flip of d|e|/de can change are left out
(near_kink_rows), hinges near their switch point are avoided by the batch choice.  Prints one JSON line.
usage: python3 tools/diag_transr_precision.py [E R D B]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import oracle
import torch_ref
from test_gpu_models import make_engine, rand_batch, batch_without_ties
from openkeonspark_amd import _lib


def main():
    E, R, D, B = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (3000, 24, 200, 4096)
    n = 1
    rng = np.random.default_rng(11)
    params = oracle.init_params(oracle.TRANSR, E, R, D, D, seed=4)
    for k in params:
        params[k] = (params[k] * 3).astype(np.float32)
    orc = oracle.Model("transr", E, R, D, D, margin=0.9, params=params)
    bh, bt, br = batch_without_ties(orc, lambda: rand_batch(rng, E, R, B, n, 0, 0.0, distinct=True), B, n)
    _, g64 = torch_ref.loss_and_grads("transr", params, bh, bt, br, B, n, 0.9, D, D)
    kink, n_near = torch_ref.near_kink_rows("transr", params, bh, bt, br, B, n, D, D, tol=1e-6)
    _, g_o = orc.grad(bh, bt, br, B, n)
    L = _lib.lib()
    out = {"E": E, "R": R, "D": D, "B": B, "near_kink_elements": int(n_near)}
    dev = torch.from_numpy(np.stack([bh, bt, br]).astype(np.int32)).cuda()

    def err(g, name):
        res = {}
        for k in ("ent_embeddings", "transfer_matrix", "rel_embeddings"):
            ref = g64[k]
            got = np.asarray(g[k], np.float64).reshape(ref.shape)
            keep = np.ones(ref.shape[0], bool)
            for row in kink.get(k, ()):
                keep[row] = False
            d = np.abs(got[keep] - ref[keep])
            scale = np.abs(ref[keep]).max()
            res[k] = {"max_abs_over_max": float(d.max() / scale), "rms_over_rms": float(np.sqrt((d ** 2).mean()) / np.sqrt((ref[keep] ** 2).mean())),
                      "rows": int(keep.sum())}
        out[name] = res

    err(g_o, "oracle_fp32_cpu")
    for tag, val in (("engine_fp32_mfma", 0), ("engine_bf16x3", 1)):
        L.kge_set_option(b"transr_bf16x3", val)
        con = make_engine("transr", E, R, D, n, 0, margin=0.9, params=params, Dr=D)
        con.forward_backward(dev, B, B, B * n)
        torch.cuda.synchronize()
        err(con.get_gradients(), tag)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
