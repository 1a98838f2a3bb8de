#!/usr/bin/env python3
"""Link-prediction ranker throughput (MR / Hits@10 half of the BASELINE metric): FB15k-237-shaped synthetic KG with
its public test / valid cardinalities (20 466 / 17 535), every test triple ranked against all 14 541 entities on both
sides, raw + filtered.  usage: bench_lp.py [MODEL] [DIM]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "TransE"
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    import torch
    import openkeonspark_amd as pkg
    from openkeonspark_amd.synthetic import FB15K237, generate_triples, write_openke_dir
    d = "/tmp/okes_fb15k237_lp/"
    if not os.path.exists(d + ".complete"):
        spec = FB15K237
        n_test, n_valid = 20466, 17535
        h, t, r = generate_triples(spec["entities"], spec["relations"], spec["train"] + n_test + n_valid, spec["seed"])
        n = spec["train"]
        write_openke_dir(d, spec["entities"], spec["relations"], h[:n], t[:n], r[:n])
        for name, lo, hi in (("test2id.txt", n, n + n_test), ("valid2id.txt", n + n_test, n + n_test + n_valid)):
            with open(d + name, "w") as f:
                f.write("%d\n" % (hi - lo))
                np.savetxt(f, np.stack([h[lo:hi], t[lo:hi], r[lo:hi]], axis=1), fmt="%d")
        open(d + ".complete", "w").write("ok\n")
    con = pkg.Config()
    con.set_in_path(d); con.set_work_threads(8); con.set_bern(1); con.set_dimension(dim); con.set_nbatches(8)
    con.set_ent_neg_rate(25); con.set_alpha(0.001); con.set_opt_method("Adam")
    con.init()
    con.init_link_prediction()
    con.set_model_and_session(getattr(pkg, model))
    for _ in range(30):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    con.link_prediction(0, 256)          # warm-up (filter index upload)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, met = con.link_prediction()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = out.shape[0]
    print(json.dumps({"model": model, "dim": dim, "test_triples": n, "entities": con.entTotal, "seconds": round(dt, 4),
                      "test_triples_per_s": n / dt, "candidate_scores_per_s": 2.0 * n * con.entTotal / dt,
                      "MR_filter_tail": met["r_filter_rank"], "Hits10_filter_tail": met["r_filter_tot"]}))


if __name__ == "__main__":
    main()
