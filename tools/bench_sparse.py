"""Rehearsal of BASELINE config #5 (synthetic KG, 50M entities, TransE dim 512, sparse touched-row exchange) on
ONE GPU: the full-size entity table (102 GB) lives in HBM, the triple count is scaled to what the host can index
in the time a gpurun call allows.  Prints one JSON line; `--profile-steps` keeps the run short under rocprofv3."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=50_000_000)
    ap.add_argument("--relations", type=int, default=1000)
    ap.add_argument("--triples", type=int, default=20_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--batch", type=int, default=200_000)
    ap.add_argument("--neg", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--ent-exponent", type=float, default=0.8, help="Zipf exponent of entity popularity (0 = uniform)")
    ap.add_argument("--dense", action="store_true", help="dense count image instead of the sparse-row path")
    ap.add_argument("--force-dp", action="store_true", help="one-rank RCCL group + Config.force_data_parallel: the table-sharded multi-GPU step, every exchange a copy")
    ap.add_argument("--model", default="TransE", help="TransE | TransH | TransD (the latter two: --dense = gradient tables + sweep, else row-wise SGD in place)")
    ap.add_argument("--opt", default="SGD", help="SGD | Adam (TF1 dense sweep, parity; implies --dense) | LazyAdam (touched rows only, NON-PARITY)")
    a = ap.parse_args()
    import numpy as np
    import torch
    import openkeonspark_amd as ok
    from openkeonspark_amd.synthetic import generate_triples
    if a.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t0 = time.time()
    if a.ent_exponent == 0.0:   # uniform popularity: plain draws (the Zipf inverse-CDF search takes minutes at 500 M triples)
        rng = np.random.default_rng(5)
        h = rng.integers(0, a.entities, a.triples, dtype=np.int64)
        t = rng.integers(0, a.entities, a.triples, dtype=np.int64)
        r = rng.integers(0, a.relations, a.triples, dtype=np.int64)
    else:
        h, t, r = generate_triples(a.entities, a.relations, a.triples, seed=5, dup_frac=0.0, ent_exponent=a.ent_exponent)
    t_gen = time.time() - t0
    con = ok.Config()
    con.set_work_threads(a.threads)
    con.set_bern(1)
    con.set_dimension(a.dim)
    con.set_ent_neg_rate(a.neg)
    con.set_rel_neg_rate(0)
    con.set_alpha(0.01)
    con.set_margin(1.0)
    con.set_opt_method(a.opt)
    con.set_nbatches(max(1, a.triples // a.batch))
    con.sparse_rows = False if a.opt == "Adam" else not a.dense
    t0 = time.time()
    con.init_from_arrays(a.entities, a.relations, h, t, r)
    t_index = time.time() - t0
    del h, t, r
    t0 = time.time()
    con.set_model_and_session(getattr(ok, a.model))
    if a.force_dp:
        con.force_data_parallel = True
        con.init_distributed()
    torch.cuda.synchronize()
    t_init = time.time() - t0
    for _ in range(a.warmup):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        loss = con.train_step(sync=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    B = con.batch_size
    print(json.dumps({"workload": "synthetic KG %dM entities / %dM triples %s dim=%d %s %d neg/pos, %s" % (
        a.entities // 1_000_000, a.triples // 1_000_000, a.model, a.dim, a.opt, a.neg,
        ("dense image / gradient tables + sweep" if (a.dense or a.opt == "Adam") else "sparse rows") + (", table-sharded step on a one-rank RCCL group" if a.force_dp else "")),
        "ent_exponent": a.ent_exponent, "batch": B, "ms_per_step": 1e3 * dt / a.steps, "positives_per_s": B * a.steps / dt, "loss": float(loss.item()),
        "hbm_allocated_GB": torch.cuda.max_memory_allocated() / 1e9,
        "seconds": {"generate": round(t_gen, 1), "index": round(t_index, 1), "table_init": round(t_init, 1)}}))


if __name__ == "__main__":
    main()
