"""Diagnostic: fused vs two-call TransE count step on the small golden graph: which rows differ after ONE step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fused_counts import engine, state
from oracle import oracle
path = os.path.join(ROOT, "tests", "golden", "kg_small")
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 200
opt = sys.argv[2] if len(sys.argv) > 2 else "Adam"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
res = {}
s0 = None
for tag in "ab":
    con = engine(path, dim, 10, 5, opt, fused=(tag == "a"), streams=s0)
    if s0 is None:
        s0 = con.get_stream_states()
    kg = oracle.KG(path, work_threads=8, bern=1)
    kg.set_stream_states(s0)
    bh, bt, br, _ = kg.sampling(600, 5, 0)
    losses = [con.train_step() for _ in range(steps)]
    print(tag, losses)
    res[tag] = state(con)
E = 1000
orc = oracle.Model("transe", 1000, 20, dim, dim, margin=1.0, seed=0)
hm = orc.hinge_margins(bh, bt, br, 600, 5)
act = hm >= 0
cnt = np.zeros(E, int)
for b in range(600):
    if act[b].any():
        cnt[bh[b]] += 1; cnt[bt[b]] += 1
    for k in range(5):
        if act[b, k]:
            j = 600 * (k + 1) + b
            x = bh[j] if bh[j] != bh[b] else bt[j]
            cnt[x] += 1
for k in res["a"]:
    d = np.abs(res["a"][k].astype(np.float64) - res["b"][k])
    rows = np.nonzero(d.reshape(d.shape[0], -1).max(1) > 0)[0]
    print(k, "max diff", d.max(), "rows differing", len(rows), "of", d.shape[0])
    if "ent" in k and len(rows):
        print("   record counts of differing rows (first 20):", cnt[rows[:20]].tolist(), " elements differing in first row:", int((d[rows[0]] > 0).sum()),
              " rel diff:", float((d[rows[0]] / (np.abs(res['b'][k][rows[0]]) + 1e-30)).max()))
        print("   count histogram all rows:", np.bincount(np.minimum(cnt, 30))[:31].tolist())
