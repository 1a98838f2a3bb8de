#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE'S OWN sampler.

Run in the build container only (needs /root/reference):

    make -C oracle            # compiles /root/reference/base/Base.cpp -> oracle/_ref/Base.so
    python tests/golden/make_golden.py

What is committed is data only: three tiny knowledge graphs in OpenKE text format (made up here,
not taken from the reference, which ships none) and, for each, the outputs of the reference's
`sampling` / getters for a grid of settings (tests/golden/*.npz), plus SHA-256 digests of long
sampling streams on the FB15k-237-shaped synthetic graph (tests/golden/fb_digests.json).
Base.so keeps one dataset per process, so every (graph, workThreads, bern) cell runs in a child.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

GRID_W = [1, 2, 3, 8]
GRID_BERN = [0, 1]
GRID_SHAPE = [(7, 1, 0), (64, 2, 1), (64, 25, 0), (50, 3, 0)]  # (B, negRate, negRelRate)
CALLS = 3


def make_tiny_graphs():
    from openkeonspark_amd.synthetic import write_openke_dir
    rng = np.random.default_rng(20261003)
    # kg_tiny: hub entity 0 whose (0, rel 0) group covers most entities; relation 5 never used;
    # explicit duplicate lines
    E, R = 30, 6
    h = list(rng.integers(0, E, 90)); t = list(rng.integers(0, E, 90)); r = list(rng.integers(0, 5, 90))
    for tail in range(2, 27):
        h.append(0); t.append(tail); r.append(0)
    for head in range(5, 20):
        h.append(head); t.append(1); r.append(2)
    for i in range(0, 24, 2):  # duplicates
        h.append(h[i]); t.append(t[i]); r.append(r[i])
    write_openke_dir(os.path.join(HERE, "kg_tiny"), E, R, np.array(h), np.array(t), np.array(r))
    # kg_small: 1000 entities, skewed
    from openkeonspark_amd.synthetic import generate_triples
    hs, ts, rs = generate_triples(1000, 20, 6000, seed=7, dup_frac=0.01)
    write_openke_dir(os.path.join(HERE, "kg_small"), 1000, 20, hs, ts, rs)
    # kg_incr: incremental mode, the last 37 lines are the "new batch" (Reader.h:61-67, Base.cpp:101-103)
    hi, ti, ri = generate_triples(60, 4, 260, seed=11, dup_frac=0.02)
    write_openke_dir(os.path.join(HERE, "kg_incr"), 60, 4, hi, ti, ri, new_batch_total=37)


def write_eval_files(kg, E, R, seed):
    """test2id / valid2id / type_constrain / ontology_constrain in the formats Reader.h:186-449 reads
    (made up here; the reference ships no data)."""
    rng = np.random.default_rng(seed)
    d = os.path.join(HERE, kg)
    train = np.loadtxt(os.path.join(d, "train2id.txt"), skiprows=1, dtype=np.int64)
    def some(n):
        rows = train[rng.integers(0, len(train), n)].copy()           # start from known triples ...
        flip = rng.random(n) < 0.7
        rows[flip, 1] = rng.integers(0, E, int(flip.sum()))          # ... and corrupt most tails
        return rows
    test, valid = some(40), some(30)
    for name, arr in (("test2id.txt", test), ("valid2id.txt", valid)):
        with open(os.path.join(d, name), "w") as f:
            f.write("%d\n" % len(arr))
            np.savetxt(f, arr, fmt="%d")
    allt = np.concatenate([train, test, valid])
    with open(os.path.join(d, "type_constrain.txt"), "w") as f:
        f.write("%d\n" % R)
        for r in range(R):
            m = allt[:, 2] == r
            heads = sorted(set(allt[m, 0].tolist()) | set(rng.integers(0, E, 3).tolist()))
            tails = sorted(set(allt[m, 1].tolist()) | set(rng.integers(0, E, 3).tolist()))
            rng.shuffle(heads); rng.shuffle(tails)                    # the loader sorts them itself
            f.write("%d\t%d%s\n" % (r, len(heads), "".join("\t%d" % x for x in heads)))
            f.write("%d\t%d%s\n" % (r, len(tails), "".join("\t%d" % x for x in tails)))
    ents = sorted(rng.choice(E, size=min(E, 25), replace=False).tolist())
    with open(os.path.join(d, "ontology_constrain.txt"), "w") as f:
        f.write("%d\n" % len(ents))
        for e in ents:
            sup = rng.choice(E, size=int(rng.integers(0, 6)), replace=False).tolist()
            sub = rng.choice(E, size=int(rng.integers(0, 6)), replace=False).tolist()
            f.write("%d\t%d%s\n" % (e, len(sup), "".join("\t%d" % x for x in sup)))
            f.write("%d\t%d%s\n" % (e, len(sub), "".join("\t%d" % x for x in sub)))


def lp_worker(kg_dir, out_path, n_triples, seed):
    """Reference testHead / testTail on seeded score vectors (quantised so that ties occur)."""
    from oracle.oracle import ReferenceSampler
    ref = ReferenceSampler(kg_dir, work_threads=1, bern=0)
    ref.init_link_prediction()
    rng = np.random.default_rng(seed)
    E = ref.entTotal
    n = min(n_triples, ref.testTotal)
    scores = np.round(rng.standard_normal((n, 2, E)).astype(np.float32) * 2, 1)
    out = np.zeros((n, 2, 8), np.int64)
    triples = np.zeros((n, 3), np.int64)
    for i in range(n):
        hb = ref.batch(i, head=False)
        triples[i] = (hb[0][0], None or 0, hb[2][0])
        tb = ref.batch(i, head=True)
        triples[i] = (hb[0][0], tb[1][0], hb[2][0])                  # (h, t, r) of the i-th sorted test triple
        # make the target's own score mid-range so that some candidates beat it
        out[i, 0] = ref.rank(i, scores[i, 0], head=True)
        out[i, 1] = ref.rank(i, scores[i, 1], head=False)
    np.savez_compressed(out_path, totals=np.array([ref.testTotal, ref.validTotal, ref.tripleTotal], np.int64),
                        triples=triples, scores=scores, out=out)


def tc_worker(kg_dir, out_path, seed):
    """Reference triple classification (Test.h:252-444) in a fresh process: type-constrained negatives
    drawn with libc rand(), thresholds, accuracy, ROC counts, on seeded score arrays."""
    import ctypes
    from oracle.oracle import ReferenceSampler, _p
    ref = ReferenceSampler(kg_dir, work_threads=1, bern=0)     # randReset consumed ONE libc draw
    ref.init_link_prediction()
    L = ref.L
    vp = ctypes.c_void_p
    L.getValidBatch.argtypes = [vp] * 6; L.getTestBatch.argtypes = [vp] * 6
    L.getBestThreshold.argtypes = [vp] * 3
    L.test_triple_classification.argtypes = [vp] * 4
    L.get_n_interval.argtypes = [ctypes.c_int64, vp, vp]; L.get_n_interval.restype = ctypes.c_int64
    L.get_TPFP.argtypes = [ctypes.c_int64, vp, vp, vp, vp]; L.get_TPFP.restype = ctypes.POINTER(ctypes.c_int64)
    V, T, R = ref.validTotal, ref.testTotal, ref.relTotal
    valid = [np.zeros(V, np.int64) for _ in range(6)]
    test = [np.zeros(T, np.int64) for _ in range(6)]
    L.getValidBatch(*[_p(a) for a in valid])
    L.getTestBatch(*[_p(a) for a in test])
    rng = np.random.default_rng(seed)
    vpos = np.round(rng.normal(2.0, 0.7, V), 3).astype(np.float32); vneg = np.round(rng.normal(3.0, 0.7, V), 3).astype(np.float32)
    tpos = np.round(rng.normal(2.0, 0.7, T), 3).astype(np.float32); tneg = np.round(rng.normal(3.0, 0.7, T), 3).astype(np.float32)
    thresh = np.full(R, -1.0, np.float32)
    L.getBestThreshold(_p(thresh), _p(vpos), _p(vneg))
    acc = np.zeros(1, np.float32)
    L.test_triple_classification(_p(thresh), _p(tpos), _p(tneg), _p(acc))
    n_int = np.array([L.get_n_interval(r, _p(vpos), _p(vneg)) for r in range(R)], np.int64)
    tpfp = {}
    for r in range(R):
        ptr = L.get_TPFP(r, _p(vpos), _p(vneg), _p(tpos), _p(tneg))
        if ptr:
            tpfp["tpfp_%d" % r] = np.array([ptr[i] for i in range(int(n_int[r] + 1) * 2)], np.int64)
    np.savez_compressed(out_path, valid=np.stack(valid), test=np.stack(test), vpos=vpos, vneg=vneg, tpos=tpos, tneg=tneg,
                        thresh=thresh, acc=acc, n_interval=n_int, **tpfp)


def worker(kg_dir, W, bern, out_path, shapes, calls):
    from oracle.oracle import ReferenceSampler
    ref = ReferenceSampler(kg_dir, work_threads=W, bern=bern)
    out = {
        "totals": np.array([ref.entTotal, ref.relTotal, ref.trainTotal, ref.trainTotal_, ref.batchTotal], np.int64),
        "seeds": ref.stream_states(),
        "left_mean": ref.left_mean(), "right_mean": ref.right_mean(),
        "by_head": ref.sorted_copy("head").astype(np.int32),
        "by_tail": ref.sorted_copy("tail").astype(np.int32),
        "by_rel": ref.sorted_copy("rel").astype(np.int32),
    }
    # the calls are consecutive on ONE rng state, in grid order
    for si, (B, n, nr) in enumerate(shapes):
        for c in range(calls):
            h, t, r, y = ref.sampling(B, n, nr)
            out["s%d_c%d" % (si, c)] = np.stack([h, t, r]).astype(np.int32)
            out["y%d_c%d" % (si, c)] = y
    out["final_states"] = ref.stream_states()
    np.savez_compressed(out_path, **out)


def digest_worker(kg_dir, W, bern, B, n, nr, calls, out_path):
    from oracle.oracle import ReferenceSampler
    ref = ReferenceSampler(kg_dir, work_threads=W, bern=bern)
    hsh = hashlib.sha256()
    first = None
    for c in range(calls):
        h, t, r, y = ref.sampling(B, n, nr)
        hsh.update(h.tobytes()); hsh.update(t.tobytes()); hsh.update(r.tobytes())
        if c == 0:
            first = [h[:4].tolist(), t[:4].tolist(), r[:4].tolist(), h[B:B + 4].tolist(), t[B:B + 4].tolist()]
    json.dump({"sha256": hsh.hexdigest(), "first": first,
               "totals": [ref.entTotal, ref.relTotal, ref.trainTotal, ref.trainTotal_, ref.batchTotal],
               "final_states": [int(x) for x in ref.stream_states()]}, open(out_path, "w"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        a = json.loads(sys.argv[2])
        worker(a["kg"], a["W"], a["bern"], a["out"], a["shapes"], a["calls"])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--lp":
        a = json.loads(sys.argv[2])
        lp_worker(a["kg"], a["out"], a["n"], a["seed"])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--tc":
        a = json.loads(sys.argv[2])
        tc_worker(a["kg"], a["out"], a["seed"])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--digest":
        a = json.loads(sys.argv[2])
        digest_worker(a["kg"], a["W"], a["bern"], a["B"], a["n"], a["nr"], a["calls"], a["out"])
        return
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    make_tiny_graphs()
    for kg in ("kg_tiny", "kg_small", "kg_incr"):
        for W in GRID_W:
            for bern in GRID_BERN:
                out = os.path.join(HERE, "%s_W%d_bern%d.npz" % (kg, W, bern))
                arg = dict(kg=os.path.join(HERE, kg) + "/", W=W, bern=bern, out=out, shapes=GRID_SHAPE, calls=CALLS)
                subprocess.check_call([sys.executable, __file__, "--worker", json.dumps(arg)],
                                      stdout=subprocess.DEVNULL)
                print("wrote", os.path.relpath(out, ROOT))
    # link-prediction inputs + reference testHead/testTail outputs
    for kg, (E, R) in {"kg_tiny": (30, 6), "kg_small": (1000, 20)}.items():
        write_eval_files(kg, E, R, seed=len(kg))
        out = os.path.join(HERE, "lp_%s.npz" % kg)
        arg = dict(kg=os.path.join(HERE, kg) + "/", out=out, n=30, seed=5)
        subprocess.check_call([sys.executable, __file__, "--lp", json.dumps(arg)], stdout=subprocess.DEVNULL)
        print("wrote", os.path.relpath(out, ROOT))
        out = os.path.join(HERE, "tc_%s.npz" % kg)
        arg = dict(kg=os.path.join(HERE, kg) + "/", out=out, seed=9)
        subprocess.check_call([sys.executable, __file__, "--tc", json.dumps(arg)], stdout=subprocess.DEVNULL)
        print("wrote", os.path.relpath(out, ROOT))
    # FB15k-237-shaped synthetic graph: generated (not committed), digests committed
    from openkeonspark_amd.synthetic import make_dataset, FB15K237
    fb = make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)
    digests = {}
    for name, (W, bern, B, n, nr, calls) in {
        "cfg1_W1_uniform_B2721_n1": (1, 0, 2721, 1, 0, 100),
        "cfg1_W8_uniform_B2721_n1": (8, 0, 2721, 1, 0, 100),
        "cfg2_W8_bern_B2721_n25": (8, 1, 2721, 25, 0, 20),
        "W8_bern_B4096_n2_nr1": (8, 1, 4096, 2, 1, 20),
        "cfg2_W8_bern_B68028_n25": (8, 1, 68028, 25, 0, 2),
    }.items():
        tmp = "/tmp/okes_digest_%s.json" % name
        arg = dict(kg=fb, W=W, bern=bern, B=B, n=n, nr=nr, calls=calls, out=tmp)
        subprocess.check_call([sys.executable, __file__, "--digest", json.dumps(arg)], stdout=subprocess.DEVNULL)
        d = json.load(open(tmp))
        d.update(W=W, bern=bern, B=B, n=n, nr=nr, calls=calls)
        digests[name] = d
        print("digest", name, d["sha256"][:16])
    json.dump(digests, open(os.path.join(HERE, "fb_digests.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
