"""Seeded synthetic knowledge graphs in the OpenKE on-disk format.

The reference ships no datasets, so benchmarks and parity tests use
shape-matched synthetic KGs written in the text format `importTrainFiles` reads
(/root/reference/base/Reader.h:27-100, README.md:67-74): ``entity2id.txt`` / ``relation2id.txt``
whose first line is the count, and ``train2id.txt`` = N followed by N lines ``head tail rel``.
"""
import os

import numpy as np

# public cardinalities of the datasets BASELINE.json names
FB15K237 = dict(entities=14541, relations=237, train=272115, seed=237)
WN18RR = dict(entities=40943, relations=11, train=86835, seed=18)


def _zipf_draw(rng, n_items, exponent, size):
    """Inverse-CDF draw from a Zipf(exponent) law over a random permutation of n_items ids."""
    w = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), exponent)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(size), side="left")
    ranks = np.minimum(ranks, n_items - 1)
    perm = rng.permutation(n_items)
    return perm[ranks].astype(np.int64)


def generate_triples(entities, relations, train, seed, ent_exponent=0.8, rel_exponent=1.0, dup_frac=0.001):
    """-> (h, t, r) int64 arrays of length `train`; about dup_frac of the lines repeat an earlier
    line so that the loader's dedup (Reader.h:106-123) is exercised."""
    rng = np.random.default_rng(seed)
    h = _zipf_draw(rng, entities, ent_exponent, train)
    t = _zipf_draw(rng, entities, ent_exponent, train)
    r = _zipf_draw(rng, relations, rel_exponent, train)
    n_dup = int(train * dup_frac)
    if n_dup > 0 and train > 1:
        dst = rng.integers(train // 2, train, n_dup)
        src = rng.integers(0, train // 2, n_dup)
        h[dst], t[dst], r[dst] = h[src], t[src], r[src]
    return h, t, r


def write_openke_dir(path, entities, relations, h, t, r, new_batch_total=None):
    """Write entity2id.txt, relation2id.txt, train2id.txt (and batch2id.txt in incremental mode)."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "entity2id.txt"), "w") as f:
        f.write("%d\n" % entities)
        f.write("".join("e%d\t%d\n" % (i, i) for i in range(min(entities, 100000))))
    with open(os.path.join(path, "relation2id.txt"), "w") as f:
        f.write("%d\n" % relations)
        f.write("".join("r%d\t%d\n" % (i, i) for i in range(min(relations, 100000))))
    arr = np.stack([h, t, r], axis=1)
    with open(os.path.join(path, "train2id.txt"), "w") as f:
        f.write("%d\n" % len(h))
        np.savetxt(f, arr, fmt="%d")
    if new_batch_total is not None:
        # only the first line is read by the loader (Reader.h:61-67); the triples themselves are
        # the last new_batch_total lines of train2id.txt (main_spark.py:152-174 appends them)
        with open(os.path.join(path, "batch2id.txt"), "w") as f:
            f.write("%d\n" % new_batch_total)
            np.savetxt(f, arr[len(h) - new_batch_total:], fmt="%d")
    return path


def make_dataset(path, spec=None, **overrides):
    """Create (once) a synthetic OpenKE directory; returns the path with a trailing slash."""
    spec = dict(spec or FB15K237)
    spec.update(overrides)
    if not path.endswith("/"):
        path += "/"
    marker = os.path.join(path, ".complete")
    if not os.path.exists(marker):
        h, t, r = generate_triples(spec["entities"], spec["relations"], spec["train"], spec["seed"])
        write_openke_dir(path, spec["entities"], spec["relations"], h, t, r)
        open(marker, "w").write("ok\n")
    return path


# ---------------------------------------------------------------------------------------------------
# Graphs with LEARNABLE structure, for metric-level checks (MR / Hits@10 of a trained model): the Zipf
# graphs above have independent heads, tails and relations, so link prediction on them can only learn
# entity popularity.  Here every entity has one of `types` latent types, relation r maps head type a to
# tail type (a * mul_r + add_r) mod types, and a triple's tail is drawn (Zipf inside the type) from the
# entities of the mapped type: a translational model that places types at distinct centroids ranks the
# right tail among ~entities/types candidates instead of ~entities/2.
# ---------------------------------------------------------------------------------------------------
FB15K237_TYPED = dict(entities=14541, relations=237, train=272115, valid=17535, test=20466, types=64, seed=2370)
SMALL_TYPED = dict(entities=1200, relations=12, train=20000, valid=300, test=400, types=12, seed=120)


def generate_typed_triples(entities, relations, count, types, seed, ent_exponent=0.8, rel_exponent=1.0):
    rng = np.random.default_rng(seed)
    ent_type = rng.integers(0, types, entities)
    by_type = [np.nonzero(ent_type == c)[0] for c in range(types)]
    add = rng.integers(0, types, relations)
    h = _zipf_draw(rng, entities, ent_exponent, count)
    r = _zipf_draw(rng, relations, rel_exponent, count)
    t_type = (ent_type[h] + add[r]) % types
    t = np.empty(count, np.int64)
    for c in range(types):
        m = np.nonzero(t_type == c)[0]
        if len(m):
            members = by_type[c] if len(by_type[c]) else np.arange(entities)
            t[m] = members[_zipf_draw(rng, len(members), ent_exponent, len(m))]
    return h, t, r, ent_type


def make_typed_dataset(path, spec=None, **overrides):
    """train2id / valid2id / test2id (disjoint draws of the same generator; valid and test triples that
    also occur in train are kept, as in real splits they would not be -- the filtered ranks handle
    them) plus type_constrain.txt in the format Reader.h:317-330 reads.  Returns the path."""
    spec = dict(spec or FB15K237_TYPED)
    spec.update(overrides)
    if not path.endswith("/"):
        path += "/"
    marker = os.path.join(path, ".complete")
    if os.path.exists(marker):
        return path
    E, R = spec["entities"], spec["relations"]
    total = spec["train"] + spec["valid"] + spec["test"]
    h, t, r, _ = generate_typed_triples(E, R, total, spec["types"], spec["seed"])
    n_tr, n_va = spec["train"], spec["valid"]
    write_openke_dir(path, E, R, h[:n_tr], t[:n_tr], r[:n_tr])
    for name, sl in (("valid2id.txt", slice(n_tr, n_tr + n_va)), ("test2id.txt", slice(n_tr + n_va, total))):
        with open(os.path.join(path, name), "w") as f:
            f.write("%d\n" % (sl.stop - sl.start))
            np.savetxt(f, np.stack([h[sl], t[sl], r[sl]], axis=1), fmt="%d")
    with open(os.path.join(path, "type_constrain.txt"), "w") as f:   # per relation: heads seen, then tails seen
        f.write("%d\n" % R)
        order = np.argsort(r, kind="stable")
        bounds = np.searchsorted(r[order], np.arange(R + 1))
        for rel in range(R):
            idx = order[bounds[rel]:bounds[rel + 1]]
            heads = np.unique(h[idx]); tails = np.unique(t[idx])
            f.write("%d\t%d%s\n" % (rel, len(heads), "".join("\t%d" % x for x in heads)))
            f.write("%d\t%d%s\n" % (rel, len(tails), "".join("\t%d" % x for x in tails)))
    open(marker, "w").write("ok\n")
    return path
