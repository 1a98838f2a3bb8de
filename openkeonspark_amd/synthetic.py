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
