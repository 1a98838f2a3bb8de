"""ctypes front-end of the CPU oracle (oracle/kge_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  It is the checker, never the product: ``openkeonspark_amd`` does not import
it and has no CPU fallback.

Parity status: sampler PINNED against the reference's compiled C++ (oracle/_ref/Base.so and the
fixtures under tests/golden/); model arithmetic PARITY UNPINNED versus TensorFlow 1.x (see the
header of kge_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkge_oracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "Base.so")

TRANSE, TRANSH, TRANSR, TRANSD = 0, 1, 2, 3
MODEL_IDS = {"transe": TRANSE, "transh": TRANSH, "transr": TRANSR, "transd": TRANSD}

_lib = None


def build(force=False):
    """Compile libkge_oracle.so (and oracle/_ref/Base.so when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "kge_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, f32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
        L.orc_kg_load.restype = vp
        L.orc_kg_load.argtypes = [ctypes.c_char_p]
        L.orc_kg_from_arrays.restype = vp
        L.orc_kg_from_arrays.argtypes = [i64, i64, i64, vp, vp, vp, i64]
        L.orc_kg_free.argtypes = [vp]
        for name in ("ent_total", "rel_total", "train_total", "train_total_dup", "batch_total"):
            fn = getattr(L, "orc_kg_" + name)
            fn.restype = i64
            fn.argtypes = [vp]
        for name in ("left_mean", "right_mean", "by_head", "by_tail", "by_rel"):
            fn = getattr(L, "orc_kg_" + name)
            fn.restype = vp
            fn.argtypes = [vp]
        L.orc_set_work_threads.argtypes = [vp, i64]
        L.orc_set_bern.argtypes = [vp, i64]
        L.orc_rand_reset.argtypes = [vp]
        L.orc_stream_state.restype = ctypes.c_uint64
        L.orc_stream_state.argtypes = [vp, i64]
        L.orc_set_stream_state.argtypes = [vp, i64, ctypes.c_uint64]
        L.orc_sampling.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64]
        L.orc_sampling_parallel.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64]
        L.orc_loss.restype = f32
        L.orc_loss.argtypes = [vp, vp, vp, vp, i64, i64]
        L.orc_scores.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp]
        L.orc_grad.restype = f32
        L.orc_grad.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, ci]
        L.orc_sgd_step_sequential.restype = f32
        L.orc_sgd_step_sequential.argtypes = [vp, vp, vp, vp, i64, i64, f32]
        L.orc_sgd_apply_dense.argtypes = [vp, vp, i64, f32]
        L.orc_adam_apply_dense.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32]
        L.orc_predict.argtypes = [vp, vp, vp, vp, i64, vp]
        L.orc_eval_load.restype = vp
        L.orc_eval_load.argtypes = [ctypes.c_char_p]
        for name in ("test_total", "valid_total", "triple_total"):
            fn = getattr(L, "orc_eval_" + name)
            fn.restype = i64
            fn.argtypes = [vp]
        L.orc_eval_test_triple.argtypes = [vp, i64, vp]
        L.orc_test_rank.argtypes = [vp, i64, vp, ci, vp]
        L.orc_max_threads.restype = ci
        L.orc_libc_rand_init.argtypes = [vp]
        L.orc_libc_rand_next.restype = ctypes.c_int32
        L.orc_libc_rand_next.argtypes = [vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def libc_rand_sequence(n):
    """First n outputs of an unseeded glibc rand() (Random.h:9-13 seeds)."""
    L = lib()
    buf = ctypes.create_string_buffer(512)
    L.orc_libc_rand_init(buf)
    return [int(L.orc_libc_rand_next(buf)) for _ in range(n)]


class KG:
    """Sampler-side oracle: loader (Reader.h:27-179) + rng (Random.h) + sampling (Base.cpp:74-172)."""

    def __init__(self, path=None, arrays=None, work_threads=8, bern=0):
        L = lib()
        if path is not None:
            if not path.endswith("/"):
                path += "/"
            self._h = L.orc_kg_load(path.encode())
            if not self._h:
                raise FileNotFoundError(path)
        else:
            E, R, h, t, r, nb = arrays
            h = np.ascontiguousarray(h, dtype=np.int64)
            t = np.ascontiguousarray(t, dtype=np.int64)
            r = np.ascontiguousarray(r, dtype=np.int64)
            self._h = L.orc_kg_from_arrays(E, R, len(h), _p(h), _p(t), _p(r), nb)
        self.entTotal = L.orc_kg_ent_total(self._h)
        self.relTotal = L.orc_kg_rel_total(self._h)
        self.trainTotal = L.orc_kg_train_total(self._h)
        self.trainTotal_ = L.orc_kg_train_total_dup(self._h)
        self.batchTotal = L.orc_kg_batch_total(self._h)
        self.work_threads = work_threads
        L.orc_set_work_threads(self._h, work_threads)
        L.orc_set_bern(self._h, bern)
        L.orc_rand_reset(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kg_free(self._h)
            self._h = None

    def _farr(self, fn, n):
        ptr = fn(self._h)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(n,)).copy()

    def left_mean(self):
        return self._farr(lib().orc_kg_left_mean, self.relTotal)

    def right_mean(self):
        return self._farr(lib().orc_kg_right_mean, self.relTotal)

    def sorted_copy(self, which):
        """(trainTotal,3) int64 array of (h, r, t) in 'head' | 'tail' | 'rel' order (Triple.h:18-28)."""
        fn = getattr(lib(), "orc_kg_by_" + which)
        ptr = fn(self._h)
        a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int64)), shape=(self.trainTotal, 3))
        return a.copy()

    def stream_states(self):
        return np.array([lib().orc_stream_state(self._h, i) for i in range(self.work_threads)], dtype=np.uint64)

    def set_stream_states(self, states):
        for i, s in enumerate(states):
            lib().orc_set_stream_state(self._h, i, int(s))

    def sampling(self, B, neg=1, negrel=0, parallel=False):
        """parallel=True: one OS thread per virtual thread, as the reference's pthreads (same output)."""
        n = B * (1 + neg + negrel)
        h = np.zeros(n, np.int64)
        t = np.zeros(n, np.int64)
        r = np.zeros(n, np.int64)
        y = np.zeros(n, np.float32)
        (lib().orc_sampling_parallel if parallel else lib().orc_sampling)(self._h, _p(h), _p(t), _p(r), _p(y), B, neg, negrel)
        return h, t, r, y


class _CModel(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int), ("E", ctypes.c_int64), ("R", ctypes.c_int64),
                ("De", ctypes.c_int), ("Dr", ctypes.c_int), ("margin", ctypes.c_float),
                ("negative_rel", ctypes.c_int),
                ("ent", ctypes.c_void_p), ("rel", ctypes.c_void_p),
                ("aux_rel", ctypes.c_void_p), ("aux_ent", ctypes.c_void_p)]


# table order used by the C side: 0 ent_embeddings, 1 rel_embeddings, 2 relation-side auxiliary,
# 3 entity-side auxiliary.  Names are the reference's variable names (TransE.py:21-22,
# TransH.py:28, TransR.py:31, TransD.py:39-40).
TABLE_NAMES = {
    TRANSE: ["ent_embeddings", "rel_embeddings"],
    TRANSH: ["ent_embeddings", "rel_embeddings", "normal_vectors"],
    TRANSR: ["ent_embeddings", "rel_embeddings", "transfer_matrix"],
    TRANSD: ["ent_embeddings", "rel_embeddings", "rel_transfer", "ent_transfer"],
}


def table_shapes(model, E, R, De, Dr):
    shapes = {"ent_embeddings": (E, De), "rel_embeddings": (R, Dr)}
    if model == TRANSH:
        shapes["normal_vectors"] = (R, Dr)
    if model == TRANSR:
        shapes["transfer_matrix"] = (R, De * Dr)
    if model == TRANSD:
        shapes["rel_transfer"] = (R, Dr)
        shapes["ent_transfer"] = (E, De)
    return shapes


def xavier_normal(rng, shape):
    """tf.contrib.layers.xavier_initializer(uniform=False) (TransE.py:21): truncated normal with
    stddev sqrt(1.3*2/(fan_in+fan_out)), fan_in=rows, fan_out=cols, resampled beyond 2 stddev."""
    rows, cols = shape
    std = np.sqrt(2.6 / (rows + cols))
    a = rng.standard_normal(shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(a) > 2.0
    return (a * std).astype(np.float32)


def init_params(model, E, R, De, Dr, seed=0):
    rng = np.random.default_rng(seed)
    return {k: xavier_normal(rng, s) for k, s in table_shapes(model, E, R, De, Dr).items()}


def adam_lr_t(lr, beta1, beta2, t):
    """TF1 AdamOptimizer._apply_sparse_shared: lr * sqrt(1 - beta2_power) / (1 - beta1_power) in
    fp32, with the powers kept by repeated fp32 multiplication (adam.py _finish)."""
    f = np.float32
    b1p, b2p = f(beta1), f(beta2)
    for _ in range(t - 1):
        b1p = f(b1p * f(beta1))
        b2p = f(b2p * f(beta2))
    return f(f(lr) * np.sqrt(f(1) - b2p, dtype=np.float32) / (f(1) - b1p))


class Model:
    """Model-side oracle: loss / gradients / optimiser step for TransE/H/R/D."""

    def __init__(self, model, E, R, De, Dr=None, margin=1.0, negative_rel=0, params=None, seed=0):
        self.model = MODEL_IDS[model] if isinstance(model, str) else model
        self.E, self.R, self.De, self.Dr = E, R, De, (Dr if Dr is not None else De)
        self.margin = float(margin)
        self.negative_rel = int(negative_rel)
        self.names = TABLE_NAMES[self.model]
        if params is None:
            params = init_params(self.model, E, R, self.De, self.Dr, seed)
        self.params = {k: np.ascontiguousarray(params[k], dtype=np.float32).copy() for k in self.names}
        self.adam_m = {k: np.zeros_like(v) for k, v in self.params.items()}
        self.adam_v = {k: np.zeros_like(v) for k, v in self.params.items()}
        self.step = 0

    def _c(self):
        tabs = [self.params[k] for k in self.names] + [None] * (4 - len(self.names))
        c = _CModel(self.model, self.E, self.R, self.De, self.Dr, self.margin, self.negative_rel,
                    _p(tabs[0]), _p(tabs[1]),
                    _p(tabs[2]) if tabs[2] is not None else None,
                    _p(tabs[3]) if tabs[3] is not None else None)
        return c

    @staticmethod
    def _batch(bh, bt, br):
        return (np.ascontiguousarray(bh, dtype=np.int64), np.ascontiguousarray(bt, dtype=np.int64),
                np.ascontiguousarray(br, dtype=np.int64))

    def loss(self, bh, bt, br, B, N):
        bh, bt, br = self._batch(bh, bt, br)
        c = self._c()
        return float(lib().orc_loss(ctypes.byref(c), _p(bh), _p(bt), _p(br), B, N))

    def hinge_margins(self, bh, bt, br, B, N):
        """p_b - n_bk + margin for every (b,k): a value within rounding of 0 is a tie that two fp32
        implementations may legitimately resolve differently (the hinge switches a whole gradient row)."""
        bh, bt, br = self._batch(bh, bt, br)
        c = self._c()
        ps = np.zeros(B, np.float32)
        ns = np.zeros(B * N, np.float32)
        lib().orc_scores(ctypes.byref(c), _p(bh), _p(bt), _p(br), B, N, _p(ps), _p(ns))
        return ps[:, None] - ns.reshape(B, N) + np.float32(self.margin)

    def grad(self, bh, bt, br, B, N, denom=None, nthreads=1):
        """-> (loss, {table: dense dL/dtable}).  denom defaults to B*N (reduce_mean)."""
        bh, bt, br = self._batch(bh, bt, br)
        c = self._c()
        g = {k: np.zeros_like(self.params[k]) for k in self.names}
        arr = (ctypes.c_void_p * 4)(*[g[k].ctypes.data for k in self.names] + [None] * (4 - len(self.names)))
        loss = lib().orc_grad(ctypes.byref(c), _p(bh), _p(bt), _p(br), B, N,
                              denom if denom is not None else B * N, arr, nthreads)
        return float(loss), g

    def sgd_step(self, bh, bt, br, B, N, lr, sequential=False, nthreads=1):
        """One GradientDescentOptimizer step (distribute_training.py:98-101).  sequential=True
        applies every IndexedSlices row one by one as scatter_sub does; False sums duplicates first."""
        self.step += 1
        if sequential:
            bh, bt, br = self._batch(bh, bt, br)
            c = self._c()
            return float(lib().orc_sgd_step_sequential(ctypes.byref(c), _p(bh), _p(bt), _p(br), B, N, lr))
        loss, g = self.grad(bh, bt, br, B, N, nthreads=nthreads)
        for k in self.names:
            lib().orc_sgd_apply_dense(_p(self.params[k]), _p(g[k]), g[k].size, lr)
        return loss

    def apply_sgd(self, g, lr):
        """GradientDescentOptimizer on an already summed gradient (as returned by grad())."""
        self.step += 1
        for k in self.names:
            lib().orc_sgd_apply_dense(_p(self.params[k]), _p(g[k]), g[k].size, lr)

    def apply_adam(self, g, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        """AdamOptimizer (TF1 sparse = dense sweep) on an already summed gradient."""
        self.step += 1
        lr_t = adam_lr_t(lr, beta1, beta2, self.step)
        for k in self.names:
            lib().orc_adam_apply_dense(_p(self.params[k]), _p(self.adam_m[k]), _p(self.adam_v[k]), _p(g[k]),
                                       g[k].size, lr_t, beta1, beta2, eps)

    def adam_step(self, bh, bt, br, B, N, lr, beta1=0.9, beta2=0.999, eps=1e-8, nthreads=1):
        """One AdamOptimizer step on IndexedSlices gradients (distribute_training.py:95-96,101)."""
        self.step += 1
        loss, g = self.grad(bh, bt, br, B, N, nthreads=nthreads)
        lr_t = adam_lr_t(lr, beta1, beta2, self.step)
        for k in self.names:
            lib().orc_adam_apply_dense(_p(self.params[k]), _p(self.adam_m[k]), _p(self.adam_v[k]), _p(g[k]),
                                       g[k].size, lr_t, beta1, beta2, eps)
        return loss

    def predict(self, ph, pt, pr):
        ph, pt, pr = self._batch(ph, pt, pr)
        out = np.zeros(len(ph), np.float32)
        c = self._c()
        lib().orc_predict(ctypes.byref(c), _p(ph), _p(pt), _p(pr), len(ph), _p(out))
        return out


class Eval:
    """Link-prediction oracle: importTestFiles / importTypeFiles / importOntologyFiles
    (Reader.h:186-449) + testHead / testTail (Test.h:31-249)."""

    def __init__(self, path):
        if not path.endswith("/"):
            path += "/"
        self._h = lib().orc_eval_load(path.encode())
        if not self._h:
            raise FileNotFoundError(path)
        self.testTotal = lib().orc_eval_test_total(self._h)
        self.validTotal = lib().orc_eval_valid_total(self._h)
        self.tripleTotal = lib().orc_eval_triple_total(self._h)

    def test_triple(self, i):
        """(h, t, r) of the i-th test triple in the reference's (r,h,t)-sorted order (Reader.h:256)."""
        out = np.zeros(3, np.int64)
        lib().orc_eval_test_triple(self._h, i, _p(out))
        return tuple(int(x) for x in out)

    def rank(self, i, scores, head):
        """testHead (head=True) / testTail: 8 int64 as the reference returns them."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        out = np.zeros(8, np.int64)
        lib().orc_test_rank(self._h, i, _p(scores), 1 if head else 0, _p(out))
        return out


class ReferenceSampler:
    """The reference's own Base.so (compiled by oracle/Makefile into oracle/_ref/), driven the way
    Config.py:30-31,160-170,347 drives it.  Process-global state: ONE dataset per process."""

    def __init__(self, path, work_threads=8, bern=0):
        if not os.path.exists(REF_LIB_PATH):
            raise FileNotFoundError(REF_LIB_PATH)
        if not path.endswith("/"):
            path += "/"
        L = ctypes.cdll.LoadLibrary(REF_LIB_PATH)
        L.sampling.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3
        for fn in ("getEntityTotal", "getRelationTotal", "getTrainTotal", "getTrainTotal_", "getBatchTotal"):
            getattr(L, fn).restype = ctypes.c_int64
        L.setWorkThreads.argtypes = [ctypes.c_int64]
        L.setBern.argtypes = [ctypes.c_int64]
        self.L = L
        L.setInPath(ctypes.create_string_buffer(path.encode(), len(path) * 2))
        L.setBern(bern)
        L.setWorkThreads(work_threads)
        L.randReset()
        L.importTrainFiles()
        self.entTotal = L.getEntityTotal()
        self.relTotal = L.getRelationTotal()
        self.trainTotal = L.getTrainTotal()
        self.trainTotal_ = L.getTrainTotal_()
        self.batchTotal = L.getBatchTotal()
        self.work_threads = work_threads

    def _global_ptr(self, name, ctype):
        return ctypes.cast(ctypes.c_void_p.in_dll(self.L, name).value, ctypes.POINTER(ctype))

    def left_mean(self):
        return np.ctypeslib.as_array(self._global_ptr("left_mean", ctypes.c_float), shape=(self.relTotal,)).copy()

    def right_mean(self):
        return np.ctypeslib.as_array(self._global_ptr("right_mean", ctypes.c_float), shape=(self.relTotal,)).copy()

    def sorted_copy(self, which):
        name = {"head": "trainHead", "tail": "trainTail", "rel": "trainRel"}[which]
        return np.ctypeslib.as_array(self._global_ptr(name, ctypes.c_int64), shape=(self.trainTotal, 3)).copy()

    def stream_states(self):
        return np.ctypeslib.as_array(self._global_ptr("next_random", ctypes.c_uint64), shape=(self.work_threads,)).copy()

    def sampling(self, B, neg=1, negrel=0):
        n = B * (1 + neg + negrel)
        h = np.zeros(n, np.int64)
        t = np.zeros(n, np.int64)
        r = np.zeros(n, np.int64)
        y = np.zeros(n, np.float32)
        self.L.sampling(_p(h), _p(t), _p(r), _p(y), B, neg, negrel)
        return h, t, r, y

    # --- link prediction, driven as Config.py:74-80,34-39 and distribute_training.py:465-475 do ---
    def init_link_prediction(self):
        L = self.L
        L.importTestFiles(); L.importTypeFiles(); L.importOntologyFiles()
        for fn in ("getTestTotal", "getValidTotal", "getTripleTotal"):
            getattr(L, fn).restype = ctypes.c_int64
        L.getTailBatch.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.getHeadBatch.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.testTail.argtypes = [ctypes.c_int64, ctypes.c_void_p]
        L.testTail.restype = ctypes.POINTER(ctypes.c_int64 * 8)
        L.testHead.argtypes = [ctypes.c_int64, ctypes.c_void_p]
        L.testHead.restype = ctypes.POINTER(ctypes.c_int64 * 8)
        self.testTotal = L.getTestTotal()
        self.validTotal = L.getValidTotal()
        self.tripleTotal = L.getTripleTotal()

    def batch(self, i, head):
        h = np.zeros(self.entTotal, np.int64); t = np.zeros(self.entTotal, np.int64); r = np.zeros(self.entTotal, np.int64)
        (self.L.getHeadBatch if head else self.L.getTailBatch)(i, _p(h), _p(t), _p(r))
        return h, t, r

    def rank(self, i, scores, head):
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        res = (self.L.testHead if head else self.L.testTail)(i, _p(scores))
        return np.array(list(res.contents), dtype=np.int64)
