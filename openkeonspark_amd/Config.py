"""`Config`: the reference's binding / hyper-parameter / session object (/root/reference/Config.py)
re-hosted on the MI355X engine.

Same constructor, setters, attributes, `init()` and `sampling()` as the reference
(Config.py:17-72,153-210,225-347), so `distribute_training.get_conf`-style callers work unchanged
(distribute_training.py:32-71).  What the reference did with a TensorFlow session --
`sess.run([train_op, loss, global_step], feed_dict)` (distribute_training.py:282) built by
`create_model` (distribute_training.py:74-105) -- is `train_step()` here: sample on the device,
run the fused forward/backward operator, exchange gradients over RCCL when data-parallel, apply
SGD or TF-parity Adam.  Nothing in this module computes on the CPU; without the HIP library or a
GPU the device calls raise.
"""
import ctypes
import json
import os

import numpy as np

from . import _lib
from ._lib import KgeError


class Config(object):
    '''
    use ctypes to call the engine's C ABI from python and set essential parameters.
    '''

    def __init__(self, cpp_lib_path=None, init_new_entities=False):
        self.init_new_entities = init_new_entities
        if init_new_entities == False:
            # C library (Config.py:28-31): defaults to the in-tree MI355X engine
            if cpp_lib_path is None:
                cpp_lib_path = _lib.LIB_PATH
            self.lib = _lib.load(os.path.abspath(cpp_lib_path))
            # other parameters (Config.py:53-72)
            self.in_path = None
            self.out_path = None
            self.bern = 0
            self.hidden_size = 64
            self.ent_size = self.hidden_size
            self.rel_size = self.hidden_size
            self.train_times = 0
            self.margin = 1.0
            self.nbatches = 0
            self.negative_ent = 1
            self.negative_rel = 0
            self.workThreads = 8
            self.alpha = 0.001
            self.exportName = None
            self.importName = None
            self.opt_method = "SGD"
            self.test_link_prediction = False
            self.test_triple_classification = False
            self.valid_triple_classification = False
            # engine-side additions
            self.seed = 0                 # parameter initialisation seed
            self.device = "cuda"
            self.adam_beta1, self.adam_beta2, self.adam_epsilon = 0.9, 0.999, 1e-8  # TF1 AdamOptimizer defaults
            self.trainModel = None
            self.global_step = 0
            self.rank, self.world_size = 0, 1
            self._pg = None

    # ------------------------------------------------------------------------------------------
    # Config.py:153-210
    # ------------------------------------------------------------------------------------------
    def init(self):
        '''
        prepare for train and test
        '''
        if self.init_new_entities == False:
            self.trainModel = None
            if self.in_path != None:
                path = self.in_path if self.in_path.endswith("/") else self.in_path + "/"
                self.lib.kge_clear_error()
                self.lib.setInPath(path.encode())
                self.lib.setBern(self.bern)
                self.lib.setWorkThreads(self.workThreads)
                self.lib.randReset()
                self.lib.importTrainFiles()
                _lib.raise_if_error(self.lib)
                self.relTotal = self.lib.getRelationTotal()
                self.entTotal = self.lib.getEntityTotal()
                self.trainTotal = self.lib.getTrainTotal_()
                self.testTotal = self.lib.getTestTotal()
                self.validTotal = self.lib.getValidTotal()
                self.bt = self.lib.getBatchTotal()
                self.set_mini_batch()
                self._alloc_batch_buffers()
            if self.test_link_prediction:
                self.init_link_prediction()
            # triple-classification inputs (Test.h:266-444) are not built yet

    def init_link_prediction(self):
        r'''
        import essential files for link prediction (Config.py:74-80); type / ontology constraints are
        optional here (the reference crashes without them)
        '''
        path = self.in_path if self.in_path.endswith("/") else self.in_path + "/"
        self.lib.kge_clear_error()
        self.lib.importTestFiles()
        if os.path.exists(path + "type_constrain.txt"):
            self.lib.importTypeFiles()
        if os.path.exists(path + "ontology_constrain.txt"):
            self.lib.importOntologyFiles()
        _lib.raise_if_error(self.lib)
        self.testTotal = self.lib.getTestTotal()
        self.validTotal = self.lib.getValidTotal()

    def init_from_arrays(self, ent_total, rel_total, h, t, r, new_batch_total=0):
        """Same as init() with the training triples (file order) given as arrays instead of files."""
        h = np.ascontiguousarray(h, dtype=np.int64)
        t = np.ascontiguousarray(t, dtype=np.int64)
        r = np.ascontiguousarray(r, dtype=np.int64)
        self.lib.kge_clear_error()
        self.lib.setBern(self.bern)
        self.lib.setWorkThreads(self.workThreads)
        self.lib.randReset()
        _lib.check(self.lib.kge_import_train_arrays(ent_total, rel_total, len(h), h.ctypes.data, t.ctypes.data,
                                                    r.ctypes.data, new_batch_total), self.lib)
        self.relTotal = self.lib.getRelationTotal()
        self.entTotal = self.lib.getEntityTotal()
        self.trainTotal = self.lib.getTrainTotal_()
        self.testTotal = self.validTotal = 0
        self.bt = self.lib.getBatchTotal()
        self.set_mini_batch()
        self._alloc_batch_buffers()

    def _alloc_batch_buffers(self):
        # Config.py:172-180
        self.batch_seq_size = self.batch_size * (1 + self.negative_ent + self.negative_rel)
        self.batch_h = np.zeros(self.batch_seq_size, dtype=np.int64)
        self.batch_t = np.zeros(self.batch_seq_size, dtype=np.int64)
        self.batch_r = np.zeros(self.batch_seq_size, dtype=np.int64)
        self.batch_y = np.zeros(self.batch_seq_size, dtype=np.float32)
        self.batch_h_addr = self.batch_h.__array_interface__['data'][0]
        self.batch_t_addr = self.batch_t.__array_interface__['data'][0]
        self.batch_r_addr = self.batch_r.__array_interface__['data'][0]
        self.batch_y_addr = self.batch_y.__array_interface__['data'][0]

    def set_mini_batch(self):
        '''
        Set mini batch used during training (Config.py:189-210)
        '''
        tot = self.bt if self.bt > 0 else self.trainTotal
        if self.nbatches > 0:
            self.batch_size = int(tot / self.nbatches)
        else:
            self.batch_size = tot
            while self.batch_size > 9999:
                self.batch_size = int(self.batch_size / 10)
            self.nbatches = int(tot / self.batch_size)
        print("Batch size is {}".format(self.batch_size))
        print("Number of batches: {}".format(self.nbatches))

    # ------------------------------------------------------------------------------------------
    # getters / setters (Config.py:213-340, 425-429)
    # ------------------------------------------------------------------------------------------
    def get_ent_total(self):
        return self.entTotal

    def get_rel_total(self):
        return self.relTotal

    def set_opt_method(self, method):
        self.opt_method = method

    def set_test_link_prediction(self, flag):
        self.test_link_prediction = flag

    def set_test_triple_classification(self, flag):
        self.test_triple_classification = flag

    def set_valid_triple_classification(self, flag):
        self.valid_triple_classification = flag

    def set_alpha(self, alpha):
        self.alpha = alpha

    def set_in_path(self, path):
        self.in_path = path

    def set_out_files(self, path):
        self.out_path = path

    def set_bern(self, bern):
        self.bern = bern

    def set_dimension(self, dim):
        self.hidden_size = dim
        self.ent_size = dim
        self.rel_size = dim

    def set_ent_dimension(self, dim):
        self.ent_size = dim

    def set_rel_dimension(self, dim):
        self.rel_size = dim

    def set_train_times(self, times):
        self.train_times = times

    def set_nbatches(self, nbatches):
        self.nbatches = nbatches

    def set_margin(self, margin):
        self.margin = margin

    def set_ent_neg_rate(self, rate):
        self.negative_ent = rate

    def set_rel_neg_rate(self, rate):
        self.negative_rel = rate

    def set_import_files(self, path):
        self.importName = path

    def set_export_files(self, path):
        self.exportName = path

    def set_work_threads(self, threads):
        """Number of VIRTUAL sampler threads (rng streams / batch slices, Setting.h:36-39).  The
        reference hard-codes 8 (Config.py:65); data-parallel ranks split them."""
        self.workThreads = threads

    def set_model(self, model):
        self.model = model

    # ------------------------------------------------------------------------------------------
    # sampling: Base.so-compatible host path (Config.py:343-347)
    # ------------------------------------------------------------------------------------------
    def sampling(self):
        '''
        Call the engine for batch sampling into the numpy buffers (same bits as the reference)
        '''
        self.lib.kge_clear_error()
        self.lib.sampling(self.batch_h_addr, self.batch_t_addr, self.batch_r_addr, self.batch_y_addr,
                          self.batch_size, self.negative_ent, self.negative_rel)
        _lib.raise_if_error(self.lib)

    # ------------------------------------------------------------------------------------------
    # "session": parameters, optimiser state, train / test steps
    # ------------------------------------------------------------------------------------------
    def set_model_and_session(self, model):
        '''
        Create the model's device tables and the optimiser state (Config.py:447-461 +
        distribute_training.create_model, :74-105).
        '''
        import torch
        self.model = model
        self._shard_plan = self._plan_entity_shard(model)
        self.trainModel = self.model(config=self, define=True)
        m = self.trainModel
        self._desc = m.descriptor()
        self._tables = [m.parameter_lists[n] for n in m.table_names]
        self._adam = self.opt_method in ("Adam", "adam")  # distribute_training.py:95
        # opt-in, NON-PARITY: Adam on the touched rows only (tf.contrib.opt.LazyAdamOptimizer's rule) for tables whose dense
        # TF1-Adam sweep -- 32 bytes per element per step, the reference's semantics -- would dominate the step.  Rides on the
        # sparse-row path; `_has_slots` = "m / v tables and beta powers exist" (both Adam flavours)
        self._lazy_adam = self.opt_method in ("LazyAdam", "lazyadam", "lazy_adam")
        self._has_slots = self._adam or self._lazy_adam
        # TransE: exact integer sign-count gradients instead of fp32 atomics (include/kge_mi355.h)
        n_neg = self.negative_ent + self.negative_rel
        self.use_counts = bool(getattr(self, "use_counts", True)) and bool(
            self.lib.kge_transe_counts_supported(ctypes.byref(self._desc), n_neg))
        # sparse-row mode: no dense gradient / count image at all.  The int8 records are reduced into a compact
        # [touched rows, D] image and only those rows are updated; across ranks the records themselves are
        # all-gathered (the sparse touched-row exchange of BASELINE config #5).  SGD only: TF1's sparse Adam
        # sweeps every row of m, v and the table each step, which is the dense path by definition.
        requested = getattr(self, "sparse_rows", None)   # None = automatic, True / False = the caller's wish
        table_bytes, sparse = self._wants_sparse_rows(n_neg)
        self.sparse_rows = (bool(sparse) or self._lazy_adam) and self.use_counts and not self._adam
        # TransH / TransD (and TransE outside the sign-count path) with SGD: the touched rows are updated in place from float
        # gradient records (kge_forward_backward_sgd_rows) -- no gradient tables, no sweep.  On request, or by itself for tables
        # beyond 2 GB (measured at 4 GB, dim 200, B = 131 072, n = 1: TransH 1.63 -> 0.70 ms, TransD 2.83 -> 0.97 ms per step and
        # half the memory, profiles/r03_h_*; the dense form's sweep scales with the table, so the two tie near 1 GB)
        vector_model = m.model_id in (_lib.TRANSE, _lib.TRANSH, _lib.TRANSD)
        self.sparse_inplace = bool(not self.sparse_rows and vector_model and not self._has_slots and
                                   (requested or (requested is None and table_bytes > (2 << 30))))
        if requested and not (self.sparse_rows or self.sparse_inplace):
            raise KgeError("sparse_rows needs TransE / TransH / TransD with SGD (TransE on the sign-count path also with LazyAdam)")
        if self._lazy_adam and not self.sparse_rows:
            raise KgeError("LazyAdam (touched rows only, NON-PARITY) needs TransE on the sign-count path: 1..63 negatives")
        self._grads = [] if (self.sparse_rows or self.sparse_inplace) else [torch.zeros_like(t) for t in self._tables]
        if self._has_slots:
            self._adam_m = [torch.zeros_like(t) for t in self._tables]
            self._adam_v = [torch.zeros_like(t) for t in self._tables]
            self._beta1_power = np.float32(self.adam_beta1)
            self._beta2_power = np.float32(self.adam_beta2)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._refresh_pointers()
        self._dist_ready = 0
        self._dev_batch = None
        self._dev_batch2 = None
        self._side_stream = None
        self._prefetched = None
        self.global_step = 0
        self._sparse_buf = None
        # TransE steps with fewer gradient rows than this take the single fused fp32-atomic kernel (launch-bound
        # regime, tools/sweep_paths.py); 0 = always the exact, run-to-run reproducible count pipeline
        self.counts_min_records = int(getattr(self, "counts_min_records",
                                              os.environ.get("KGE_COUNTS_MIN_RECORDS", 1 << 16)))
        # None = automatic: on in data-parallel runs (batch i+1 is drawn while step i's gradient exchange waits on the wire) and,
        # at 1 GPU, on the sign-count path, where the sampler starts right behind the emit kernel and runs beside the small
        # kernels after it (measured +4 %); off on the other 1-GPU paths (there it only competes with the step's own kernels)
        self.prefetch_sampling = getattr(self, "prefetch_sampling", None)
        self._prefetch_auto = self.prefetch_sampling is None      # (init_distributed decides again once the world size is known)
        if self._prefetch_auto:
            self.prefetch_sampling = self._prefetch_default()
        if self.use_counts and not self.sparse_rows:
            self._counts = torch.zeros((self.entTotal + self.relTotal, self.hidden_size), dtype=torch.int32,
                                       device=self.device)
        self._setup_partition()

    def _wants_sparse_rows(self, n_neg):
        """(table bytes, sparse rows wanted?) -- the caller's wish (`sparse_rows` True / False) or, left at None, the measured rule."""
        table_bytes = (self.entTotal + self.relTotal) * self.hidden_size * 4
        sparse = getattr(self, "sparse_rows", None)
        if sparse is None:
            # Measured cross-over on one MI355X (tools/sparse_crossover.sh, profiles/r03_sparse_crossover.jsonl: dim 512,
            # B = 131 072, n = 1, tables of 0.26 .. 8.2 GB): the dense step costs the batch's work plus a sweep of the whole
            # table and count image, the sparse step the batch's work plus its sort -- they tie at 0.26 GB (0.68 vs 0.70 ms),
            # sparse rows win by 18 % at 0.5 GB and 5.4x at 8 GB.  The tie sits where the table is ~0.3x the bytes of the rows
            # a step touches (1.07 GB there); below 128 MB the dense path's image fits the caches and it is kept.
            touched_bytes = self.batch_size * (3 + n_neg) * self.hidden_size * 4
            threshold = getattr(self, "sparse_threshold_bytes", None)
            if threshold is None:
                threshold = max(128 << 20, int(0.3 * touched_bytes))
            sparse = table_bytes > int(threshold)
        return table_bytes, sparse

    def _plan_entity_shard(self, model):
        """If this process already belongs to a torch.distributed world of N > 1 ranks and the step will be the table-sharded
        sparse one (TransE on the sign-count path, SGD or LazyAdam), the rows [lo, hi) of the entity table this rank will own:
        the model then draws ONLY those rows (Model.embedding_def) instead of the whole table that _setup_shards would cut
        down -- 102 GB per rank at BASELINE config #5.  None otherwise (the table is created whole, as before)."""
        try:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                return None
            W, g = dist.get_world_size(), dist.get_rank()
        except Exception:
            return None
        if W <= 1 or W > 64 or self.hidden_size % 4:
            return None
        probe = model(config=self, define=False)
        if probe.model_id != _lib.TRANSE or self.opt_method in ("Adam", "adam"):
            return None
        n_neg = self.negative_ent + self.negative_rel
        desc = probe.descriptor()
        if not (bool(getattr(self, "use_counts", True)) and bool(self.lib.kge_transe_counts_supported(ctypes.byref(desc), n_neg))):
            return None
        lazy = self.opt_method in ("LazyAdam", "lazyadam", "lazy_adam")
        if not (bool(self._wants_sparse_rows(n_neg)[1]) or lazy):
            return None
        from .parallel import chunk_size
        chunk = chunk_size(self.entTotal, W)
        return dict(world=W, rank=g, chunk=chunk, lo=min(g * chunk, self.entTotal), hi=min((g + 1) * chunk, self.entTotal))

    # --- data-parallel partition: rank g owns virtual threads [g*W/G, (g+1)*W/G) ---
    def init_distributed(self, process_group=None):
        """Join the (already initialised) torch.distributed world: one process per GPU, RCCL."""
        import torch.distributed as dist
        self._pg = process_group
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        if self.trainModel is not None:
            self._setup_partition()
        if getattr(self, "_prefetch_auto", getattr(self, "prefetch_sampling", None) is None):
            self.prefetch_sampling = self._prefetch_default()

    def _prefetch_default(self):
        n_neg = self.negative_ent + self.negative_rel
        counts_path = bool(getattr(self, "use_counts", False)) and not getattr(self, "sparse_rows", False) and \
            self.batch_size * (3 + n_neg) >= int(getattr(self, "counts_min_records", 1 << 16)) * self.world_size
        pair_path = getattr(self, "_desc", None) is not None and \
            bool(self.lib.kge_pair_path_active(ctypes.byref(self._desc), max(self._n_local, 1) if hasattr(self, "_n_local") else self.batch_size, n_neg))
        transr = getattr(self, "trainModel", None) is not None and self.trainModel.model_id == _lib.TRANSR
        dense_one_gpu = not getattr(self, "sparse_rows", False) and not getattr(self, "sparse_inplace", False)
        return bool(self.world_size > 1 or counts_path or pair_path or transr or dense_one_gpu)

    @property
    def _dp(self):
        """Does a step go through the data-parallel exchange?  More than one rank -- or `force_data_parallel` on a one-rank
        process group: the rehearsal of the RCCL path (collectives on device memory, asynchronous work handles, the in-place
        all-gather) on a box with a single GPU; results equal the plain single-process step bit for bit."""
        return self.world_size > 1 or bool(getattr(self, "force_data_parallel", False))

    def _setup_partition(self):
        from .parallel import thread_range
        lo, hi = thread_range(self.rank, self.world_size, self.workThreads)
        self._thread_lo, self._thread_hi = lo, hi
        first = ctypes.c_int64(0)
        self._n_local = self.lib.kge_slice_positions(self.batch_size, lo, hi, ctypes.byref(first))
        self._first_pos = first.value
        if self._dp and self.trainModel is not None and getattr(self, "_dist_ready", 0) != self.world_size:
            if getattr(self, "sparse_inplace", False):
                pass          # replicated tables, the step's gradient records all-gathered (_records_step): nothing to lay out
            elif self.sparse_rows:
                self._setup_shards()
            else:
                self._setup_flat_buffers()
            self._dist_ready = self.world_size

    def _refresh_pointers(self):
        self.lib.kge_set_option(b"tables_changed", 1)
        self._tab_ptrs = _lib.table_ptrs([t.data_ptr() for t in self._tables])
        self._grad_ptrs = _lib.table_ptrs([g.data_ptr() for g in self._grads])
        self._numel = (ctypes.c_int64 * _lib.KGE_MAX_TABLES)(*[t.numel() for t in self._tables])
        if self._has_slots:
            self._adam_m_ptrs = _lib.table_ptrs([t.data_ptr() for t in self._adam_m])
            self._adam_v_ptrs = _lib.table_ptrs([t.data_ptr() for t in self._adam_v])

    def _setup_flat_buffers(self):
        """Data-parallel layout of the dense path: all tables in ONE flat fp32 buffer cut into world_size equal chunks,
        rank g owning chunk g.  Per step the gradient image is reduce-scattered, the optimizer runs on the owned chunk only
        and the updated chunks are all-gathered (what the reference's ps tasks do for their share of the variables,
        distribute_training.py:193-196).  TransE: the chunk is a whole number of rows of the [(E+R), D] row space, because
        the sign-count update needs whole rows; other models: any 4-element boundary."""
        import torch
        from .parallel import chunk_size
        W = self.world_size
        names = list(self.trainModel.table_names)
        numels = [t.numel() for t in self._tables]
        # PIECES: the flat buffer is cut into K consecutive segments, each of them exchanged and updated like a small data-parallel
        # problem of its own (rank g owns the g-th W-th of every segment), so that the optimizer on piece k and its all-gather run
        # while piece k+1 is still on the wire.  One piece for FB15k-237-sized tables (the exchange is latency-, not size-bound
        # there); four from 64 MB of image on.  `dp_pieces` overrides (tests force 2 on a small graph).
        image_bytes = sum(numels) * 4
        K = int(getattr(self, "dp_pieces", 0)) or (4 if image_bytes >= (64 << 20) else 1)
        if self.use_counts:     # ent_embeddings then rel_embeddings, back to back: rows of one [(E+R), D] space
            D = self.hidden_size
            chunk_rows = chunk_size(self.entTotal + self.relTotal, W, 4 * K)
            if chunk_rows * W == self.entTotal + self.relTotal:
                chunk_rows += 4 * K        # at least one spare row at the end: the loss rides in it (train_step)
            chunk = chunk_rows * D
            offs = [0, numels[0]]
        else:
            offs, off = [], 0
            for n in numels:
                offs.append(off)
                off += -(-n // 64) * 64
            chunk = chunk_size(off + 4, W, 4 * K)     # (+ 4: a spare tail slot for the loss)
        total = chunk * W
        dev = self._tables[0].device
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_m = torch.zeros(total, dtype=torch.float32, device=dev) if self._adam else None
        flat_v = torch.zeros(total, dtype=torch.float32, device=dev) if self._adam else None
        for i, name in enumerate(names):
            shape = self._tables[i].shape
            view = flat_p[offs[i]:offs[i] + numels[i]].view(shape)
            view.copy_(self._tables[i])
            self._tables[i] = view
            self.trainModel.parameter_lists[name] = view
            setattr(self.trainModel, name, view)
            self._grads[i] = flat_g[offs[i]:offs[i] + numels[i]].view(shape)
            if self._adam:
                for flat, slots in ((flat_m, self._adam_m), (flat_v, self._adam_v)):
                    v = flat[offs[i]:offs[i] + numels[i]].view(shape)
                    v.copy_(slots[i])
                    slots[i] = v
        self._flat_p, self._flat_g, self._flat_m, self._flat_v = flat_p, flat_g, flat_m, flat_v
        self._chunk = chunk
        self._pieces = K
        seg, own = total // K, chunk // K                     # elements per piece, and per (piece, rank)
        # piece k: segment [k*seg, (k+1)*seg) of the flat buffers; this rank's share of it; where that share sits in the owned images
        self._piece_seg = [(k * seg, (k + 1) * seg) for k in range(K)]
        self._piece_own = [(k * seg + self.rank * own, k * seg + (self.rank + 1) * own) for k in range(K)]
        self._own_len = own
        self._grads_own = torch.zeros(chunk, dtype=torch.float32, device=dev)
        self._loss_tail = total - 1                           # spare element (no table reaches it) of the LAST rank's last piece
        # the loss scalar LIVES in that slot: the step's kernels write this rank's partial loss there, the exchange replaces it by
        # the global one (no copy in either direction)
        self._loss = flat_p[self._loss_tail:]
        if self.use_counts:
            rows_total = chunk_rows * W
            seg_r, own_r = rows_total // K, chunk_rows // K
            live = self.entTotal + self.relTotal
            self._counts = torch.zeros((rows_total, self.hidden_size), dtype=torch.int32, device=dev)
            self._counts_own = torch.zeros((chunk_rows, self.hidden_size), dtype=torch.int32, device=dev)
            self._piece_rows = [(min(k * seg_r + self.rank * own_r, live), min(k * seg_r + (self.rank + 1) * own_r, live)) for k in range(K)]
            self._own_rows_len = own_r
        self._opt_state_synced = True
        self._refresh_pointers()

    def _setup_shards(self):
        """Table-sharded sparse mode (config #5 on N GPUs): rank g keeps the entity rows [g*chunk, (g+1)*chunk) only; the
        relation table stays replicated.  The shard is cut from the table this rank initialised with the common seed, so
        the union of the shards is the single-process table."""
        import torch
        from .parallel import chunk_size
        if self.hidden_size % 4:
            raise KgeError("the table-sharded sparse mode needs an embedding width that is a multiple of 4")
        if self.world_size > 64:
            raise KgeError("the table-sharded sparse mode supports up to 64 ranks")
        W, g, E, D = self.world_size, self.rank, self.entTotal, self.hidden_size
        chunk = chunk_size(E, W)
        lo, hi = min(g * chunk, E), min((g + 1) * chunk, E)
        if getattr(self, "_ent_is_shard", False):     # the model drew this rank's rows only (_plan_entity_shard); the moments were made shard-sized
            plan = self._shard_plan
            if (plan["world"], plan["rank"], plan["chunk"]) != (W, g, chunk) or self._tables[0].shape[0] != chunk:
                raise KgeError("the entity table was created as the shard of rank %d of %d, the process group has rank %d of %d"
                               % (plan["rank"], plan["world"], g, W))
            self._shard = dict(chunk=chunk, lo=lo, hi=hi)
            self._refresh_pointers()
            return
        full = self._tables[0]
        shard = torch.zeros((chunk, D), dtype=torch.float32, device=full.device)
        if hi > lo:
            shard[:hi - lo].copy_(full[lo:hi])
        self._tables[0] = shard
        self.trainModel.parameter_lists["ent_embeddings"] = shard
        self.trainModel.ent_embeddings = shard
        del full
        if self._lazy_adam:     # the moments of the entity rows are sharded with them (owner computes); the relation table's stay replicated
            for slots in (self._adam_m, self._adam_v):
                old = slots[0]
                slots[0] = torch.zeros((chunk, D), dtype=torch.float32, device=shard.device)
                if hi > lo:
                    slots[0][:hi - lo].copy_(old[lo:hi])
                del old
        torch.cuda.empty_cache()
        self._shard = dict(chunk=chunk, lo=lo, hi=hi)
        self._refresh_pointers()

    def _stream(self):
        """The current HIP stream of this Config's device, as the C ABI takes it.  Read through the raw accessor: building a
        torch.cuda.Stream object per call (torch.cuda.current_stream()) was 60 % of the host time of a step at the reference's
        batch sizes, where the host, not the GPU, set the step time (tools/host_bound_small.py)."""
        import torch
        if getattr(self, "_dev_index_of", None) != self.device:       # (callers may set con.device after construction)
            d = torch.device(self.device)
            self._dev_index = d.index if d.index is not None else torch.cuda.current_device()
            self._dev_index_of = self.device
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # (a private accessor: fall back to the public object if it ever goes)
        return ctypes.c_void_p(raw(self._dev_index) if raw is not None else torch.cuda.current_stream().cuda_stream)

    def _ensure_dev_batch(self, stride):
        import torch
        n = stride * (1 + self.negative_ent + self.negative_rel)
        if self._dev_batch is None or self._dev_batch.shape[2] != n:
            self._dev_batch = torch.zeros((2, 3, n), dtype=torch.int32, device=self.device)  # two slots: double buffer
        return self._dev_batch

    def sample_device(self, slot=0):
        """Sample this rank's slice of the next batch on the device; returns (int32[3,n] tensor, n_pos)."""
        stride = max(self._n_local, 1)
        buf = self._ensure_dev_batch(stride)[slot]
        nl = ctypes.c_int64(0)
        _lib.check(self.lib.kge_sampling_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(),
                                                self.batch_size, self.negative_ent, self.negative_rel,
                                                self._thread_lo, self._thread_hi, stride, ctypes.byref(nl),
                                                self._stream()), self.lib)
        return buf, nl.value

    def _next_sampled_batch(self):
        """The batch for this step, sampled one step ahead on a side stream: the sampler depends on the rng
        streams and the dataset only, never on the parameters, so batch i+1 is drawn while step i's
        reduction / gradient exchange / update run.  Same batches, same order, same bits."""
        import torch
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
            self._slot = 0
            self._prefetched = None
            self.lib.kge_set_option(b"record_emit_event", 1)
        if self._prefetched is None:
            dev, n_pos = self.sample_device(self._slot)
        else:
            dev, n_pos, ev = self._prefetched
            if ev is not None:            # (None: drawn on this stream by a sampler that rode in a launch of the step)
                torch.cuda.current_stream().wait_event(ev)
        return dev, n_pos

    def _attach_next_batch(self):
        """Arm the sampler of the NEXT batch so that it rides in this step's bucket-scatter launch (kge_sampling_attach): the
        sign-count path's way of drawing batch i+1 during step i -- one stream, no events.  Call _flush_next_batch() once the
        step's forward launches are enqueued."""
        stride = max(self._n_local, 1)
        self._slot ^= 1
        buf = self._ensure_dev_batch(stride)[self._slot]
        nl = ctypes.c_int64(0)
        _lib.check(self.lib.kge_sampling_attach(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), self.batch_size,
                                                self.negative_ent, self.negative_rel, self._thread_lo, self._thread_hi, stride,
                                                ctypes.byref(nl), self._stream()), self.lib)
        self._prefetched = (buf, nl.value, None)

    def _flush_next_batch(self):
        _lib.check(self.lib.kge_sampling_flush(self._stream()), self.lib)

    def _prefetch_next_batch(self, behind_emit=False):
        """behind_emit=True (the sign-count path): the side stream waits for THIS step's emit kernel only -- the one
        bandwidth-bound kernel of the step, which a concurrent sampler would slow down by what it takes -- and the sampler
        then runs beside the small latency-bound kernels after it (bucketing, segmented sum, apply).  The other slot's last
        readers (the previous step's kernels) are older than that emit kernel."""
        import torch
        main = torch.cuda.current_stream()
        done = None
        if not behind_emit:
            done = torch.cuda.Event()
            done.record(main)                  # the consumers of the OTHER slot (previous step) were enqueued before this
        self._slot ^= 1
        with torch.cuda.stream(self._side_stream):
            if behind_emit:
                rc = _lib.check(self.lib.kge_stream_wait_emit(ctypes.c_void_p(self._side_stream.cuda_stream)), self.lib)
                if rc == _lib.NO_EVENT:            # this step recorded no emit event: wait for everything enqueued so far
                    done = torch.cuda.Event()
                    done.record(main)
            if done is not None:
                self._side_stream.wait_event(done)  # slot reuse: its last reader is older than `done`
            dev, n_pos = self.sample_device(self._slot)
            ev = torch.cuda.Event()
            ev.record(self._side_stream)
        self._prefetched = (dev, n_pos, ev)

    def forward_backward(self, dev_batch, n_pos, stride, denom, sampler_shaped=False):
        """Add dLoss/dTables of the batch into the gradient accumulators; loss -> self._loss.
        sampler_shaped=True (a device-sampled batch): paths with a separate exact pass for other batches skip it
        (include/kge_mi355.h kge_forward_backward_sampled)."""
        fn = self.lib.kge_forward_backward_sampled if sampler_shaped else self.lib.kge_forward_backward
        _lib.check(fn(ctypes.byref(self._desc), self._tab_ptrs,
                                                 dev_batch[0].data_ptr(), dev_batch[1].data_ptr(),
                                                 dev_batch[2].data_ptr(), n_pos,
                                                 self.negative_ent + self.negative_rel, stride, denom,
                                                 self._grad_ptrs, self._loss.data_ptr(), self._stream()), self.lib)

    def _adam_lr_t(self):
        f = np.float32
        return f(f(self.alpha) * np.sqrt(f(1) - self._beta2_power, dtype=np.float32) / (f(1) - self._beta1_power))

    def _adam_advance(self):
        f = np.float32
        self._beta1_power = f(self._beta1_power * f(self.adam_beta1))
        self._beta2_power = f(self._beta2_power * f(self.adam_beta2))

    def apply_gradients(self, own=False, piece=0):
        """GradientDescentOptimizer / AdamOptimizer on the summed gradients (distribute_training.py:95-101).
        own=True (data-parallel): on this rank's chunk of the flat parameter buffer only, from its reduce-scattered
        gradient chunk."""
        st = self._stream()
        if own:       # one piece of this rank's share (train_step's exchange loop calls it per piece and advances Adam once)
            lo, hi = self._piece_own[piece]
            n = hi - lo
            if hi > self._loss_tail:
                n -= 4                                        # the spare tail slot (it holds the loss, not a parameter)
            p, g = self._flat_p.data_ptr() + 4 * lo, self._grads_own.data_ptr() + 4 * piece * self._own_len
            if self._adam:
                _lib.check(self.lib.kge_adam_update(p, self._flat_m.data_ptr() + 4 * lo, self._flat_v.data_ptr() + 4 * lo, g, n,
                                                    float(self._adam_lr_t()), self.adam_beta1, self.adam_beta2,
                                                    self.adam_epsilon, st), self.lib)
                self._opt_state_synced = False
            else:
                _lib.check(self.lib.kge_sgd_update(p, g, n, float(self.alpha), st), self.lib)
            return
        elif self._adam:
            _lib.check(self.lib.kge_adam_update_tables(len(self._tables), self._tab_ptrs, self._adam_m_ptrs, self._adam_v_ptrs,
                                                       self._grad_ptrs, self._numel, float(self._adam_lr_t()), self.adam_beta1,
                                                       self.adam_beta2, self.adam_epsilon, st), self.lib)
            self._adam_advance()
        else:
            _lib.check(self.lib.kge_sgd_update_tables(len(self._tables), self._tab_ptrs, self._grad_ptrs, self._numel,
                                                      float(self.alpha), st), self.lib)
        self.global_step += 1

    def _dp_exchange(self, image, own_image, apply_piece, counts):
        """The data-parallel part of a dense step: reduce-scatter of the gradient image, the optimizer on the owned share,
        all-gather of the updated parameters -- piece by piece (see _setup_flat_buffers), every collective asynchronous, so
        that piece k is updated and sent while piece k+1 is still being summed.  The loss needs no collective of its own: it
        is added into a spare tail slot of the image (TransE's int32 count image: as four 16-bit limbs of a 2^-32 fixed-point
        value, kge_loss_to_limbs), summed by the reduce-scatter like everything else, written by its owner -- the last rank --
        into the spare tail slot of the parameter buffer, and reaches every rank with the all-gather."""
        from .parallel import reduce_scatter_sum, all_gather_chunks
        st, W, K = self._stream(), self.world_size, self._pieces
        flat_img, flat_own = image.view(-1), own_image.view(-1)
        per_seg, per_own = flat_img.numel() // K, flat_own.numel() // K
        ride = flat_img.numel() >= 4 and (not counts or self.hidden_size >= 4)
        last = self.rank == W - 1
        nat = self._stream_rccl() if (K == 1 and ride) else None
        self.comm_fence("nat" if nat is not None else "pg")
        if nat is not None:
            # one piece (tables below 64 MB): the two collectives go onto the engine's own stream, in order with its kernels
            # (parallel.StreamRccl) -- no process-group stream, no event hand-offs
            if counts and not getattr(self, "_limbs_from_emit", False):
                _lib.check(self.lib.kge_loss_to_limbs(self._loss.data_ptr(), flat_img[flat_img.numel() - self.hidden_size:].data_ptr(), st), self.lib)
            elif not counts:
                flat_img[-1:].copy_(self._loss)
            nat.reduce_scatter_sum(flat_own, flat_img, st)
            if last:
                if counts:
                    _lib.check(self.lib.kge_limbs_to_loss(flat_own[flat_own.numel() - self.hidden_size:].data_ptr(), self._loss.data_ptr(), st), self.lib)
                else:
                    self._loss.copy_(flat_own[-1:])
            apply_piece(0)
            (slo, shi), (lo, hi) = self._piece_seg[0], self._piece_own[0]
            nat.all_gather_chunks(self._flat_p[slo:shi], self._flat_p[lo:hi], st)
            image.zero_()
            if self._adam:
                self._adam_advance()
            self.global_step += 1
            return
        if ride:
            if counts and not getattr(self, "_limbs_from_emit", False):     # (a sampled batch: the emit kernel has written the limbs itself)
                _lib.check(self.lib.kge_loss_to_limbs(self._loss.data_ptr(), flat_img[flat_img.numel() - self.hidden_size:].data_ptr(), st), self.lib)
            else:
                flat_img[-1:].copy_(self._loss)
        rs = [reduce_scatter_sum(flat_own[k * per_own:(k + 1) * per_own], flat_img[k * per_seg:(k + 1) * per_seg], self._pg, async_op=True)
              for k in range(K)]
        if not ride:
            from .parallel import allreduce_sum
            allreduce_sum([self._loss], self._pg)
        ag = []
        for k in range(K):
            if rs[k] is not None:
                rs[k].wait()
            if ride and last and k == K - 1:      # the summed loss -> the parameter buffer's tail slot (before the all-gather of this piece)
                if counts:
                    _lib.check(self.lib.kge_limbs_to_loss(flat_own[flat_own.numel() - self.hidden_size:].data_ptr(),
                                                          self._flat_p[self._loss_tail:].data_ptr(), st), self.lib)
                else:
                    self._loss.copy_(flat_own[-1:])
            apply_piece(k)
            (slo, shi), (lo, hi) = self._piece_seg[k], self._piece_own[k]
            ag.append(all_gather_chunks(self._flat_p[slo:shi], self._flat_p[lo:hi], self._pg, async_op=True))
        image.zero_()
        if self._adam:
            self._adam_advance()
        self.global_step += 1
        for w in ag:
            if w is not None:
                w.wait()
        # (ride: self._loss IS the tail slot of the parameter buffer -- the all-gather has just delivered the global loss into it)

    def comm_fence(self, kind):
        """Two RCCL communicators live in a data-parallel process: torch's process group ("pg": the sharded step, optimizer-state
        gathers, evaluation reductions, the caller's own barriers) and the engine's own (parallel.StreamRccl, "nat": the dense
        step's reduce-scatter / all-gather on the engine's stream).  Collectives of two communicators must never be in flight
        together on a device: each spins on its peers, and two ranks that get them scheduled in opposite order wait on each other
        for ever.  The rule enforced here: every collective is issued through ONE communicator per step path, and whenever the
        communicator CHANGES (first step after a checkpoint gather, an evaluation, ...) the device is drained first -- every
        collective this rank has enqueued so far has then completed, so its peers have all entered it, before a collective of
        the other communicator is enqueued.  Every rank runs the same program, so all ranks fence at the same points.  Call it with
        "pg" before using torch.distributed collectives of your own between training steps."""
        last = getattr(self, "_last_comm", None)
        if last is not None and last != kind:
            import torch
            torch.cuda.synchronize(self.device)
        self._last_comm = kind

    def _stream_rccl(self):
        """The communicator for collectives on the engine's stream (parallel.StreamRccl): "nccl" backend, `stream_rccl` not switched
        off (KGE_STREAM_RCCL=0 switches it off from the environment).  Created on first use -- collectively: every rank reaches the
        first data-parallel step together -- and PROBED before it is trusted: a reduce-scatter and an all-gather of known values
        through it must give the known sums on every rank, else (or on any error) all ranks fall back to the process group."""
        if getattr(self, "_nat_rccl", None) is None:
            import torch.distributed as dist
            want = getattr(self, "stream_rccl", os.environ.get("KGE_STREAM_RCCL", "1") != "0") and dist.is_initialized() and \
                dist.get_backend(self._pg) == "nccl"
            nat = False
            if want:
                import sys
                import torch
                from .parallel import StreamRccl
                self.comm_fence("pg")
                good = 0
                try:
                    nat = StreamRccl(self._pg)
                    torch.cuda.synchronize(self.device)
                    W, g, st = nat.world, nat.rank, self._stream()
                    src = torch.arange(4 * W, dtype=torch.int32, device=self.device) + g          # rank g contributes i + g
                    own = torch.empty(4, dtype=torch.int32, device=self.device)
                    nat.reduce_scatter_sum(own, src, st)
                    full = torch.empty(4 * W, dtype=torch.int32, device=self.device)
                    full[4 * g:4 * g + 4] = own
                    nat.all_gather_chunks(full, full[4 * g:4 * g + 4], st)
                    torch.cuda.synchronize(self.device)
                    want_full = W * torch.arange(4 * W, dtype=torch.int32) + W * (W - 1) // 2      # sum_g (i + g)
                    good = int(torch.equal(full.cpu(), want_full))
                    if not good:
                        print("StreamRccl probe gave wrong sums: collectives stay on the process group's stream", file=sys.stderr)
                except Exception as exc:       # still RCCL either way: the process-group path is the documented alternative
                    print("StreamRccl unavailable (%s): collectives stay on the process group's stream" % exc, file=sys.stderr)
                # every rank must take the same path (a rank on the other communicator would wait for ever): agree on the minimum
                ok = torch.tensor([good], dtype=torch.int32, device=self.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self._pg)
                if int(ok.item()) == 0:
                    if nat:
                        nat.close()
                    nat = False
            self._nat_rccl = nat
        return self._nat_rccl or None

    def sync_optimizer_state(self):
        """Data-parallel Adam keeps m and v current on their owner only; gather them before they are read as whole tables
        (checkpoints)."""
        if self._dp and self._adam and not self.sparse_rows and not getattr(self, "_opt_state_synced", True):
            from .parallel import all_gather_chunks
            self.comm_fence("pg")
            for (slo, shi), (lo, hi) in zip(self._piece_seg, self._piece_own):
                all_gather_chunks(self._flat_m[slo:shi], self._flat_m[lo:hi], self._pg)
                all_gather_chunks(self._flat_v[slo:shi], self._flat_v[lo:hi], self._pg)
            self._opt_state_synced = True

    def train_step(self, batch_h=None, batch_t=None, batch_r=None, batch_y=None, sync=True):
        '''
        Perform a single training step (Config.py:464-475 / distribute_training.py:274-282).
        With no arguments the batch is sampled on the device; with the reference's four arrays the
        given batch (layout of Config.batch_h/t/r) is trained on.  Returns the loss (float) when
        sync=True, else the device scalar.
        '''
        import torch
        n_neg = self.negative_ent + self.negative_rel
        if getattr(self, "_ent_is_shard", False) and not hasattr(self, "_shard"):
            raise KgeError("the entity table was created as this rank's shard of a row-sharded table (the process belongs to a "
                           "torch.distributed world): call init_distributed() before training")
        if batch_h is None:
            dev, n_pos = self._next_sampled_batch() if self.prefetch_sampling else self.sample_device()
            stride = max(self._n_local, 1)
        else:
            if self._dp:      # (also a one-rank group under force_data_parallel: the sharded step has no check for hand-made negatives)
                raise KgeError("feeding a host batch is single-process only")
            host = np.stack([np.asarray(batch_h), np.asarray(batch_t), np.asarray(batch_r)]).astype(np.int32)
            self._check_ids(host)
            dev = torch.from_numpy(host).to(self.device)
            n_pos = host.shape[1] // (1 + n_neg)
            stride = n_pos
        denom = self.batch_size * n_neg if batch_h is None else n_pos * n_neg
        # small steps are launch-bound: the single fused atomic kernel beats the multi-stage count pipeline
        # (same decision on every rank: it depends on the global batch only)
        big = (self.batch_size if batch_h is None else n_pos) * (3 + n_neg) >= self.counts_min_records * self.world_size
        if self.sparse_inplace and self._dp:
            self._records_step(dev, n_pos, stride, denom)
            if batch_h is None and self.prefetch_sampling:
                self._prefetch_next_batch()
        elif self.sparse_inplace:
            _lib.check(self.lib.kge_forward_backward_sgd_rows(
                ctypes.byref(self._desc), self._tab_ptrs, dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), n_pos, n_neg,
                stride, denom, float(self.alpha), self._loss.data_ptr(), self._stream()), self.lib)
            if batch_h is not None:       # a hand-fed batch may hold negatives the in-place update cannot take (the sampler draws none)
                skipped = ctypes.c_int32(0)
                _lib.check(self.lib.kge_sgd_rows_skipped(ctypes.byref(skipped)), self.lib)
                if skipped.value:
                    raise KgeError("sparse_rows: %d negatives are not single-slot corruptions of their positive and were left out "
                                   "of this step; train such batches with sparse_rows=False" % skipped.value)
            self.global_step += 1
            if batch_h is None and self.prefetch_sampling:
                self._prefetch_next_batch()
        elif self.sparse_rows:
            if self._dp:
                self._sharded_step(dev, n_pos, stride, denom)
            else:
                self._sparse_step(dev, n_pos, stride, denom, check_shape=batch_h is not None)
            if batch_h is None and self.prefetch_sampling:
                self._prefetch_next_batch()
        elif self.use_counts and big:
            ahead = batch_h is None and self.prefetch_sampling
            if ahead:
                self._attach_next_batch()              # rides in the scatter launch enqueued by forward_counts
            # data-parallel: the emit kernel of a sampled batch also writes the loss as limbs into the count image's spare tail slot
            self._limbs_from_emit = bool(self._dp and batch_h is None and self.hidden_size >= 4)
            target = self._counts.view(-1)[self._counts.numel() - self.hidden_size:].data_ptr() if self._limbs_from_emit else 0
            if target != getattr(self, "_limbs_target", 0):
                self.lib.kge_loss_limbs_target(ctypes.c_void_p(target) if target else None)
                self._limbs_target = target
            # one process: forward, segmented sum and optimizer in ONE call (kge_transe_train_step_counts) -- the rows whose records one
            # team can hold are summed in registers and updated there, without the count image; `fused_counts = False` keeps the
            # two-call form (the data-parallel step needs the image: it is what the ranks exchange)
            one_call = not self._dp and bool(getattr(self, "fused_counts", True))
            try:
                if one_call:
                    self.step_counts(dev, n_pos, stride, denom, sampler_shaped=batch_h is None)
                else:
                    self.forward_counts(dev, n_pos, stride, denom, sampler_shaped=batch_h is None)
            finally:
                if ahead:                              # launched on its own if the step's path had no scatter kernel -- and
                    self._flush_next_batch()           # also when the forward call failed: an armed sampler never outlives its buffers
            if one_call:
                pass
            elif self._dp:
                # int32 SUM is exact: rank g receives the summed counts of ITS rows, updates them, and the updated rows go round
                self._dp_exchange(self._counts, self._counts_own, lambda k: self.apply_counts(denom, own=True, piece=k), counts=True)
                self.tables_changed()
            else:
                self.apply_counts(denom)
        else:
            # one GPU: the next batch's sampler rides in a launch of the step that leaves most of the chip idle -- TransR's relation
            # scatter, the record paths' bucket scatter, the atomic path's forward/backward kernel itself (a few hundred
            # latency-bound workgroups at the reference's batch sizes); a path without such a launch runs it on its own afterwards
            ride = batch_h is None and self.prefetch_sampling and not self._dp
            if ride:
                self._attach_next_batch()
            try:
                self.forward_backward(dev, n_pos, stride, denom, sampler_shaped=batch_h is None)
            finally:
                if ride:
                    self._flush_next_batch()
            if batch_h is None and self.prefetch_sampling and not ride:
                self._prefetch_next_batch(behind_emit=bool(self.lib.kge_pair_path_active(ctypes.byref(self._desc), n_pos, n_neg)))
            if self._dp:
                self._dp_exchange(self._flat_g, self._grads_own, lambda k: self.apply_gradients(own=True, piece=k), counts=False)
            else:
                self.apply_gradients()
        self.trainModel.loss = self._loss
        return float(self._loss.item()) if sync else self._loss

    def persistent_supported(self):
        """Can train_steps() run its steps inside one persistent launch (csrc/persist.hip)?  Single process, the dense
        fp32-accumulator path of TransE / TransH / TransD at a launch-latency-bound step size."""
        if self._dp or self.sparse_rows or self.sparse_inplace or self.hidden_size > 256:
            return False
        if self.trainModel.model_id not in (_lib.TRANSE, _lib.TRANSH, _lib.TRANSD):
            return False
        n_neg = self.negative_ent + self.negative_rel
        if self.use_counts and self.batch_size * (3 + n_neg) >= self.counts_min_records:
            return False       # TransE at a large step: the exact integer-count pipeline is the faster path
        slots = {_lib.TRANSE: 3 + n_neg, _lib.TRANSH: 4 + n_neg, _lib.TRANSD: 6 + 2 * n_neg}[self.trainModel.model_id]
        return self.batch_size * slots < int(getattr(self, "persistent_max_rows", 1 << 16))

    def persistent_preferred(self):
        """Is the persistent launch the FASTER way to run this configuration?  Measured (profiles/r03_a_other_configs.jsonl and the
        bench line's config-#1 leg): TransE at its auto batch 26-27 us/step against 27-29 as separate launches (round 2: 27.8
        against 35.2 -- the sampler now rides in the forward/backward launch and the host side of a step is cheaper); TransH /
        TransD steps are bound by the fp32-atomic rate of their gradient rows either way and the separate launches are quicker
        (66-70 vs 77-80 us at config #3's batch)."""
        return self.persistent_supported() and self.trainModel.model_id == _lib.TRANSE

    def train_steps(self, n_steps, persistent=None):
        """`n_steps` iterations of the training loop body (distribute_training.py:267-283): sample, forward / backward,
        update.  Returns the losses (numpy float32 [n_steps]).  Where persistent_preferred() (or with persistent=True where
        persistent_supported()), all the steps run inside ONE persistent launch; otherwise as n_steps calls of train_step()."""
        import torch
        n_steps = int(n_steps)
        if n_steps <= 0:
            return np.zeros(0, np.float32)
        use = self.persistent_preferred() if persistent is None else bool(persistent)
        if use and not self.persistent_supported():
            raise KgeError("train_steps(persistent=True): this configuration has no persistent-launch path")
        if not use:
            out = [self.train_step(sync=False).clone() for _ in range(n_steps)]   # (the loss scalar is one reused device buffer)
            return torch.stack([o.reshape(()) for o in out]).cpu().numpy()
        if self._prefetched is not None:
            raise KgeError("train_steps: a batch was sampled ahead by train_step(); use one or the other in a run")
        f = np.float32
        lr = np.empty(n_steps, np.float32)
        powers = None
        if self._adam:
            saved = (self._beta1_power, self._beta2_power)
            for i in range(n_steps):
                lr[i] = self._adam_lr_t()
                self._adam_advance()
            powers = (self._beta1_power, self._beta2_power)
            self._beta1_power, self._beta2_power = saved       # committed below, once the launch is known to have completed
        else:
            lr[:] = f(self.alpha)
        losses = torch.empty(n_steps, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.kge_train_steps_persistent(
            ctypes.byref(self._desc), self._tab_ptrs, self._grad_ptrs, self._adam_m_ptrs if self._adam else None,
            self._adam_v_ptrs if self._adam else None, self.batch_size, self.negative_ent, self.negative_rel, n_steps,
            1 if self._adam else 0, lr.ctypes.data, self.adam_beta1, self.adam_beta2, self.adam_epsilon, losses.data_ptr(),
            self._stream()), self.lib)
        out = losses.cpu().numpy()          # synchronises: the launch has drained
        flag = ctypes.c_int32(0)
        _lib.check(self.lib.kge_persistent_aborted(ctypes.byref(flag)), self.lib)
        if flag.value:
            raise KgeError("persistent training launch aborted: a grid barrier did not complete (is another process holding "
                           "compute units of this GPU?); the tables are in an intermediate state")
        if powers is not None:
            self._beta1_power, self._beta2_power = powers
        self.global_step += n_steps
        self._loss.copy_(losses[-1:])
        self.trainModel.loss = self._loss
        return out

    def _records_step(self, dev_batch, n_pos, stride, denom):
        """Row-wise SGD in place across ranks (TransH / TransD, or TransE off the sign-count path, with sparse_rows): the tables
        are replicated; every rank turns ITS slice of the batch into float gradient records, the records (rows + destination
        keys) are all-gathered -- the sparse touched-row exchange: only rows a step touches travel -- and every rank adds -lr * the
        per-row sums of ALL records to its replica.  Every rank reduces the same records in the same order, so the replicas stay
        bit-identical; against the single-process step the per-row sums differ in fp32 order only.  Replaces the scatter_sub
        updates the reference's workers send to its parameter servers (distribute_training.py:99-101,193-196)."""
        import torch
        from .parallel import all_gather_chunks, allreduce_sum, max_slice_positions
        n_neg = self.negative_ent + self.negative_rel
        W, D = self.world_size, self.hidden_size
        slots = {_lib.TRANSE: 3 + n_neg, _lib.TRANSH: 4 + n_neg, _lib.TRANSD: 6 + 2 * n_neg}[self.trainModel.model_id]
        b = getattr(self, "_rec_buf", None)
        if b is None or b["slots"] != slots:
            per = max_slice_positions(self.lib, self.batch_size, W, self.workThreads) * slots     # records per rank's slice
            b = dict(slots=slots, per=per,
                     rec=torch.zeros((W * per, D), dtype=torch.float32, device=self.device),
                     dst=torch.full((W * per,), -1, dtype=torch.int32, device=self.device))
            self._rec_buf = b
        per, off = b["per"], self.rank * b["per"]
        _lib.check(self.lib.kge_forward_backward_records(
            ctypes.byref(self._desc), self._tab_ptrs, dev_batch[0].data_ptr(), dev_batch[1].data_ptr(), dev_batch[2].data_ptr(),
            n_pos, n_neg, stride, denom, self.batch_size, b["rec"].data_ptr(), b["dst"].data_ptr(), off, per,
            self._loss.data_ptr(), self._stream()), self.lib)
        if W > 1:
            self.comm_fence("pg")
            all_gather_chunks(b["rec"].view(-1), b["rec"][off:off + per].view(-1), self._pg)
            all_gather_chunks(b["dst"], b["dst"][off:off + per], self._pg)
            allreduce_sum([self._loss], self._pg)
        _lib.check(self.lib.kge_float_records_apply(
            ctypes.byref(self._desc), self._tab_ptrs, b["rec"].data_ptr(), b["dst"].data_ptr(), W * per, self.batch_size, n_neg,
            float(self.alpha), self._stream()), self.lib)
        self.global_step += 1

    def _sparse_step(self, dev_batch, n_pos, stride, denom, check_shape=False):
        """Sparse-row TransE step on ONE GPU: emit int8 records -> sort by destination row -> compact per-row counts ->
        SGD on the touched rows.  Bitwise the same update as the dense count image (integer sums).  N GPUs:
        _sharded_step."""
        import torch
        n_neg = self.negative_ent + self.negative_rel
        D = self.hidden_size
        m_local = stride * (3 + n_neg)
        dw = int(self.lib.kge_transe_record_dwords(ctypes.byref(self._desc)))
        buf = self._sparse_buf
        if buf is None or buf.get("m_local") != m_local:
            dev = self.device
            buf = dict(m_local=m_local,
                       rec=torch.empty((m_local, dw), dtype=torch.int32, device=dev),
                       dst=torch.empty(m_local, dtype=torch.int32, device=dev),
                       rows=torch.empty(m_local, dtype=torch.int32, device=dev),
                       row_counts=torch.empty((m_local, D), dtype=torch.int32, device=dev),
                       n_rows=torch.zeros(1, dtype=torch.int32, device=dev))
            self._sparse_buf = buf
        st = self._stream()
        buf["dst"].fill_(-1)
        _lib.check(self.lib.kge_transe_emit_records(
            ctypes.byref(self._desc), self._tables[0].data_ptr(), self._tables[1].data_ptr(),
            dev_batch[0].data_ptr(), dev_batch[1].data_ptr(), dev_batch[2].data_ptr(), n_pos, n_neg,
            dev_batch.shape[1] // (1 + n_neg), denom, buf["rec"].data_ptr(), buf["dst"].data_ptr(), None, None,
            self._loss.data_ptr(), st), self.lib)
        if check_shape:
            nd = ctypes.c_int32(0)
            _lib.check(self.lib.kge_transe_deferred_groups(ctypes.byref(nd)), self.lib)
            if nd.value:
                raise KgeError("sparse_rows: %d groups have negatives that are not single-slot corruptions of their "
                               "positive; train such batches with sparse_rows=False" % nd.value)
        rec, dst = buf["rec"], buf["dst"]
        if self._lazy_adam:
            _lib.check(self.lib.kge_transe_reduce_records(
                ctypes.byref(self._desc), rec.data_ptr(), dst.data_ptr(), dst.numel(), buf["rows"].data_ptr(),
                buf["row_counts"].data_ptr(), buf["n_rows"].data_ptr(), st), self.lib)
            _lib.check(self.lib.kge_transe_apply_rows_adam_lazy(
                ctypes.byref(self._desc), self._tables[0].data_ptr(), self._tables[1].data_ptr(), self._adam_m[0].data_ptr(),
                self._adam_m[1].data_ptr(), self._adam_v[0].data_ptr(), self._adam_v[1].data_ptr(), buf["rows"].data_ptr(),
                buf["row_counts"].data_ptr(), buf["n_rows"].data_ptr(), dst.numel(), denom, float(self._adam_lr_t()),
                self.adam_beta1, self.adam_beta2, self.adam_epsilon, st), self.lib)
            self._adam_advance()
        elif getattr(self, "sparse_fused", True) and D % 4 == 0:
            # reduce + apply in one pass: only chunk-boundary rows go through the compact count image
            _lib.check(self.lib.kge_transe_reduce_apply_records_sgd(
                ctypes.byref(self._desc), rec.data_ptr(), dst.data_ptr(), dst.numel(), self._tables[0].data_ptr(),
                self._tables[1].data_ptr(), buf["rows"].data_ptr(), buf["row_counts"].data_ptr(), buf["n_rows"].data_ptr(),
                denom, float(self.alpha), st), self.lib)
        else:
            _lib.check(self.lib.kge_transe_reduce_records(
                ctypes.byref(self._desc), rec.data_ptr(), dst.data_ptr(), dst.numel(), buf["rows"].data_ptr(),
                buf["row_counts"].data_ptr(), buf["n_rows"].data_ptr(), st), self.lib)
            _lib.check(self.lib.kge_transe_apply_rows_sgd(
                ctypes.byref(self._desc), self._tables[0].data_ptr(), self._tables[1].data_ptr(), buf["rows"].data_ptr(),
                buf["row_counts"].data_ptr(), buf["n_rows"].data_ptr(), dst.numel(), denom, float(self.alpha), st),
                self.lib)
        self.global_step += 1

    def _desc_with(self, **fields):
        """A copy of the model descriptor with some fields replaced (row spaces of a cache / a shard)."""
        d = self._desc
        vals = {name: getattr(d, name) for name, _ in d._fields_}
        vals.update(fields)
        return _lib.ModelDesc(*[vals[name] for name, _ in d._fields_])

    def _shard_buffers(self, M, n_recv_rows, n_recv_rec):
        """Workspace of the sharded step, grown geometrically (receive sizes vary from step to step)."""
        import torch
        D, W, dev = self.hidden_size, self.world_size, self.device
        dw = int(self.lib.kge_transe_record_dwords(ctypes.byref(self._desc)))
        b = self._sparse_buf if isinstance(self._sparse_buf, dict) and self._sparse_buf.get("sharded") else dict(sharded=True, M=0, rr=0, rc=0)
        i32 = torch.int32
        if M > b["M"] or "req" not in b:
            cap = max(int(M * 1.25), 64)
            b.update(M=cap, req=torch.empty(cap, dtype=i32, device=dev), slot_of=torch.empty(cap, dtype=i32, device=dev),
                     send_ids=torch.empty(cap, dtype=i32, device=dev), cache=torch.empty((cap, D), dtype=torch.float32, device=dev),
                     rec=torch.empty((cap, dw), dtype=i32, device=dev), dst=torch.empty(cap, dtype=i32, device=dev),
                     ids2=torch.empty(cap, dtype=i32, device=dev), slot_of2=torch.empty(cap, dtype=i32, device=dev),
                     send_rows2=torch.empty(cap, dtype=i32, device=dev), send_rec=torch.empty((cap, dw), dtype=i32, device=dev),
                     counts=torch.zeros(W, dtype=i32, device=dev), cursor=torch.zeros(W, dtype=i32, device=dev),
                     rel_counts=torch.zeros((self.relTotal, D), dtype=i32, device=dev),
                     rel_rows=torch.arange(self.entTotal, self.entTotal + self.relTotal, dtype=i32, device=dev),
                     n_rel=torch.full((1,), self.relTotal, dtype=i32, device=dev), n_rows=torch.zeros(1, dtype=i32, device=dev), dw=dw)
        if n_recv_rows > b["rr"]:
            cap = max(int(n_recv_rows * 1.25), 64)
            b.update(rr=cap, recv_ids=torch.empty(cap, dtype=i32, device=dev), rows_out=torch.empty((cap, D), dtype=torch.float32, device=dev))
        if n_recv_rec > b["rc"]:
            cap = max(int(n_recv_rec * 1.25), 64)
            b.update(rc=cap, recv_rows2=torch.empty(cap, dtype=i32, device=dev), recv_rec=torch.empty((cap, dw), dtype=i32, device=dev),
                     rows=torch.empty(cap, dtype=i32, device=dev), row_counts=torch.empty((cap, D), dtype=i32, device=dev))
        self._sparse_buf = b
        return b

    def _sharded_step(self, dev_batch, n_pos, stride, denom):
        """Sparse-row TransE step on a SHARDED entity table (world_size > 1): every stage is O(local batch).
        request ids -> all-to-all -> owners gather rows -> all-to-all -> emit int8 records against the fetched rows ->
        all-to-all (row id, record) -> owners sort / sum / apply their rows; relation counts all-reduced (csrc/shard.hip).
        Integer sums and one per-row update formula: the union of the shards equals the single-process table bit for bit."""
        import torch
        from . import parallel as par
        L, st, W, D, pg = self.lib, self._stream(), self.world_size, self.hidden_size, self._pg
        self.comm_fence("pg")
        n_neg = self.negative_ent + self.negative_rel
        sh = self._shard
        chunk = sh["chunk"]
        M = n_pos * (3 + n_neg)
        b = self._shard_buffers(M, 0, 0)
        dw = b["dw"]
        h, t, r = dev_batch[0], dev_batch[1], dev_batch[2]
        bstride = dev_batch.shape[1] // (1 + n_neg)
        # 1. which entity rows does this slice touch, and who owns them
        _lib.check(L.kge_shard_requests(h.data_ptr(), t.data_ptr(), r.data_ptr(), n_pos, n_neg, bstride, b["req"].data_ptr(), st), L)
        _lib.check(L.kge_shard_count(b["req"].data_ptr(), M, chunk, W, b["counts"].data_ptr(), st), L)
        send, recv, gmax = par.exchange_counts_max(b["counts"], pg)
        h_counts = (ctypes.c_int64 * W)(*send)
        _lib.check(L.kge_shard_scatter(b["req"].data_ptr(), M, chunk, W, h_counts, b["cursor"].data_ptr(), b["send_ids"].data_ptr(),
                                       b["slot_of"].data_ptr(), st), L)
        n_send, n_recv = sum(send), sum(recv)
        b = self._shard_buffers(M, n_recv, 0)
        # 2. ids out, rows back
        par.all_to_all_rows(b["recv_ids"], b["send_ids"], recv, send, pg, max_rows=gmax)
        _lib.check(L.kge_shard_gather_rows(self._tables[0].data_ptr(), b["recv_ids"].data_ptr(), n_recv, sh["lo"], chunk, D,
                                           b["rows_out"].data_ptr(), st), L)
        par.all_to_all_rows(b["cache"], b["rows_out"], send, recv, pg, max_rows=gmax)
        # 3. the unchanged emit kernel over the fetched rows: the batch in terms of cache slots
        if self._dev_batch2 is None or self._dev_batch2.shape != dev_batch.shape:
            self._dev_batch2 = torch.zeros_like(dev_batch)
        h2, t2 = self._dev_batch2[0], self._dev_batch2[1]
        _lib.check(L.kge_shard_remap_batch(h.data_ptr(), t.data_ptr(), n_pos, n_neg, bstride, b["slot_of"].data_ptr(), h2.data_ptr(),
                                           t2.data_ptr(), st), L)
        desc2 = self._desc_with(ent_total=max(n_send, 1))
        _lib.check(L.kge_transe_emit_records(ctypes.byref(desc2), b["cache"].data_ptr(), self._tables[1].data_ptr(), h2.data_ptr(),
                                             t2.data_ptr(), r.data_ptr(), n_pos, n_neg, bstride, denom, b["rec"].data_ptr(),
                                             b["dst"].data_ptr(), None, None, self._loss.data_ptr(), st), L)
        # 4. relation rows: dense int32 image, all-reduced (the relation table is replicated)
        b["rel_counts"].zero_()
        if self._lazy_adam:
            # which relations have a record on THIS rank -- the relation-slot destinations of its active groups (before the reduce below
            # rewrites the destination keys in place); summed over the ranks with the counts.  Entry R collects the inactive groups.
            if "rel_live" not in b:
                b["rel_live"] = torch.zeros(self.relTotal + 1, dtype=torch.int32, device=self.device)
            b["rel_live"].zero_()
            if n_pos > 0:
                d = b["dst"][2 * n_pos:3 * n_pos]
                idx = torch.where(d >= 0, d - int(desc2.ent_total), torch.full_like(d, self.relTotal))
                b["rel_live"].index_fill_(0, idx.long(), 1)
        if D % 4 == 0 and self.negative_rel == 0 and n_pos > 0:
            # the relation-side records are slot 2 of every group -- records [2 n_pos, 3 n_pos): ordered by relation, summed by
            # segments into <= R compact rows, scattered into the image (per-element atomics took 0.4 ms of a 3 ms step)
            cap = min(n_pos, self.relTotal) + 8
            if b.get("relc_cap", 0) < cap:
                b.update(relc_cap=cap, relc_rows=torch.empty(cap, dtype=torch.int32, device=self.device),
                         relc_counts=torch.empty((cap, D), dtype=torch.int32, device=self.device))
            _lib.check(L.kge_transe_reduce_records(ctypes.byref(desc2), b["rec"][2 * n_pos:3 * n_pos].data_ptr(),
                                                   b["dst"][2 * n_pos:3 * n_pos].data_ptr(), n_pos, b["relc_rows"].data_ptr(),
                                                   b["relc_counts"].data_ptr(), b["n_rows"].data_ptr(), st), L)
            _lib.check(L.kge_shard_scatter_count_rows(b["relc_rows"].data_ptr(), b["relc_counts"].data_ptr(), b["n_rows"].data_ptr(),
                                                      cap, desc2.ent_total, self.relTotal, D, b["rel_counts"].data_ptr(), st), L)
        else:
            _lib.check(L.kge_shard_relation_counts(b["rec"].data_ptr(), b["dst"].data_ptr(), M, desc2.ent_total, self.relTotal, dw, D,
                                                   b["rel_counts"].data_ptr(), st), L)
        if self._lazy_adam:
            par.allreduce_sum([b["rel_counts"], self._loss, b["rel_live"]], pg)
        else:
            par.allreduce_sum([b["rel_counts"], self._loss], pg)
        # 5. entity records to their owners
        _lib.check(L.kge_shard_record_ids(b["dst"].data_ptr(), M, n_send, b["send_ids"].data_ptr(), b["ids2"].data_ptr(), st), L)
        _lib.check(L.kge_shard_count(b["ids2"].data_ptr(), M, chunk, W, b["counts"].data_ptr(), st), L)
        send2, recv2, gmax2 = par.exchange_counts_max(b["counts"], pg)
        h_counts2 = (ctypes.c_int64 * W)(*send2)
        _lib.check(L.kge_shard_scatter(b["ids2"].data_ptr(), M, chunk, W, h_counts2, b["cursor"].data_ptr(), b["send_rows2"].data_ptr(),
                                       b["slot_of2"].data_ptr(), st), L)
        _lib.check(L.kge_shard_pack_records(b["rec"].data_ptr(), b["slot_of2"].data_ptr(), M, dw, b["send_rec"].data_ptr(), st), L)
        n_recv2 = sum(recv2)
        b = self._shard_buffers(M, n_recv, n_recv2)
        par.all_to_all_rows(b["recv_rows2"], b["send_rows2"], recv2, send2, pg, max_rows=gmax2)
        par.all_to_all_rows(b["recv_rec"], b["send_rec"], recv2, send2, pg, max_rows=gmax2)
        # 6. owner: sort by row, segmented sum, SGD (or lazy Adam: the shard's own moment rows) on the touched rows of the shard
        if n_recv2 > 0:
            b["recv_rows2"][:n_recv2].sub_(sh["lo"])
            desc3 = self._desc_with(ent_total=chunk, rel_total=0)
            if self._lazy_adam:
                _lib.check(L.kge_transe_reduce_records(ctypes.byref(desc3), b["recv_rec"].data_ptr(), b["recv_rows2"].data_ptr(), n_recv2,
                                                       b["rows"].data_ptr(), b["row_counts"].data_ptr(), b["n_rows"].data_ptr(), st), L)
                _lib.check(L.kge_transe_apply_rows_adam_lazy(
                    ctypes.byref(desc3), self._tables[0].data_ptr(), self._tables[1].data_ptr(), self._adam_m[0].data_ptr(),
                    self._adam_m[1].data_ptr(), self._adam_v[0].data_ptr(), self._adam_v[1].data_ptr(), b["rows"].data_ptr(),
                    b["row_counts"].data_ptr(), b["n_rows"].data_ptr(), n_recv2, denom, float(self._adam_lr_t()), self.adam_beta1,
                    self.adam_beta2, self.adam_epsilon, st), L)
            else:
                _lib.check(L.kge_transe_reduce_apply_records_sgd(
                    ctypes.byref(desc3), b["recv_rec"].data_ptr(), b["recv_rows2"].data_ptr(), n_recv2, self._tables[0].data_ptr(),
                    self._tables[1].data_ptr(), b["rows"].data_ptr(), b["row_counts"].data_ptr(), b["n_rows"].data_ptr(), denom,
                    float(self.alpha), st), L)
        # 7. every rank applies the identical relation update from the all-reduced counts
        if self._lazy_adam:
            # (only the relations SOME rank had a record for: a record-less row keeps its value and moments under the lazy rule)
            _lib.check(L.kge_transe_lazy_row_live(b["rel_live"].data_ptr()), L)
            _lib.check(L.kge_transe_apply_rows_adam_lazy(
                ctypes.byref(self._desc), self._tables[0].data_ptr(), self._tables[1].data_ptr(), self._adam_m[0].data_ptr(),
                self._adam_m[1].data_ptr(), self._adam_v[0].data_ptr(), self._adam_v[1].data_ptr(), b["rel_rows"].data_ptr(),
                b["rel_counts"].data_ptr(), b["n_rel"].data_ptr(), self.relTotal, denom, float(self._adam_lr_t()), self.adam_beta1,
                self.adam_beta2, self.adam_epsilon, st), L)
            self._adam_advance()
        else:
            _lib.check(L.kge_transe_apply_rows_sgd(ctypes.byref(self._desc), self._tables[0].data_ptr(), self._tables[1].data_ptr(),
                                                   b["rel_rows"].data_ptr(), b["rel_counts"].data_ptr(), b["n_rel"].data_ptr(),
                                                   self.relTotal, denom, float(self.alpha), st), L)
        self.global_step += 1

    def sparse_row_gradients(self):
        """(rows int32[n], counts int32[n, D]) of the last sparse step: the touched rows (entity rows first, relation
        rows offset by entTotal) and their integer sign counts (complete only with sparse_fused = False)."""
        n = int(self._sparse_buf["n_rows"].item())
        return self._sparse_buf["rows"][:n], self._sparse_buf["row_counts"][:n]

    def forward_counts(self, dev_batch, n_pos, stride, denom, sampler_shaped=False):
        """TransE sign-count forward/backward: exact int32 gradient counts -> self._counts.
        sampler_shaped=True (a device-sampled batch): no residual pass is needed (include/kge_mi355.h)."""
        resid = (None, None) if sampler_shaped else (self._grads[0].data_ptr(), self._grads[1].data_ptr())
        _lib.check(self.lib.kge_transe_forward_counts(
            ctypes.byref(self._desc), self._tables[0].data_ptr(), self._tables[1].data_ptr(),
            dev_batch[0].data_ptr(), dev_batch[1].data_ptr(), dev_batch[2].data_ptr(), n_pos,
            self.negative_ent + self.negative_rel, stride, denom, self._counts.data_ptr(),
            resid[0], resid[1], self._loss.data_ptr(), self._stream()), self.lib)

    def step_counts(self, dev_batch, n_pos, stride, denom, sampler_shaped=False):
        """TransE sign-count step in one call: forward_counts + apply_counts with the middle fused on the device
        (include/kge_mi355.h kge_transe_train_step_counts; distribute_training.py:95-101,282).  Same bits as the two calls."""
        lr = float(self._adam_lr_t()) if self._adam else float(self.alpha)
        _lib.check(self.lib.kge_transe_train_step_counts(
            ctypes.byref(self._desc), self._tab_ptrs, self._adam_m_ptrs if self._adam else None, self._adam_v_ptrs if self._adam else None,
            dev_batch[0].data_ptr(), dev_batch[1].data_ptr(), dev_batch[2].data_ptr(), n_pos, self.negative_ent + self.negative_rel,
            stride, denom, self._counts.data_ptr(), self._grad_ptrs, 1 if sampler_shaped else 0, 1 if self._adam else 0, lr,
            self.adam_beta1, self.adam_beta2, self.adam_epsilon, self._loss.data_ptr(), self._stream()), self.lib)
        if self._adam:
            self._adam_advance()
        self.global_step += 1

    def apply_counts(self, denom, own=False, piece=0):
        """Normalise-backward on the summed counts + SGD / TF1 Adam, both tables in one launch
        (distribute_training.py:95-101).  own=True (data-parallel): this rank's rows of the [(E+R), D] row space only,
        from its reduce-scattered chunk of the count image."""
        st = self._stream()
        lr = float(self._adam_lr_t()) if self._adam else float(self.alpha)
        m_ptrs = self._adam_m_ptrs if self._adam else None
        v_ptrs = self._adam_v_ptrs if self._adam else None
        if own:       # one piece of this rank's rows (the exchange loop advances Adam and the step counter once per step)
            row_lo, row_hi = self._piece_rows[piece]
            if row_hi > row_lo:
                img = self._counts_own.data_ptr() + 4 * piece * self._own_rows_len * self.hidden_size
                _lib.check(self.lib.kge_transe_apply_counts_range(
                    ctypes.byref(self._desc), self._tab_ptrs, m_ptrs, v_ptrs, img, self._grad_ptrs,
                    row_lo, row_hi, denom, 1 if self._adam else 0, lr, self.adam_beta1, self.adam_beta2,
                    self.adam_epsilon, st), self.lib)
            self._opt_state_synced = not self._adam
            return
        else:
            _lib.check(self.lib.kge_transe_apply_counts_tables(
                ctypes.byref(self._desc), self._tab_ptrs, m_ptrs, v_ptrs, self._counts.data_ptr(), self._grad_ptrs, denom,
                1 if self._adam else 0, lr, self.adam_beta1, self.adam_beta2, self.adam_epsilon, st), self.lib)
        if self._adam:
            self._adam_advance()
        self.global_step += 1

    def _check_ids(self, host):
        """Caller-supplied ids index device tables directly: an id out of range must fail here, not fault on the GPU."""
        if host.shape[1] == 0:
            return
        if host.min() < 0 or host[:2].max() >= self.entTotal or host[2].max() >= self.relTotal:
            raise KgeError("entity / relation id out of range in the supplied batch")

    def test_step(self, test_h, test_t, test_r):
        '''
        Score triples with the model's predict op (Config.py:478-488)
        '''
        import torch
        host = np.stack([np.asarray(test_h), np.asarray(test_t), np.asarray(test_r)]).astype(np.int32)
        self._check_ids(host)
        dev = torch.from_numpy(host).to(self.device)
        out = torch.empty(host.shape[1], dtype=torch.float32, device=self.device)
        _lib.check(self.lib.kge_predict(ctypes.byref(self._desc), self._tab_ptrs, dev[0].data_ptr(), dev[1].data_ptr(),
                                        dev[2].data_ptr(), host.shape[1], out.data_ptr(), self._stream()), self.lib)
        self.trainModel.predict = out
        return out.cpu().numpy()

    # ------------------------------------------------------------------------------------------
    # triple classification and the predict_* helpers (Config.py:83-151, 491-516, 574-663)
    # ------------------------------------------------------------------------------------------
    def _tc_buffers(self, prefix, n):
        for side in ("pos", "neg"):
            for col in ("h", "t", "r"):
                arr = np.zeros(n, dtype=np.int64)
                setattr(self, "%s_%s_%s" % (prefix, side, col), arr)
                setattr(self, "%s_%s_%s_addr" % (prefix, side, col), arr.ctypes.data)

    def init_valid_triple_classification(self):
        """Evaluation files + the buffers getValidBatch / getBestThreshold fill (Config.py:123-151)."""
        self.lib.kge_clear_error()
        self.lib.importTestFiles()
        self.lib.importTypeFiles()
        _lib.raise_if_error(self.lib)
        self.testTotal = self.lib.getTestTotal()
        self.validTotal = self.lib.getValidTotal()
        self._tc_buffers("valid", self.validTotal)
        self.relThresh = np.zeros(self.lib.getRelationTotal(), dtype=np.float32)
        self.relThresh_addr = self.relThresh.ctypes.data
        self.acc = np.zeros(1, dtype=np.float32)
        self.acc_addr = self.acc.ctypes.data

    def init_triple_classification(self):
        """The same plus the test-set buffers (Config.py:83-120)."""
        self.init_valid_triple_classification()
        self._tc_buffers("test", self.testTotal)

    def _fit_thresholds(self):
        """Per-relation thresholds from the validation positives and their type-constrained negatives."""
        L = self.lib
        L.getValidBatch(self.valid_pos_h_addr, self.valid_pos_t_addr, self.valid_pos_r_addr,
                        self.valid_neg_h_addr, self.valid_neg_t_addr, self.valid_neg_r_addr)
        _lib.raise_if_error(L)
        res_pos = np.ascontiguousarray(self.test_step(self.valid_pos_h, self.valid_pos_t, self.valid_pos_r).reshape(-1), dtype=np.float32)
        res_neg = np.ascontiguousarray(self.test_step(self.valid_neg_h, self.valid_neg_t, self.valid_neg_r).reshape(-1), dtype=np.float32)
        L.getBestThreshold(self.relThresh_addr, res_pos.ctypes.data, res_neg.ctypes.data)

    def test(self):
        """Triple classification on the test set with thresholds fitted on the validation set, and / or link prediction,
        as the flags say (Config.py:491-516).  Returns a dict: {"acc": ..} and / or the link-prediction metrics."""
        import time
        t0 = time.time()
        result = {}
        if self.test_triple_classification:
            if not hasattr(self, "test_pos_h"):
                self.init_triple_classification()
            L = self.lib
            self._fit_thresholds()
            L.getTestBatch(self.test_pos_h_addr, self.test_pos_t_addr, self.test_pos_r_addr,
                           self.test_neg_h_addr, self.test_neg_t_addr, self.test_neg_r_addr)
            res_pos = np.ascontiguousarray(self.test_step(self.test_pos_h, self.test_pos_t, self.test_pos_r).reshape(-1), dtype=np.float32)
            res_neg = np.ascontiguousarray(self.test_step(self.test_neg_h, self.test_neg_t, self.test_neg_r).reshape(-1), dtype=np.float32)
            L.test_triple_classification(self.relThresh_addr, res_pos.ctypes.data, res_neg.ctypes.data, self.acc_addr)
            _lib.raise_if_error(L)
            result["acc"] = float(self.acc[0])
        if self.test_link_prediction:
            result.update(self.link_prediction()[1])
        print("\nElapsed test time (seconds): {}".format(time.time() - t0))
        return result

    def _top_k(self, scores, k):
        res = np.asarray(scores).reshape(-1).argsort()[:k]
        print(res)
        return res

    def predict_head_entity(self, t, r, k):
        """The k head entities that score best for (?, t, r) (Config.py:574-593)."""
        ar = np.arange(self.entTotal)
        return self._top_k(self.test_step(ar, np.full(self.entTotal, t), np.full(self.entTotal, r)), k)

    def predict_tail_entity(self, h, r, k):
        """The k tail entities that score best for (h, ?, r) (Config.py:595-614)."""
        ar = np.arange(self.entTotal)
        return self._top_k(self.test_step(np.full(self.entTotal, h), ar, np.full(self.entTotal, r)), k)

    def predict_relation(self, h, t, k):
        """The k relations that score best for (h, t, ?) (Config.py:616-635; TransR's predict op uses the matrix of
        the FIRST relation of a call, TransR.py:83, like the reference)."""
        ar = np.arange(self.relTotal)
        return self._top_k(self.test_step(np.full(self.relTotal, h), np.full(self.relTotal, t), ar), k)

    def predict_triple(self, h, t, r, thresh=None):
        """Is (h, t, r) correct?  Score below `thresh`, or below the relation's threshold fitted on the validation set
        (Config.py:637-663).  Prints the reference's message and also returns the verdict."""
        res = float(self.test_step(np.array([h]), np.array([t]), np.array([r])).reshape(-1)[0])
        if thresh is None:
            if not hasattr(self, "valid_pos_h"):
                self.init_triple_classification()
            self._fit_thresholds()
            thresh = float(self.relThresh[r])
        ok = res < thresh
        print("triple (%d,%d,%d) is %s" % (h, t, r, "correct" if ok else "wrong"))
        return ok

    @staticmethod
    def _lp_sums(out, test_head=True):
        """Un-normalised accumulators of main_spark.py:430-448 over the rows of `out` (they add across test-set slices)."""
        d = {}
        for side, p in ((0, "r"), (1, "l")):
            if side == 1 and not test_head:
                continue
            raw, filt, raw_c, filt_c = (out[:, side, i] for i in range(4))
            for name, cnt in (("", raw), ("_filter", filt)):
                d[p + name + "_tot"] = float((cnt < 10).sum())                    # Hits@10
                d[p + "3" + name + "_tot"] = float((cnt < 3).sum())               # Hits@3
                d[p + "1" + name + "_tot"] = float((cnt < 1).sum())               # Hits@1
                d[p + name + "_rank"] = float((1 + cnt).sum())                    # MR
                d[p + name + "_reci_rank"] = float((1.0 / (1 + cnt)).sum())       # MRR
            for name, cnt in (("", raw_c), ("_filter", filt_c)):
                d[p + name + "_tot_constrain"] = float((cnt < 10).sum())
                d[p + "3" + name + "_tot_constrain"] = float((cnt < 3).sum())
                d[p + "1" + name + "_tot_constrain"] = float((cnt < 1).sum())
                d[p + name + "_rank_constrain"] = float((1 + cnt).sum())
                d[p + name + "_reci_rank_constrain"] = float((1.0 / (1 + cnt)).sum())
        return d

    @staticmethod
    def _lp_normalise(sums, count):
        n = float(max(count, 1))
        return {k: v / n for k, v in sums.items()}

    def link_prediction_distributed(self, test_head=True):
        """The whole test set, split into one contiguous range per rank (the static split of
        distribute_training.py:430-441) and reduced like main_spark.py:430-448: every rank returns the global metrics."""
        import torch
        import torch.distributed as dist
        total = self.lib.getTestTotal()
        per = (total + self.world_size - 1) // self.world_size
        lo = min(self.rank * per, total)
        hi = min(lo + per, total)
        out = np.zeros((hi - lo, 2, 8), dtype=np.int64)
        if hi > lo:
            _lib.check(self.lib.kge_link_prediction(ctypes.byref(self._desc), self._tab_ptrs, lo, hi - lo,
                                                    1 if test_head else 0, out.ctypes.data, self._stream()), self.lib)
        sums = self._lp_sums(out, test_head)
        keys = sorted(sums)
        vec = torch.tensor([sums[k] for k in keys], dtype=torch.float64)
        if self.world_size > 1:
            self.comm_fence("pg")
            if dist.get_backend(self._pg) == "nccl":
                vec = vec.to(self.device)
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self._pg)
        return self._lp_normalise(dict(zip(keys, vec.cpu().tolist())), total)

    def link_prediction(self, first=0, count=None, test_head=True):
        """Rank every test triple in [first, first+count) on the device (replaces the per-triple loop of
        distribute_training.py:465-590).  Returns (raw int64 [count,2,8] as testTail/testHead give them,
        metrics dict with the reference's accumulator names normalised by the number of triples, the way
        main_spark.py:430-448 reduces them: r_* = tail prediction, l_* = head prediction)."""
        if count is None:
            count = self.lib.getTestTotal() - first
        out = np.zeros((count, 2, 8), dtype=np.int64)
        _lib.check(self.lib.kge_link_prediction(ctypes.byref(self._desc), self._tab_ptrs, first, count,
                                                1 if test_head else 0, out.ctypes.data, self._stream()), self.lib)
        d = self._lp_normalise(self._lp_sums(out, test_head), count)
        return out, d

    # ------------------------------------------------------------------------------------------
    # parameters by the reference's variable names (Config.py:378-421)
    # ------------------------------------------------------------------------------------------
    def get_parameter_lists(self):
        return self.trainModel.parameter_lists

    def _sharded(self, var_name):
        return self._dp and getattr(self, "sparse_rows", False) and var_name == "ent_embeddings" and hasattr(self, "_shard")

    def get_parameters_by_name(self, var_name):
        if var_name in self.trainModel.parameter_lists:
            t = self.trainModel.parameter_lists[var_name]
            if self._sharded(var_name):   # collective: every rank must call it (the shards are gathered; small tables only)
                import torch
                from .parallel import all_gather_chunks
                self.comm_fence("pg")
                full = torch.empty((self._shard["chunk"] * self.world_size, t.shape[1]), dtype=t.dtype, device=t.device)
                all_gather_chunks(full.view(-1), t.reshape(-1), self._pg)
                return full[:self.entTotal].cpu().numpy()
            return t.detach().cpu().numpy()
        return None

    def get_parameters(self, mode="numpy"):
        res = {}
        for var_name in self.get_parameter_lists():
            if mode == "numpy":
                res[var_name] = self.get_parameters_by_name(var_name)
            else:
                res[var_name] = self.get_parameters_by_name(var_name).tolist()
        return res

    def save_parameters(self, path=None):
        if path == None:
            path = self.out_path
        with open(path, "w") as f:
            f.write(json.dumps(self.get_parameters("list")))

    def tables_changed(self):
        """Tell the engine that device tables were written from outside its kernels (it keeps a per-row 1/|row| table for
        the TransE emit kernel current across steps, include/kge_mi355.h "tables_changed").  Everything in this package that
        writes tables calls it; call it yourself after writing `parameter_lists[...]` tensors directly."""
        self.lib.kge_set_option(b"tables_changed", 1)

    def set_parameters_by_name(self, var_name, tensor):
        import torch
        if var_name in self.trainModel.parameter_lists:
            dst = self.trainModel.parameter_lists[var_name]
            src = torch.as_tensor(np.asarray(tensor, dtype=np.float32))
            if self._sharded(var_name):
                lo, hi = self._shard["lo"], self._shard["hi"]
                dst[:hi - lo].copy_(src.reshape(self.entTotal, -1)[lo:hi])
            else:
                dst.copy_(src.reshape(dst.shape))
            self.tables_changed()

    def set_parameters(self, lists):
        for i in lists:
            self.set_parameters_by_name(i, lists[i])

    def get_gradients(self):
        """Current contents of the dense gradient accumulators (zero between steps)."""
        return {n: g.detach().cpu().numpy() for n, g in zip(self.trainModel.table_names, self._grads)}

    def get_stream_states(self, before_prefetch=False):
        """rng stream states of the virtual sampler threads (next_random[], Random.h:6).  With sampling one step ahead
        (prefetch_sampling) the device states are those AFTER the prefetched batch was drawn; before_prefetch=True rewinds
        them by that one batch (what a checkpoint must store so that a resumed run trains on the prefetched batch too)."""
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = np.zeros(self.workThreads, dtype=np.uint64)
        _lib.check(self.lib.kge_get_stream_states(out.ctypes.data, self.workThreads), self.lib)
        if before_prefetch and getattr(self, "_prefetched", None) is not None:
            from .parallel import slice_positions
            draws = 1 + 2 * self.negative_ent + self.negative_rel          # rng draws per positive (Base.cpp:101-139)
            for i in range(self.workThreads):
                _, cnt = slice_positions(self.batch_size, self.workThreads, i, i + 1)
                out[i] = np.uint64(lcg_jump(int(out[i]), -cnt * draws))
        return out


LCG_A, LCG_C, MASK64 = 25214903917, 11, (1 << 64) - 1   # Random.h:16-19: x = x * 25214903917 + 11 (mod 2^64)


def lcg_jump(state, n):
    """The stream state n draws later (n < 0: earlier; the generator has full period 2^64, so stepping back n draws is
    stepping forward 2^64 - n)."""
    n %= 1 << 64
    a, c = LCG_A, LCG_C
    acc_a, acc_c = 1, 0
    while n:
        if n & 1:
            acc_a, acc_c = (acc_a * a) & MASK64, (acc_c * a + c) & MASK64
        a, c = (a * a) & MASK64, (c * a + c) & MASK64
        n >>= 1
    return (state * acc_a + acc_c) & MASK64
