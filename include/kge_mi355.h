/*
 * kge_mi355.h -- C ABI of libkge_mi355.so, the MI355X (gfx950) engine for the OpenKEonSpark hot path.
 *
 * Two groups of entry points:
 *
 *  (1) The hot-path subset of the reference's `Base.so` ABI, with the reference's exact unmangled
 *      names and signatures, so that /root/reference/Config.py:30-31,160-170,347 binds to this
 *      library unchanged (ctypes.cdll.LoadLibrary + the same calls).  INT = long (int64 on LP64),
 *      REAL = float (base/Setting.h:3-4).  `sampling` runs the HIP sampler and copies the batch
 *      into the caller's host buffers.
 *
 *  (2) `kge_*` entry points that replace what the reference does inside TensorFlow
 *      (`sess.run([train_op, loss, global_step], feed_dict)`, distribute_training.py:282) with
 *      device-resident operators: on-device sampling, fused gather/score/hinge/backward for
 *      TransE/H/D/R, SGD / TF-parity Adam updates, scoring for prediction.  Plain pointers and
 *      sizes only; device pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()),
 *      `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * Error convention: the reference has none (void functions, missing files only print;
 * Reader.h:36-39).  Here every kge_* function returns 0 on success or a negative code, the
 * Base-compatible void functions record a message, and `kge_last_error` returns the most recent
 * message (empty string if none).  There is NO CPU fallback: device entry points fail with
 * KGE_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef KGE_MI355_H
#define KGE_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef INT
#define INT long
#endif
#ifndef REAL
#define REAL float
#endif

/* ------------------------------------------------------------------------------------------
 * (1) Base.so-compatible entry points
 * ---------------------------------------------------------------------------------------- */
void setInPath(char *path);       /* replaces base/Setting.h:12-19  */
void setOutPath(char *path);      /* replaces base/Setting.h:21-28  */
void setWorkThreads(INT threads); /* replaces base/Setting.h:36-39: number of VIRTUAL sampler threads (rng streams + batch slices) */
INT getWorkThreads(void);         /* replaces base/Setting.h:41-44  */
void setBern(INT con);            /* replaces base/Setting.h:110-113 */
void randReset(void);             /* replaces base/Random.h:8-13: seeds one 64-bit LCG stream per virtual thread from the (continuing) unseeded glibc rand() sequence */
void importTrainFiles(void);      /* replaces base/Reader.h:26-179: parse, dedup, build the device-resident filter index */
INT getEntityTotal(void);         /* replaces base/Setting.h:63-66  */
INT getRelationTotal(void);       /* replaces base/Setting.h:68-71  */
INT getTripleTotal(void);         /* replaces base/Setting.h:73-76  */
INT getTrainTotal(void);          /* replaces base/Setting.h:78-81  (after dedup) */
INT getTrainTotal_(void);         /* replaces base/Setting.h:84-87  (file count, duplicates kept) */
INT getBatchTotal(void);          /* replaces base/Setting.h:90-93  (newBatchTotal, incremental mode) */
INT getTestTotal(void);           /* replaces base/Setting.h:95-98  */
INT getValidTotal(void);          /* replaces base/Setting.h:100-103 */
/* replaces base/Base.cpp:149-172.  Caller-owned HOST buffers of length batchSize*(1+negRate+negRelRate),
 * layout [B positives | B negatives round 0 | ... | relation negatives] (Base.cpp:109-139).
 * Bit-identical to the reference for the same workThreads / bern / call history. */
void sampling(INT *batch_h, INT *batch_t, INT *batch_r, REAL *batch_y, INT batchSize, INT negRate, INT negRelRate);

/* Evaluation subset of Base.so (link prediction) */
void importTestFiles(void);      /* replaces base/Reader.h:185-292: test2id / valid2id + the sorted union used by the filter */
void importTypeFiles(void);      /* replaces base/Reader.h:301-365: type_constrain.txt */
void importOntologyFiles(void);  /* replaces base/Reader.h:375-449: ontology_constrain.txt */
void getHeadBatch(INT index, INT *ph, INT *pt, INT *pr);  /* replaces base/Test.h:10-17 */
void getTailBatch(INT index, INT *ph, INT *pt, INT *pr);  /* replaces base/Test.h:19-26 */
/* replace base/Test.h:30-136 and :140-249.  `con` = HOST score vector of all entities; returns 8 INT:
 * [0..3] candidates scoring strictly lower (raw, filtered, type-constrained, both), [4..7] ontology
 * class of the four arg-mins.  The pointer addresses a library-owned slot reused after 64 calls (the
 * reference leaks a `new INT[8]` per call). */
INT *testHead(INT index, REAL *con);
INT *testTail(INT index, REAL *con);
/* Triple classification (replace base/Test.h:252-444).  Host routines over
 * validTotal / testTotal-long score arrays, as in the reference: negatives are the positives with a
 * type-constrained new tail drawn with the (continuing) libc rand() sequence (Corrupt.h:118-137);
 * thresholds by grid search with step 0.01f per relation; get_TPFP returns (n_interval+1)*2 INT in a
 * library-owned buffer valid until the next call (the reference leaks a new[] per call). */
void getTestBatch(INT *ph, INT *pt, INT *pr, INT *nh, INT *nt, INT *nr);
void getValidBatch(INT *ph, INT *pt, INT *pr, INT *nh, INT *nt, INT *nr);
void getBestThreshold(REAL *relThresh, REAL *score_pos, REAL *score_neg);
void test_triple_classification(REAL *relThresh, REAL *score_pos, REAL *score_neg, REAL *acc);
INT get_n_interval(INT r, REAL *score_pos, REAL *score_neg);
INT *get_TPFP(INT r, REAL *score_pos, REAL *score_neg, REAL *threshold, REAL *unused);

/* ------------------------------------------------------------------------------------------
 * (2) Engine entry points
 * ---------------------------------------------------------------------------------------- */
enum {
    KGE_OK = 0,
    KGE_NO_EVENT = 1,         /* kge_stream_wait_emit: no emit launch was recorded since the previous wait */
    KGE_ERR_NO_DEVICE = -1,   /* no usable gfx950 device / HIP runtime error */
    KGE_ERR_NO_DATASET = -2,  /* importTrainFiles / kge_import_train_arrays not done */
    KGE_ERR_BAD_ARG = -3,
    KGE_ERR_UNSUPPORTED = -4
};

enum { KGE_TRANSE = 0, KGE_TRANSH = 1, KGE_TRANSR = 2, KGE_TRANSD = 3 };

/* copies the last error message (NUL terminated, truncated to n) and returns its length */
size_t kge_last_error(char *buf, size_t n);
void kge_clear_error(void);
/* 1 when a HIP device is visible and usable, else 0 (never throws, never falls back) */
int kge_device_available(void);
const char *kge_version(void);

/* engine options (testing / measurement).
 *   "counts_force_sort": 1 = order the sign-count records with rocPRIM's radix sort instead of the
 *                        hand-written two-level counting sort (default 0)
 *   "counts_fused":      1 (default) = kge_transe_train_step_counts sums the records of a row and applies the optimizer to it in one
 *                        kernel (rows whose int8 and 2-bit record lists hold up to 64 records each; longer rows and relation rows go through the count image);
 *                        0 = the two-kernel form (segmented sum into the image, then kge_transe_apply_counts_tables).  Same bits.
 *   "ride_shares":       where an armed sampler (kge_sampling_attach) rides: percent of its workgroups for the bucket histogram /
 *                        bucket scatter / bucket sort / the launch that ends the step, one byte each (default 100 << 8: all of it in
 *                        the scatter launch; parts that no launch took are launched by kge_sampling_flush)
 *   "counts_fused_diag": measurement hook of the fused kernel (1: no record loops, 2: no row update);
 *                        any value but 0 gives WRONG results
 *   "counts_fused_cap":  test hook: rows of more than this many records take the image path (0 = the kernel's capacity)
 *   "inv_table_max_bytes": the TransE emit kernel reads 1/|row| from a per-row table rebuilt every step while
 *                        the two tables are at most this many bytes (default 256 MiB); larger tables (or 0)
 *                        compute the norms from the gathered rows
 *   "float_records":     1 (default) = kge_forward_backward stores TransE/H/D gradient rows as records and sums
 *                        them by destination after a sort; 0 = fp32 atomic adds straight into the accumulators
 *   "float_records_min": smallest number of gradient rows per step that takes the record path (default 65536: measured cross-over, tools/sweep_paths.py)
 *   "tables_changed": the caller has written device parameter tables itself (a copy into them, an all-gather, a restored
 *                     checkpoint).  The TransE emit kernel keeps a table of 1/|row| that the full-table apply kernel
 *                     (kge_transe_apply_counts_tables) refreshes row by row, so that no pre-pass over the tables is needed between
 *                     steps; every kge_* entry point that writes tables marks it stale itself, this option is for writes the
 *                     library cannot see.  Value ignored.
 *   "counts_krel": dense TransE sign-count path: group b sends its relation-side records to virtual copy b % counts_krel of the
 *                     relation rows while they are ordered, the segmented sum folds the copies back (hub rows otherwise
 *                     serialise the LDS atomics of the bucketing kernels); a power of two, default 4 (measured best of 1..64), 1 = off
 *   "transr_dgrad_records" / "transr_dgrad_records_min": TransR backward w.r.t. the entity rows: 1 (default) = the rows of
 *                     G . M_r^T are stored as float records and summed per entity by the record sort + segmented sum from
 *                     `_min` (default 32768) projected rows per step on; 0 = fp32 atomics always
 *   "inv_carry": 0 = recompute that table in front of every emit launch (test hook; default 1)
 *   "record_emit_event": 1 = record an event behind every TransE emit launch (kge_stream_wait_emit); default 0
 *   "pair_counts":       1 (default) = TransH / TransD steps of at least float_records_min entity-side rows (widths that are
 *                        multiples of 4 up to 256, at most 63 negatives, ent_total*rel_total below 2^31) take the
 *                        pair-count path: int8 sign records keyed by (entity, relation), the backward applied once per pair
 *                        (csrc/pairs.hip); 0 = float records / atomics as for the other shapes
 *   "pair_counts_min_neg": fewest negatives per positive for that path; default 0 = the measured cross-over (TransH 5, TransD 3)
 *   "index_device_min":  training sets with at least this many lines are indexed on the device (rocPRIM sorts,
 *                        same arrays bit for bit); default 4194304, 0 = always, negative = never
 *   "hub_copies":        1 (default) = on the fp32-atomic TransH/TransD path, relation-side gradient rows that would
 *                        take >= 128 adds per step are accumulated in up to 64 copies and folded (same-address
 *                        atomics serialise); 0 = straight into the accumulators
 *   "transr_bf16x3":     1 (default) = TransR's row GEMMs and wgrad (dims <= 208, multiples of 4) form the fp32 products as six
 *                        bf16 x bf16 term products of an exact three-term split on the bf16 matrix pipe (error vs fp64 equal to the
 *                        fp32 MFMA's); 0 = the fp32 MFMA kernels
 *   "transr_groups":     TransR, device-sampled batches, 2 + negatives <= 16: 1 (default) = steps with well-filled relation buckets
 *                        sort GROUPS by relation and keep a group's rows inside one 16-row sub-tile; 2 = at any size; 0 = never
 *   "transr_fuse_vec":   1 (default) = with that layout the vector stage runs inside the projection's epilogue; 0 = as its own launch
 *   "transr_v1":         test hooks for the TransR MFMA tilings (default 0 = automatic): 1 = always the 32x32x2 tiles; 2 = 16x16x4
 *                        tiles with the all-output-tiles wgrad and its 512-row spans forced; 3 = 16x16x4 tiles with the 32x32x2 wgrad
 *   "time_emit":         N > 0 = bracket every N-th launch of the TransE emit kernel with HIP events on its launch stream
 *   "fb_occ4":           1 (default) = TransH / TransD / TransR's vector stage at <= 4 elements per lane run the forward/backward body
 *                        compiled for four waves per SIMD (128 VGPRs); 0 = the uncapped build
 *   "persist_touch":     1 = the persistent launch requests all rows of a group together before its dependent gathers (default 0:
 *                        measured, no gain -- the phase is bound by same-address atomics, not by cold gathers)
 *   "persist_ahead":     1 (default) = teams without a group draw the next batch during the forward/backward phase
 *   "persist_trace":     1 = kge_train_steps_persistent stamps its phase boundaries (read with kge_persistent_trace)
 *   "persist_threads":   threads per workgroup of the persistent launch, 512 (default) or 1024
 *   "libc_rand_restart": restart the glibc-compatible seed generator, as in a fresh process (the next
 *                        randReset then yields 1804289383, 846930886, ... again) */
int kge_set_option(const char *name, INT value);
/* elapsed time of the most recent launch of a timed kernel; name = "transe_emit" (needs time_emit) */
int kge_last_kernel_ms(const char *name, float *ms);
/* mean over the launches since "time_emit" was switched on (the most recent 512 of them); one event pair per launch, read
 * back here, so nothing synchronises inside the timed region */
int kge_kernel_ms_mean(const char *name, float *mean_ms, INT *launches);

/* Same as importTrainFiles but from arrays already in memory (h,t,r in FILE ORDER, duplicates
 * kept; new_batch_total as batch2id.txt's first line, 0 = not incremental).  Restates
 * Reader.h:82-177 without the text parse. */
int kge_import_train_arrays(INT ent_total, INT rel_total, INT n, const INT *h, const INT *t, const INT *r,
                            INT new_batch_total);

/* Host-side copies of the index, for inspection and CPU tests.  `what` is one of
 *   "tails_hr"  int32[trainTotal]  tails in (h,r,t) order         (= trainHead[].t, Reader.h:125)
 *   "heads_tr"  int32[trainTotal]  heads in (t,r,h) order         (= trainTail[].h, Reader.h:126)
 *   "rels_ht"   int32[trainTotal]  relations in (h,t,r) order     (= trainRel[].r,  Reader.h:127)
 *   "pos"       int32[trainTotal_][4]  file-order triples (h,t,r,0)             (= trainList_no)
 *   "grp"       int32[trainTotal_][4]  (hr_off,hr_len,tr_off,tr_len) per file-order triple
 *   "ht"        int32[trainTotal_][2]  (ht_off,ht_len) per file-order triple
 *   "left_mean" / "right_mean" float[relationTotal]               (Reader.h:160-177)
 *   "bern_prob" float[relationTotal]  1000*right/(right+left)      (Base.cpp:117)
 * Returns the number of BYTES the array holds (copying at most `bytes` of them), <0 on error. */
int64_t kge_index_copy(const char *what, void *dst, int64_t bytes);

/* rng stream states of the virtual threads (host view; Random.h:6) */
int kge_get_stream_states(uint64_t *dst, INT n);
int kge_set_stream_states(const uint64_t *src, INT n);

/* On-device sampling of the slice of the batch owned by virtual threads [thread_lo, thread_hi)
 * (Base.cpp:85-92 partitions the batch by thread id; a data-parallel rank owns a range of them).
 * d_h/d_t/d_r: DEVICE int32 arrays of length out_stride*(1+negRate+negRelRate); the positive at
 * global batch position p goes to index p - first_position(thread_lo), negative k to that +
 * (k+1)*out_stride.  All workThreads rng streams advance exactly as one reference `sampling`
 * call would advance them, whatever the owned range, so replicas stay in step.
 * *n_local receives the number of positives written (may be NULL). */
int kge_sampling_device(int32_t *d_h, int32_t *d_t, int32_t *d_r, INT batchSize, INT negRate, INT negRelRate,
                        INT thread_lo, INT thread_hi, INT out_stride, INT *n_local, void *stream);

/* The same batch, ARMED instead of launched: the sampler depends on the rng streams and the dataset only, never on the
 * parameters, so the batch of step i+1 can be drawn while step i is being reduced.  The armed sampler rides in the launch of
 * the next kernel of the sign-count / pair-count pipeline that leaves most wave slots idle -- the bucket scatter, one
 * workgroup per CU -- as extra workgroups of that launch: no side stream, no events, no second queue for the command
 * processor to arbitrate (the step is one in-order stream of launches).  kge_sampling_flush launches an armed sampler on its
 * own when the step's path had no such kernel (and is a no-op otherwise); every other sampler entry point, the stream-state
 * accessors and the persistent launch flush it first, so batches are always drawn in order.  The rng streams are accounted
 * as advanced from the moment of the call (Base.cpp:149-172 semantics unchanged: same batches, same order, same bits). */
int kge_sampling_attach(int32_t *d_h, int32_t *d_t, int32_t *d_r, INT batchSize, INT negRate, INT negRelRate,
                        INT thread_lo, INT thread_hi, INT out_stride, INT *n_local, void *stream);
int kge_sampling_flush(void *stream);
/* number of batch positions owned by virtual threads [thread_lo, thread_hi) for this batchSize */
INT kge_slice_positions(INT batchSize, INT thread_lo, INT thread_hi, INT *first_position);

typedef struct kge_model_desc {
    int32_t model;        /* KGE_TRANSE .. KGE_TRANSD  (distribute_training.py:62-69) */
    int32_t negative_rel; /* Config.negative_rel; TransR reuses the positive's matrix when 0 (TransR.py:57) */
    int64_t ent_total, rel_total;
    int32_t ent_dim, rel_dim; /* TransE/H/D: both = hidden_size (TransD.py:37-40 ignores ent/rel_size) */
    float margin;
    int32_t reserved;
} kge_model_desc;

/* Parameter tables, by the reference's variable names (the checkpoint contract):
 *   [0] ent_embeddings [E,De]  [1] rel_embeddings [R,Dr]
 *   [2] normal_vectors [R,Dr] (TransH) | transfer_matrix [R,De*Dr] (TransR) | rel_transfer [R,Dr] (TransD)
 *   [3] ent_transfer [E,De] (TransD)
 * Dense row-major fp32, no padding. */
#define KGE_MAX_TABLES 4
int kge_table_shape(const kge_model_desc *m, int table, int64_t *rows, int64_t *cols);

/* Fused gather -> score -> margin-ranking loss -> backward for one batch already on the device.
 * Replaces the forward/backward half of sess.run(train_op) for TransE.py:26-51, TransH.py:33-69,
 * TransD.py:46-84, TransR.py:36-75.
 *   d_h,d_t,d_r : DEVICE int32[stride*(1+n_neg)], the Base.cpp:109-139 layout
 *   n_pos       : positives in this (local) batch; n_neg = negative_ent + negative_rel
 *   denom       : the reduce_mean denominator, GLOBAL batch_size*n_neg (TransE.py:51)
 *   grads[i]    : DEVICE fp32 dense accumulators shaped like tables[i]; the summed (deduplicated)
 *                 IndexedSlices gradient is ADDED into them (zero them first; the update ops
 *                 re-zero them)
 *   d_loss      : DEVICE float[1], receives sum(hinge)/denom of this local batch
 */
int kge_forward_backward(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES],
                         const int32_t *d_h, const int32_t *d_t, const int32_t *d_r,
                         INT n_pos, INT n_neg, INT stride, INT denom,
                         float *const grads[KGE_MAX_TABLES], float *d_loss, void *stream);

/* Makes `stream` wait for the most recent launch of the TransE emit kernel (or of the pair-count path's emit kernel) (option "record_emit_event" = 1 records an event
 * behind every such launch).  Config.prefetch_sampling uses it to start the next batch's sampler on a side stream as soon as the
 * emit kernel -- the one bandwidth-bound kernel of the step -- has finished, so that it runs beside the small latency-bound
 * kernels that follow (bucketing, segmented sum, apply).  Returns KGE_NO_EVENT (and makes `stream` wait for nothing) when no
 * emit launch was recorded since the previous call: the caller then orders `stream` behind the step by an event of its own. */
int kge_stream_wait_emit(void *stream);
/* Forward / backward AND plain SGD on the touched rows only, in place (TransE / TransH / TransD): the step's gradient rows are
 * stored as float records, ordered by destination row, summed by segments and added to their PARAMETER rows as -lr * sum.
 * No gradient tables and no sweep over the tables: for tables whose size, not the batch, would set the step time (the dense
 * form is kge_forward_backward + kge_sgd_update_tables; same update up to fp32 summation order -- GradientDescentOptimizer
 * leaves rows without gradient alone, distribute_training.py:99-101).  Not for Adam: TF1's Adam moves every row.
 * Negatives that are not single-slot corruptions of their positive (the reference's sampler draws no others; a hand-fed batch
 * may hold them) are SKIPPED -- their exact path adds rows atomically, which in place would race with the forward reads -- and
 * counted: kge_sgd_rows_skipped (synchronises); the caller must treat a non-zero count as an error of that step. */
int kge_forward_backward_sgd_rows(const kge_model_desc *m, float *const tables[KGE_MAX_TABLES], const int32_t *d_h, const int32_t *d_t,
                                  const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, float lr, float *d_loss, void *stream);
int kge_sgd_rows_skipped(int32_t *n_negatives);
/* The same update across N ranks (replaces the per-variable scatter_sub the reference's workers send to the parameter servers,
 * distribute_training.py:99-101,193-196): kge_forward_backward_records stores the gradient rows of THIS rank's slice of the batch
 * as float records (d_rec [*, dim], destination keys d_dst) into its slice [rec_offset, rec_offset + rec_slice) of two buffers the
 * caller owns (unused positions of the slice get key -1), the caller all-gathers the slices, and kge_float_records_apply sums ALL
 * n_records records by destination row and adds -lr * sum to the rows of every replica -- the sparse touched-row exchange; the
 * replicas stay identical because every rank reduces the same records in the same order.  n_pos_total = positives of the GLOBAL
 * batch (it fixes the virtual row space of the keys, which must be the same on all ranks); `denom` the global denominator;
 * d_loss receives this rank's share of the loss.  kge_sgd_rows_skipped applies as above. */
int kge_forward_backward_records(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], const int32_t *d_h, const int32_t *d_t,
                                 const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, INT n_pos_total, float *d_rec,
                                 int32_t *d_dst, INT rec_offset, INT rec_slice, float *d_loss, void *stream);
int kge_float_records_apply(const kge_model_desc *m, float *const tables[KGE_MAX_TABLES], const float *d_rec, int32_t *d_dst, INT n_records,
                            INT n_pos_total, INT n_neg, float lr, void *stream);
/* 1 when kge_forward_backward on a step of this shape takes the TransH / TransD pair-count path (whose emit kernel also records
 * the event above), else 0 */
int kge_pair_path_active(const kge_model_desc *m, INT n_pos, INT n_neg);

/* The same call for a batch the caller KNOWS to be sampler-shaped -- what kge_sampling_device / `sampling` produce
 * (Base.cpp:109-139): every negative differs from its positive in exactly one entity slot, or (negative_rel) in the relation.
 * Paths that route other groups to a separate exact pass (the TransH / TransD pair-count path) then skip that pass and
 * its two bookkeeping launches.  A group that breaks the promise contributes nothing on those paths. */
int kge_forward_backward_sampled(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES],
                                 const int32_t *d_h, const int32_t *d_t, const int32_t *d_r,
                                 INT n_pos, INT n_neg, INT stride, INT denom,
                                 float *const grads[KGE_MAX_TABLES], float *d_loss, void *stream);

/* The loss of a data-parallel TransE step, carried inside the int32 count image that the step reduce-scatters anyway (one
 * collective less per step): d_limbs4 receives the four 16-bit limbs of llrint(loss * 2^32); after the SUM exchange
 * kge_limbs_to_loss turns the summed limbs back into the summed loss.  Exact integer arithmetic: the result does not depend on
 * the number of ranks or on the reduction order. */
int kge_loss_to_limbs(const float *d_loss, int32_t *d_limbs4, void *stream);
int kge_limbs_to_loss(const int32_t *d_limbs4, float *d_out, void *stream);
/* From now on kge_transe_forward_counts on a device-sampled batch ALSO writes its loss as those four limbs to d_limbs4 (the emit
 * kernel's last workgroup does it: no conversion launch); NULL switches it off.  The pointer must stay valid until then. */
int kge_loss_limbs_target(int32_t *d_limbs4);

/* GradientDescentOptimizer on the summed gradient: p -= lr*g; g = 0   (distribute_training.py:98) */
int kge_sgd_update(float *d_p, float *d_g, int64_t n, float lr, void *stream);
/* TF1 AdamOptimizer._apply_sparse_shared on the summed gradient (distribute_training.py:96): every
 * row decays and moves.  lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) computed by the caller; g = 0 after. */
int kge_adam_update(float *d_p, float *d_m, float *d_v, float *d_g, int64_t n, float lr_t, float beta1,
                    float beta2, float eps, void *stream);

/* The same two updates for all tables of a model in ONE launch (n_tables <= KGE_MAX_TABLES; numel[i] elements each) */
int kge_sgd_update_tables(int32_t n_tables, float *const d_p[KGE_MAX_TABLES], float *const d_g[KGE_MAX_TABLES],
                          const INT numel[KGE_MAX_TABLES], float lr, void *stream);
int kge_adam_update_tables(int32_t n_tables, float *const d_p[KGE_MAX_TABLES], float *const d_m[KGE_MAX_TABLES],
                           float *const d_v[KGE_MAX_TABLES], float *const d_g[KGE_MAX_TABLES], const INT numel[KGE_MAX_TABLES],
                           float lr_t, float beta1, float beta2, float eps, void *stream);

/* ---- TransE sign-count path (exact integer gradients, no fp32 atomics) ----------------------
 * For the L1 score of TransE.py:11-15 the gradient w.r.t. every l2-normalised vector is (1/denom) x an
 * integer vector of signs, so the backward can be carried as exact int32 COUNTS per table row:
 *   counts[(E+R), D]  rows [0,E) = ent_embeddings, rows [E,E+R) = rel_embeddings.
 * kge_transe_forward_counts = gather/score/hinge (as kge_forward_backward) + int8 gradient records +
 * sort-and-sum into `d_counts` (must be zero on entry; kge_transe_apply_counts re-zeroes it).  The
 * counts are order-independent, so a data-parallel all-reduce of them is exact and every replica
 * stays bit-identical.  Negatives that are not sampler-shaped (more than one slot differs from the
 * positive) are differentiated exactly in fp32 into the residual accumulators d_resid_ent [E,D] /
 * d_resid_rel [R,D] (zero on entry, all-zero afterwards for sampler batches).  d_resid_ent = d_resid_rel = NULL
 * is the caller's guarantee that the batch IS sampler-shaped (it came from kge_sampling_device): no deferral
 * bookkeeping, no fp32 pass, and the loss is written by the emit kernel itself.
 * kge_transe_apply_counts: per row g = (1/denom)*inv*(S - x^<x^,S>) + resid (the normalise-backward
 * applied once to the summed counts), then SGD (adam=0, lr) or TF1 Adam (adam=1, lr = lr_t) in place. */
int kge_transe_counts_supported(const kge_model_desc *m, INT n_neg);
int kge_transe_forward_counts(const kge_model_desc *m, const float *d_ent, const float *d_rel, const int32_t *d_h,
                              const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom,
                              int32_t *d_counts, float *d_resid_ent, float *d_resid_rel, float *d_loss, void *stream);
int kge_transe_apply_counts(float *d_p, float *d_m, float *d_v, int32_t *d_counts, float *d_resid, int64_t rows, int32_t dim,
                            INT denom, int32_t adam, float lr, float beta1, float beta2, float eps, void *stream);
/* The whole single-process step of the sign-count path in one call -- what sess.run([train_op, loss, global_step]) does for TransE
 * (distribute_training.py:95-101,282): kge_transe_forward_counts followed by kge_transe_apply_counts_tables, with the middle fused
 * where the shape allows (widths that are multiples of 4, tables that take the bucket sort): the negatives' records shrink to two
 * bits per element, and the rows whose records fit one team are summed in registers and updated right there -- no count image, no
 * second pass -- by the same per-row arithmetic as the apply kernel (bit-identical results; tests/test_gpu_models.py).  Longer
 * rows, relation rows and rows without records go through `d_counts` (zero on entry, zero on return) and one apply launch.
 * sampler_shaped = 1: the batch came from kge_sampling_device (no deferral bookkeeping, no fp32 pass, d_resid tables not read);
 * 0: any batch, d_resid [E,D] / [R,D] (zero on entry and return) take the exact fp32 gradients of groups that are not
 * sampler-shaped.  d_m / d_v may be NULL for SGD (adam = 0, lr); Adam: lr = lr_t. */
int kge_transe_train_step_counts(const kge_model_desc *m, float *const d_p[2], float *const d_m[2], float *const d_v[2], const int32_t *d_h,
                                 const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, int32_t *d_counts,
                                 float *const d_resid[2], int32_t sampler_shaped, int32_t adam, float lr, float beta1, float beta2, float eps,
                                 float *d_loss, void *stream);
/* both tables ([0] = ent_embeddings, [1] = rel_embeddings; d_counts = the whole [(E+R), D] image) in one launch */
int kge_transe_apply_counts_tables(const kge_model_desc *m, float *const d_p[2], float *const d_m[2], float *const d_v[2],
                                   int32_t *d_counts, float *const d_resid[2], INT denom, int32_t adam, float lr, float beta1,
                                   float beta2, float eps, void *stream);

/* the same on the row range [row_lo, row_hi) of the [(E+R), D] row space only; d_counts_chunk = the count image OF THAT RANGE
 * (a data-parallel rank's reduce-scattered chunk): owner-computes update, replaces the replicated optimizer sweep */
int kge_transe_apply_counts_range(const kge_model_desc *m, float *const d_p[2], float *const d_m[2], float *const d_v[2],
                                  int32_t *d_counts_chunk, float *const d_resid[2], INT row_lo, INT row_hi, INT denom, int32_t adam,
                                  float lr, float beta1, float beta2, float eps, void *stream);

/* ---- TransE sign-count path, stage level: for tables too large for a dense count image and for the
 * multi-GPU exchange, where the int8 records (8x smaller than fp32 gradient rows) are the wire format ----
 *   kge_transe_record_dwords : dwords per record for this embedding width
 *   kge_transe_emit_records  : stage 1 into CALLER buffers d_rec [n_pos*(3+n_neg), dwords], d_dst [n_pos*(3+n_neg)]
 *                              (destination row in the [0,E)+[E,E+R) row space, -1 = no record)
 *   kge_transe_reduce_records: order any number of records (e.g. all ranks' records after an all-gather) by
 *                              destination and sum them into a COMPACT image: d_rows[i] = touched row,
 *                              d_row_counts[i, D] = its int32 count vector, *d_n_rows = how many (device scalar);
 *                              buffers sized for n_records rows; d_dst is overwritten
 *   kge_transe_apply_rows_sgd: SGD on the touched rows only (same arithmetic as kge_transe_apply_counts)
 *   kge_transe_deferred_groups: groups of the last emit whose negatives were not sampler-shaped (each negative
 *                              differing from its positive in exactly one slot).  With residual accumulators they
 *                              were handled by the fp32 path; with d_resid_* == NULL they were SKIPPED and the
 *                              caller must treat a non-zero count as an error.  Synchronises. */
INT kge_transe_record_dwords(const kge_model_desc *m);
int kge_transe_deferred_groups(int32_t *n_groups);
int kge_transe_emit_records(const kge_model_desc *m, const float *d_ent, const float *d_rel, const int32_t *d_h, const int32_t *d_t,
                            const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, uint32_t *d_rec, int32_t *d_dst,
                            float *d_resid_ent, float *d_resid_rel, float *d_loss, void *stream);
int kge_transe_reduce_records(const kge_model_desc *m, const uint32_t *d_rec, int32_t *d_dst, INT n_records, int32_t *d_rows,
                              int32_t *d_row_counts, int32_t *d_n_rows, void *stream);
int kge_transe_apply_rows_sgd(const kge_model_desc *m, float *d_ent, float *d_rel, const int32_t *d_rows, const int32_t *d_row_counts,
                              const int32_t *d_n_rows, INT max_rows, INT denom, float lr, void *stream);
/* NON-PARITY, opt-in ("LazyAdam"): Adam on the touched rows only -- m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2,
 * p -= lr_t m / (sqrt(v) + eps) for the rows listed in d_rows (every element of a listed row, zero gradients included), all
 * other rows and their moments left alone: tf.contrib.opt.LazyAdamOptimizer's rule, NOT what the reference trains with (TF1's
 * AdamOptimizer moves every row of the table every step, distribute_training.py:96 -- kge_transe_apply_counts / kge_adam_dense
 * are the parity path).  For tables whose dense sweep (32 bytes per element per step) would dominate the step.  lr_t =
 * lr sqrt(1 - b2^t) / (1 - b1^t) with the global step t, computed by the caller. */
/* The NEXT kge_transe_apply_rows_adam_lazy call processes listed row i only where d_row_live[i] != 0 (one call, then back to "every
 * listed row").  For the table-sharded data-parallel step: the replicated relation rows are listed in full with their all-reduced
 * counts, and a relation for which NO rank had a record must keep its row and moments (the lazy rule). */
int kge_transe_lazy_row_live(const int32_t *d_row_live);
int kge_transe_apply_rows_adam_lazy(const kge_model_desc *m, float *d_ent, float *d_rel, float *d_m_ent, float *d_m_rel, float *d_v_ent,
                                    float *d_v_rel, const int32_t *d_rows, const int32_t *d_row_counts, const int32_t *d_n_rows,
                                    INT max_rows, INT denom, float lr_t, float beta1, float beta2, float eps, void *stream);
/* reduce + apply in one pass (embedding width a multiple of 4): rows whose records all fall inside one 64-record chunk
 * of the sorted list are updated straight from the registers that hold their sum; only chunk-boundary rows go through
 * d_row_counts and a second, small pass.  Same bits as kge_transe_reduce_records + kge_transe_apply_rows_sgd.
 * d_rows / *d_n_rows as there; d_row_counts is only meaningful for the boundary rows afterwards. */
int kge_transe_reduce_apply_records_sgd(const kge_model_desc *m, const uint32_t *d_rec, int32_t *d_dst, INT n_records, float *d_ent,
                                        float *d_rel, int32_t *d_rows, int32_t *d_row_counts, int32_t *d_n_rows, INT denom, float lr,
                                        void *stream);

/* ---- Table-sharded sparse path (N GPUs, BASELINE config #5): rank g OWNS the entity rows [g*chunk, (g+1)*chunk); what the
 * reference does with ps tasks holding the variables and workers pulling rows / pushing IndexedSlices over gRPC
 * (distribute_training.py:193-196) becomes: request ids -> all-to-all -> owners gather rows -> all-to-all -> emit records
 * against the fetched rows -> all-to-all (row id, record) -> owners reduce + apply their rows.  These are the device stages
 * between the collectives (csrc/shard.hip); the collectives themselves are torch.distributed all_to_all_single (RCCL).
 *   kge_shard_requests      : d_req[slot*n_pos + b] = entity touched by record slot (slot, b) of the emit kernel, -1 if none
 *                             (slot 0/1 = the positive's head/tail, 2 = relation, 3+k = the new entity of negative k)
 *   kge_shard_count         : d_counts[o] = how many of d_ids (ids < 0 skipped) rank o = id / chunk owns   (n_owners <= 64)
 *   kge_shard_scatter       : d_sorted = the live ids grouped by owner (h_counts = HOST copy of d_counts), d_slot_of[i] = position of
 *                             d_ids[i] in d_sorted or -1; d_cursor = n_owners ints of scratch
 *   kge_shard_remap_batch   : the batch with entity ids replaced by positions in the fetched-row list (d_slot_of from the
 *                             requests), so the unchanged emit kernel runs against the fetched rows as its "entity table"
 *   kge_shard_gather_rows   : d_out[i,:] = d_table[d_ids[i] - row_lo, :]  (the owner's reply; dim % 4 == 0)
 *   kge_shard_record_ids    : d_ids[m] = global entity id of record m when its destination is a fetched-row slot, else -1
 *   kge_shard_pack_records  : d_out[d_slot_of[m], :] = d_rec[m, :] for the records that travel
 *   kge_shard_relation_counts: relation records (destination >= cache_rows) summed into the dense int32 image [R, dim]
 *                             (zero it first; all-reduced across ranks, the small relation table stays replicated) */
int kge_shard_requests(const int32_t *d_h, const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, int32_t *d_req,
                       void *stream);
int kge_shard_count(const int32_t *d_ids, INT n, INT chunk, INT n_owners, int32_t *d_counts, void *stream);
int kge_shard_scatter(const int32_t *d_ids, INT n, INT chunk, INT n_owners, const INT *h_counts, int32_t *d_cursor, int32_t *d_sorted,
                      int32_t *d_slot_of, void *stream);
int kge_shard_remap_batch(const int32_t *d_h, const int32_t *d_t, INT n_pos, INT n_neg, INT stride, const int32_t *d_slot_of, int32_t *d_h2,
                          int32_t *d_t2, void *stream);
int kge_shard_gather_rows(const float *d_table, const int32_t *d_ids, INT n, INT row_lo, INT rows, INT dim, float *d_out, void *stream);
int kge_shard_record_ids(const int32_t *d_dst, INT n_records, INT cache_rows, const int32_t *d_cache_ids, int32_t *d_ids, void *stream);
int kge_shard_pack_records(const uint32_t *d_rec, const int32_t *d_slot_of, INT n_records, INT dwords, uint32_t *d_out, void *stream);
/* The same image from a COMPACT (rows, counts) pair as kge_transe_reduce_records leaves it (dim % 4 == 0): image[row - base] =
 * counts of that row for rows in [base, base + rel_total); the image is zeroed by the caller.  Config._sharded_step reduces the
 * relation-slot records (a contiguous third of the record buffer) by sort + segmented sum and scatters the <= R rows with this --
 * the per-element int32 atomics of kge_shard_relation_counts were 0.40 ms of a 3.0 ms step at 133 k positives x dim 512. */
int kge_shard_scatter_count_rows(const int32_t *d_rows, const int32_t *d_row_counts, const int32_t *d_n_rows, INT max_rows, INT base,
                                 INT rel_total, INT dim, int32_t *d_image, void *stream);
int kge_shard_relation_counts(const uint32_t *d_rec, const int32_t *d_dst, INT n_records, INT cache_rows, INT rel_total, INT dwords, INT dim,
                              int32_t *d_counts, void *stream);

/* ---- Many training steps in ONE persistent launch (csrc/persist.hip): the loop body of distribute_training.py:267-283 --
 * sampling, forward / backward, optimizer -- at the reference's own (launch-latency bound) batch sizes.  One workgroup per CU
 * stays resident and walks n_steps steps with two XCD-hierarchical grid barriers per step; batches are bit-identical to n_steps
 * calls of `sampling` (all workThreads streams are advanced accordingly), gradients go through fp32 atomics as in
 * kge_forward_backward's small-step path.  TransE / TransH / TransD, embedding width <= 256, the whole batch on this GPU.
 *   h_lr[n_steps] : HOST array, the learning rate of each step (SGD: alpha; Adam: lr_t = alpha*sqrt(1-b2^t)/(1-b1^t))
 *   d_losses[n_steps] : DEVICE array, receives every step's loss
 * kge_persistent_aborted: 1 if a grid barrier of the last launch gave up (a bounded spin expired); the tables are then in an
 * unspecified intermediate state and the caller must treat the run as failed.  Synchronises. */
int kge_train_steps_persistent(const kge_model_desc *m, float *const tables[KGE_MAX_TABLES], float *const grads[KGE_MAX_TABLES],
                               float *const adam_m[KGE_MAX_TABLES], float *const adam_v[KGE_MAX_TABLES], INT batchSize, INT negRate,
                               INT negRelRate, INT n_steps, int32_t adam, const float *h_lr, float beta1, float beta2, float eps,
                               float *d_losses, void *stream);
int kge_persistent_aborted(int32_t *flag);
/* measurement hook (option "persist_trace"): workgroup 0's 100 MHz clock at the six phase boundaries of each of the first
 * n_steps (<= 256) steps of the last launch: [sweep start, sampling start, barrier-1 arrive, barrier-1 leave, barrier-2 arrive, leave] */
int kge_persistent_trace(uint64_t *h_out, INT n_steps);

/* Device-native link prediction for test triples [first, first+count) (replaces the loop
 * distribute_training.py:465-590: getTailBatch -> sess.run(predict) -> testTail, and the head side when
 * test_head != 0).  h_out (HOST) receives count x 2 x 8 int64: [i][0] testTail's 8-vector, [i][1]
 * testHead's (zeros if test_head == 0).  Needs importTestFiles (+ Type / Ontology files if present). */
int kge_link_prediction(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], INT first, INT count,
                        INT test_head, int64_t *h_out, void *stream);

/* predict op: score n triples.  TransE: mean over the dimension (TransE.py:58); others: sum
 * (TransH.py:82, TransR.py:87 with predict_r[0]'s matrix for all, TransD.py:98). */
int kge_predict(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], const int32_t *d_h,
                const int32_t *d_t, const int32_t *d_r, INT n, float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KGE_MI355_H */
