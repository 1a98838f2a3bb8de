// Process-global engine state shared by the translation units of libkge_mi355.so.
//
// Like the reference's Base.so (base/Reader.h:9-24, base/Setting.h:9-10,34,50-61) the library
// holds ONE dataset per process.  The host index lives in `KgIndex`; its device mirror is uploaded
// lazily on first device use so that the loader and getters also work on a box without a GPU
// (where every device entry point fails with KGE_ERR_NO_DEVICE -- there is no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kge_mi355.h"
#include "kg_index.hpp"

namespace kge {

struct DeviceIndex {
    bool uploaded = false;
    int4 *pos = nullptr;      // [train_dup]
    int4 *grp = nullptr;      // [train_dup]
    int2 *ht = nullptr;       // [train_dup]
    int32_t *tails_hr = nullptr, *heads_tr = nullptr, *rels_ht = nullptr;  // [train_uniq]
    float *bern_prob = nullptr;                                            // [rel_total]
    uint64_t *streams = nullptr;                                           // [work_threads]: the CURRENT states
    uint64_t *streams_next = nullptr;   // the other half of the same allocation: the sampler writes the advanced states there, then the two swap
    int64_t streams_cap = 0;
    int streams_sync = 0;  // 0: host copy newer (upload before use), 1: in sync, 2: device copy newer
    // staging for the Base.so-compatible host-buffer `sampling`
    int32_t *stage_i32 = nullptr;    // 3 * cap int32
    int64_t *stage_i64 = nullptr;    // 3 * cap int64 followed by cap floats
    int64_t stage_cap = 0;
    float *loss_partials = nullptr;  // per-block partial losses
    unsigned *loss_ticket = nullptr; // blocks-done counter: the last block of a forward/backward kernel adds the partials
};

struct Engine {
    std::mutex mu;
    std::string in_path = "../data/FB15K/";   // Setting.h:9
    std::string out_path = "../data/FB15K/";  // Setting.h:10
    int64_t work_threads = 1;                 // Setting.h:34
    int64_t bern = 0;                         // Setting.h:108
    KgIndex index;
    LibcRand libc;
    std::vector<uint64_t> streams;  // host view of next_random[] (Random.h:6)
    LcgJumpTable jump = make_jump_table();
    DeviceIndex dev;
    std::string last_error;
    int device_state = 0;  // 0 unknown, 1 ok, -1 none
    int record_emit_event = 0;  // 1: an event is recorded right behind every launch of the TransE emit kernel (kge_stream_wait_emit)
    hipEvent_t emit_done = nullptr;
    unsigned long long emit_seq = 0, emit_waited = 0;   // events recorded / the last one a side stream was made to wait for
    int time_emit = 0;          // N > 0: record HIP events around every N-th launch of the TransE emit kernel (kge_last_kernel_ms / kge_kernel_ms_mean)
    static constexpr int kEmitRing = 512;   // event pairs: the timed launches of a bench run are read back AFTER the run, no sync inside it
    hipEvent_t ev_emit0[kEmitRing] = {}, ev_emit1[kEmitRing] = {};
    long long emit_launches = 0;   // timed launches since time_emit was switched on
    long long emit_seen = 0;       // all launches since then (time_emit = N times every N-th)
    // per-row 1/|row| table of the vectorised TransE emit kernel.  inv_valid = 1: it holds the norms of the CURRENT contents of
    // inv_for_ent / inv_for_rel -- set by the pre-pass, kept by the full-table apply kernel
    // (which rewrites the entry of every row it changes), cleared by every other entry point that writes tables and by
    // kge_set_option("tables_changed") for writes the library cannot see (Config.set_parameters, restore, all-gathers)
    float *inv_norm = nullptr;
    int64_t inv_cap = 0;
    const float *inv_for_ent = nullptr, *inv_for_rel = nullptr;
    int inv_valid = 0;
    int32_t *loss_limbs = nullptr;   // kge_loss_limbs_target: where the TransE emit kernel also writes its loss as limbs (null = nowhere)
    int counts_krel = 4;        // dense TransE path: virtual copies of the relation rows in the record sort (1 = none; measured 1/2/4/8/16/64: 4 best); a power of two
    int inv_carry = 1;          // 0 = always recompute the table in front of the emit kernel (test hook)
    int64_t inv_table_max_bytes = int64_t(256) << 20;  // TransE emit: per-row inverse-norm table only while the tables are this small
    int float_records = 1;              // TransH / TransD (and TransE without counts): record + segmented-sum path instead of fp32 atomics
    int64_t float_records_min = 1 << 16; // ... from this many gradient rows per step (below it the atomic kernel alone is quicker)
    int64_t index_device_min = int64_t(1) << 22;  // training sets from this many lines on are indexed on the device (index_build.hip); < 0 = never
    int transr_dgrad_records = 1;      // TransR dgrad: entity-gradient rows as float records + segmented sum instead of fp32 atomics ...
    int64_t transr_dgrad_records_min = 1 << 15;   // ... from this many (scored triple, side) slots per step
    int transr_bf16x3 = 1;      // TransR row GEMMs (projection, dgrad): fp32 products as six bf16 x bf16 term products of an exact three-term split, on the bf16 matrix pipe (transr.hip rows_gemm3_kernel); 0 = the fp32 MFMA kernels
    int transr_groups = 1;      // TransR, device-sampled batches, 2 + n <= 16: groups (not jobs) sorted by relation, a group's rows side by side in one 16-row sub-tile (transr.hip GemmArgs); 1 = steps with well-filled buckets, 2 = always, 0 = never
    int transr_fuse_vec = 1;    // group layout: the vector stage inside the projection's epilogue (0: transr_vec_kernel on P)
    int transr_lean = 1;        // TransR vector stage: the float4-per-lane kernel that also zero-fills GP (0 = generic fwdbwd_kernel + memset)
    int pair_counts = 1;        // TransH / TransD: int8 sign records keyed by (entity, relation) + per-pair backward (pairs.hip) instead of float records
    int pair_counts_min_neg = 0;   // ... from this many negatives per positive; 0 = the measured cross-over (TransH 5, TransD 3)
    int hub_copies = 1;         // atomic TransH/TransD path: spread the relation-side rows over copies when a row takes >= 128 adds per step
    int lp_v1 = 0;              // test hook: link prediction through the generic predict kernel on materialised candidate batches
    int transr_v1 = 0;          // test hook: 1 = the 32x32x2 / 32-row-tile TransR kernels even where the v2 tiles apply; 2 = v2 with its all-tiles wgrad forced
    int fb_occ4 = 1;            // projecting models at <= 4 elements per lane: the forward/backward body compiled for four waves per SIMD
    int persist_ahead = 1;      // persistent launch: idle teams sample the next batch during the forward/backward phase
    int persist_touch = 0;      // persistent launch: a group's rows requested together before its dependent gathers (measured: no gain)
    int persist_trace = 0;      // measurement hook: the persistent launch stamps its phase boundaries (kge_persistent_trace)
    int persist_threads = 512;  // threads per workgroup of the persistent launch (512, or 1024: spills at its 128-register cap, measured slower)
    int counts_force_sort = 0;  // test hook: take the sort+segsum reduction even for small tables
    // percent of an armed sampler's workgroups riding in bkt_hist / bkt_scatter / bkt_sort / the launch that ends the step (byte 0..3).
    // Default: all of it in the scatter launch.  Spread 20/35/20/25 over the four it was SLOWER (their sum 68 -> 76 us per bench step):
    // a part of the sampler takes a full latency chain (~8 us) however small it is, and it did not hide behind the host kernels.
    int ride_shares = 100 << 8;
    int counts_fused = 1;       // kge_transe_train_step_counts: 1 = segmented sum and optimizer in one kernel (segapply_kernel); 0 = segsum + apply kernels
    int counts_fused_diag = 0;  // measurement hook, see SegApplyArgs::diag
    int counts_fused_cap = 0;   // test hook: rows of more than this many records go through the count image (0 = the kernel's capacity, 3 x team width)
};

Engine &engine();
// some table was (or may have been) written by a path that does not refresh the emit kernel's 1/|row| table: it is stale from now on
inline void tables_written() { engine().inv_valid = 0; }
// sampler.hip: the next batch's sampler riding in another kernel's launch (kge_sampling_attach)
struct SamplerArgs;
bool take_attached_sampler(SamplerArgs &a, unsigned &blocks, float share = 1.0f);   // true: `a` / `blocks` describe the part taken (sampler.hip)
int flush_attached_sampler(hipStream_t stream);                 // launches an armed sampler on its own
int attach_sampler(int32_t *d_h, int32_t *d_t, int32_t *d_r, int64_t B, int64_t neg, int64_t negrel, int64_t thread_lo,
                   int64_t thread_hi, int64_t out_stride, int64_t *n_local_out, hipStream_t stream);
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
bool device_ok();
// uploads index / rng streams if needed; returns KGE_OK or an error code
int ensure_device_index();
int hip_check(hipError_t e, const char *what);

constexpr int kMaxLossBlocks = 4096;

// How a negative relates to its positive.  A slot is "same" when the vector the score uses for it is
// the positive's: same entity AND same projection context (TransH/D: same relation; TransR: same
// matrix, which is the positive's whenever negative_rel == 0, TransR.py:57-60; TransE: no context).
// fast = exactly one of the three vectors differs -> the other two stay in registers.
struct NegClass { bool same_h, same_t, same_r, fast; };
template <int MODEL>
__host__ __device__ inline NegClass classify_negative(long long h, long long t, long long r, long long nh,
                                                               long long nt, long long nr, int negative_rel) {
    NegClass c;
    c.same_r = nr == r;
    const bool ctx_same = MODEL == KGE_TRANSE ? true : (MODEL == KGE_TRANSR ? (negative_rel == 0 || c.same_r) : c.same_r);
    c.same_h = nh == h && ctx_same;
    c.same_t = nt == t && ctx_same;
    c.fast = ((int)!c.same_h + (int)!c.same_t + (int)!c.same_r) == 1;
    return c;
}



// float-record reduction (transe_counts.hip): destination row space = up to four tables back to back
struct FloatRowSpace {   // virtual row space of models.hip's FbArgs::frec records
    float *g_ent, *g_rel, *g_auxr, *g_auxe;
    long long E, R;             // entity rows [0,E), ent_transfer rows [E, hub_base) when hub_base == 2E
    long long hub_base, hub_rows;   // copies of { rel [R] | auxr [R] } from hub_base on, hub_rows rows per copy
    long long rows;             // total virtual rows
    float scale = 1.0f;         // a run's sum is added as scale * sum (-lr with the PARAMETER tables as targets: sparse-row SGD in place)
};
bool pair_path_active(const kge_model_desc &m, int64_t n_pos, int64_t n_neg);
int sgd_rows_skipped(int32_t *out);
int float_records_workspace(int64_t M, int D, float *&rec, int32_t *&dst);
// rec_ext / dst_ext: records held by the caller (the gathered records of a data-parallel step) instead of the workspace's
// deterministic: stable sort + one team per run + ordered fold of the hub copies (bit-identical on every rank that reduces the same records)
int float_records_reduce(int64_t M, int D, const FloatRowSpace &rs, hipStream_t stream, const float *rec_ext = nullptr, int32_t *dst_ext = nullptr,
                         bool deterministic = false);
int launch_forward_backward_records(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                                    const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride, int64_t denom, int64_t n_pos_total,
                                    float *d_rec, int32_t *d_dst, int64_t rec_offset, int64_t rec_slice, float *d_loss, hipStream_t stream);
int launch_float_records_apply(const kge_model_desc &m, float *const tables[4], const float *d_rec, int32_t *d_dst, int64_t M_total,
                               int64_t n_pos_total, int64_t n_neg, float lr, hipStream_t stream);

// device-side index build (index_build.hip)
bool device_index_build_supported(int64_t E, int64_t R, int64_t n);
std::string build_index_device(KgIndex &ix, DeviceIndex &dev, int64_t E, int64_t R, int64_t new_batch, int64_t n, const int64_t *h,
                               const int64_t *t, const int64_t *r);

// ---- launchers implemented in the .hip files ------------------------------------------------
int launch_sampler(int32_t *d_h, int32_t *d_t, int32_t *d_r, int64_t B, int64_t neg, int64_t negrel, int64_t thread_lo,
                   int64_t thread_hi, int64_t out_stride, int64_t *n_local, hipStream_t stream);
int launch_widen(const int32_t *src3, int64_t *dst3_and_y, int64_t B, int64_t total, hipStream_t stream);
int launch_forward_backward(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                            const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride, int64_t denom,
                            float *const grads[4], float *d_loss, hipStream_t stream, bool sampler_shaped = false, float inplace_lr = 0.f);
int launch_predict(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                   const int32_t *d_r, int64_t n, float *d_out, hipStream_t stream);
int launch_lp_table(const kge_model_desc &m, const float *const tables[4], const float *P_all, int64_t r, float *T, hipStream_t stream);
int launch_lp_scores(const kge_model_desc &m, const float *const tables[4], const float *T, int64_t r, const int32_t *d_req_fixed,
                     const int32_t *d_req_head, int64_t n_req, float *d_scores, hipStream_t stream);
int transr_project_all(const kge_model_desc &m, const float *const tables[4], int64_t r, float *P_out, hipStream_t stream);
int launch_sgd(float *p, float *g, int64_t n, float lr, hipStream_t stream);
int launch_sgd_tables(int n_tables, float *const *p, float *const *g, const int64_t *numel, float lr, hipStream_t stream);
int launch_adam_tables(int n_tables, float *const *p, float *const *m, float *const *v, float *const *g, const int64_t *numel,
                       float lr_t, float b1, float b2, float eps, hipStream_t stream);
int launch_adam(float *p, float *m, float *v, float *g, int64_t n, float lr_t, float b1, float b2, float eps,
                hipStream_t stream);

}  // namespace kge
