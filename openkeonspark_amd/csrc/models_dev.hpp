// Device-side building blocks of the vector models (TransE / TransH / TransD, and TransR's vector stage): argument
// block, loss hand-off, entity-side / relation-context forward and backward, and the per-group forward/backward body.
// Shared by models.hip (one launch per stage) and persist.hip (many training steps inside one persistent launch).
#pragma once
#include "engine.hpp"
#include "team.hpp"

namespace kge {

struct FbArgs {
    const float *ent, *rel, *auxr, *auxe;  // tables
    float *g_ent, *g_rel, *g_auxr, *g_auxe;  // dense gradient accumulators
    const int32_t *bh, *bt, *br;
    long long n_pos, n_neg, stride;
    int D;
    float margin, unit;
    float *loss_partials;
    // TransR vector stage (transr.hip): entity sides are rows of the projected buffer P (one row per
    // canonical (scored triple, side) slot) and their gradients are STORED to GP, not added to g_ent
    const float *P;
    float *GP;
    int negative_rel;
    // TransE sign-count path (transe_counts.hip): int8 gradient records + destination rows
    uint32_t *rec;
    int32_t *dst;
    // rec2 != nullptr (the fused single-process step, transe_counts.hip segapply_kernel): the record of NEGATIVE slot 3 + k of group
    // b is not an int8 record in `rec` but a 2-bit one here -- record (k * n_pos + b), one BYTE per lane and float4 chunk (L * Q
    // bytes per record), element j of the lane's four as the 2-bit field (sign + 1) in {0, 1, 2} at bits 2j.  A negative's
    // gradient w.r.t. its one new normalised vector is +-sign(e): two bits per element say it all, and the negatives are 25 of
    // the 28 records of a bench group (64 B instead of 256 B each).  Slots 0..2 (sums of up to 2 n signs) stay int8 in `rec`.
    // In this mode the destination KEYS are 2 * row (int8 records) and 2 * row + 1 (2-bit records): ordered by key, a row's
    // records come as two homogeneous lists and the reducer needs no test per record.
    uint8_t *rec2;
    int ent_total, rel_total, krel;
    // indirection for the deferred groups of the sign-count path: when group_list != nullptr the
    // kernel walks group_list[0 .. *group_count) instead of 0 .. n_pos
    int32_t *group_list;
    int32_t *group_count;
    const float *inv_norm;  // [E+R] 1/|row| of ent_embeddings then rel_embeddings, refreshed per step (vectorised emit)
    // float-record path (TransH / TransD, float_records in transe_counts.hip): instead of an atomic row add, a
    // gradient row is STORED as record m = slot*n_pos + b with its destination in the VIRTUAL row space
    //   ent [0,E) | ent_transfer [E,2E) (TransD) | hub_k copies of { rel [R] | normal vectors / rel_transfer [R] }
    // group b writes its relation-side rows into copy b % hub_k, so that the records of a hub relation (WN18RR has
    // 11 relations) spread over many sort buckets; the segmented sum folds the copies back onto the real row
    float *loss_out;        // see finish_loss
    unsigned *loss_ticket;
    float *frec;
    int32_t *fdst;
    long long hub_base;   // first row of copy 0
    int hub_k, hub_rows;  // copies, rows per copy (R or 2R)
    // atomic path on a KG with few relations (WN18RR: 11): thousands of groups per step add into the same R rows and
    // same-address atomics serialise; group b adds into copy b % hub_k of [hub_k][R][D] buffers, folded afterwards
    float *copies_rel, *copies_auxr;
    // pair-count path (pairs.hip): per record the projection coefficient a and 1/|projected| of its (entity, relation) pair
    // (negative 1/|.| = the norm was clipped), so that the per-pair backward need not recompute them
    float2 *pair_aux;
    // row-wise SGD in place (kge_forward_backward_sgd_rows): the accumulators ARE the parameter tables, so a negative that is not a
    // single-slot corruption (its exact path adds rows atomically) must not run: it is skipped and counted here
    int32_t *skipped;
    // data-parallel TransE count path: the loss also goes out as four 16-bit limbs of a 2^-32 fixed-point value, into the spare tail
    // slot of the int32 count image that the reduce-scatter sums (kge_loss_limbs_target; same encoding as kge_loss_to_limbs)
    int32_t *loss_limbs;
    // a launch whose grid carries workgroups of ANOTHER job behind its own (the next batch's sampler riding along, models.hip
    // fwdbwd_ride_kernel): the number of workgroups that are this kernel's -- what the loss ticket and the group loop count with
    // instead of gridDim.x; 0 = the whole grid
    int loss_blocks;
};

int ensure_loss_buffers();
void guard_loss_stream(hipStream_t stream);   // loss partials / tickets are process-global: serialise launches across streams

// Per-block partial hinge sums -> loss = sum / denom (TransE.py:51).  With a.loss_out set, the LAST block to
// finish (ticket counter) adds the partials in the fixed order loss_finalize_kernel uses and writes the loss,
// so no separate launch is needed; otherwise the partials are left for loss_finalize_kernel.
// (finish_loss_sh: the 256-float scratch of the last workgroup's reduction is the caller's -- a kernel whose LDS budget has no
// kilobyte to spare hands in a buffer it is done with)
template <int TEAMS>
__device__ __forceinline__ void finish_loss_sh(const FbArgs &a, float *red, float lsum, int lane, int team_in_block, float *sh) {
    __shared__ int is_last;
    const unsigned nblk = a.loss_blocks ? (unsigned)a.loss_blocks : gridDim.x;   // this kernel's own workgroups
    if (lane == 0) red[team_in_block] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TEAMS; i++) s += red[i];
        is_last = 0;
        if (a.loss_out) {
            // HARDWARE ASSUMPTION (gfx950; not promised by the HIP / LLVM memory model): relaxed agent-scope atomics are
            // performed at the memory side and are therefore coherent across XCDs by themselves, and a returning atomic
            // followed by s_waitcnt vmcnt(0) has been performed before the ticket add that follows it is issued.  The
            // buffers are process-global: launches are serialised across streams by guard_loss_stream (models.hip).  Every
            // loss value of the GPU test-suite (hundreds of launches against the oracle at 1e-5) goes through this path.
            // No fences: an agent-scope release fence writes back the whole XCD L2 on gfx950 (measured: the emit
            // kernel went from 119 to 237 us).  Memory-side atomics are coherent across XCDs by themselves: publish
            // the partial with a RETURNING exchange, wait for it, then take a ticket.
            unsigned *slot = reinterpret_cast<unsigned *>(a.loss_partials) + blockIdx.x;
            (void)__hip_atomic_exchange(slot, __float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the exchange has been performed
            // Two-level ticket: same-address memory-side atomics serialise at ~8 ns each, so thousands of blocks on ONE
            // counter cost more than the kernel they replace (measured +17 us on a 22 us kernel); 32 blocks share a
            // sub-counter (loss_ticket[1 + group]) and only the last of each group touches the master (loss_ticket[0]).
            const unsigned group = blockIdx.x >> 5, n_groups = (nblk + 31) >> 5;
            const unsigned in_group = min(32u, nblk - (group << 5));
            if (__hip_atomic_fetch_add(a.loss_ticket + 1 + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1) {
                __hip_atomic_store(a.loss_ticket + 1 + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                is_last = __hip_atomic_fetch_add(a.loss_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1 ? 1 : 0;
            }
        } else {
            a.loss_partials[blockIdx.x] = s;
        }
    }
    if (!a.loss_out) return;
    __syncthreads();
    if (!is_last) return;
    const unsigned *part = reinterpret_cast<const unsigned *>(a.loss_partials);
    float s = 0.f;   // same fixed order as loss_finalize_kernel
    for (int i = threadIdx.x; i < (int)nblk; i += 256)
        s += __uint_as_float(__hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.loss_out[0] = sh[0] * a.unit;
        if (a.loss_limbs) {
            const double x = (double)(sh[0] * a.unit) * 4294967296.0;
            const long long v = x >= 0.0 ? (long long)(x + 0.5) : 0;
            for (int i = 0; i < 4; i++) a.loss_limbs[i] = (int32_t)((v >> (16 * i)) & 0xFFFF);
        }
        __hip_atomic_store(a.loss_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int TEAMS>
__device__ __forceinline__ void finish_loss(const FbArgs &a, float *red, float lsum, int lane, int team_in_block) {
    __shared__ float sh[256];
    finish_loss_sh<TEAMS>(a, red, lsum, lane, team_in_block, sh);
}

// one gradient row: atomic add into the dense accumulator, or (REC, m >= 0) a plain 4*D-byte record store
template <bool REC, int L, int C>
__device__ __forceinline__ void put_row(const Team<L, C> &tm, const FbArgs &a, float *gtab, long long row, long long vrow, long long m,
                                        const float (&v)[C]) {
    if (REC && m >= 0) {
        float *p = a.frec + m * a.D;
#pragma unroll
        for (int c = 0; c < C; c++) { const int e = tm.lane + L * c; if (e < a.D) p[e] = v[c]; }
        if (tm.lane == 0) a.fdst[m] = (int32_t)vrow;
    } else {
        tm.add(gtab, row, v);
    }
}

// ---- integer sign vectors (sign-count paths: TransE emit in models.hip, pair-count emit in pairs.hip) ----
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int sign_of_bits(float e) {
    int r;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(__builtin_bit_cast(int, e)));
    return r;
}
__device__ __forceinline__ s16x2 pack16(int lo, int hi) { s16x2 r; r.x = (short)lo; r.y = (short)hi; return r; }
// the four low bytes of two int16 pairs -> one record dword
__device__ __forceinline__ uint32_t bytes_of(s16x2 lo, s16x2 hi) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
}

// One entity side (h or t slot) of a scored triple: raw row(s), projected+normalised vector.
template <int C>
struct Side {
    float raw[C];  // ent_embeddings row          (unused after projection for TransE)
    float aux[C];  // ent_transfer row             (TransD only)
    float nrm[C];  // l2_normalize(projected)
    float inv, a;  // 1/|projected| , projection coefficient (e.w  or  e.e_p)
    bool uc;
};

// Projection onto the relation context (rows already in s.raw / s.aux).  TransH.py:12-14: e - (e.w^)w^ ; TransD.py:23-25: e + (e.e_p) r_p
template <int MODEL, int L, int C>
__device__ __forceinline__ void side_project(const Team<L, C> &tm, const float (&cw)[C], Side<C> &s) {
    float xp[C];
    if constexpr (MODEL == KGE_TRANSE || MODEL == KGE_TRANSR) {
#pragma unroll
        for (int c = 0; c < C; c++) xp[c] = s.raw[c];
        s.a = 0.f;
    } else if constexpr (MODEL == KGE_TRANSH) {
        s.a = tm.dot(s.raw, cw);
#pragma unroll
        for (int c = 0; c < C; c++) xp[c] = s.raw[c] - s.a * cw[c];
    } else {
        s.a = tm.dot(s.raw, s.aux);
#pragma unroll
        for (int c = 0; c < C; c++) xp[c] = s.raw[c] + s.a * cw[c];
    }
    tm.normalize(xp, s.nrm, s.inv, s.uc);
}

template <int MODEL, int L, int C>
__device__ __forceinline__ void side_forward(const Team<L, C> &tm, const FbArgs &a, long long row, const float (&cw)[C],
                                             Side<C> &s) {
    if constexpr (MODEL == KGE_TRANSR) tm.load(a.P, row, s.raw);  // row = slot of the projected buffer
    else tm.load(a.ent, row, s.raw);
    if constexpr (MODEL == KGE_TRANSD) tm.load(a.auxe, row, s.aux);
    side_project<MODEL, L, C>(tm, cw, s);
}

// Backward of one entity side given G = dL/d(normalised projected vector).  Adds the row gradient(s)
// and accumulates the relation-context gradient into acw (TransH: d/dw^, TransD: d/dr_p).
// m = record index of the entity row (TransD: its transfer row is the next SLOT, m + n_pos); -1 = atomic add
template <int MODEL, int L, int C, bool REC = false>
__device__ __forceinline__ void side_backward(const Team<L, C> &tm, const FbArgs &a, long long row, const Side<C> &s,
                                              const float (&G)[C], const float (&cw)[C], float (&acw)[C], long long m = -1) {
    float gxp[C];
    tm.normalize_bwd(s.nrm, G, s.inv, s.uc, gxp);
    if constexpr (MODEL == KGE_TRANSE) {
        put_row<REC>(tm, a, a.g_ent, row, row, m, gxp);
    } else if constexpr (MODEL == KGE_TRANSR) {
        tm.store(a.GP, row, gxp);  // each canonical slot is written by exactly one team
    } else if constexpr (MODEL == KGE_TRANSH) {
        float d = tm.dot(gxp, cw);
        float gx[C];
#pragma unroll
        for (int c = 0; c < C; c++) { gx[c] = gxp[c] - d * cw[c]; acw[c] -= d * s.raw[c] + s.a * gxp[c]; }
        put_row<REC>(tm, a, a.g_ent, row, row, m, gx);
    } else {
        float d = tm.dot(gxp, cw);
        float gx[C], gv[C];
#pragma unroll
        for (int c = 0; c < C; c++) { gx[c] = gxp[c] + d * s.aux[c]; gv[c] = d * s.raw[c]; acw[c] += s.a * gxp[c]; }
        put_row<REC>(tm, a, a.g_ent, row, row, m, gx);
        put_row<REC>(tm, a, a.g_auxe, row, (long long)a.ent_total + row, m >= 0 ? m + a.n_pos : -1, gv);
    }
}

// Relation context of a triple: normalised relation vector + projection vector.
template <int C>
struct Ctx {
    float rn[C];  // l2_normalize(rel_embeddings[r])
    float cw[C];  // TransH: l2_normalize(normal_vectors[r]); TransD: rel_transfer[r]
    float inv_r, inv_w;
    bool uc_r, uc_w;
};

template <int MODEL, int L, int C>
__device__ __forceinline__ void ctx_forward(const Team<L, C> &tm, const FbArgs &a, long long r, Ctx<C> &cx) {
    float raw[C];
    tm.load(a.rel, r, raw);
    tm.normalize(raw, cx.rn, cx.inv_r, cx.uc_r);
    cx.inv_w = 1.f; cx.uc_w = true;
    if constexpr (MODEL == KGE_TRANSH) {
        tm.load(a.auxr, r, raw);
        tm.normalize(raw, cx.cw, cx.inv_w, cx.uc_w);
    } else if constexpr (MODEL == KGE_TRANSD) {
        tm.load(a.auxr, r, cx.cw);
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) cx.cw[c] = 0.f;
    }
}

// Adds the relation-side gradients: Gr = dL/d rn, acw = accumulated dL/d cw.
// m = record index of the rel_embeddings row; the context row (normal vector / rel_transfer) is the next slot
template <int MODEL, int L, int C, bool REC = false>
__device__ __forceinline__ void ctx_backward(const Team<L, C> &tm, const FbArgs &a, long long r, const Ctx<C> &cx,
                                             const float (&Gr)[C], const float (&acw)[C], long long m = -1, long long hub = 0) {
    // hub = first virtual row of this group's copy of the relation-side rows
    float g[C];
    tm.normalize_bwd(cx.rn, Gr, cx.inv_r, cx.uc_r, g);
    // atomic path with hub copies: `hub` is the copy index
    float *grel = (!REC && a.copies_rel) ? a.copies_rel + hub * (long long)a.rel_total * a.D : a.g_rel;
    float *gaux = (!REC && a.copies_auxr) ? a.copies_auxr + hub * (long long)a.rel_total * a.D : a.g_auxr;
    put_row<REC>(tm, a, grel, r, hub + r, m, g);
    const long long m2 = m >= 0 ? m + a.n_pos : -1;
    if constexpr (MODEL == KGE_TRANSH) {
        tm.normalize_bwd(cx.cw, acw, cx.inv_w, cx.uc_w, g);
        put_row<REC>(tm, a, gaux, r, hub + a.rel_total + r, m2, g);
    } else if constexpr (MODEL == KGE_TRANSD) {
        put_row<REC>(tm, a, gaux, r, hub + a.rel_total + r, m2, acw);
    }
}

// slot layout of the float-record path (record m = slot*n_pos + b):
//   TransE: h, t, r, then one slot per negative            (3 + n)
//   TransH: h, t, r, w, then one slot per negative          (4 + n)
//   TransD: h, h_p, t, t_p, r, r_p, then two per negative   (6 + 2n)
template <int MODEL> struct RecSlots {
    static constexpr int ent_w = MODEL == KGE_TRANSD ? 2 : 1;                 // slots per entity side
    static constexpr int h = 0, t = ent_w, r = 2 * ent_w;
    static constexpr int group = 2 * ent_w + (MODEL == KGE_TRANSE ? 1 : 2);   // slots of the positive's shared rows
};

template <int L, int C>
__device__ __forceinline__ float l1_score(const Team<L, C> &tm, const float (&hn)[C], const float (&rn)[C],
                                          const float (&tn)[C], float (&sg)[C]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) { float e = hn[c] + rn[c] - tn[c]; s += fabsf(e); sg[c] = sgn(e); }
    return team_sum<L>(s);
}

// A negative that the fast path cannot use (more than one slot differs, or the relation differs
// for a projecting model): score it from scratch and, if its hinge is active, add all its gradients
// separately (gs = -unit).  Kept frugal in registers -- the entity sides are re-gathered for the
// backward instead of being held -- because this rare path must not cost the common one occupancy.
template <int MODEL, int L, int C>
__device__ __forceinline__ bool standalone_negative(const Team<L, C> &tm, const FbArgs &a, long long nh, long long nt,
                                                    long long nr, float p, float &hinge) {
    // nh / nt are ROW HANDLES (entity ids, or projected-buffer slots for TransR); nr the relation id
    Ctx<C> cx;
    ctx_forward<MODEL, L, C>(tm, a, nr, cx);
    float sg[C];
    float nk;
    {
        float hn[C];
        {
            Side<C> sx;
            side_forward<MODEL, L, C>(tm, a, nh, cx.cw, sx);
#pragma unroll
            for (int c = 0; c < C; c++) hn[c] = sx.nrm[c];
        }
        Side<C> sx;
        side_forward<MODEL, L, C>(tm, a, nt, cx.cw, sx);
        nk = l1_score<L, C>(tm, hn, cx.rn, sx.nrm, sg);
    }
    float v = p - nk + a.margin;
    if (!(v >= 0.f)) { hinge = 0.f; return false; }
    hinge = v;
    float G[C], acw[C];
#pragma unroll
    for (int c = 0; c < C; c++) { G[c] = -a.unit * sg[c]; acw[c] = 0.f; }
    {
        Side<C> sx;
        side_forward<MODEL, L, C>(tm, a, nh, cx.cw, sx);
        side_backward<MODEL, L, C>(tm, a, nh, sx, G, cx.cw, acw);
    }
    {
        Side<C> sx;
        side_forward<MODEL, L, C>(tm, a, nt, cx.cw, sx);
        float Gt[C];
#pragma unroll
        for (int c = 0; c < C; c++) Gt[c] = -G[c];
        side_backward<MODEL, L, C>(tm, a, nt, sx, Gt, cx.cw, acw);
    }
    ctx_backward<MODEL, L, C>(tm, a, nr, cx, G, acw);
    return true;
}

// team-uniform broadcast of lane `src` (index inside the team)
template <int L>
__device__ __forceinline__ int team_bcast(int v, int src) {
    if constexpr (L == 64) return __builtin_amdgcn_readlane(v, src);  // wave == team: uniform result in an SGPR
    else return __shfl(v, src, L);
}

// One positive and all its negatives: forward, hinge, backward, gradient rows out (atomic adds or float records).
template <int MODEL, int L, int C, bool REC>
__device__ __forceinline__ void fwdbwd_group(const Team<L, C> &tm, const FbArgs &a, const long long b, float &lsum) {
    using RS = RecSlots<MODEL>;
    const long long h = a.bh[b], t = a.bt[b], r = a.br[b];
    const long long hub = REC ? a.hub_base + (long long)(b % a.hub_k) * a.hub_rows : (a.copies_rel ? b % a.hub_k : 0);
    Ctx<C> cx;
    ctx_forward<MODEL, L, C>(tm, a, r, cx);
    // row handles of the two entity sides: the entity id, or (TransR) the slot of the projected
    // vector of (scored triple s, side): 2*s + side with s = k*n_pos + b
    const long long row_h = MODEL == KGE_TRANSR ? 2 * b : h;
    const long long row_t = MODEL == KGE_TRANSR ? 2 * b + 1 : t;
    Side<C> sh, st;
    side_forward<MODEL, L, C>(tm, a, row_h, cx.cw, sh);
    side_forward<MODEL, L, C>(tm, a, row_t, cx.cw, st);
    float sp[C];
    const float p = l1_score<L, C>(tm, sh.nrm, cx.rn, st.nrm, sp);
    // gradients w.r.t. the three shared normalised vectors, in units of `unit`
    float Ah[C], At[C], Ar[C], acw[C];
#pragma unroll
    for (int c = 0; c < C; c++) { Ah[c] = 0.f; At[c] = 0.f; Ar[c] = 0.f; acw[c] = 0.f; }
    int cnt = 0;
    for (long long k = 0; k < a.n_neg; k++) {
        const long long j = b + (k + 1) * a.stride;
        const long long nh = a.bh[j], nt = a.bt[j], nr = a.br[j];
        const NegClass nc = classify_negative<MODEL>(h, t, r, nh, nt, nr, a.negative_rel);
        const long long s_neg = (k + 1) * a.n_pos + b;
        const long long nrow_h = MODEL == KGE_TRANSR ? 2 * s_neg : nh;
        const long long nrow_t = MODEL == KGE_TRANSR ? 2 * s_neg + 1 : nt;
        const long long m_neg = REC ? (RS::group + RS::ent_w * k) * a.n_pos + b : -1;   // this negative's record(s)
        bool wrote = false;
        if (!nc.fast) {
            float hinge;
            if (a.skipped) { if (tm.lane == 0) atomicAdd(a.skipped, 1); }
            else if (standalone_negative<MODEL, L, C>(tm, a, nrow_h, nrow_t, nr, p, hinge)) { cnt++; lsum += hinge; }
            if (REC && tm.lane == 0) {
#pragma unroll
                for (int w = 0; w < RS::ent_w; w++) a.fdst[m_neg + w * a.n_pos] = -1;
            }
            continue;
        }
        float sg[C];
        if (!nc.same_h) {  // head corrupted (corrupt_tail keeps t, Base.cpp:123-126)
            Side<C> sx;
            side_forward<MODEL, L, C>(tm, a, nrow_h, cx.cw, sx);
            float nk = l1_score<L, C>(tm, sx.nrm, cx.rn, st.nrm, sg);
            float v = p - nk + a.margin;
            if (v >= 0.f) {
                cnt++; lsum += v;
                float G[C];
#pragma unroll
                for (int c = 0; c < C; c++) { G[c] = -a.unit * sg[c]; At[c] += sg[c]; Ar[c] -= sg[c]; }
                side_backward<MODEL, L, C, REC>(tm, a, nrow_h, sx, G, cx.cw, acw, m_neg);
                wrote = true;
            }
        } else if (!nc.same_t) {  // tail corrupted (corrupt_head keeps h, Base.cpp:119-121)
            Side<C> sx;
            side_forward<MODEL, L, C>(tm, a, nrow_t, cx.cw, sx);
            float nk = l1_score<L, C>(tm, sh.nrm, cx.rn, sx.nrm, sg);
            float v = p - nk + a.margin;
            if (v >= 0.f) {
                cnt++; lsum += v;
                float G[C];
#pragma unroll
                for (int c = 0; c < C; c++) { G[c] = a.unit * sg[c]; Ah[c] -= sg[c]; Ar[c] -= sg[c]; }
                side_backward<MODEL, L, C, REC>(tm, a, nrow_t, sx, G, cx.cw, acw, m_neg);
                wrote = true;
            }
        } else {  // relation vector corrupted while both projected entities are shared (TransE; TransR with negative_rel == 0)
            float raw[C], xn[C], inv; bool uc;
            tm.load(a.rel, nr, raw);
            tm.normalize(raw, xn, inv, uc);
            float nk = l1_score<L, C>(tm, sh.nrm, xn, st.nrm, sg);
            float v = p - nk + a.margin;
            if (v >= 0.f) {
                cnt++; lsum += v;
                float G[C], g[C];
#pragma unroll
                for (int c = 0; c < C; c++) { G[c] = -a.unit * sg[c]; Ah[c] -= sg[c]; At[c] += sg[c]; }
                tm.normalize_bwd(xn, G, inv, uc, g);
                put_row<REC>(tm, a, a.g_rel, nr, hub + nr, m_neg, g);
                wrote = true;
            }
        }
        if (REC && !wrote && tm.lane == 0) {   // hinge inactive: no record in this negative's slot(s)
#pragma unroll
            for (int w = 0; w < RS::ent_w; w++) a.fdst[m_neg + w * a.n_pos] = -1;
        }
    }
    if (REC && cnt == 0 && tm.lane == 0) {
#pragma unroll
        for (int sl = 0; sl < RS::group; sl++) a.fdst[sl * a.n_pos + b] = -1;
    }
    if (cnt > 0) {
        const float fc = (float)cnt;
        float Gh[C], Gt[C], Gr[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            Gh[c] = a.unit * (Ah[c] + fc * sp[c]);
            Gt[c] = a.unit * (At[c] - fc * sp[c]);
            Gr[c] = a.unit * (Ar[c] + fc * sp[c]);
        }
        side_backward<MODEL, L, C, REC>(tm, a, row_h, sh, Gh, cx.cw, acw, REC ? RS::h * a.n_pos + b : -1);
        side_backward<MODEL, L, C, REC>(tm, a, row_t, st, Gt, cx.cw, acw, REC ? RS::t * a.n_pos + b : -1);
        ctx_backward<MODEL, L, C, REC>(tm, a, r, cx, Gr, acw, REC ? RS::r * a.n_pos + b : -1, hub);
    }
}

template <int MODEL, int L, int C, bool REC>
__device__ __forceinline__ void fwdbwd_body(const FbArgs &a);

template <int MODEL, int L, int C, bool REC = false>
__global__ __launch_bounds__(256) void fwdbwd_kernel(FbArgs a) { fwdbwd_body<MODEL, L, C, REC>(a); }

// the same body compiled for four waves per SIMD (<= 128 VGPRs): the projecting models at C <= 4 are latency-bound on
// their reduction chains, and a fourth resident wave hides more of it than the registers it gives up cost
template <int MODEL, int L, int C, bool REC = false>
__global__ __launch_bounds__(256, 4) void fwdbwd_kernel_occ4(FbArgs a) { fwdbwd_body<MODEL, L, C, REC>(a); }

template <int MODEL, int L, int C, bool REC>
__device__ __forceinline__ void fwdbwd_body(const FbArgs &a) {
    constexpr int TEAMS = 256 / L;
    __shared__ float red[TEAMS];
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    const int team_in_block = threadIdx.x / L;
    float lsum = 0.f;
    const long long n_groups = a.group_list ? (long long)a.group_count[0] : a.n_pos;
    const long long own_blocks = a.loss_blocks ? a.loss_blocks : (long long)gridDim.x;
    for (long long gi = (long long)blockIdx.x * TEAMS + team_in_block; gi < n_groups; gi += own_blocks * TEAMS) {
        const long long b = a.group_list ? (long long)a.group_list[gi] : gi;
        fwdbwd_group<MODEL, L, C, REC>(tm, a, b, lsum);
    }
    finish_loss<TEAMS>(a, red, lsum, tm.lane, team_in_block);
}

}  // namespace kge
