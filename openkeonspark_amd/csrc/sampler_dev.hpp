// Device side of the negative sampler shared by sampler.hip (one launch per batch) and persist.hip (sampling inside a
// persistent multi-step launch): the 64-bit LCG with jump-ahead (Random.h:16-34), exact modulo without 64-bit division,
// the closed form of the filtered pick (Corrupt.h:25-36), and one scored triple of a batch (Base.cpp:95-140).
#pragma once
#include "engine.hpp"

namespace kge {

static __constant__ LcgJumpTable c_jump;     // one copy per translation unit that samples (sampler.hip, persist.hip)
static bool g_jump_uploaded = false;

static inline int upload_jump_table() {
    if (g_jump_uploaded) return KGE_OK;
    int rc = hip_check(hipMemcpyToSymbol(HIP_SYMBOL(c_jump), &engine().jump, sizeof(LcgJumpTable)), "upload jump table");
    if (rc == KGE_OK) g_jump_uploaded = true;
    return rc;
}

struct SamplerArgs {
    const int4 *pos;
    const int4 *grp;
    const int2 *ht;
    const int32_t *tails_hr, *heads_tr, *rels_ht;
    const float *bern_prob;
    const uint64_t *streams;
    uint64_t *streams_next;   // sample_kernel: every stream's state after this batch (null: not written)
    long long W, B;           // virtual threads, global batch (for streams_next)
    int32_t *out_h, *out_t, *out_r;
    long long per_thread;  // positions per virtual thread: B/W, or B/W+1 when W does not divide B
    long long pos_lo;      // first global batch position written by this launch
    long long n_local;     // positions written by this launch
    long long out_stride;
    long long train_dup, new_batch;
    unsigned long long pick_div, pick_magic;   // divisor of the positive pick and floor((2^64-1)/divisor)
    int ent_total, rel_total;
    int neg, negrel, bern;
    int kshift;            // log2 of the lane slots per positive
    // a PART of the sampler's grid riding in another kernel's launch (take_attached_sampler): the rider's extra workgroup i runs
    // workgroup ride_first + i of ride_total
    unsigned ride_first, ride_total;
};

__device__ __forceinline__ uint64_t lcg_step(uint64_t s) { return s * kLcgMul + kLcgAdd; }

__device__ __forceinline__ uint64_t lcg_skip(uint64_t s, uint64_t n) {
    for (int j = 0; n != 0; ++j, n >>= 1)
        if (n & 1) s = c_jump.mulA[j] * s + c_jump.addC[j];
    return s;
}

// s % d for a 64-bit LCG state, exact, without the 64-bit division sequence (~100 instructions each, three per thread):
// d known on the host -> multiply-high by m = floor((2^64-1)/d), at most two corrections
__device__ __forceinline__ uint64_t mod_magic(uint64_t s, uint64_t d, uint64_t m) {
    uint64_t r = s - __umul64hi(s, m) * d;
    while (r >= d) r -= d;
    return r;
}
// d < 2^31 known only per thread: two rounds of fp64 reciprocal division; each quotient is < 2^32, so the fp64
// estimate is within one of the truth and one correction step each makes it exact
__device__ __forceinline__ uint32_t mod_u64_u32(uint64_t s, uint32_t d) {
    if (d == 0) return 0;   // a group that already contains every candidate: the reference divides by zero (SIGFPE) here
    const double rcp = 1.0 / (double)d;
    const uint32_t hi = (uint32_t)(s >> 32), lo = (uint32_t)s;
    uint32_t q1 = (uint32_t)((double)hi * rcp);
    int64_t r1 = (int64_t)hi - (int64_t)q1 * d;
    if (r1 < 0) r1 += d;
    if (r1 >= (int64_t)d) r1 -= d;
    const uint64_t x = ((uint64_t)r1 << 32) | lo;                  // < d * 2^32
    const double xd = (double)(uint32_t)r1 * 4294967296.0 + (double)lo;
    uint64_t q2 = (uint64_t)(xd * rcp);
    int64_t r2 = (int64_t)(x - q2 * d);
    if (r2 < 0) r2 += d;
    if (r2 < 0) r2 += d;
    if (r2 >= (int64_t)d) r2 -= d;
    if (r2 >= (int64_t)d) r2 -= d;
    return (uint32_t)r2;
}

// Corrupt.h:25-36 in closed form: the tmp-th id (0-based) that is NOT in the strictly increasing
// list vals[0..len) is tmp + #{j : vals[j] - j <= tmp}; the predicate is monotone in j.
__device__ __forceinline__ int filtered_pick(const int32_t *__restrict__ vals, int len, long long tmp) {
    int lo = 0, hi = len;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((long long)vals[mid] - mid <= tmp) lo = mid + 1; else hi = mid;
    }
    return (int)(tmp + lo);
}

// The same pick with ONE memory round trip for groups of up to four known ids (most (h, r) / (t, r) groups of a sparse KG):
// the four candidates are requested together (clamped, unconditional) and the monotone predicate is counted; longer lists
// fall through to the binary search.
__device__ __forceinline__ int filtered_pick_short(const int32_t *__restrict__ vals, int len, long long tmp) {
    if (len > 4) return filtered_pick(vals, len, tmp);
    if (len <= 0) return (int)tmp;
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = vals[j < len ? j : len - 1];
    int lo = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) lo += (j < len && (long long)v[j] - j <= tmp) ? 1 : 0;
    return (int)(tmp + lo);
}

// s advanced by n < 2^16 steps, n different in every lane: a masked multiply-add per bit with the first jump entries read
// through compile-time offsets (scalar loads the compiler batches), as many rounds as the widest n of the wave needs
__device__ __forceinline__ uint64_t lcg_skip_lanes(uint64_t s, unsigned n) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (!__any((n >> j) != 0u)) break;
        const uint64_t t = c_jump.mulA[j] * s + c_jump.addC[j];
        s = ((n >> j) & 1u) ? t : s;
    }
    return s;
}

// One scored triple of the batch: slot k of the positive at global batch position p (k = 0 the positive, 1..neg entity
// negatives, then relation negatives), drawn exactly as virtual thread `id` of the reference draws it (Base.cpp:95-140).
// `skip_batches` whole batches of this thread's slice are skipped first (a persistent launch samples step s from the
// states the launch started with).
__device__ __forceinline__ void sample_slot(const SamplerArgs &a, long long p, long long k, unsigned long long skip_draws, int &oh, int &ot,
                                            int &orr) {
    const long long id = (long long)((unsigned)p / (unsigned)a.per_thread);   // owning virtual thread (Base.cpp:85-92); B < 2^31
    const long long off = p - id * a.per_thread;  // index inside its slice
    const unsigned long long draws = 1ull + 2ull * a.neg + a.negrel;
    uint64_t s = lcg_skip(a.streams[id], skip_draws + (unsigned long long)off * draws);
    s = lcg_step(s);  // Base.cpp:101-106: which training triple
    long long i = (long long)mod_magic(s, a.pick_div, a.pick_magic) + (a.new_batch > 0 ? a.train_dup - a.new_batch : 0);
    const int4 tr = a.pos[i];  // (h, t, r, -)
    const int4 gq = a.grp[i];  // loaded together with it (not after the coin): one memory latency instead of two
    oh = tr.x; ot = tr.y; orr = tr.z;
    if (k >= 1 && k <= a.neg) {
        s = lcg_skip(s, 2ull * (unsigned long long)(k - 1));
        s = lcg_step(s);  // Base.cpp:118: head-or-tail coin, compared in float
        const float prob = a.bern ? a.bern_prob[orr] : 500.0f;
        const bool keep_head = (float)(s % 1000ull) < prob;
        s = lcg_step(s);  // Corrupt.h:25: the one draw of the corruption
        if (keep_head) {  // corrupt_head(h, r): new TAIL outside tails(h,r)
            long long tmp = (long long)mod_u64_u32(s, (uint32_t)(a.ent_total - gq.y));
            ot = min(filtered_pick(a.tails_hr + gq.x, gq.y, tmp), a.ent_total - 1);   // (clamp: only reachable in that degenerate case)
        } else {          // corrupt_tail(t, r): new HEAD outside heads(t,r)
            long long tmp = (long long)mod_u64_u32(s, (uint32_t)(a.ent_total - gq.w));
            oh = min(filtered_pick(a.heads_tr + gq.z, gq.w, tmp), a.ent_total - 1);
        }
    } else if (k > a.neg) {  // Base.cpp:133-139: corrupt_rel(h, t)
        s = lcg_skip(s, 2ull * a.neg + (unsigned long long)(k - 1 - a.neg));
        s = lcg_step(s);
        const int2 g = a.ht[i];
        long long tmp = (long long)mod_u64_u32(s, (uint32_t)(a.rel_total - g.y));
        orr = min(filtered_pick(a.rels_ht + g.x, g.y, tmp), a.rel_total - 1);
    }
}

// every stream's state after this batch, into the OTHER half of the double buffer (the launch reads only the current half,
// so no ordering between blocks is needed and no separate launch either); the host swaps the halves
__device__ __forceinline__ void write_next_streams(const SamplerArgs &a, int kp, long long block, long long n_blocks) {
    for (long long id = block * 256 + threadIdx.x; id < a.W; id += n_blocks * 256) {
        long long lef = id * a.per_thread, rig = lef + a.per_thread;
        if (rig > a.B) rig = a.B;
        if (lef > a.B) lef = a.B;
        a.streams_next[id] = lcg_skip(a.streams[id], (unsigned long long)(rig - lef) * (unsigned long long)(kp + a.neg));
    }
}

// More than 64 slots per positive (over 63 negatives): one independent thread per slot, each with its own full jump.
__device__ __forceinline__ void sample_block_wide(const SamplerArgs &a, long long block, long long n_blocks) {
    const int kshift = a.kshift, kp = 1 + a.neg + a.negrel;
    write_next_streams(a, kp, block, n_blocks);
    for (long long g = block * 256 + threadIdx.x; (g >> kshift) < a.n_local; g += n_blocks * 256) {
        const long long b = g >> kshift;
        const long long k = g & ((1 << kshift) - 1);
        if (k >= kp) continue;
        int oh, ot, orr;
        sample_slot(a, a.pos_lo + b, k, 0ull, oh, ot, orr);
        const long long o = b + k * a.out_stride;
        a.out_h[o] = oh; a.out_t[o] = ot; a.out_r[o] = orr;
    }
}

constexpr int kBernLds = 2048;

// Up to 64 slots per positive (the usual case).  The 1+neg+negrel draws of one positive sit in ADJACENT lanes (k = 0 the
// positive, 1..neg entity negatives, then relation negatives; padded to a power of two <= 64): they read the same pos / grp
// record and search the same groups, so those loads coalesce.  What the slots of a WAVE share is computed once:
//   * the long jump (up to 64 table steps: slice offset x draws per positive) is done for the wave's FIRST positive only, on
//     wave-uniform values (scalar unit); a lane then advances by the few draws between that state and its own slot -- at
//     most 64 positives' worth, a masked multiply-add per bit (lcg_skip_lanes) -- instead of repeating the long jump;
//   * the training-triple pick of a positive (one 64-bit modulo) is made by its k = 0 lane and handed to the others;
//   * the Bernoulli table sits in LDS (one dependent global load less per negative).
// Same draws in the same order as Base.cpp:95-140, so the batch is bit-identical to sample_slot's.
// `block` of `n_blocks` 256-thread workgroups: the body of sample_kernel, also run by workgroups that ride along in another
// kernel's launch (transe_counts.hip: the bucket scatter carries the NEXT batch's sampler, see kge_sampling_attach).
__device__ __forceinline__ void sample_block(const SamplerArgs &a, long long block, long long n_blocks, float *bern_lds) {
    const int kshift = a.kshift, kp = 1 + a.neg + a.negrel, kmask = (1 << kshift) - 1;
    const unsigned long long draws = 1ull + 2ull * a.neg + a.negrel;
    write_next_streams(a, kp, block, n_blocks);
    const bool bern_in_lds = a.bern && a.rel_total <= kBernLds;
    if (bern_in_lds) {
        for (int i = threadIdx.x; i < a.rel_total; i += 256) bern_lds[i] = a.bern_prob[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long long total = a.n_local << kshift;
    const long long wave0 = block * 256 + (__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63);
    for (long long g0 = wave0; g0 < total; g0 += n_blocks * 256) {
        // ---- wave-uniform: state in front of the first draw of the wave's first positive ----
        const long long p0 = a.pos_lo + (g0 >> kshift);
        const long long id0 = (long long)((unsigned)p0 / (unsigned)a.per_thread);   // owning virtual thread (Base.cpp:85-92); B < 2^31
        const long long off0 = p0 - id0 * a.per_thread;
        const uint64_t base0 = lcg_skip(a.streams[id0], (unsigned long long)off0 * draws);
        // ---- per lane ----
        const long long g = g0 + lane;
        long long b = g >> kshift;
        const int k = (int)(g & kmask);
        const bool live = b < a.n_local && k < kp;
        if (b >= a.n_local) b = a.n_local - 1;
        const int kk = k < kp ? k : kp - 1;
        const long long p = a.pos_lo + b;
        const long long id = (long long)((unsigned)p / (unsigned)a.per_thread);
        const long long off = p - id * a.per_thread;
        const bool same = id == id0;                  // (a wave may cross into the next virtual thread's slice)
        uint64_t s = same ? base0 : a.streams[id];
        unsigned ahead = (unsigned)((same ? off - off0 : off) * (long long)draws);   // < 64 positives' draws
        // draw 0 of a positive picks the training triple; entity negative k uses draws 1 + 2(k-1) (coin) and the next one
        // (corruption); relation negative k uses draw 1 + 2 neg + (k - 1 - neg)   (Base.cpp:101-139)
        if (kk >= 1) ahead += kk <= a.neg ? 1u + 2u * (unsigned)(kk - 1) : 1u + 2u * (unsigned)a.neg + (unsigned)(kk - 1 - a.neg);
        s = lcg_skip_lanes(s, ahead);
        s = lcg_step(s);                              // k = 0: the pick; entity negative: the coin; relation negative: its draw
        const long long pick = (long long)mod_magic(s, a.pick_div, a.pick_magic) + (a.new_batch > 0 ? a.train_dup - a.new_batch : 0);
        const long long i = __shfl((int)pick, lane & ~kmask);      // the positive's k = 0 lane holds the real one (train_dup < 2^31)
        const int4 tr = a.pos[i];  // (h, t, r, -)
        const int4 gq = a.grp[i];  // loaded together with it (not after the coin): one memory latency instead of two
        int oh = tr.x, ot = tr.y, orr = tr.z;
        if (kk >= 1 && kk <= a.neg) {
            const float prob = a.bern ? (bern_in_lds ? bern_lds[orr] : a.bern_prob[orr]) : 500.0f;
            const bool keep_head = (float)(s % 1000ull) < prob;      // Base.cpp:118: compared in float
            s = lcg_step(s);                                           // Corrupt.h:25: the one draw of the corruption
            if (keep_head) {  // corrupt_head(h, r): new TAIL outside tails(h,r)
                const long long tmp = (long long)mod_u64_u32(s, (uint32_t)(a.ent_total - gq.y));
                ot = min(filtered_pick_short(a.tails_hr + gq.x, gq.y, tmp), a.ent_total - 1);   // (clamp: only reachable in that degenerate case)
            } else {          // corrupt_tail(t, r): new HEAD outside heads(t,r)
                const long long tmp = (long long)mod_u64_u32(s, (uint32_t)(a.ent_total - gq.w));
                oh = min(filtered_pick_short(a.heads_tr + gq.z, gq.w, tmp), a.ent_total - 1);
            }
        } else if (kk > a.neg) {  // Base.cpp:133-139: corrupt_rel(h, t)
            const int2 gr = a.ht[i];
            const long long tmp = (long long)mod_u64_u32(s, (uint32_t)(a.rel_total - gr.y));
            orr = min(filtered_pick_short(a.rels_ht + gr.x, gr.y, tmp), a.rel_total - 1);
        }
        if (live) {
            const long long o = b + (long long)k * a.out_stride;
            a.out_h[o] = oh; a.out_t[o] = ot; a.out_r[o] = orr;
        }
    }
}


// workgroup `i` of the part of an armed sampler that rides in this launch
__device__ __forceinline__ void sample_block_ride(const SamplerArgs &a, long long i, float *bern_lds) {
    sample_block(a, (long long)a.ride_first + i, (long long)a.ride_total, bern_lds);
}

}  // namespace kge
