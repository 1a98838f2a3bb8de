// Device negative sampler: one GPU thread per SCORED TRIPLE (positive or negative).
//
// The reference fills a batch with `workThreads` pthreads, each walking its slice sequentially on
// its own LCG stream (base/Base.cpp:74-143, base/Random.h:16-19).  Every positive consumes a FIXED
// number of draws (1 + 2*negRate + negRelRate), so the state in front of any draw is a jump-ahead
// of the slice's start state; that turns the sequential walk into B*(1+n+nr) independent threads
// whose output is bit-identical to the reference.  The filter is the reference's: draw k uniformly
// from the complement of the known tails / heads / relations (base/Corrupt.h:7-101), here as one
// monotone search over a flat int32 group (the two group-locating searches were done at load time,
// kg_index.hpp).
#include "engine.hpp"

namespace kge {

__constant__ LcgJumpTable c_jump;
static bool g_jump_uploaded = false;

struct SamplerArgs {
    const int4 *pos;
    const int4 *grp;
    const int2 *ht;
    const int32_t *tails_hr, *heads_tr, *rels_ht;
    const float *bern_prob;
    const uint64_t *streams;
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

__global__ __launch_bounds__(256) void sample_kernel(SamplerArgs a) {
    // The 1+neg+negrel draws of one positive sit in ADJACENT lanes (k = 0 positive, 1..neg entity negatives, then
    // relation negatives; padded to a power of two <= 64): they all read the same pos / grp record and search the same
    // groups, so those loads coalesce into one request instead of 1+neg requests from as many different blocks.
    const int kshift = a.kshift, kp = 1 + a.neg + a.negrel;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; (g >> kshift) < a.n_local; g += (long long)gridDim.x * blockDim.x) {
        const long long b = g >> kshift;
        const long long k = g & ((1 << kshift) - 1);
        if (k >= kp) continue;
        const long long p = a.pos_lo + b;             // global batch position
        const long long id = (long long)((unsigned)p / (unsigned)a.per_thread);   // owning virtual thread (Base.cpp:85-92); B < 2^31
        const long long off = p - id * a.per_thread;  // index inside its slice
        const unsigned long long draws = 1ull + 2ull * a.neg + a.negrel;
        uint64_t s = lcg_skip(a.streams[id], (unsigned long long)off * draws);
        s = lcg_step(s);  // Base.cpp:101-106: which training triple
        long long i = (long long)mod_magic(s, a.pick_div, a.pick_magic) + (a.new_batch > 0 ? a.train_dup - a.new_batch : 0);
        const int4 tr = a.pos[i];  // (h, t, r, -)
        const int4 gq = a.grp[i];  // loaded together with it (not after the coin): one memory latency instead of two
        int oh = tr.x, ot = tr.y, orr = tr.z;
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
        const long long o = b + k * a.out_stride;
        a.out_h[o] = oh; a.out_t[o] = ot; a.out_r[o] = orr;
    }
}

// After a batch every stream has moved by (slice length) * (draws per positive).
__global__ void advance_streams_kernel(uint64_t *streams, long long W, long long B, long long per_thread,
                                       unsigned long long draws) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= W) return;
    long long lef = id * per_thread, rig = lef + per_thread;
    if (rig > B) rig = B;
    if (lef > B) lef = B;
    streams[id] = lcg_skip(streams[id], (unsigned long long)(rig - lef) * draws);
}

// int32 device batch -> the reference's int64 h/t/r + float y host layout (Base.cpp:109-139: y=+1 for
// the B positives, -1 for every negative).
__global__ void widen_kernel(const int32_t *__restrict__ src, long long *__restrict__ dst, float *__restrict__ y,
                             long long B, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        dst[i] = src[i];
        dst[total + i] = src[total + i];
        dst[2 * total + i] = src[2 * total + i];
        y[i] = i < B ? 1.0f : -1.0f;
    }
}

int launch_sampler(int32_t *d_h, int32_t *d_t, int32_t *d_r, int64_t B, int64_t neg, int64_t negrel, int64_t thread_lo,
                   int64_t thread_hi, int64_t out_stride, int64_t *n_local_out, hipStream_t stream) {
    Engine &e = engine();
    int rc = ensure_device_index();
    if (rc) return rc;
    const int64_t W = e.work_threads;
    if (B <= 0 || neg < 0 || negrel < 0 || thread_lo < 0 || thread_hi > W || thread_lo > thread_hi)
        return fail(KGE_ERR_BAD_ARG, "kge_sampling_device: bad batch/thread range");
    if (e.index.train_dup <= 0) return fail(KGE_ERR_NO_DATASET, "sampling: empty training set");
    if (!g_jump_uploaded) {
        rc = hip_check(hipMemcpyToSymbol(HIP_SYMBOL(c_jump), &e.jump, sizeof(LcgJumpTable)), "upload jump table");
        if (rc) return rc;
        g_jump_uploaded = true;
    }
    int64_t lo, hi, tmp;
    if (thread_lo == thread_hi) { lo = hi = 0; }
    else { thread_slice(B, W, thread_lo, lo, tmp); thread_slice(B, W, thread_hi - 1, tmp, hi); }
    const int64_t n_local = hi - lo;
    if (n_local_out) *n_local_out = n_local;
    if (out_stride < n_local) return fail(KGE_ERR_BAD_ARG, "kge_sampling_device: out_stride smaller than the slice");
    const int64_t per_thread = (B % W == 0) ? B / W : B / W + 1;
    if (n_local > 0) {
        SamplerArgs a;
        a.pos = e.dev.pos; a.grp = e.dev.grp; a.ht = e.dev.ht;
        a.tails_hr = e.dev.tails_hr; a.heads_tr = e.dev.heads_tr; a.rels_ht = e.dev.rels_ht;
        a.bern_prob = e.dev.bern_prob; a.streams = e.dev.streams;
        a.out_h = d_h; a.out_t = d_t; a.out_r = d_r;
        a.per_thread = per_thread; a.pos_lo = lo; a.n_local = n_local; a.out_stride = out_stride;
        a.train_dup = e.index.train_dup; a.new_batch = e.index.new_batch;
        a.ent_total = (int)e.index.ent_total; a.rel_total = (int)e.index.rel_total;
        a.neg = (int)neg; a.negrel = (int)negrel; a.bern = e.bern ? 1 : 0;
        a.pick_div = (unsigned long long)(a.new_batch > 0 ? a.new_batch : a.train_dup);
        a.pick_magic = ~0ull / a.pick_div;
        int kshift = 0;
        while ((1 << kshift) < 1 + neg + negrel) kshift++;
        a.kshift = kshift;
        int64_t blocks = ((n_local << kshift) + 255) / 256;
        if (blocks > (1 << 20)) blocks = 1 << 20;
        hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    }
    hipLaunchKernelGGL(advance_streams_kernel, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, stream, e.dev.streams,
                       (long long)W, (long long)B, (long long)per_thread,
                       (unsigned long long)(1 + 2 * neg + negrel));
    e.dev.streams_sync = 2;  // device copy is now the newer one
    return hip_check(hipGetLastError(), "sampler launch");
}

int launch_widen(const int32_t *src3, int64_t *dst3_and_y, int64_t B, int64_t total, hipStream_t stream) {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(widen_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src3, (long long *)dst3_and_y,
                       (float *)(dst3_and_y + 3 * total), (long long)B, (long long)total);
    return hip_check(hipGetLastError(), "widen launch");
}

}  // namespace kge
