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
#include "sampler_dev.hpp"

namespace kge {

// every stream's state after this batch, into the OTHER half of the double buffer (the launch reads only the current half,
// so no ordering between blocks is needed and no separate launch either); the host swaps the halves
__device__ __forceinline__ void write_next_streams(const SamplerArgs &a, int kp) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < a.W; id += (long long)gridDim.x * blockDim.x) {
        long long lef = id * a.per_thread, rig = lef + a.per_thread;
        if (rig > a.B) rig = a.B;
        if (lef > a.B) lef = a.B;
        a.streams_next[id] = lcg_skip(a.streams[id], (unsigned long long)(rig - lef) * (unsigned long long)(kp + a.neg));
    }
}

// More than 64 slots per positive (over 63 negatives): one independent thread per slot, each with its own full jump.
__global__ __launch_bounds__(256) void sample_kernel_wide(SamplerArgs a) {
    const int kshift = a.kshift, kp = 1 + a.neg + a.negrel;
    write_next_streams(a, kp);
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; (g >> kshift) < a.n_local; g += (long long)gridDim.x * blockDim.x) {
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
__global__ __launch_bounds__(256) void sample_kernel(SamplerArgs a) {
    __shared__ float bern_lds[kBernLds];
    const int kshift = a.kshift, kp = 1 + a.neg + a.negrel, kmask = (1 << kshift) - 1;
    const unsigned long long draws = 1ull + 2ull * a.neg + a.negrel;
    write_next_streams(a, kp);
    const bool bern_in_lds = a.bern && a.rel_total <= kBernLds;
    if (bern_in_lds) {
        for (int i = threadIdx.x; i < a.rel_total; i += 256) bern_lds[i] = a.bern_prob[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const long long total = a.n_local << kshift;
    const long long wave0 = (long long)blockIdx.x * 256 + (__builtin_amdgcn_readfirstlane(threadIdx.x) & ~63);
    for (long long g0 = wave0; g0 < total; g0 += (long long)gridDim.x * 256) {
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

// The same advance in place, for a launch whose own slice is empty (a data-parallel rank without positions).
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
    if ((rc = upload_jump_table())) return rc;
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
        a.streams_next = e.dev.streams_next; a.W = W; a.B = B;
        if (kshift <= 6) hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(sample_kernel_wide, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        std::swap(e.dev.streams, e.dev.streams_next);
    } else
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
