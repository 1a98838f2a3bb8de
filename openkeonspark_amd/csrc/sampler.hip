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

__global__ __launch_bounds__(256) void sample_kernel(SamplerArgs a) {
    // The 1+neg+negrel draws of one positive sit in ADJACENT lanes (k = 0 positive, 1..neg entity negatives, then
    // relation negatives; padded to a power of two <= 64): they all read the same pos / grp record and search the same
    // groups, so those loads coalesce into one request instead of 1+neg requests from as many different blocks.
    const int kshift = a.kshift, kp = 1 + a.neg + a.negrel;
    // After a batch every stream has moved by (slice length) * (draws per positive).  The advanced states go to the
    // OTHER half of the double buffer (this launch reads only the current half, so no ordering between blocks is
    // needed and no separate launch either); the host swaps the halves.
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < a.W; id += (long long)gridDim.x * blockDim.x) {
        long long lef = id * a.per_thread, rig = lef + a.per_thread;
        if (rig > a.B) rig = a.B;
        if (lef > a.B) lef = a.B;
        a.streams_next[id] = lcg_skip(a.streams[id], (unsigned long long)(rig - lef) * (unsigned long long)(kp + a.neg));
    }
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
        hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
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
