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

__global__ __launch_bounds__(256) void sample_kernel_wide(SamplerArgs a) { sample_block_wide(a, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void sample_kernel(SamplerArgs a) {
    __shared__ float bern_lds[kBernLds];
    sample_block(a, blockIdx.x, gridDim.x, bern_lds);
}

// what is left of an armed sampler after parts of it rode in other launches
__global__ __launch_bounds__(256) void sample_kernel_part(SamplerArgs a) {
    __shared__ float bern_lds[kBernLds];
    sample_block_ride(a, blockIdx.x, bern_lds);
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

// Validates the call and fills the launch description.  blocks = 0: this rank's slice is empty (only the streams advance).
static int build_sampler(int32_t *d_h, int32_t *d_t, int32_t *d_r, int64_t B, int64_t neg, int64_t negrel, int64_t thread_lo,
                         int64_t thread_hi, int64_t out_stride, int64_t *n_local_out, SamplerArgs &a, unsigned &blocks, bool &wide) {
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
    a = SamplerArgs();
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
    a.streams_next = e.dev.streams_next; a.W = W; a.B = B;
    wide = kshift > 6;
    int64_t nb = n_local > 0 ? ((n_local << kshift) + 255) / 256 : 0;
    if (nb > (1 << 20)) nb = 1 << 20;
    blocks = (unsigned)nb;
    return KGE_OK;
}

// The sampler of the NEXT batch, armed by kge_sampling_attach and not launched yet: the bucket-scatter launch of the step in
// progress takes it along as extra workgroups (transe_counts.hip), kge_sampling_flush launches it on its own otherwise.
static bool g_att_armed = false, g_att_wide = false;
static SamplerArgs g_att;
static unsigned g_att_blocks = 0, g_att_next = 0;     // workgroups of the armed sampler; the first one not handed out yet

// `share` of the armed sampler's workgroups (1 = all that is left) for the caller's launch: a.ride_first / a.ride_total say which,
// `blocks` how many extra workgroups the launch needs (sample_block_ride).  The small kernels of a step each leave most of the
// chip idle; the sampler -- a latency-bound pointer chase independent of the step -- is spread over several of them.
bool take_attached_sampler(SamplerArgs &a, unsigned &blocks, float share) {
    if (!g_att_armed || g_att_wide || g_att_blocks == 0) return false;
    const unsigned left = g_att_blocks - g_att_next;
    unsigned want = share >= 1.0f ? left : (unsigned)(share * (float)g_att_blocks + 0.5f);
    if (want > left) want = left;
    if (want == 0) return false;
    a = g_att; a.ride_first = g_att_next; a.ride_total = g_att_blocks;
    blocks = want;
    g_att_next += want;
    if (g_att_next >= g_att_blocks) g_att_armed = false;
    return true;
}

int flush_attached_sampler(hipStream_t stream) {
    if (!g_att_armed) return KGE_OK;
    g_att_armed = false;
    if (g_att_wide) hipLaunchKernelGGL(sample_kernel_wide, dim3(g_att_blocks), dim3(256), 0, stream, g_att);
    else if (g_att_next == 0) hipLaunchKernelGGL(sample_kernel, dim3(g_att_blocks), dim3(256), 0, stream, g_att);
    else {
        SamplerArgs a = g_att;
        a.ride_first = g_att_next; a.ride_total = g_att_blocks;
        hipLaunchKernelGGL(sample_kernel_part, dim3(g_att_blocks - g_att_next), dim3(256), 0, stream, a);
    }
    return hip_check(hipGetLastError(), "sampler launch");
}

int attach_sampler(int32_t *d_h, int32_t *d_t, int32_t *d_r, int64_t B, int64_t neg, int64_t negrel, int64_t thread_lo,
                   int64_t thread_hi, int64_t out_stride, int64_t *n_local_out, hipStream_t stream) {
    Engine &e = engine();
    int rc = flush_attached_sampler(stream);    // at most one armed sampler: batches are drawn in order
    if (rc) return rc;
    SamplerArgs a; unsigned blocks; bool wide;
    if ((rc = build_sampler(d_h, d_t, d_r, B, neg, negrel, thread_lo, thread_hi, out_stride, n_local_out, a, blocks, wide))) return rc;
    if (blocks == 0) {       // empty slice: nothing to carry, the streams advance now
        hipLaunchKernelGGL(advance_streams_kernel, dim3((unsigned)((a.W + 63) / 64)), dim3(64), 0, stream, e.dev.streams,
                           (long long)a.W, (long long)B, (long long)a.per_thread, (unsigned long long)(1 + 2 * neg + negrel));
        e.dev.streams_sync = 2;
        return hip_check(hipGetLastError(), "sampler launch");
    }
    g_att = a; g_att_blocks = blocks; g_att_next = 0; g_att_wide = wide; g_att_armed = true;
    std::swap(e.dev.streams, e.dev.streams_next);     // the armed launch reads the current half and writes the other one
    e.dev.streams_sync = 2;
    return KGE_OK;
}

int launch_sampler(int32_t *d_h, int32_t *d_t, int32_t *d_r, int64_t B, int64_t neg, int64_t negrel, int64_t thread_lo,
                   int64_t thread_hi, int64_t out_stride, int64_t *n_local_out, hipStream_t stream) {
    Engine &e = engine();
    int rc = flush_attached_sampler(stream);    // an armed sampler draws the batch BEFORE this one
    if (rc) return rc;
    SamplerArgs a; unsigned blocks; bool wide;
    if ((rc = build_sampler(d_h, d_t, d_r, B, neg, negrel, thread_lo, thread_hi, out_stride, n_local_out, a, blocks, wide))) return rc;
    if (blocks > 0) {
        if (!wide) hipLaunchKernelGGL(sample_kernel, dim3(blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(sample_kernel_wide, dim3(blocks), dim3(256), 0, stream, a);
        std::swap(e.dev.streams, e.dev.streams_next);
    } else
        hipLaunchKernelGGL(advance_streams_kernel, dim3((unsigned)((a.W + 63) / 64)), dim3(64), 0, stream, e.dev.streams,
                           (long long)a.W, (long long)B, (long long)a.per_thread,
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
