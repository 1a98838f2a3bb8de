// Owner-routed exchange for the TABLE-SHARDED sparse TransE path (BASELINE config #5 on N GPUs).
//
// The reference shards its variables over parameter-server tasks (distribute_training.py:193-196,
// replica_device_setter + GreedyLoadBalancingStrategy) and workers pull the rows they need / push
// IndexedSlices gradients over gRPC.  Here rank g OWNS the entity rows [g*chunk, (g+1)*chunk) -- the
// 102 GB entity table of config #5 is never replicated -- and one step is
//
//   sample (local slice) -> REQUEST ids bucketed by owner -> all-to-all ids -> owners GATHER rows ->
//   all-to-all rows -> emit int8 sign records against the fetched rows (the unchanged emit kernel: the
//   batch is remapped to slots of the fetched-row cache) -> entity records bucketed by owner ->
//   all-to-all (row id, record) -> owners sort / sum / apply THEIR rows only
//
// so that every per-rank stage is O(local batch).  The relation table (R x D, a few MB) stays
// replicated: its records are summed into a dense int32 image that is all-reduced.
// This file holds the device side of that routing: request lists, the counting sort by owner, the
// row gather, record packing and the relation count image.  All kernels are plain HBM-bound
// integer / copy work (16-byte accesses, one pass each).
#include "engine.hpp"

namespace kge {
namespace {

constexpr int kMaxOwners = 64;

// req[slot*n_pos + b]: the entity whose row record slot (slot, b) of the emit kernel touches, -1 if none
//   slot 0 = the positive's head, 1 = its tail, 2 = its relation (not an entity), 3+k = the NEW entity of negative k
//   (a relation-corrupted negative, Base.cpp:133-139, introduces no entity)
__global__ __launch_bounds__(256) void shard_requests_kernel(const int32_t *__restrict__ bh, const int32_t *__restrict__ bt,
                                                             const int32_t *__restrict__ br, long long n_pos, int n_neg, long long stride,
                                                             int32_t *__restrict__ req) {
    const long long total = n_pos * (3 + n_neg);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long slot = i / n_pos, b = i - slot * n_pos;
        int v = -1;
        if (slot == 0) v = bh[b];
        else if (slot == 1) v = bt[b];
        else if (slot >= 3) {
            const long long j = b + (slot - 2) * stride;      // negative k = slot-3 sits at b + (k+1)*stride
            const int nh = bh[j], nt = bt[j];
            if (nh != bh[b]) v = nh; else if (nt != bt[b]) v = nt;
        }
        req[i] = v;
    }
}

// counts[o] = number of ids owned by rank o (id / chunk), ids < 0 skipped
__global__ __launch_bounds__(256) void shard_count_kernel(const int32_t *__restrict__ ids, long long n, int chunk, int owners,
                                                          int32_t *__restrict__ counts) {
    __shared__ int hist[kMaxOwners];
    if (threadIdx.x < kMaxOwners) hist[threadIdx.x] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int v = ids[i];
        if (v >= 0) atomicAdd(&hist[min(v / chunk, owners - 1)], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < owners && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], hist[threadIdx.x]);
}

struct Offsets { int v[kMaxOwners]; };

// sorted[pos] = ids[i], slot_of[i] = pos, grouped by owner (any order inside a group: downstream sums are integers)
__global__ __launch_bounds__(256) void shard_scatter_kernel(const int32_t *__restrict__ ids, long long n, int chunk, int owners, Offsets off,
                                                            int32_t *__restrict__ cursor, int32_t *__restrict__ sorted,
                                                            int32_t *__restrict__ slot_of) {
    __shared__ int hist[kMaxOwners], base[kMaxOwners];
    constexpr int PER = 8;
    const long long tile = (long long)blockIdx.x * (256 * PER);
    if (threadIdx.x < kMaxOwners) hist[threadIdx.x] = 0;
    __syncthreads();
    int v[PER], rank_in[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const long long i = tile + threadIdx.x + 256 * k;
        v[k] = i < n ? ids[i] : -1;
        rank_in[k] = v[k] >= 0 ? atomicAdd(&hist[min(v[k] / chunk, owners - 1)], 1) : 0;
    }
    __syncthreads();
    if ((int)threadIdx.x < owners) base[threadIdx.x] = hist[threadIdx.x] ? off.v[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], hist[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const long long i = tile + threadIdx.x + 256 * k;
        if (i >= n) continue;
        int pos = -1;
        if (v[k] >= 0) {
            pos = base[min(v[k] / chunk, owners - 1)] + rank_in[k];
            sorted[pos] = v[k];
        }
        slot_of[i] = pos;
    }
}

// the batch rewritten in terms of CACHE SLOTS: entity "ids" become positions in the fetched-row list; a kept side of a
// negative carries its positive's slot, so the emit kernel's classification (same_h / same_t) is unchanged
__global__ __launch_bounds__(256) void shard_remap_kernel(const int32_t *__restrict__ bh, const int32_t *__restrict__ bt, long long n_pos,
                                                          int n_neg, long long stride, const int32_t *__restrict__ slot_of,
                                                          int32_t *__restrict__ h2, int32_t *__restrict__ t2) {
    const long long total = n_pos * (1 + n_neg);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long k1 = i / n_pos, b = i - k1 * n_pos;     // k1 = 0: the positive, k1 = k+1: negative k
        const int sh = slot_of[b], st = slot_of[n_pos + b];
        const long long j = b + k1 * stride;
        int oh = sh, ot = st;
        if (k1 > 0) {
            const int sn = slot_of[(2 + k1) * n_pos + b];       // slot 3+k
            if (bh[j] != bh[b]) oh = sn; else if (bt[j] != bt[b]) ot = sn;
        }
        h2[j] = oh; t2[j] = ot;
    }
}

// out[i, :] = table[ids[i] - row_lo, :]   (D % 4 == 0; one float4 per lane)
__global__ __launch_bounds__(256) void shard_gather_rows_kernel(const float *__restrict__ table, const int32_t *__restrict__ ids, long long n,
                                                                long long row_lo, long long rows, int D4, float *__restrict__ out) {
    const long long total = n * D4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / D4, c = i - r * D4;
        long long src = (long long)ids[r] - row_lo;
        src = src < 0 ? 0 : (src >= rows ? rows - 1 : src);    // ids come from a peer: never index outside the shard
        reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(table)[src * D4 + c];
    }
}

// ids2[m] = global entity id of record m if it is a live ENTITY record (dst[m] is a cache slot), else -1
__global__ __launch_bounds__(256) void shard_record_ids_kernel(const int32_t *__restrict__ dst, long long M, long long cache_rows,
                                                               const int32_t *__restrict__ cache_ids, int32_t *__restrict__ ids2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x) {
        const int d = dst[i];
        ids2[i] = (d >= 0 && d < cache_rows) ? cache_ids[d] : -1;
    }
}

// out[slot_of[m], :] = rec[m, :] for the records that travel (dw dwords each)
__global__ __launch_bounds__(256) void shard_pack_kernel(const uint32_t *__restrict__ rec, const int32_t *__restrict__ slot_of, long long M,
                                                         int dw, uint32_t *__restrict__ out) {
    const long long total = M * dw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / dw, c = i - m * dw;
        const int pos = slot_of[m];
        if (pos >= 0) out[(long long)pos * dw + c] = rec[i];
    }
}

// relation records (dst >= cache_rows) summed into the dense int32 image [R, D] (natural element order: byte e of a
// record is element e when D % 4 == 0); one wave per record, int32 atomics (R rows: a few MB, L2 resident)
__global__ __launch_bounds__(256) void shard_rel_counts_kernel(const uint32_t *__restrict__ rec, const int32_t *__restrict__ dst, long long M,
                                                               long long cache_rows, long long R, int dw, int D, int32_t *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    for (long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long long)gridDim.x * 4) {
        const long long d = dst[m];
        if (d < cache_rows || d >= cache_rows + R) continue;
        int32_t *row = counts + (d - cache_rows) * D;
        for (int w = lane; w < dw; w += 64) {
            const uint32_t word = rec[m * dw + w];
            if (!word) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int v = (int)(int8_t)(word >> (8 * j));
                const int e = 4 * w + j;
                if (v != 0 && e < D) atomicAdd(row + e, v);
            }
        }
    }
}

// image[rows[i] - base, :] = counts[i, :] for the *n_rows_p compact rows (image zeroed by the caller; rows outside [base, base + R) skipped)
__global__ __launch_bounds__(256) void shard_scatter_count_rows_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ counts,
                                                                       const int32_t *__restrict__ n_rows_p, long long base, long long R, int D4,
                                                                       int32_t *__restrict__ image) {
    const long long total = (long long)n_rows_p[0] * D4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long u = i / D4, c = i - u * D4;
        const long long r = (long long)rows[u] - base;
        if (r >= 0 && r < R) reinterpret_cast<int4 *>(image)[r * D4 + c] = reinterpret_cast<const int4 *>(counts)[i];
    }
}

unsigned blocks_for(long long n, int per_block = 256, long long cap = 16384) {
    long long b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace
}  // namespace kge

using namespace kge;

extern "C" {

int kge_shard_requests(const int32_t *d_h, const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, int32_t *d_req,
                       void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_requests: no usable HIP device");
    if (n_pos < 0 || n_neg < 1 || stride < n_pos || !d_req) return fail(KGE_ERR_BAD_ARG, "kge_shard_requests: bad arguments");
    if (n_pos == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_requests_kernel, dim3(blocks_for(n_pos * (3 + n_neg))), dim3(256), 0, (hipStream_t)stream, d_h, d_t, d_r,
                       (long long)n_pos, (int)n_neg, (long long)stride, d_req);
    return hip_check(hipGetLastError(), "shard requests launch");
}

int kge_shard_count(const int32_t *d_ids, INT n, INT chunk, INT n_owners, int32_t *d_counts, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_count: no usable HIP device");
    if (n < 0 || chunk <= 0 || n_owners < 1 || n_owners > kMaxOwners || !d_counts) return fail(KGE_ERR_BAD_ARG, "kge_shard_count: bad arguments (1..64 owners)");
    int rc = hip_check(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)n_owners, (hipStream_t)stream), "zero owner counts");
    if (rc || n == 0) return rc;
    hipLaunchKernelGGL(shard_count_kernel, dim3(blocks_for(n, 2048, 2048)), dim3(256), 0, (hipStream_t)stream, d_ids, (long long)n, (int)chunk,
                       (int)n_owners, d_counts);
    return hip_check(hipGetLastError(), "shard count launch");
}

int kge_shard_scatter(const int32_t *d_ids, INT n, INT chunk, INT n_owners, const INT *h_counts, int32_t *d_cursor, int32_t *d_sorted,
                      int32_t *d_slot_of, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_scatter: no usable HIP device");
    if (n < 0 || chunk <= 0 || n_owners < 1 || n_owners > kMaxOwners || !h_counts || !d_cursor || !d_sorted || !d_slot_of)
        return fail(KGE_ERR_BAD_ARG, "kge_shard_scatter: bad arguments (1..64 owners)");
    Offsets off;
    long long acc = 0;
    for (int o = 0; o < kMaxOwners; o++) { off.v[o] = (int)acc; if (o < n_owners) acc += h_counts[o]; }
    if (acc > n) return fail(KGE_ERR_BAD_ARG, "kge_shard_scatter: owner counts exceed the number of ids");
    int rc = hip_check(hipMemsetAsync(d_cursor, 0, sizeof(int32_t) * (size_t)n_owners, (hipStream_t)stream), "zero owner cursors");
    if (rc || n == 0) return rc;
    hipLaunchKernelGGL(shard_scatter_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, d_ids, (long long)n,
                       (int)chunk, (int)n_owners, off, d_cursor, d_sorted, d_slot_of);
    return hip_check(hipGetLastError(), "shard scatter launch");
}

int kge_shard_remap_batch(const int32_t *d_h, const int32_t *d_t, INT n_pos, INT n_neg, INT stride, const int32_t *d_slot_of, int32_t *d_h2,
                          int32_t *d_t2, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_remap_batch: no usable HIP device");
    if (n_pos < 0 || n_neg < 1 || stride < n_pos || !d_slot_of || !d_h2 || !d_t2) return fail(KGE_ERR_BAD_ARG, "kge_shard_remap_batch: bad arguments");
    if (n_pos == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_remap_kernel, dim3(blocks_for(n_pos * (1 + n_neg))), dim3(256), 0, (hipStream_t)stream, d_h, d_t, (long long)n_pos,
                       (int)n_neg, (long long)stride, d_slot_of, d_h2, d_t2);
    return hip_check(hipGetLastError(), "shard remap launch");
}

int kge_shard_gather_rows(const float *d_table, const int32_t *d_ids, INT n, INT row_lo, INT rows, INT dim, float *d_out, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_gather_rows: no usable HIP device");
    if (n < 0 || rows <= 0 || dim <= 0 || dim % 4 || !d_table || !d_out) return fail(KGE_ERR_BAD_ARG, "kge_shard_gather_rows: bad arguments (dim % 4 == 0)");
    if (n == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_gather_rows_kernel, dim3(blocks_for(n * (dim / 4), 256, 65536)), dim3(256), 0, (hipStream_t)stream, d_table, d_ids,
                       (long long)n, (long long)row_lo, (long long)rows, (int)(dim / 4), d_out);
    return hip_check(hipGetLastError(), "shard gather launch");
}

int kge_shard_record_ids(const int32_t *d_dst, INT n_records, INT cache_rows, const int32_t *d_cache_ids, int32_t *d_ids, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_record_ids: no usable HIP device");
    if (n_records < 0 || cache_rows < 0 || !d_dst || !d_ids) return fail(KGE_ERR_BAD_ARG, "kge_shard_record_ids: bad arguments");
    if (n_records == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_record_ids_kernel, dim3(blocks_for(n_records)), dim3(256), 0, (hipStream_t)stream, d_dst, (long long)n_records,
                       (long long)cache_rows, d_cache_ids, d_ids);
    return hip_check(hipGetLastError(), "shard record ids launch");
}

int kge_shard_pack_records(const uint32_t *d_rec, const int32_t *d_slot_of, INT n_records, INT dwords, uint32_t *d_out, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_pack_records: no usable HIP device");
    if (n_records < 0 || dwords <= 0 || !d_rec || !d_slot_of || !d_out) return fail(KGE_ERR_BAD_ARG, "kge_shard_pack_records: bad arguments");
    if (n_records == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_pack_kernel, dim3(blocks_for(n_records * dwords, 256, 65536)), dim3(256), 0, (hipStream_t)stream, d_rec, d_slot_of,
                       (long long)n_records, (int)dwords, d_out);
    return hip_check(hipGetLastError(), "shard pack launch");
}

int kge_shard_relation_counts(const uint32_t *d_rec, const int32_t *d_dst, INT n_records, INT cache_rows, INT rel_total, INT dwords, INT dim,
                              int32_t *d_counts, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_relation_counts: no usable HIP device");
    if (n_records < 0 || rel_total <= 0 || dwords <= 0 || dim <= 0 || dim % 4 || dim > 4 * dwords || !d_counts)
        return fail(KGE_ERR_BAD_ARG, "kge_shard_relation_counts: bad arguments (dim % 4 == 0)");
    if (n_records == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_rel_counts_kernel, dim3(blocks_for(n_records, 4, 16384)), dim3(256), 0, (hipStream_t)stream, d_rec, d_dst,
                       (long long)n_records, (long long)cache_rows, (long long)rel_total, (int)dwords, (int)dim, d_counts);
    return hip_check(hipGetLastError(), "shard relation counts launch");
}

int kge_shard_scatter_count_rows(const int32_t *d_rows, const int32_t *d_row_counts, const int32_t *d_n_rows, INT max_rows, INT base,
                                 INT rel_total, INT dim, int32_t *d_image, void *stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_shard_scatter_count_rows: no usable HIP device");
    if (!d_rows || !d_row_counts || !d_n_rows || !d_image || max_rows < 0 || rel_total <= 0 || dim <= 0 || dim % 4)
        return fail(KGE_ERR_BAD_ARG, "kge_shard_scatter_count_rows: bad arguments (dim % 4 == 0)");
    if (max_rows == 0) return KGE_OK;
    hipLaunchKernelGGL(shard_scatter_count_rows_kernel, dim3(blocks_for(max_rows * (dim / 4))), dim3(256), 0, (hipStream_t)stream, d_rows,
                       d_row_counts, d_n_rows, (long long)base, (long long)rel_total, (int)(dim / 4), d_image);
    return hip_check(hipGetLastError(), "shard scatter count rows launch");
}

}  // extern "C"
