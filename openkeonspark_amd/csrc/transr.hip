// TransR (TransR.py:16-87): h_r = h . M_r with a [ent_size x rel_size] matrix per relation.
//
// The reference gathers one 160 kB matrix per TRIPLE (`embedding_lookup(transfer_matrix, pos_r)`,
// TransR.py:52) and runs batched [1 x de] x [de x dr] matmuls -- 435 MB of matrix traffic per step
// at config #4.  Here the projections are bucketed BY RELATION so each matrix is streamed once per
// bucket tile and the work is GEMM-shaped for the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 1e-5 parity without a reduced-precision path):
//
//   prep      one thread per (scored triple, side): entity id, matrix relation, canonical? (a side
//             whose projected vector equals the positive's is an alias and is not projected again)
//   sort      stable radix sort of the canonical jobs by matrix relation (rocPRIM)
//   bounds    bucket starts + (relation, 32-row tile) map
//   project   P[slot]  = ent[e] . M_r                      MFMA, 32 x 128 output tile per workgroup
//   vector    models.hip fwdbwd_kernel<TRANSR>: normalise, L1 score, hinge, backward -> GP[slot], g_rel
//   dgrad     g_ent[e] += GP[slot] . M_r^T                 MFMA + fp32 atomics (rows of `de` floats)
//   wgrad     g_M[r]   += X_r^T . GP_r  over the bucket    MFMA, buckets split over workgroups, fp32 atomics
//
// FLOPs per scored triple: fwd 2 projections 4.de.dr, bwd 8.de.dr.
#include <cstring>
#include <type_traits>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "engine.hpp"
#include "models_dev.hpp"
#include "sampler_dev.hpp"

namespace kge {

int launch_transr_vector_stage(const float *rel, float *g_rel, const float *P, float *GP, const int32_t *d_h,
                               const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                               int64_t denom, int rel_dim, float margin, int negative_rel, float *d_loss,
                               hipStream_t stream, bool lean, bool sampler_shaped, int64_t rel_total);
bool transr_lean_vector_stage(int rel_dim);
int launch_transr_predict_stage(const float *rel, const float *P, const int32_t *d_r, int64_t n, int rel_dim, float *d_out,
                                hipStream_t stream);

namespace {

struct TrWork {
    float *P = nullptr, *GP = nullptr;
    int32_t *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr, *job_ent = nullptr, *row_ent = nullptr;
    int32_t *bucket_start = nullptr;  // [R+2]
    int32_t *tile_rel = nullptr, *tile_row0 = nullptr, *n_tiles = nullptr;
    int32_t *bucket_rows = nullptr;   // group layout: start of every relation's ROW space ([R + 2]; bucket_start holds the group starts there)
    int64_t cap_rel_rows = 0, pad_ready = -1;
    int32_t *rel_hist = nullptr;      // two alternating pairs of [kRelBins] bucket sizes + [kRelBins] scatter cursors
    int rel_parity = 0;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int64_t cap_slots = 0, cap_dim = 0, cap_rel = 0, cap_tiles = 0;
};
TrWork g_w;

template <typename T>
int grow(T *&p, size_t count, const char *what) {
    if (p) (void)hipFree(p);
    p = nullptr;
    return hip_check(hipMalloc(&p, sizeof(T) * (count ? count : 1)), what);
}

int ensure_work(int64_t slots, int64_t dr, int64_t R) {
    int rc;
    if (slots > g_w.cap_slots || dr > g_w.cap_dim) {
        int64_t s = slots > g_w.cap_slots ? slots : g_w.cap_slots, d = dr > g_w.cap_dim ? dr : g_w.cap_dim;
        if ((rc = grow(g_w.P, (size_t)s * d, "transr P"))) return rc;
        if ((rc = grow(g_w.GP, (size_t)s * d, "transr GP"))) return rc;
        if ((rc = grow(g_w.keys, (size_t)s, "transr keys"))) return rc;
        if ((rc = grow(g_w.keys2, (size_t)s, "transr keys2"))) return rc;
        if ((rc = grow(g_w.vals, (size_t)s, "transr vals"))) return rc;
        if ((rc = grow(g_w.vals2, (size_t)s, "transr vals2"))) return rc;
        if ((rc = grow(g_w.job_ent, (size_t)s, "transr job_ent"))) return rc;
        if ((rc = grow(g_w.row_ent, (size_t)s, "transr row_ent"))) return rc;
        size_t bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, bytes, g_w.keys, g_w.keys2, g_w.vals, g_w.vals2, (size_t)s, 0, 32, nullptr);
        if (bytes > g_w.sort_tmp_bytes) {
            if (g_w.sort_tmp) (void)hipFree(g_w.sort_tmp);
            g_w.sort_tmp = nullptr;
            if ((rc = hip_check(hipMalloc(&g_w.sort_tmp, bytes), "transr sort temp"))) return rc;
            g_w.sort_tmp_bytes = bytes;
        }
        g_w.cap_slots = s; g_w.cap_dim = d; g_w.pad_ready = -1;
    }
    if (R > g_w.cap_rel) {
        if ((rc = grow(g_w.bucket_start, (size_t)R + 2, "transr bucket_start"))) return rc;
        g_w.cap_rel = R;
    }
    int64_t tiles = slots / 32 + R + 2;
    if (tiles > g_w.cap_tiles) {
        if ((rc = grow(g_w.tile_rel, (size_t)tiles, "transr tile_rel"))) return rc;
        if ((rc = grow(g_w.tile_row0, (size_t)tiles, "transr tile_row0"))) return rc;
        if (!g_w.n_tiles && (rc = grow(g_w.n_tiles, 1, "transr n_tiles"))) return rc;
        g_w.cap_tiles = tiles;
    }
    return KGE_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void prep_kernel(const int32_t *__restrict__ bh, const int32_t *__restrict__ bt, const int32_t *__restrict__ br,
                            long long n_pos, long long n_neg, long long stride, int negative_rel, int R,
                            int32_t *__restrict__ keys, int32_t *__restrict__ vals, int32_t *__restrict__ job_ent, int32_t *__restrict__ rec_dst) {
    const long long total = 2 * n_pos * (1 + n_neg);
    for (long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += (long long)gridDim.x * blockDim.x) {
        if (rec_dst) rec_dst[slot] = -1;      // (dgrad's record mode: positions that no tile covers -- the alias slots -- carry no record)
        const long long s = slot >> 1;
        const int side = (int)(slot & 1);
        const long long k = s / n_pos, b = s - k * n_pos;
        const long long h = bh[b], t = bt[b], r = br[b];
        long long e = side ? t : h, key = r;
        if (k > 0) {
            const long long j = b + k * stride;
            const long long nh = bh[j], nt = bt[j], nr = br[j];
            const NegClass nc = classify_negative<KGE_TRANSR>(h, t, r, nh, nt, nr, negative_rel);
            const bool canonical = nc.fast ? (side ? !nc.same_t : !nc.same_h) : true;
            e = side ? nt : nh;
            key = canonical ? (negative_rel == 0 ? r : nr) : R;  // TransR.py:57-65: positive's matrix when negative_rel == 0
        }
        keys[slot] = (int32_t)key;
        vals[slot] = (int32_t)slot;
        job_ent[slot] = (int32_t)e;
    }
}

// group layout: one key per GROUP (the positive's relation); also blanks the record keys of the row space (rows without a record)
__global__ void group_keys_kernel(const int32_t *__restrict__ br, long long n_pos, int32_t *__restrict__ keys, int32_t *__restrict__ vals,
                                  int32_t *__restrict__ rec_dst, long long n_rows) {
    const long long n = n_pos > n_rows ? n_pos : n_rows;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (i < n_pos) { keys[i] = br[i]; vals[i] = (int32_t)i; }
        if (rec_dst && i < n_rows) rec_dst[i] = -1;
    }
}

// bucket_start[r] = first sorted position with key >= r (r = 0..R+1), then the (relation, row tile) map
__global__ __launch_bounds__(1024) void bounds_kernel(const int32_t *__restrict__ sorted_keys, int J, int R,
                                                      int32_t *__restrict__ bucket_start, int32_t *__restrict__ tile_rel,
                                                      int32_t *__restrict__ tile_row0, int32_t *__restrict__ n_tiles, int tile_shift) {
    // tile_shift: log2 of the rows per tile (5 = the 32-row tiles of rows_gemm_kernel, 7 = the 128-row tiles of the v2 kernels)
    const int tile_rows = 1 << tile_shift;
    __shared__ int chunk_tiles[1024];
    for (int r = threadIdx.x; r <= R + 1; r += blockDim.x) {
        int lo = 0, hi = J;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (sorted_keys[mid] < r) lo = mid + 1; else hi = mid; }
        bucket_start[r] = lo;
    }
    __syncthreads();
    const int per = (R + (int)blockDim.x - 1) / (int)blockDim.x;
    const int r0 = threadIdx.x * per, r1 = min(R, r0 + per);
    int mine = 0;
    for (int r = r0; r < r1; r++) mine += (bucket_start[r + 1] - bucket_start[r] + tile_rows - 1) >> tile_shift;
    chunk_tiles[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < (int)blockDim.x; i++) { int v = chunk_tiles[i]; chunk_tiles[i] = acc; acc += v; }
        n_tiles[0] = acc;
    }
    __syncthreads();
    int t = chunk_tiles[threadIdx.x];
    for (int r = r0; r < r1; r++)
        for (int row = bucket_start[r]; row < bucket_start[r + 1]; row += tile_rows) { tile_rel[t] = r; tile_row0[t] = row; t++; }
}

// ---------------------------------------------------------------------------------------------
// Counting sort of the jobs by matrix relation (R + 1 keys: the last one collects the alias slots) in two launches, with the
// bucket starts and the (relation, row tile) map as by-products -- instead of rocPRIM's radix sort (4-8 launches) plus the
// single-workgroup bounds kernel: at the reference's batch sizes those launches were a sixth of the TransR step.
// The order inside a bucket is whatever the cursors hand out (projection and dgrad are per-row; wgrad sums a bucket's rows
// in a different order from run to run, as its atomics already did).
// ---------------------------------------------------------------------------------------------
constexpr int kRelBins = 4096;        // (R + 1) * kRelSub must fit the LDS histogram
constexpr int kRelTile = 4096;        // jobs per workgroup of the scatter
// Every relation is counted in kRelSub adjacent bins, a job taking the bin of its thread (tid % kRelSub): the Zipf-head relation of a
// skewed graph holds a sixth of the jobs and its one LDS counter serialised the wave's atomics (SQ_LDS_BANK_CONFLICT 75-91 % of the
// LDS cycles of both kernels).  The sub-bins of a relation are adjacent in the sorted order, so together they are its bucket.
constexpr int kRelSub = 4;

__global__ __launch_bounds__(256) void rel_count_kernel(const int32_t *__restrict__ keys, int J, int bins, int32_t *__restrict__ hist,
                                                        int32_t *__restrict__ next_pair) {
    __shared__ int h[kRelBins];
    if (blockIdx.x == 0)      // the histogram + cursor pair the NEXT call will use (the pairs alternate): no memset launch per step
        for (int i = threadIdx.x; i < 2 * kRelBins; i += 256) next_pair[i] = 0;
    for (int i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    for (int i = blockIdx.x * kRelTile + threadIdx.x; i < min(J, (blockIdx.x + 1) * kRelTile); i += 256)
        atomicAdd(&h[keys[i] * kRelSub + (threadIdx.x & (kRelSub - 1))], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ __launch_bounds__(256) void rel_scatter_kernel(const int32_t *__restrict__ keys, const int32_t *__restrict__ vals, int J, int R,
                                                          int32_t *__restrict__ hist, int32_t *__restrict__ cursor,
                                                          int32_t *__restrict__ sorted_vals, int32_t *__restrict__ bucket_start,
                                                          int32_t *__restrict__ tile_rel, int32_t *__restrict__ tile_row0,
                                                          int32_t *__restrict__ n_tiles, int tile_shift, SamplerArgs ride, int n_own, int n_ride,
                                                          int gps, int32_t *__restrict__ bucket_rows) {
    // gps > 0 (group layout): the keys are groups; bucket_start stays in group units and the row space -- every relation's groups in
    // sub-tiles of 16 rows holding gps groups each -- gets its own starts (bucket_rows, [R + 2]); the tile map is in ROW units
    // workgroups beyond the scatter's own carry the NEXT batch's sampler (kge_sampling_attach): this launch has a dozen to a
    // hundred workgroups and leaves the chip idle, the sampler is independent of everything in the step
    if ((int)blockIdx.x >= n_own) {
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - n_own, bern_lds);
        return;
    }
    const int bins = (R + 1) * kRelSub;
    __shared__ int start[kRelBins + 1];      // exclusive scan of the histogram
    __shared__ int cnt[kRelBins];            // this tile's histogram, then its base per bin
    __shared__ int part[256];
    // exclusive scan: 16 consecutive bins per thread, then the 256 partial sums
    constexpr int PER = kRelBins / 256;
    int local[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { const int i = threadIdx.x * PER + k; local[k] = i < bins ? hist[i] : 0; sum += local[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) { const int i = threadIdx.x * PER + k; if (i <= kRelBins) start[i] = run; run += local[k]; }
    if (threadIdx.x == 255) start[kRelBins] = run;
    for (int i = threadIdx.x; i < bins; i += 256) cnt[i] = 0;
    __syncthreads();
    // this tile
    constexpr int ITEMS = kRelTile / 256;
    int key[ITEMS], rank[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const int i = blockIdx.x * kRelTile + threadIdx.x + 256 * k;
        key[k] = i < J ? keys[i] * kRelSub + (int)(threadIdx.x & (kRelSub - 1)) : -1;
        rank[k] = key[k] >= 0 ? atomicAdd(&cnt[key[k]], 1) : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256) { const int c = cnt[i]; cnt[i] = c ? start[i] + atomicAdd(&cursor[i], c) : 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const int i = blockIdx.x * kRelTile + threadIdx.x + 256 * k;
        if (key[k] >= 0) sorted_vals[cnt[key[k]] + rank[k]] = vals[i];
    }
    if (blockIdx.x != 0) return;
    // workgroup 0: bucket starts (bucket_start[r] = first position with key >= r, r = 0 .. R+1) and the tile map
    for (int r = threadIdx.x; r <= R + 1; r += 256) bucket_start[r] = r <= R ? start[r * kRelSub] : J;
    const int tile_rows = 1 << tile_shift;
    // relation r's bucket = its kRelSub sub-bins; thread t maps the relations t*RPT .. t*RPT + RPT-1
    constexpr int RPT = PER / kRelSub;
    if (gps > 0) {
        int rows_of[RPT], rows_mine = 0, tiles_g = 0;
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int r = threadIdx.x * RPT + k;
            rows_of[k] = r < R ? 16 * ((start[(r + 1) * kRelSub] - start[r * kRelSub] + gps - 1) / gps) : 0;
            rows_mine += rows_of[k];
            tiles_g += (rows_of[k] + tile_rows - 1) >> tile_shift;
        }
        __syncthreads();
        part[threadIdx.x] = rows_mine;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        int row_run = part[threadIdx.x] - rows_mine;
        const int rows_total = part[255];
        __syncthreads();
        part[threadIdx.x] = tiles_g;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        int tix = part[threadIdx.x] - tiles_g;
        if (threadIdx.x == 255) n_tiles[0] = part[255];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int r = threadIdx.x * RPT + k;
            if (r < R) {
                bucket_rows[r] = row_run;
                for (int row = row_run; row < row_run + rows_of[k]; row += tile_rows) { tile_rel[tix] = r; tile_row0[tix] = row; tix++; }
                row_run += rows_of[k];
            }
        }
        for (int r = R + (int)threadIdx.x; r <= R + 1; r += 256) bucket_rows[r] = rows_total;
        return;
    }
    int tiles_mine = 0;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int r = threadIdx.x * RPT + k;
        if (r < R) tiles_mine += (start[(r + 1) * kRelSub] - start[r * kRelSub] + tile_rows - 1) >> tile_shift;
    }
    __syncthreads();
    part[threadIdx.x] = tiles_mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int tix = part[threadIdx.x] - tiles_mine;
    if (threadIdx.x == 255) n_tiles[0] = part[255];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int r = threadIdx.x * RPT + k;
        if (r < R)
            for (int row = start[r * kRelSub]; row < start[(r + 1) * kRelSub]; row += tile_rows) { tile_rel[tix] = r; tile_row0[tix] = row; tix++; }
    }
}

// predict: every triple uses the matrix of predict_r[0] (TransR.py:83): one bucket holding all slots
__global__ void predict_prep_kernel(const int32_t *__restrict__ bh, const int32_t *__restrict__ bt, const int32_t *__restrict__ br,
                                    long long n, int R, int32_t *__restrict__ vals, int32_t *__restrict__ job_ent,
                                    int32_t *__restrict__ bucket_start, int32_t *__restrict__ tile_rel,
                                    int32_t *__restrict__ tile_row0, int32_t *__restrict__ n_tiles) {
    const long long total = 2 * n;
    const int r0 = br[0];
    for (long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += (long long)gridDim.x * blockDim.x) {
        vals[slot] = (int32_t)slot;
        job_ent[slot] = (slot & 1) ? bt[slot >> 1] : bh[slot >> 1];
        if ((slot & 31) == 0) { tile_rel[slot >> 5] = r0; tile_row0[slot >> 5] = (int32_t)slot; }
    }
    if (blockIdx.x == 0) {
        for (int r = threadIdx.x; r <= R + 1; r += blockDim.x) bucket_start[r] = r <= r0 ? 0 : (int32_t)total;
        if (threadIdx.x == 0) n_tiles[0] = (int)((total + 31) >> 5);
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA tiles.  v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// the 16 accumulator registers hold D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 32;    // K chunk staged in LDS
constexpr int TN = 128;   // output columns per workgroup (4 waves x 32)

enum { GEMM_PROJECT = 0, GEMM_DGRAD = 1 };

struct GemmArgs {
    const float *ent;       // [E, De]
    const float *mat;       // [R, De*Dr]
    const float *GP;        // [slots, Dr]
    float *P;               // [slots, Dr]
    float *g_ent;           // [E, De]
    const int32_t *sorted_slots, *job_ent, *bucket_start, *tile_rel, *tile_row0, *n_tiles;
    int De, Dr;
    // dgrad, large steps: the output row of sorted job position p is STORED as float record p (rec_out[p * De ..]) with its entity
    // in rec_dst[p], and summed per entity afterwards by the float-record sort + segmented sum (transe_counts.hip) -- memory-side
    // fp32 atomics run at ~1.1 TB/s whatever their shape, and 82 MB of them per step were a third of the dgrad kernel
    float *rec_out;
    int32_t *rec_dst;
    int32_t *row_ent;       // v3: entity of sorted job position p, written by the projection for wgrad3_kernel (one lookup instead of slot -> entity)
    // group layout (sampler-shaped batches, negative_rel == 0, 2 + n <= 16): GROUPS, not (triple, side) jobs, are sorted by relation, and
    // the U = 2 + n canonical rows of a group (h, t, the new entity of each negative) sit side by side inside one 16-row sub-tile of
    // the relation's row space (gps = 16 / U groups per sub-tile, the rest of a sub-tile are pad rows: slot `pad_slot`, a zero GP row).
    // The projection derives its rows from the sorted groups and writes sorted_slots / job_ent / row_ent for dgrad and wgrad.
    const int32_t *sorted_groups, *group_start, *bh, *bt;
    int32_t *sorted_slots_w, *job_ent_w;
    long long n_pos, stride;
    int U, gps, pad_slot;
    // the vector stage inside the projection's epilogue (group layout): P never leaves the CU -- a group's projected rows lie side by
    // side in the wave's LDS staging block; what the projection stores is GP (TransR.py:19-23, Model.py loss, backward through the
    // normalisation), the loss partial goes through finish_loss (fb), the relation gradient as one row of atomics per wave and tile
    int fuse_vec;
    const float *rel;
    float *g_rel, *GP_w;
    FbArgs fb;
};

template <int MODE>
__global__ __launch_bounds__(256) void rows_gemm_kernel(GemmArgs a) {
    const int tile = blockIdx.x;
    if (tile >= a.n_tiles[0]) return;
    __shared__ float As[32][KC + 1];
    __shared__ float Bs[KC][TN + 1];
    __shared__ int s_slot[32], s_ent[32];
    const int r = a.tile_rel[tile];
    const int row0 = a.tile_row0[tile];
    const int rows = min(32, a.bucket_start[r + 1] - row0);
    const int K = MODE == GEMM_PROJECT ? a.De : a.Dr;
    const int N = MODE == GEMM_PROJECT ? a.Dr : a.De;
    const int col0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 32) {
        int sl = tid < rows ? a.sorted_slots[row0 + tid] : -1;
        s_slot[tid] = sl;
        s_ent[tid] = sl >= 0 ? a.job_ent[sl] : -1;
    }
    __syncthreads();
    const float *M = a.mat + (long long)r * a.De * a.Dr;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += KC) {
        for (int idx = tid; idx < 32 * KC; idx += 256) {
            const int i = idx / KC, kk = idx - i * KC, kg = k0 + kk;
            float v = 0.f;
            if (i < rows && kg < K)
                v = MODE == GEMM_PROJECT ? a.ent[(long long)s_ent[i] * a.De + kg] : a.GP[(long long)s_slot[i] * a.Dr + kg];
            As[i][kk] = v;
        }
        if (MODE == GEMM_PROJECT) {  // B[k][j] = M[k][col0+j] : coalesced along j
            for (int idx = tid; idx < KC * TN; idx += 256) {
                const int kk = idx / TN, j = idx - kk * TN, kg = k0 + kk, cg = col0 + j;
                Bs[kk][j] = (kg < K && cg < N) ? M[(long long)kg * a.Dr + cg] : 0.f;
            }
        } else {  // B[k][j] = M[col0+j][k] : read along k, store transposed
            for (int idx = tid; idx < KC * TN; idx += 256) {
                const int j = idx / KC, kk = idx - j * KC, kg = k0 + kk, cg = col0 + j;
                Bs[kk][j] = (kg < K && cg < N) ? M[(long long)cg * a.Dr + kg] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < KC; k2 += 2) {
            const float av = As[lane & 31][k2 + (lane >> 5)];
            const float bv = Bs[k2 + (lane >> 5)][wave * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int cg = col0 + wave * 32 + (lane & 31);
    if (cg < N) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (row < rows) {
                if (MODE == GEMM_PROJECT) a.P[(long long)s_slot[row] * a.Dr + cg] = acc[reg];
                else __builtin_amdgcn_global_atomic_fadd_f32(
                        (__attribute__((address_space(1))) float *)(a.g_ent + (long long)s_ent[row] * a.De + cg), acc[reg]);
            }
        }
    }
}

// g_M[r][i][j] += sum over the bucket's rows of ent[e_row][i] * GP[slot_row][j].
// Work unit = WG_TILES consecutive 32-row tiles of the (relation-sorted) job list, so a hub relation's
// bucket is split over many workgroups; a workgroup flushes its 32 x 128 partial with fp32 atomics
// whenever the relation changes inside its span (Zipf-skewed relations: one bucket can hold 15% of
// the batch, a single owner per relation would serialise on it).
constexpr int WG_TILES = 8;

// sole = this workgroup's span holds the relation's WHOLE bucket: it is the only writer of its output block of g_M[r] (zero
// on entry), so the partial is stored, not added with memory-side atomics (config #4's batch: 46 rows per relation, every
// relation has one owner, and 38 MB of fp32 atomics per step become plain stores)
__device__ __forceinline__ void wgrad_flush(const GemmArgs &a, float *__restrict__ g_mat, int r, int i0, int jg, int lane,
                                            const f32x16 &acc, bool sole) {
    if (jg >= a.Dr) return;
    float *G = g_mat + (long long)r * a.De * a.Dr;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int ig = i0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (ig < a.De) {
            if (sole) G[(long long)ig * a.Dr + jg] = acc[reg];
            else __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(G + (long long)ig * a.Dr + jg), acc[reg]);
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_kernel(GemmArgs a, float *__restrict__ g_mat, int tiles_i, int tile_rows, int wg_tiles) {
    // tile_rows: height of the tiles in the map (32, or 128 when the map was built for the v2 kernels)
    const int grp = blockIdx.x / tiles_i, it = blockIdx.x - grp * tiles_i;
    const int n_tiles = a.n_tiles[0];
    const int t0 = grp * wg_tiles;
    if (t0 >= n_tiles) return;
    const int t1 = min(t0 + wg_tiles, n_tiles);
    __shared__ float As[KC][32 + 1];
    __shared__ float Bs[KC][TN + 1];
    __shared__ int s_slot[KC], s_ent[KC];
    const int i0 = it * 32, j0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jg = j0 + wave * 32 + (lane & 31);
    f32x16 acc = {0};
    int r_cur = a.tile_rel[t0];
    // rows [span_lo, span_hi) of the relation-sorted job list belong to this workgroup
    const int span_lo = a.tile_row0[t0];
    const int span_hi = a.tile_row0[t1 - 1] + min(tile_rows, a.bucket_start[a.tile_rel[t1 - 1] + 1] - a.tile_row0[t1 - 1]);
#define KGE_SOLE(rel_) (a.bucket_start[rel_] >= span_lo && a.bucket_start[(rel_) + 1] <= span_hi)
    for (int t = t0; t < t1; t++) {
        const int r = a.tile_rel[t];
        if (r != r_cur) {
            wgrad_flush(a, g_mat, r_cur, i0, jg, lane, acc, KGE_SOLE(r_cur));
            acc = f32x16{0};
            r_cur = r;
        }
        const int trow0 = a.tile_row0[t];
        const int trows = min(tile_rows, a.bucket_start[r + 1] - trow0);
        for (int c0 = 0; c0 < trows; c0 += 32) {
            const int row0 = trow0 + c0;
            const int rows = min(32, trows - c0);
            __syncthreads();
            if (tid < KC) {
                int sl = tid < rows ? a.sorted_slots[row0 + tid] : -1;
                s_slot[tid] = sl;
                s_ent[tid] = sl >= 0 ? a.job_ent[sl] : -1;
            }
            __syncthreads();
            for (int idx = tid; idx < KC * 32; idx += 256) {
                const int kk = idx >> 5, i = idx & 31;
                As[kk][i] = (kk < rows && i0 + i < a.De) ? a.ent[(long long)s_ent[kk] * a.De + i0 + i] : 0.f;
            }
            for (int idx = tid; idx < KC * TN; idx += 256) {
                const int kk = idx / TN, j = idx - kk * TN;
                Bs[kk][j] = (kk < rows && j0 + j < a.Dr) ? a.GP[(long long)s_slot[kk] * a.Dr + j0 + j] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < KC; k2 += 2) {
                const float av = As[k2 + (lane >> 5)][lane & 31];
                const float bv = Bs[k2 + (lane >> 5)][wave * 32 + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
    }
    wgrad_flush(a, g_mat, r_cur, i0, jg, lane, acc, KGE_SOLE(r_cur));
#undef KGE_SOLE
}


// ---------------------------------------------------------------------------------------------
// v2 tiles (training path): v_mfma_f32_16x16x4_f32, 16-wide granularity so that dim 200 pads to 208 (4 %)
// instead of 256 (28 %), 64-row tiles per workgroup (each wave owns 16 rows and ALL <= 13 column tiles, so a
// relation's matrix is read once per 64 rows), operands staged in LDS in layouts whose MFMA reads are
// bank-conflict free:
//   A operand: lane l supplies A[i = l&15][k = l>>4];  B operand: B[k = l>>4][j = l&15];
//   accumulator reg v of lane l holds D[i = 4*(l>>4) + v][j = l&15].
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: the staging register arrays must only ever be indexed by constants (a rolled loop sends them to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// keep-or-zero of a staged float4, component by component: `cond ? reg[u] : zero` on the whole vector becomes a select of two
// ADDRESSES, which drags the register array it indexes into scratch memory (176 bytes per lane in the projection kernel:
// every K chunk then went global -> scratch -> scratch load -> LDS)
__device__ __forceinline__ float4 keep_if(bool c, const float4 &v) { return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f); }

constexpr int RM2 = 64;          // rows per workgroup
constexpr int KC2 = 40;          // K chunk: 10 MFMA k-steps
constexpr int NT2 = 13;          // 16-column tiles per workgroup: 208 columns
constexpr int LDA2 = KC2 + 2;    // [i][k] layout, stride 42: lanes (i, k..k+1) hit 32 distinct banks
constexpr int LDB2 = NT2 * 16;   // [k][j] layout, stride 208 = 16 mod 32: lanes (k..k+1, j) hit 32 distinct banks

// Staging: every thread issues ALL its global loads of a chunk back to back into registers (unconditional, with
// clamped addresses -- the compiler serialises conditional loads, one memory latency each), the chunk after the one
// being multiplied is in flight during the MFMA loop, and invalid elements are zeroed on the way into LDS.
// The MFMA block is branch-free (a condition per tile puts every MFMA in its own basic block): the kernels are
// instantiated for NT = 7 or 13 column tiles (dims <= 112 / <= 208, multiples of 4) and always multiply all of them --
// padding rows and columns are zero in LDS; other shapes take the 32x32x2 kernels.
// RT2 16-row sub-tiles per wave: a workgroup covers 64*RT2 rows, so a relation's 160 kB matrix (which comes from the
// Infinity Cache, not HBM) is re-read once per 128 rows: 39 flop per byte instead of 24.
constexpr int RT2 = 2;
constexpr int RW2 = RM2 * RT2;   // rows per workgroup (128)

// One K chunk (KC2 / 4 = 10 k-steps) of a wave's NS x NT accumulator tiles.  Software-pipelined by hand: the NS + NT operand reads
// of k-step s+1 are issued BEFORE the NS*NT MFMAs of k-step s (two register sets used alternately), so an LDS round trip is
// always covered by a full k-step of matrix work.  The compiler's own schedule for the plain loop was "one ds_read, s_waitcnt
// lgkmcnt(0), its MFMAs" repeated -- the matrix pipe idle for an LDS latency per group of four MFMAs (MFMA busy 44 % / 30 % of the
// project / dgrad kernels, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES, profiles/r03_*).  NS + NT = 15 reads in flight is also what the
// 4-bit lgkmcnt counter can express exactly.
template <int MODE, int NT, int NS>
__device__ __forceinline__ void gemm2_mfma_block(const float *__restrict__ As, const float *__restrict__ Bs, f32x4 (&acc)[RT2][NT], int wave, int lane) {
    constexpr int LDBN = NT * 16;
    constexpr int STEPS = KC2 / 4;
    const int kq = lane >> 4, j = lane & 15;
    const float *a_base[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++) a_base[s2] = As + ((4 * s2 + wave) * 16 + j) * LDA2 + kq;
    const float *b_base = MODE == GEMM_PROJECT ? Bs + kq * LDBN + j : Bs + j * LDA2 + kq;
    float av[2][NS], bv[2][NT];
    auto fetch = [&](int buf, int st) {
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) av[buf][s2] = a_base[s2][4 * st];
#pragma unroll
        for (int t = 0; t < NT; t++) bv[buf][t] = MODE == GEMM_PROJECT ? b_base[4 * st * LDBN + t * 16] : b_base[t * 16 * LDA2 + 4 * st];
    };
    fetch(0, 0);
    static_for<0, STEPS>([&](auto sc) {
        constexpr int st = decltype(sc)::value;
        constexpr int cur = st & 1;
        if constexpr (st + 1 < STEPS) fetch(cur ^ 1, st + 1);
        __builtin_amdgcn_sched_barrier(0);      // keep the next step's reads in front of this step's matrix work
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) acc[s2][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][s2], bv[cur][t], acc[s2][t], 0, 0, 0);
    });
}

template <int MODE, int NT>
__global__ __launch_bounds__(256, 2) void rows_gemm2_kernel(GemmArgs a) {   // two workgroups per CU: their barriers and load waits overlap
    constexpr int LDBN = NT * 16;    // [k][j] stride: 112 or 208 floats, both = 16 mod 32 banks
    const int tile = blockIdx.x;
    if (tile >= a.n_tiles[0]) return;
    __shared__ __attribute__((aligned(16))) float As[RW2 * LDA2];
    __shared__ __attribute__((aligned(16))) float Bs[(MODE == GEMM_PROJECT) ? KC2 * LDBN : LDBN * LDA2];
    __shared__ int s_slot[RW2], s_ent[RW2];
    const int r = a.tile_rel[tile];
    const int row0 = a.tile_row0[tile];
    const int rows = min(RW2, a.bucket_start[r + 1] - row0);
    const int K = MODE == GEMM_PROJECT ? a.De : a.Dr;
    // blockIdx.y: column block of NT*16 output columns (one block covers everything when NT*16 >= the output width; at small
    // batches -- one 128-row tile per relation, fewer tiles than CUs -- the 7-tile instantiation is launched with two column
    // blocks so that twice as many workgroups, each with half the matrix to stream, are in flight)
    const int j0 = blockIdx.y * LDBN;
    const int ncols = min(LDBN, (MODE == GEMM_PROJECT ? a.Dr : a.De) - j0);     // multiple of 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < RW2) {
        int sl = a.sorted_slots[row0 + min(tid, rows - 1)];   // padding rows repeat the last live row (never stored)
        s_slot[tid] = sl;
        s_ent[tid] = a.job_ent[sl];
    }
    __syncthreads();
    const float *M = a.mat + (long long)r * a.De * a.Dr;
    f32x4 acc[RT2][NT];
#pragma unroll
    for (int s2 = 0; s2 < RT2; s2++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[s2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // sub-tile s2 of wave w covers rows (4*s2 + w)*16 ..: interleaved, so a short tile keeps all four waves busy
    const int n_live = (wave * 16 < rows ? 1 : 0) + ((4 + wave) * 16 < rows ? 1 : 0);
    constexpr int AQ = KC2 / 4;                       // float4 per A row chunk
    constexpr int NA = (RW2 * AQ + 255) / 256;        // A float4 loads per thread (5)
    constexpr int BROWS = MODE == GEMM_PROJECT ? KC2 : LDBN;
    constexpr int BQ = MODE == GEMM_PROJECT ? LDBN / 4 : KC2 / 4;
    constexpr int NB = (BROWS * BQ + 255) / 256;      // B float4 loads per thread (9 at NT = 13)
    float4 ra[NA], rb[NB];
#define KGE_LOAD_CHUNK(k0_)                                                                                                  \
    {                                                                                                                        \
        static_for<0, NA>([&](auto uc) {                                                                                     \
            constexpr int u = decltype(uc)::value;                                                                           \
            const int idx = min(tid + 256 * u, RW2 * AQ - 1);                                                                \
            const int i = idx / AQ, q = idx - i * AQ;                                                                        \
            const int kg = min((k0_) + 4 * q, K - 4);                                                                        \
            const float *src = MODE == GEMM_PROJECT ? a.ent + (long long)s_ent[i] * a.De + kg                                \
                                                    : a.GP + (long long)s_slot[i] * a.Dr + kg;                               \
            ra[u] = *reinterpret_cast<const float4 *>(src);                                                                  \
        });                                                                                                                  \
        static_for<0, NB>([&](auto uc) {                                                                                     \
            constexpr int u = decltype(uc)::value;                                                                           \
            const int idx = min(tid + 256 * u, BROWS * BQ - 1);                                                              \
            const int rr = idx / BQ, q = idx - rr * BQ;                                                                      \
            const float *src = MODE == GEMM_PROJECT ? M + (long long)min((k0_) + rr, K - 1) * a.Dr + j0 + min(4 * q, ncols - 4) \
                                                    : M + (long long)(j0 + min(rr, ncols - 1)) * a.Dr + min((k0_) + 4 * q, K - 4);  \
            rb[u] = *reinterpret_cast<const float4 *>(src);                                                                  \
        });                                                                                                                  \
    }
    KGE_LOAD_CHUNK(0)
    for (int k0 = 0; k0 < K; k0 += KC2) {
        if (k0 > 0) __syncthreads();          // the previous chunk's MFMA reads are done
        {
            static_for<0, NA>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int idx = tid + 256 * u;
                if (idx < RW2 * AQ) {
                    const int i = idx / AQ, q = idx - i * AQ;
                    const float4 v = keep_if(i < rows && k0 + 4 * q < K, ra[u]);
                    float2 *dst = reinterpret_cast<float2 *>(&As[i * LDA2 + 4 * q]);   // stride 42 floats: 8-byte aligned
                    dst[0] = make_float2(v.x, v.y);
                    dst[1] = make_float2(v.z, v.w);
                }
            });
            static_for<0, NB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int idx = tid + 256 * u;
                if (idx < BROWS * BQ) {
                    const int rr = idx / BQ, q = idx - rr * BQ;
                    if (MODE == GEMM_PROJECT) {
                        *reinterpret_cast<float4 *>(&Bs[rr * LDBN + 4 * q]) = keep_if(k0 + rr < K && 4 * q < ncols, rb[u]);
                    } else {
                        const float4 v = keep_if(rr < ncols && k0 + 4 * q < K, rb[u]);
                        float2 *dst = reinterpret_cast<float2 *>(&Bs[rr * LDA2 + 4 * q]);
                        dst[0] = make_float2(v.x, v.y);
                        dst[1] = make_float2(v.z, v.w);
                    }
                }
            });
        }
        __syncthreads();
        if (k0 + KC2 < K) KGE_LOAD_CHUNK(k0 + KC2)   // in flight during the MFMA loop
        // a wave with rows in both of its sub-tiles multiplies both; a tile of at most 64 rows (the tail of a relation's bucket:
        // ~1 tile in 8 at B = 34 014, nearly every tile at the reference's own batch) only has first sub-tiles, and the
        // one-sub-tile instantiation on the SAME accumulator array does half the matrix work
        // (7-tile instantiations only: with 13 column tiles a second block variant pushes the kernel past 256 VGPRs into scratch)
        if constexpr (NT <= 7) {
            if (rows > RM2) gemm2_mfma_block<MODE, NT, RT2>(As, Bs, acc, wave, lane);
            else if (n_live > 0) gemm2_mfma_block<MODE, NT, 1>(As, Bs, acc, wave, lane);
        } else {
            if (n_live > 0) gemm2_mfma_block<MODE, NT, RT2>(As, Bs, acc, wave, lane);
        }
    }
#undef KGE_LOAD_CHUNK
    if (MODE == GEMM_DGRAD && a.rec_out && blockIdx.y == 0 && tid < rows) a.rec_dst[row0 + tid] = s_ent[tid];
#pragma unroll
    for (int s2 = 0; s2 < RT2; s2++) {
        if ((4 * s2 + wave) * 16 >= rows) continue;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int j = t * 16 + (lane & 15);
            if (j < ncols) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int row = (4 * s2 + wave) * 16 + 4 * (lane >> 4) + v;
                    if (row < rows) {
                        if (MODE == GEMM_PROJECT) a.P[(long long)s_slot[row] * a.Dr + j0 + j] = acc[s2][t][v];
                        else if (a.rec_out) a.rec_out[(long long)(row0 + row) * a.De + j0 + j] = acc[s2][t][v];
                        else __builtin_amdgcn_global_atomic_fadd_f32(
                                (__attribute__((address_space(1))) float *)(a.g_ent + (long long)s_ent[row] * a.De + j0 + j), acc[s2][t][v]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// v3 row GEMMs: fp32 products on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
// The fp32 MFMA runs at the vector rate (64 flop / clk / SIMD = 1/16 of the bf16 MFMA): at B = 34 014 the v2 K loop is 89 % matrix-
// pipe time and the kernels cannot get past ~57 % of the fp32 peak.  Every fp32 value splits EXACTLY into three bf16 terms
// (x = x1 + x2 + x3: x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2 -- 8 + 8 + 8 significand bits, the residuals are exact in
// fp32), every bf16 x bf16 product is exact in the pipe's fp32 accumulation, and a.b = sum_ij a_i b_j.  Six of the nine term
// products are formed -- (1,1) (1,2) (2,1) (2,2) (1,3) (3,1); the three dropped ones are below 2^-25 |a b|, under the rounding of
// ONE fp32 accumulate (2^-24) -- so a k-block of 32 costs 6 x 16 = 96 cycles on a SIMD where the fp32 MFMA takes 8 x 32 = 256.
// Measured against fp64 the result is as close as the fp32-MFMA kernels' (tests/test_gpu_models.py, both tilings vs the oracle).
//
// LDS image of an operand: [term][row or column][32 bf16 of the k-block] = 64-byte rows, the 16-byte chunk index XORed with
// (row >> 2) & 3, so the ds_read_b128 of an MFMA fragment (16 rows x one chunk) and the staging writes spread over all banks.
// The split is done on the way into LDS (5.5 VALU per element, done once per element and k-block by the staging threads).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KB3 = 32;            // k-block: one bf16 MFMA per term pair
constexpr int RW3 = 128;           // rows per workgroup (wave w: sub-tiles w and w + 4, as v2)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32 (round to nearest even)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// two fp32 values -> their three bf16 terms, packed pairwise (low half: a)
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = pack_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(ra, rb);
    p3 = pack_bf16(ra - __uint_as_float(p2 << 16), rb - __uint_as_float(p2 & 0xffff0000u));
}
__device__ __forceinline__ void split3_8(const float (&v)[8], uint4 &t1, uint4 &t2, uint4 &t3) {
    split3_pair(v[0], v[1], t1.x, t2.x, t3.x);
    split3_pair(v[2], v[3], t1.y, t2.y, t3.y);
    split3_pair(v[4], v[5], t1.z, t2.z, t3.z);
    split3_pair(v[6], v[7], t1.w, t2.w, t3.w);
}
// byte offset of chunk `ch` (8 bf16) of row `row` of term plane `term` (planes of `plane_rows` rows)
__device__ __forceinline__ int lds3_off(int plane_rows, int term, int row, int ch) {
    return ((term * plane_rows + row) << 6) + ((ch ^ ((row >> 2) & 3)) << 4);
}

// one k-block of a wave's NS x NT accumulator tiles: the A fragments are in registers (af), the B fragments of column tile t + 1 are
// requested in front of the 6 NS matrix instructions of tile t
template <int NT, int NS>
__device__ __forceinline__ void gemm3_mfma_block(const bf16x8 (&af)[RT2][3], const unsigned char *__restrict__ Bs, f32x4 (&acc)[RT2][NT], int lane) {
    constexpr int NCOL = NT * 16;
    const int r16 = lane & 15, g = lane >> 4;
    bf16x8 bf[2][3];
    auto fetch = [&](int buf, int ct) {
#pragma unroll
        for (int t = 0; t < 3; t++) bf[buf][t] = *reinterpret_cast<const bf16x8 *>(Bs + lds3_off(NCOL, t, ct * 16 + r16, g));
    };
    fetch(0, 0);
    static_for<0, NT>([&](auto tc) {
        constexpr int ct = decltype(tc)::value;
        constexpr int cur = ct & 1;
        if constexpr (ct + 1 < NT) fetch(cur ^ 1, ct + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) {
            f32x4 c = acc[s2][ct];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][0], bf[cur][2], c, 0, 0, 0);   // small terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][2], bf[cur][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][1], bf[cur][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][0], bf[cur][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][1], bf[cur][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][0], bf[cur][0], c, 0, 0, 0);
            acc[s2][ct] = c;
        }
    });
}

// Structure of a workgroup (128 rows of one relation x all output columns, four waves, wave w: rows 16 w .. and 64 + 16 w ..):
//  * the A operand never touches LDS: a lane of a 16x16x32 fragment holds 8 consecutive k of ONE row, and a wave's rows are its
//    own, so every lane reads its 32 bytes straight from the gathered row (ent / GP), one k-block ahead, and splits them in
//    registers;
//  * the B operand (the relation's matrix, shared by the four waves) is split on the way into a DOUBLE-buffered LDS image: one
//    barrier per k-block, the image of block k + 1 is written while other waves still multiply block k, its global loads were
//    issued a whole k-block earlier;
//  * the output tile goes back through LDS (each wave its own 16 rows) so that every row leaves as one contiguous run of float4.
// Rows beyond the tile's live rows and k beyond K need no masking on the A side: B is zero for k >= K, A reads are clamped to valid
// finite data, and the output rows / columns of the padding are never stored.
template <int MODE, int NT>
__global__ __launch_bounds__(256, 2) void rows_gemm3_kernel(GemmArgs a) {
    constexpr int NCOL = NT * 16;
    constexpr int BIMG = 3 * NCOL * 64;      // bytes of one B image
    constexpr int LDC = NCOL + 4;            // output staging row (floats): 4 (mod 8), the transposing writes of a half-wave hit 32 banks
    const int tile = blockIdx.x;
    __shared__ float red_loss[4];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * BIMG];
    __shared__ int s_slot[RW3], s_ent[RW3];
    if (tile >= a.n_tiles[0]) {
        if (MODE == GEMM_PROJECT && a.fuse_vec)     // every workgroup reports a loss partial
            finish_loss_sh<4>(a.fb, red_loss, 0.f, threadIdx.x & 63, threadIdx.x >> 6, reinterpret_cast<float *>(Bs));
        return;
    }
    const int r = a.tile_rel[tile];
    const int row0 = a.tile_row0[tile];
    const int rows = min(RW3, a.bucket_start[r + 1] - row0);
    const int K = MODE == GEMM_PROJECT ? a.De : a.Dr;
    const int ncols = MODE == GEMM_PROJECT ? a.Dr : a.De;            // multiple of 4, <= NCOL
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const float *M = a.mat + (long long)r * a.De * a.Dr;
    // B staging tasks.  dgrad (B[k = j][col = i] = M[i][j], k contiguous in memory): (column c, chunk q), up to four per thread.
    // Projection (B[k = i][col = j] = M[i][j], k strided): (column quad jq, chunk q) = 8 rows of 4 columns, transposed in registers.
    constexpr int NBD = (NCOL * 4 + 255) / 256;
    constexpr int JQ = NCOL / 4;
    float4 rb[8];
    const int pj = tid % JQ, pq = tid / JQ;         // projection B task (pq < 4: live)
#define KGE_LOADB3(k0_)                                                                                                       \
    {                                                                                                                         \
        if constexpr (MODE == GEMM_PROJECT) {                                                                                 \
            static_for<0, 8>([&](auto ec) {                                                                                   \
                constexpr int e = decltype(ec)::value;                                                                        \
                rb[e] = *reinterpret_cast<const float4 *>(M + (long long)min((k0_) + 8 * min(pq, 3) + e, K - 1) * a.Dr + min(4 * pj, ncols - 4)); \
            });                                                                                                               \
        } else {                                                                                                              \
            static_for<0, NBD>([&](auto uc) {                                                                                 \
                constexpr int u = decltype(uc)::value;                                                                        \
                const int idx = min(tid + 256 * u, NCOL * 4 - 1), c = idx >> 2, q = idx & 3;                                   \
                const float *src = M + (long long)min(c, ncols - 1) * a.Dr;                                                    \
                rb[2 * u] = *reinterpret_cast<const float4 *>(src + min((k0_) + 8 * q, K - 4));                                \
                rb[2 * u + 1] = *reinterpret_cast<const float4 *>(src + min((k0_) + 8 * q + 4, K - 4));                        \
            });                                                                                                               \
        }                                                                                                                     \
    }
#define KGE_PUTB3(col_, ch_, v_)                                                                                              \
    {                                                                                                                         \
        uint4 t1, t2, t3;                                                                                                     \
        split3_8(v_, t1, t2, t3);                                                                                             \
        *reinterpret_cast<uint4 *>(img + lds3_off(NCOL, 0, (col_), (ch_))) = t1;                                              \
        *reinterpret_cast<uint4 *>(img + lds3_off(NCOL, 1, (col_), (ch_))) = t2;                                              \
        *reinterpret_cast<uint4 *>(img + lds3_off(NCOL, 2, (col_), (ch_))) = t3;                                              \
    }
    auto store_b = [&](unsigned char *img, int k0) {
        if constexpr (MODE == GEMM_PROJECT) {
            if (pq < 4) {
                float4 m[8];
                static_for<0, 8>([&](auto ec) {
                    constexpr int e = decltype(ec)::value;
                    m[e] = keep_if(k0 + 8 * pq + e < K && 4 * pj < ncols, rb[e]);      // (columns beyond the width: zero, so that the padding columns of P are)
                });
                { const float v[8] = {m[0].x, m[1].x, m[2].x, m[3].x, m[4].x, m[5].x, m[6].x, m[7].x}; KGE_PUTB3(4 * pj, pq, v) }
                { const float v[8] = {m[0].y, m[1].y, m[2].y, m[3].y, m[4].y, m[5].y, m[6].y, m[7].y}; KGE_PUTB3(4 * pj + 1, pq, v) }
                { const float v[8] = {m[0].z, m[1].z, m[2].z, m[3].z, m[4].z, m[5].z, m[6].z, m[7].z}; KGE_PUTB3(4 * pj + 2, pq, v) }
                { const float v[8] = {m[0].w, m[1].w, m[2].w, m[3].w, m[4].w, m[5].w, m[6].w, m[7].w}; KGE_PUTB3(4 * pj + 3, pq, v) }
            }
        } else {
            static_for<0, NBD>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int idx = tid + 256 * u;
                if (idx < NCOL * 4) {
                    const int c = idx >> 2, q = idx & 3;
                    const float4 lo = keep_if(k0 + 8 * q < K, rb[2 * u]), hi = keep_if(k0 + 8 * q + 4 < K, rb[2 * u + 1]);
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    KGE_PUTB3(c, q, v)
                }
            });
        }
    };
    KGE_LOADB3(0)            // the matrix does not depend on the index chain below
    if (tid < RW3) {
        if (MODE == GEMM_PROJECT && a.sorted_groups) {
            // group layout: row -> (sub-tile, group in it, member u): u = 0 the head, 1 the tail, 1 + k the new entity of negative k
            const int rel_row = row0 + tid - a.bucket_start[r];
            const int w = rel_row & 15, q = w / a.U, u = w - q * a.U;
            const int gi = (rel_row >> 4) * a.gps + q;
            const int g0 = a.group_start[r];
            int sl = a.pad_slot, e = 0;
            if (tid < rows && q < a.gps && gi < a.group_start[r + 1] - g0) {
                const long long b = a.sorted_groups[g0 + gi];
                const int h = a.bh[b];
                if (u == 0) { sl = (int)(2 * b); e = h; }
                else if (u == 1) { sl = (int)(2 * b + 1); e = a.bt[b]; }
                else {
                    const long long j = b + (long long)(u - 1) * a.stride;
                    const int nh = a.bh[j];
                    const bool new_head = nh != h;
                    sl = (int)(2 * ((long long)(u - 1) * a.n_pos + b) + (new_head ? 0 : 1));
                    e = new_head ? nh : a.bt[j];
                }
                a.job_ent_w[sl] = e;
            }
            s_slot[tid] = sl;
            s_ent[tid] = e;
            if (tid < rows) a.sorted_slots_w[row0 + tid] = sl;
        } else {
            const int sl = a.sorted_slots[row0 + min(tid, rows - 1)];     // padding rows repeat the last live row (never stored)
            s_slot[tid] = sl;
            s_ent[tid] = a.job_ent[sl];
        }
    }
    __syncthreads();
    const float *arow[RT2];
#pragma unroll
    for (int s2 = 0; s2 < RT2; s2++) {
        const int row = (4 * s2 + wave) * 16 + r16;
        arow[s2] = MODE == GEMM_PROJECT ? a.ent + (long long)s_ent[row] * a.De : a.GP + (long long)s_slot[row] * a.Dr;
    }
    const bool two = rows > RM2;                                       // (workgroup-uniform) second sub-tiles exist
    const bool live = wave * 16 < rows;                                // this wave has rows at all
    float4 ra[RT2][2];
#define KGE_LOADA3(k0_)                                                                                                       \
    {                                                                                                                         \
        ra[0][0] = *reinterpret_cast<const float4 *>(arow[0] + min((k0_) + 8 * g, K - 4));                                     \
        ra[0][1] = *reinterpret_cast<const float4 *>(arow[0] + min((k0_) + 8 * g + 4, K - 4));                                 \
        if (two) {                                                                                                            \
            ra[1][0] = *reinterpret_cast<const float4 *>(arow[1] + min((k0_) + 8 * g, K - 4));                                 \
            ra[1][1] = *reinterpret_cast<const float4 *>(arow[1] + min((k0_) + 8 * g + 4, K - 4));                             \
        }                                                                                                                     \
    }
    KGE_LOADA3(0)
    f32x4 acc[RT2][NT];
#pragma unroll
    for (int s2 = 0; s2 < RT2; s2++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[s2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    store_b(Bs, 0);
    if (KB3 < K) KGE_LOADB3(KB3)
    __syncthreads();
    int kb = 0;
    for (int k0 = 0; k0 < K; k0 += KB3, kb ^= 1) {
        const bool more = k0 + KB3 < K;
        bf16x8 af[RT2][3];
#pragma unroll
        for (int s2 = 0; s2 < RT2; s2++) {
            if (s2 == 0 || two) {
                const float v[8] = {ra[s2][0].x, ra[s2][0].y, ra[s2][0].z, ra[s2][0].w, ra[s2][1].x, ra[s2][1].y, ra[s2][1].z, ra[s2][1].w};
                uint4 t1, t2, t3;
                split3_8(v, t1, t2, t3);
                af[s2][0] = __builtin_bit_cast(bf16x8, t1); af[s2][1] = __builtin_bit_cast(bf16x8, t2); af[s2][2] = __builtin_bit_cast(bf16x8, t3);
            } else {
                af[s2][0] = af[0][0]; af[s2][1] = af[0][1]; af[s2][2] = af[0][2];      // (13-tile instantiation: multiplied, never stored)
            }
        }
        if (more) KGE_LOADA3(k0 + KB3)                  // in flight during this block's matrix work
        if constexpr (NT <= 7) {
            if (two) gemm3_mfma_block<NT, RT2>(af, Bs + kb * BIMG, acc, lane);
            else if (live) gemm3_mfma_block<NT, 1>(af, Bs + kb * BIMG, acc, lane);
        } else {
            if (live) gemm3_mfma_block<NT, RT2>(af, Bs + kb * BIMG, acc, lane);
        }
        if (more) {
            store_b(Bs + (kb ^ 1) * BIMG, k0 + KB3);   // (that image was last read before the previous barrier)
            if (k0 + 2 * KB3 < K) KGE_LOADB3(k0 + 2 * KB3)
        }
        __syncthreads();
    }
#undef KGE_LOADA3
#undef KGE_LOADB3
#undef KGE_PUTB3
    const bool has_pads = a.sorted_groups != nullptr;           // group layout: rows of slot a.pad_slot carry nothing
    if (MODE == GEMM_DGRAD && a.rec_out && tid < rows) a.rec_dst[row0 + tid] = has_pads && s_slot[tid] == a.pad_slot ? -1 : s_ent[tid];
    if (MODE == GEMM_PROJECT && a.row_ent && tid < rows) a.row_ent[row0 + tid] = s_ent[tid];
    if (MODE == GEMM_DGRAD && !a.rec_out) {            // small steps: fp32 atomics straight from the accumulators
#pragma unroll
        for (int s2 = 0; s2 < RT2; s2++) {
            if ((4 * s2 + wave) * 16 >= rows) continue;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int j = t * 16 + r16;
                if (j < ncols) {
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int row = (4 * s2 + wave) * 16 + 4 * g + v;
                        if (row < rows && !(has_pads && s_slot[row] == a.pad_slot)) __builtin_amdgcn_global_atomic_fadd_f32(
                                (__attribute__((address_space(1))) float *)(a.g_ent + (long long)s_ent[row] * a.De + j), acc[s2][t][v]);
                    }
                }
            }
        }
        return;
    }
    // rows out through LDS (the B images are free: every wave is past the last barrier): wave-private 16 x LDC staging
    float *Cs = reinterpret_cast<float *>(Bs) + wave * 16 * LDC;
    const int qn = ncols >> 2;
    if (MODE == GEMM_PROJECT && a.fuse_vec) {
        // ---- the vector stage on the staged rows.  FOUR groups at a time: the 16 lanes of a DPP row are a team, lane l16 of it holds
        // columns l16, l16 + 16, ... of its group's rows (NT per row), every reduction is four DPP adds inside the row and serves four
        // groups.  The rows are turned into their gradients IN PLACE in the staging block, which then leaves as GP rows. ----
        const int Dr = a.Dr;
        const int team = lane >> 4, l16 = lane & 15;
        bool cv[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) cv[j] = l16 + 16 * j < Dr;
        // (the padding columns of the staged rows are exact zeros -- the matrix image's padding columns are -- so rows are read and
        // written whole: a mask per element compiled into an exec-mask branch around every LDS access)
        auto ldrow = [&](const float *p, float (&x)[NT]) {
#pragma unroll
            for (int j = 0; j < NT; j++) x[j] = p[l16 + 16 * j];
        };
        auto dotT = [&](const float (&x)[NT], const float (&y)[NT]) {
            float sdot = 0.f;
#pragma unroll
            for (int j = 0; j < NT; j++) sdot += x[j] * y[j];
            return team_sum<16>(sdot);
        };
        auto normalizeT = [&](float (&x)[NT], float &inv, bool &uc) {
            const float ss = dotT(x, x);
            uc = ss >= 1e-12f;
            inv = 1.0f / sqrtf(uc ? ss : 1e-12f);
#pragma unroll
            for (int j = 0; j < NT; j++) x[j] *= inv;
        };
        const float unit = a.fb.unit;
        // gx = inv (unit G - y <y, unit G>), written over gv
        auto normalize_bwdT = [&](const float (&y)[NT], float (&gv)[NT], float inv, bool uc) {
            float d = dotT(y, gv) * unit;
            if (!uc) d = 0.f;
#pragma unroll
            for (int j = 0; j < NT; j++) gv[j] = inv * (unit * gv[j] - d * y[j]);
        };
        float *s_grel = reinterpret_cast<float *>(Bs) + 4 * 16 * LDC;      // behind the four waves' staging blocks: the tile's relation sign sums
        for (int i = tid; i < NCOL; i += 256) s_grel[i] = 0.f;
        __syncthreads();
        float rn[NT], Sr[NT];
        float inv_r; bool uc_r;
#pragma unroll
        for (int j = 0; j < NT; j++) { rn[j] = cv[j] ? a.rel[(long long)r * Dr + l16 + 16 * j] : 0.f; Sr[j] = 0.f; }
        normalizeT(rn, inv_r, uc_r);
        float lsum = 0.f;
        const int n_groups_r = a.group_start[r + 1] - a.group_start[r];
        const int rel_row0 = row0 - a.bucket_start[r];
        const int U = a.U, gps = a.gps;
#pragma unroll
        for (int s2 = 0; s2 < RT2; s2++) {
            const int base = (4 * s2 + wave) * 16;
            if (base >= rows) continue;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int v = 0; v < 4; v++) Cs[(4 * g + v) * LDC + 16 * t + r16] = acc[s2][t][v];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int live_groups = min(gps, n_groups_r - ((rel_row0 + base) >> 4) * gps);      // >= 1: the sub-tile exists
            for (int q0 = 0; q0 < live_groups; q0 += 4) {
                const bool active = q0 + team < live_groups;
                const int q = active ? q0 + team : live_groups - 1;         // idle teams shadow the last group (nothing stored)
                float *rowp = Cs + (q * U) * LDC;
                float hn[NT], tn[NT];
                float inv_h, inv_t; bool uc_h, uc_t;
                ldrow(rowp, hn); ldrow(rowp + LDC, tn);
                normalizeT(hn, inv_h, uc_h); normalizeT(tn, inv_t, uc_t);
                float gh[NT], gt[NT], sp_keep[NT];     // dL/d(h^, t^) / unit: integer sign sums (r^'s is gh - gt - cnt sp); the positive's signs
                float pacc = 0.f;
#pragma unroll
                for (int j = 0; j < NT; j++) { const float ev = hn[j] + rn[j] - tn[j]; pacc += fabsf(ev); sp_keep[j] = sgn(ev); gh[j] = 0.f; gt[j] = 0.f; }
                const float pscore = team_sum<16>(pacc);
                int cnt = 0;
                for (int k = 0; k < U - 2; k++) {
                    const bool new_head = (s_slot[base + q * U + 2 + k] & 1) == 0;      // the corrupted side's slot: even = head side
                    float xn[NT], sg[NT];
                    float inv; bool uc;
                    ldrow(rowp + (2 + k) * LDC, xn);
                    normalizeT(xn, inv, uc);
                    float nacc = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        const float ev = new_head ? xn[j] + rn[j] - tn[j] : hn[j] + rn[j] - xn[j];
                        nacc += fabsf(ev); sg[j] = sgn(ev);
                    }
                    const float hv = pscore - team_sum<16>(nacc) + a.fb.margin;
                    const bool on = hv >= 0.f;                                          // tf.maximum routes a tie to the hinge
                    if (on && active && l16 == 0) lsum += hv;
                    cnt += on ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        const float sj = on ? sg[j] : 0.f;
                        if (new_head) gt[j] += sj; else gh[j] -= sj;
                        sg[j] = new_head ? -sj : sj;
                    }
                    normalize_bwdT(xn, sg, inv, uc);                                    // (an inactive hinge: zeros in, zeros out)
                    if (active) {
#pragma unroll
                        for (int j = 0; j < NT; j++) rowp[(2 + k) * LDC + l16 + 16 * j] = sg[j];
                    }
                }
                const float fc = (float)cnt;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    gh[j] += fc * sp_keep[j]; gt[j] -= fc * sp_keep[j];
                    if (active) Sr[j] += gh[j] - gt[j] - fc * sp_keep[j];      // = -(all the negatives' signs) + cnt sp
                }
                normalize_bwdT(hn, gh, inv_h, uc_h);
                normalize_bwdT(tn, gt, inv_t, uc_t);
                if (active) {
#pragma unroll
                    for (int j = 0; j < NT; j++) { rowp[l16 + 16 * j] = gh[j]; rowp[LDC + l16 + 16 * j] = gt[j]; }
                }
            }
            if (lane < 16 && a.row_ent) a.row_ent[row0 + base + lane] = s_ent[base + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int idx = lane; idx < 16 * qn; idx += 64) {             // the gradients out as GP rows (pad rows keep the pad slot's zeros)
                const int rl = idx / qn, q4 = idx - rl * qn;
                const int sl = s_slot[base + rl];
                if (base + rl < rows && sl != a.pad_slot)
                    *reinterpret_cast<float4 *>(a.GP_w + (long long)sl * Dr + 4 * q4) = *reinterpret_cast<const float4 *>(&Cs[rl * LDC + 4 * q4]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int j = 0; j < NT; j++) if (cv[j] && Sr[j] != 0.f) atomicAdd(&s_grel[l16 + 16 * j], Sr[j]);
        const float wsum = team_sum<64>(l16 == 0 ? lsum : 0.f);        // the wave's hinge sum (the four teams' lane 0)
        __syncthreads();                                               // s_grel complete; every wave is done with its staging rows
        if (wave == 0) {   // g_rel[r] += inv_r (unit S - unit <r^, S> r^): the normalise-backward is linear in S, so once per tile
            float Sg[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) Sg[j] = cv[j] ? s_grel[l16 + 16 * j] : 0.f;
            normalize_bwdT(rn, Sg, inv_r, uc_r);
            if (team == 0) {
#pragma unroll
                for (int j = 0; j < NT; j++)
                    if (cv[j] && Sg[j] != 0.f) __builtin_amdgcn_global_atomic_fadd_f32(
                            (__attribute__((address_space(1))) float *)(a.g_rel + (long long)r * Dr + l16 + 16 * j), Sg[j]);
            }
        }
        finish_loss_sh<4>(a.fb, red_loss, wsum, lane, wave, reinterpret_cast<float *>(Bs));
        return;
    }
#pragma unroll
    for (int s2 = 0; s2 < RT2; s2++) {
        const int base = (4 * s2 + wave) * 16;
        if (base >= rows) continue;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) Cs[(4 * g + v) * LDC + 16 * t + r16] = acc[s2][t][v];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int idx = lane; idx < 16 * qn; idx += 64) {
            const int rl = idx / qn, q = idx - rl * qn;
            if (base + rl < rows) {
                const float4 v = *reinterpret_cast<const float4 *>(&Cs[rl * LDC + 4 * q]);
                float *dst = MODE == GEMM_PROJECT ? a.P + (long long)s_slot[base + rl] * a.Dr : a.rec_out + (long long)(row0 + base + rl) * a.De;
                *reinterpret_cast<float4 *>(dst + 4 * q) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// wgrad v2 (dims 196..208, multiples of 4: the full 13 x 13 output tile grid): g_M[r][i][j] += sum over rows of
// ent[e_row][i] * GP[slot_row][j].  A workgroup owns SPAN2 consecutive 128-row tiles of the relation-sorted job list
// (512 rows between flushes) and one HALF of the output row tiles (blockIdx.y: tiles 0..6 / 7..12) with every column
// tile: 2 x 13 accumulator tiles per wave (104 registers), which leaves room for two workgroups per CU -- with all
// 169 tiles in one workgroup (172 + staging registers, one workgroup per CU) 63 % of the wave cycles were parked on
// s_waitcnt.  Wave w takes local row tiles w and w+4 (the 8th does not exist: computed on a clamped tile, dropped at the
// flush, which keeps the MFMA block branch-free).  Rows are staged 16 at a time, the next 16 in flight during the MFMA
// loop.  A relation whose whole bucket lies inside the span has one owner: its matrix gradient is stored, not added with
// atomics (the accumulator is zero).
constexpr int SPAN2 = 4;
constexpr int WK2 = 32;                 // rows per staged chunk (8 k-steps: half the barriers of 16-row chunks; 41 KB of LDS, two workgroups per CU)
constexpr int WH2 = 7;                  // output row tiles per half
constexpr int LDX2 = WH2 * 16;          // [row k][i] stride of the staged X half: 112 = 16 mod 32 banks

__global__ __launch_bounds__(256, 2) void wgrad2_kernel(GemmArgs a, float *__restrict__ g_mat, int span) {
    // span: 128-row tiles per workgroup (SPAN2 = 512 rows between flushes for well-filled buckets; 1 at the reference's batch
    // sizes, where a tile is one relation's whole bucket and the chip needs every tile as a workgroup of its own)
    const int n_tiles = a.n_tiles[0];
    const int t0 = blockIdx.x * span;
    if (t0 >= n_tiles) return;
    const int t1 = min(t0 + span, n_tiles);
    __shared__ __attribute__((aligned(16))) float Xs[WK2 * LDX2];    // [row k][i - i0]
    __shared__ __attribute__((aligned(16))) float Gs[WK2 * LDB2];    // [row k][j], stride 208
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = blockIdx.y;
    const int i0 = half * LDX2;                                       // first output row (= X column) of this half
    const int n_it = half == 0 ? WH2 : NT2 - WH2;                     // row tiles that exist in this half (7 / 6)
    f32x4 acc[2][NT2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int t2 = 0; t2 < NT2; t2++) acc[s2][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
    // rows [span_lo, span_hi) of the sorted job list belong to this workgroup
    const int span_lo = a.tile_row0[t0];
    const int span_hi = a.tile_row0[t1 - 1] + min(RW2, a.bucket_start[a.tile_rel[t1 - 1] + 1] - a.tile_row0[t1 - 1]);
    constexpr int QX = LDX2 / 4, QG = LDB2 / 4;       // float4 per staged row: 28 of X, 52 of GP
    constexpr int NX = (WK2 * QX + 255) / 256;        // float4 loads per thread: 2 of X
    constexpr int NG = (WK2 * QG + 255) / 256;        //                          4 of GP
    float4 rx[NX], rg[NG];
    const int qx = (min(a.De, i0 + LDX2) - i0) / 4, qg = a.Dr / 4;   // valid float4 per row
#define KGE_WLOAD(row_first_, crow_)                                                                          \
    {                                                                                                         \
        static_for<0, NX>([&](auto uc) {                                                                      \
            constexpr int u = decltype(uc)::value;                                                            \
            const int idx = min(tid + 256 * u, WK2 * QX - 1);                                                 \
            const int kk = idx / QX, q = idx - kk * QX;                                                       \
            const int e = a.job_ent[a.sorted_slots[(row_first_) + min(kk, (crow_) - 1)]];                     \
            rx[u] = *reinterpret_cast<const float4 *>(a.ent + (long long)e * a.De + i0 + 4 * min(q, qx - 1)); \
        });                                                                                                   \
        static_for<0, NG>([&](auto uc) {                                                                      \
            constexpr int u = decltype(uc)::value;                                                            \
            const int idx = min(tid + 256 * u, WK2 * QG - 1);                                                 \
            const int kk = idx / QG, q = idx - kk * QG;                                                       \
            const int sl = a.sorted_slots[(row_first_) + min(kk, (crow_) - 1)];                               \
            rg[u] = *reinterpret_cast<const float4 *>(a.GP + (long long)sl * a.Dr + 4 * min(q, qg - 1));      \
        });                                                                                                   \
    }
#define KGE_WPUT(i_, j_, v_)                                                                                  \
    if ((i_) < a.De && (j_) < a.Dr && (v_) != 0.f) {                                                          \
        if (sole) G[(long long)(i_) * a.Dr + (j_)] = (v_);                                                    \
        else __builtin_amdgcn_global_atomic_fadd_f32(                                                         \
                (__attribute__((address_space(1))) float *)(G + (long long)(i_) * a.Dr + (j_)), (v_));        \
    }
#define KGE_WFLUSH(rel_)                                                                                      \
    {                                                                                                         \
        float *G = g_mat + (long long)(rel_) * a.De * a.Dr;                                                   \
        const bool sole = a.bucket_start[rel_] >= span_lo && a.bucket_start[(rel_) + 1] <= span_hi;           \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; s2++) {                                                    \
            _Pragma("unroll") for (int t2 = 0; t2 < NT2; t2++) {                                              \
                const int j = t2 * 16 + (lane & 15);                                                          \
                if (wave + 4 * s2 < n_it) {                                                                   \
                    _Pragma("unroll") for (int v = 0; v < 4; v++) {                                           \
                        const int i = i0 + (wave + 4 * s2) * 16 + 4 * (lane >> 4) + v;                        \
                        KGE_WPUT(i, j, acc[s2][t2][v])                                                        \
                    }                                                                                         \
                }                                                                                             \
                acc[s2][t2] = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
            }                                                                                                 \
        }                                                                                                     \
    }
    int t = t0, c0 = 0;
    int rel = a.tile_rel[t];
    int r_cur = rel;
    int rows_t = min(RW2, a.bucket_start[rel + 1] - a.tile_row0[t]);
    int row_first = a.tile_row0[t];
    int crow = min(WK2, rows_t);
    KGE_WLOAD(row_first, crow)
    bool first = true;
    while (true) {
        if (rel != r_cur) {
            KGE_WFLUSH(r_cur)
            r_cur = rel;
        }
        if (!first) __syncthreads();
        first = false;
        {
            static_for<0, NX>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int idx = tid + 256 * u;
                if (idx < WK2 * QX) {
                    const int kk = idx / QX, q = idx - kk * QX;
                    *reinterpret_cast<float4 *>(&Xs[kk * LDX2 + 4 * q]) = keep_if(kk < crow && q < qx, rx[u]);
                }
            });
            static_for<0, NG>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int idx = tid + 256 * u;
                if (idx < WK2 * QG) {
                    const int kk = idx / QG, q = idx - kk * QG;
                    *reinterpret_cast<float4 *>(&Gs[kk * LDB2 + 4 * q]) = keep_if(kk < crow && q < qg, rg[u]);
                }
            });
        }
        __syncthreads();
        // the next chunk: same tile, or the first chunk of the next tile of the span
        int nt_ = t, nc = c0 + WK2;
        if (nc >= rows_t) { nt_ = t + 1; nc = 0; }
        const bool more = nt_ < t1;
        int n_first = 0, n_crow = 1, n_rel = rel, n_rows = rows_t;
        if (more) {
            n_rel = a.tile_rel[nt_];
            const int nrow0 = a.tile_row0[nt_];
            n_rows = min(RW2, a.bucket_start[n_rel + 1] - nrow0);
            n_first = nrow0 + nc;
            n_crow = min(WK2, n_rows - nc);
            KGE_WLOAD(n_first, n_crow)
        }
        {   // the chunk's WK2 / 4 k-steps, operand reads of step s+1 in front of the MFMAs of step s (see gemm2_mfma_block)
            constexpr int STEPS = WK2 / 4;
            const float *xa[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) xa[s2] = Xs + (lane >> 4) * LDX2 + min(wave + 4 * s2, WH2 - 1) * 16 + (lane & 15);   // A[i][k] = X[row k][i]
            const float *gb = Gs + (lane >> 4) * LDB2 + (lane & 15);                                                            // B[k][j] = GP[row k][j]
            float av[2][2], bv[2][NT2];
            auto fetch = [&](int buf, int st) {
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) av[buf][s2] = xa[s2][4 * st * LDX2];
#pragma unroll
                for (int t2 = 0; t2 < NT2; t2++) bv[buf][t2] = gb[4 * st * LDB2 + t2 * 16];
            };
            fetch(0, 0);
            static_for<0, STEPS>([&](auto sc) {
                constexpr int st = decltype(sc)::value;
                constexpr int cur = st & 1;
                if constexpr (st + 1 < STEPS) fetch(cur ^ 1, st + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t2 = 0; t2 < NT2; t2++)
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) acc[s2][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][s2], bv[cur][t2], acc[s2][t2], 0, 0, 0);
            });
        }
        if (!more) break;
        t = nt_; c0 = nc; row_first = n_first; crow = n_crow; rel = n_rel; rows_t = n_rows;
    }
    KGE_WFLUSH(r_cur)
#undef KGE_WLOAD
#undef KGE_WFLUSH
#undef KGE_WPUT
}

// ---------------------------------------------------------------------------------------------
// wgrad v3: g_M[r][i][j] += sum over rows of ent[e_row][i] * GP[slot_row][j] with the products on the bf16 matrix pipe (the
// three-term split of rows_gemm3_kernel).  The reduction index is the ROW here, so both MFMA operands are transposes of what
// memory holds (A[i][k = row] = X[row][i], B[k = row][j] = GP[row][j]).  The staged rows go into LDS row-major -- a thread splits one
// float4 of one row and writes three 8-byte pieces, ten such tasks per thread and 32-row chunk, no transposition in registers -- and
// the fragments come out with ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns and each lane receives one COLUMN's
// four rows, two such reads are a lane's 8 consecutive k of a 16x16x32 operand.
// Image: [term][row k (32)][columns], rows of 256 B (the 112-column X half) / 512 B (GP, 208 columns), the 32-byte segment (16
// columns) of row k rotated by (k & 3) + 4 (k >> 3 & 1): the eight 4 x 16 blocks one transposed read of a 32-lane half touches
// (rows k .. k + 3 of lane groups 8 rows apart) lie in eight different 32-byte bank groups.
// Workgroup = the span / half decomposition of wgrad2_kernel.  The ids of a chunk's rows (sorted slot, entity) travel one chunk
// ahead of its data through a small LDS table, so the data loads issued beside a chunk's matrix work depend on nothing in flight.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int WK3 = 32;
// image rows: the X half 10 segments of 32 B (7 of columns + 3 of rotation: 320 B), GP 16 (13 + 3: 512 B).  The segment of row k is
// rotated by k & 3 only and never wraps, so every address is (a per-lane base) + (a compile-time offset): the kernel is bound by
// instruction issue (20 % of its wave cycles active, 16 % matrix pipe), and with a wrapping rotation every one of a chunk's 90
// fragment reads and 33 staging writes recomputed its segment -- some 1 500 of the ~2 000 vector instructions of a chunk.  (Rows
// 8 apart now share their bank group: the transposed reads of a 32-lane half are 2-way conflicted; the LDS array is not the limit.)
constexpr int XROWB = 320, GROWB = 512;
constexpr int XTERM = WK3 * XROWB, GTERM = WK3 * GROWB;     // bytes of one term plane

// byte offset of columns c .. c + 3 (c % 4 == 0) of row k of term plane t:  t * term_bytes + k * rowb + 32 (c / 16 + (k & 3)) + 2 (c % 16)
// a lane's base for the fragments of k-rows 8G .. 8G+7: lane 4q + p of a 16-lane group addresses row 8G + q (+ 4 for the second read),
// columns 4p .. of the tile; tile t of term `term`: + term * term_bytes + 32 t, the second read + 4 * rowb -- all compile-time
__device__ __forceinline__ int frag3_base(int rowb, int lane) {
    const int r16 = lane & 15, q = r16 >> 2, pp = r16 & 3, G = lane >> 4;
    return (8 * G + q) * rowb + (q << 5) + 8 * pp;
}
template <int ROWB, int TERMB>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *lane_base, int term, int tile) {
    const unsigned char *p = lane_base + term * TERMB + tile * 32;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * ROWB));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
}

__global__ __launch_bounds__(256, 2) void wgrad3_kernel(GemmArgs a, float *__restrict__ g_mat, int span) {
    const int n_tiles = a.n_tiles[0];
    const int t0 = blockIdx.x * span;
    if (t0 >= n_tiles) return;
    const int t1 = min(t0 + span, n_tiles);
    __shared__ __attribute__((aligned(16))) unsigned char Xs[3 * XTERM];
    __shared__ __attribute__((aligned(16))) unsigned char Gs[3 * GTERM];
    __shared__ int s_ids[2][WK3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = blockIdx.y;
    const int i0 = half * LDX2;
    const int n_it = half == 0 ? WH2 : NT2 - WH2;
    f32x4 acc[2][NT2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int t2 = 0; t2 < NT2; t2++) acc[s2][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int span_lo = a.tile_row0[t0];
    const int span_hi = a.tile_row0[t1 - 1] + min(RW2, a.bucket_start[a.tile_rel[t1 - 1] + 1] - a.tile_row0[t1 - 1]);
    const int qx = (min(a.De, i0 + LDX2) - i0) / 4, qg = a.Dr / 4;   // valid float4 per row
    // staging tasks: a row of a chunk is 80 float4 -- 28 of the X half, 52 of GP; thread t < 240 owns column quad t % 80 of the rows
    // t / 80 + 3 u (u = 0 .. 10): its table (X or GP), its column offset and its place in an image row never change
    constexpr int QROW = LDX2 / 4 + LDB2 / 4;  // 80
    constexpr int NTASK = (WK3 + 2) / 3;       // 11 row slots per thread (the last one: rows 30, 31)
    float4 rs[NTASK];
    const int cq80 = tid % QROW, k0 = tid / QROW;              // k0 = 3: no tasks
    const bool isx = cq80 < LDX2 / 4;
    const int cq = isx ? cq80 : cq80 - LDX2 / 4;
    const float *tab = isx ? a.ent + i0 + 4 * min(cq, qx - 1) : a.GP + 4 * min(cq, qg - 1);
    const long long ld = isx ? a.De : a.Dr;
    // this thread's column part of every staging address; the row part of task u (row k0 + 3 u) is k * rowb + 32 (k & 3)
    unsigned char *img = (isx ? Xs : Gs) + (((4 * cq) >> 4) << 5) + ((4 * cq) & 15) * 2;
    const int rowb = isx ? XROWB : GROWB, termb = isx ? XTERM : GTERM;
    // chunk descriptors: the one being multiplied (d0), the one whose data is in flight (d1), the one whose ids are in flight (d2)
    struct Chunk { int row_first, crow, rel, t, c; bool valid; };
    auto chunk_at = [&](int t, int c) {
        Chunk d; d.t = t; d.c = c; d.valid = t < t1;
        d.rel = 0; d.row_first = 0; d.crow = 1;
        if (d.valid) {
            d.rel = a.tile_rel[t];
            const int row0 = a.tile_row0[t];
            const int rows_t = min(RW2, a.bucket_start[d.rel + 1] - row0);
            d.row_first = row0 + c;
            d.crow = min(WK3, rows_t - c);
        }
        return d;
    };
    auto chunk_after = [&](const Chunk &d) {
        if (!d.valid) return d;
        const int rows_t = min(RW2, a.bucket_start[d.rel + 1] - a.tile_row0[d.t]);
        int t = d.t, c = d.c + WK3;
        if (c >= rows_t) { t++; c = 0; }
        return chunk_at(t, c);
    };
    int my_sl = 0, my_e = 0;
    auto load_ids = [&](const Chunk &d) {           // two independent loads (row_ent: written by the projection kernel)
        if (tid < WK3 && d.valid) {
            const int p = d.row_first + min(tid, d.crow - 1);
            my_sl = a.sorted_slots[p];
            my_e = a.row_ent[p];
        }
    };
    auto put_ids = [&]() { if (tid < WK3) { s_ids[0][tid] = my_sl; s_ids[1][tid] = my_e; } };
    auto load_data = [&](const Chunk &d) {
        if (k0 < 3) {
            static_for<0, NTASK>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int k = min(k0 + 3 * u, d.crow - 1);
                const int id = s_ids[isx ? 1 : 0][k];
                rs[u] = *reinterpret_cast<const float4 *>(tab + (long long)id * ld);
            });
        }
    };
    auto store_data = [&](const Chunk &d) {
        if (k0 < 3) {
            static_for<0, NTASK>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const int k = k0 + 3 * u;
                if (u < NTASK - 1 || k < WK3) {
                    // columns beyond the table's width hold clamped duplicates: they only reach output rows / columns that are never
                    // stored.  Rows beyond the chunk must be zero (a short last chunk: the rare, wave-uniform case).
                    const float4 v = d.crow == WK3 ? rs[u] : keep_if(k < d.crow, rs[u]);
                    uint2 p1, p2, p3;
                    split3_pair(v.x, v.y, p1.x, p2.x, p3.x);
                    split3_pair(v.z, v.w, p1.y, p2.y, p3.y);
                    unsigned char *dst = img + k * rowb + ((k & 3) << 5);
                    *reinterpret_cast<uint2 *>(dst) = p1;
                    *reinterpret_cast<uint2 *>(dst + termb) = p2;
                    *reinterpret_cast<uint2 *>(dst + 2 * termb) = p3;
                }
            });
        }
    };
#define KGE_W3PUT(i_, j_, v_)                                                                                 \
    if ((i_) < a.De && (j_) < a.Dr && (v_) != 0.f) {                                                          \
        if (sole) G[(long long)(i_) * a.Dr + (j_)] = (v_);                                                    \
        else __builtin_amdgcn_global_atomic_fadd_f32(                                                         \
                (__attribute__((address_space(1))) float *)(G + (long long)(i_) * a.Dr + (j_)), (v_));        \
    }
#define KGE_W3FLUSH(rel_)                                                                                     \
    {                                                                                                         \
        float *G = g_mat + (long long)(rel_) * a.De * a.Dr;                                                   \
        const bool sole = a.bucket_start[rel_] >= span_lo && a.bucket_start[(rel_) + 1] <= span_hi;           \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; s2++) {                                                    \
            _Pragma("unroll") for (int t2 = 0; t2 < NT2; t2++) {                                              \
                const int j = t2 * 16 + (lane & 15);                                                          \
                if (wave + 4 * s2 < n_it) {                                                                   \
                    _Pragma("unroll") for (int v = 0; v < 4; v++) {                                           \
                        const int i = i0 + (wave + 4 * s2) * 16 + 4 * (lane >> 4) + v;                        \
                        KGE_W3PUT(i, j, acc[s2][t2][v])                                                       \
                    }                                                                                         \
                }                                                                                             \
                acc[s2][t2] = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
            }                                                                                                 \
        }                                                                                                     \
    }
    Chunk d0 = chunk_at(t0, 0), d1 = chunk_after(d0), d2 = chunk_after(d1);
    int r_cur = d0.rel;
    load_ids(d0);
    put_ids();
    __syncthreads();
    load_data(d0);
    load_ids(d1);
    bool first = true;
    const unsigned char *xbase = Xs + frag3_base(XROWB, lane), *gbase = Gs + frag3_base(GROWB, lane);
    while (true) {
        if (!first) __syncthreads();          // the previous chunk's fragment reads (and its readers of s_ids) are done
        first = false;
        store_data(d0);
        put_ids();                            // the ids of d1's rows (loaded a chunk ago)
        if (d0.rel != r_cur) {                // (here, between the staging stores and the next loads, no staged row is live in registers)
            KGE_W3FLUSH(r_cur)
            r_cur = d0.rel;
        }
        __syncthreads();
        if (d1.valid) load_data(d1);          // in flight during the matrix work
        load_ids(d2);
        {
            bf16x8 af[2][3];
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
                for (int t = 0; t < 3; t++) af[s2][t] = tr_frag<XROWB, XTERM>(xbase, t, min(wave + 4 * s2, WH2 - 1));
            bf16x8 bf[2][3];
            auto fetch = [&](int buf, int jt) {
#pragma unroll
                for (int t = 0; t < 3; t++) bf[buf][t] = tr_frag<GROWB, GTERM>(gbase, t, jt);
            };
            fetch(0, 0);
            static_for<0, NT2>([&](auto tc) {
                constexpr int jt = decltype(tc)::value;
                constexpr int cur = jt & 1;
                if constexpr (jt + 1 < NT2) fetch(cur ^ 1, jt + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    f32x4 c = acc[s2][jt];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][0], bf[cur][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][2], bf[cur][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][1], bf[cur][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][0], bf[cur][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][1], bf[cur][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s2][0], bf[cur][0], c, 0, 0, 0);
                    acc[s2][jt] = c;
                }
            });
        }
        if (!d1.valid) break;
        d0 = d1; d1 = d2; d2 = chunk_after(d2);
    }
    KGE_W3FLUSH(r_cur)
#undef KGE_W3FLUSH
#undef KGE_W3PUT
}

int bits_for(int64_t v) { int b = 1; while ((int64_t(1) << b) <= v) b++; return b; }

}  // namespace

int launch_forward_backward_transr(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h,
                                   const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                                   int64_t denom, float *const grads[4], float *d_loss, hipStream_t stream, bool sampler_shaped) {
    const int De = m.ent_dim, Dr = m.rel_dim;
    const int64_t R = m.rel_total;
    if (Dr > 1024) return fail(KGE_ERR_UNSUPPORTED, "TransR rel_dim > 1024 is not supported");
    if (n_pos == 0) return hip_check(hipMemsetAsync(d_loss, 0, sizeof(float), stream), "zero loss");
    const int64_t slots = 2 * n_pos * (1 + n_neg);
    if (slots >= (int64_t(1) << 31)) return fail(KGE_ERR_UNSUPPORTED, "TransR batch too large");
    int rc = ensure_work(slots, Dr, R);
    if (rc) return rc;
    int blocks = (int)((slots + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    // v2 kernels (16x16x4 MFMA, 128-row tiles): dims multiples of 4 up to 208, one workgroup covers all columns
    const bool v2 = De % 4 == 0 && Dr % 4 == 0 && De >= 4 && Dr >= 4 && De <= LDB2 && Dr <= LDB2 && engine().transr_v1 != 1;
    // ---- group layout (GemmArgs): device-sampled batches with the positive's matrix for every negative, 2 + n <= 16 rows per group ----
    {
        const int U = 2 + (int)n_neg;
        const bool v3g = v2 && engine().transr_bf16x3 && engine().transr_v1 == 0 && engine().transr_groups;
        // well-filled buckets only (the same bound as wgrad's spans): at the reference's own batch (~46 rows per relation) the longer
        // epilogue of the one tile a relation has costs more than the vector-stage launch it replaces (192 vs 177 us at B = 2 721);
        // transr_groups = 2 takes the layout at any size (tests)
        const bool big = n_pos * U >= 64 * R || engine().transr_groups == 2;
        if (v3g && big && sampler_shaped && m.negative_rel == 0 && U <= 16 && (R + 1) * kRelSub <= kRelBins && !engine().counts_force_sort) {
            const int gps = 16 / U;
            const int64_t rows_max = 16 * ((n_pos + gps - 1) / gps + R + 1);       // sum over relations of 16 ceil(groups / gps)
            if ((rc = ensure_work(slots + 1 > rows_max ? slots + 1 : rows_max, Dr, R))) return rc;
            if (R + 2 > g_w.cap_rel_rows) { if ((rc = grow(g_w.bucket_rows, (size_t)R + 2, "transr bucket_rows"))) return rc; g_w.cap_rel_rows = R + 2; }
            if (g_w.pad_ready != slots) {     // the pad slot: a zero GP row, entity 0 (never written by the vector stage)
                if ((rc = hip_check(hipMemsetAsync(g_w.GP + (size_t)slots * Dr, 0, sizeof(float) * (size_t)Dr, stream), "zero pad GP row"))) return rc;
                if ((rc = hip_check(hipMemsetAsync(g_w.job_ent + slots, 0, sizeof(int32_t), stream), "pad entity"))) return rc;
                g_w.pad_ready = slots;
            }
            float *drec = nullptr;
            int32_t *ddst = nullptr;
            const bool records = engine().transr_dgrad_records && slots >= engine().transr_dgrad_records_min;
            if (records && (rc = float_records_workspace(rows_max, De, drec, ddst))) return rc;
            const int64_t span_n = n_pos > rows_max ? n_pos : (records ? rows_max : n_pos);
            int gb = (int)((span_n + 255) / 256);
            if (gb > 4096) gb = 4096;
            hipLaunchKernelGGL(group_keys_kernel, dim3(gb), dim3(256), 0, stream, d_r, (long long)n_pos, g_w.keys, g_w.vals, ddst,
                               (long long)(records ? rows_max : 0));
            if (!g_w.rel_hist) {
                if ((rc = grow(g_w.rel_hist, 4 * (size_t)kRelBins, "transr relation histogram"))) return rc;
                if ((rc = hip_check(hipMemset(g_w.rel_hist, 0, sizeof(int32_t) * 4 * kRelBins), "zero relation histogram"))) return rc;
            }
            int32_t *pair = g_w.rel_hist + (g_w.rel_parity ? 2 * kRelBins : 0), *other = g_w.rel_hist + (g_w.rel_parity ? 0 : 2 * kRelBins);
            g_w.rel_parity ^= 1;
            const unsigned tiles = (unsigned)((n_pos + kRelTile - 1) / kRelTile);
            hipLaunchKernelGGL(rel_count_kernel, dim3(tiles), dim3(256), 0, stream, g_w.keys, (int)n_pos, ((int)R + 1) * kRelSub, pair, other);
            SamplerArgs ride = {};
            unsigned n_ride = 0;
            if (upload_jump_table() == KGE_OK && !take_attached_sampler(ride, n_ride)) n_ride = 0;
            hipLaunchKernelGGL(rel_scatter_kernel, dim3(tiles + n_ride), dim3(256), 0, stream, g_w.keys, g_w.vals, (int)n_pos, (int)R, pair,
                               pair + kRelBins, g_w.keys2, g_w.bucket_start, g_w.tile_rel, g_w.tile_row0, g_w.n_tiles, 7, ride,
                               (int)tiles, (int)n_ride, gps, g_w.bucket_rows);
            GemmArgs ga = {};
            ga.ent = tables[0]; ga.mat = tables[2]; ga.GP = g_w.GP; ga.P = g_w.P; ga.g_ent = grads[0];
            ga.sorted_slots = g_w.vals2; ga.job_ent = g_w.job_ent; ga.bucket_start = g_w.bucket_rows;
            ga.tile_rel = g_w.tile_rel; ga.tile_row0 = g_w.tile_row0; ga.n_tiles = g_w.n_tiles;
            ga.De = De; ga.Dr = Dr;
            ga.rec_out = drec; ga.rec_dst = ddst; ga.row_ent = g_w.row_ent;
            ga.sorted_groups = g_w.keys2; ga.group_start = g_w.bucket_start; ga.bh = d_h; ga.bt = d_t;
            ga.sorted_slots_w = g_w.vals2; ga.job_ent_w = g_w.job_ent; ga.n_pos = n_pos; ga.stride = stride;
            ga.U = U; ga.gps = gps; ga.pad_slot = (int)slots;
            const unsigned max_tiles = (unsigned)(rows_max / RW3 + R + 1);
            // the vector stage in the projection's epilogue: widths the one-float4-per-lane layout covers, a grid the loss hand-off can count
            const bool fuse = engine().transr_fuse_vec && Dr <= 256 && max_tiles <= (unsigned)kMaxLossBlocks;
            if (fuse) {
                if ((rc = ensure_loss_buffers())) return rc;
                guard_loss_stream(stream);
                ga.fuse_vec = 1; ga.rel = tables[1]; ga.g_rel = grads[1]; ga.GP_w = g_w.GP;
                ga.fb.loss_partials = engine().dev.loss_partials; ga.fb.loss_out = d_loss; ga.fb.loss_ticket = engine().dev.loss_ticket;
                ga.fb.unit = 1.0f / (float)denom; ga.fb.margin = m.margin;
            }
            if (Dr <= 112) hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_PROJECT, 7>), dim3(max_tiles), dim3(256), 0, stream, ga);
            else hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_PROJECT, 13>), dim3(max_tiles), dim3(256), 0, stream, ga);
            if (!fuse) {
                const bool lean = transr_lean_vector_stage(Dr);
                if (!lean && (rc = hip_check(hipMemsetAsync(g_w.GP, 0, sizeof(float) * (size_t)slots * Dr, stream), "zero GP"))) return rc;
                rc = launch_transr_vector_stage(tables[1], grads[1], g_w.P, g_w.GP, d_h, d_t, d_r, n_pos, n_neg, stride, denom, Dr, m.margin,
                                                m.negative_rel, d_loss, stream, lean, sampler_shaped, R);
                if (rc) return rc;
            }
            if (De <= 112) hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_DGRAD, 7>), dim3(max_tiles), dim3(256), 0, stream, ga);
            else hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_DGRAD, 13>), dim3(max_tiles), dim3(256), 0, stream, ga);
            if (records) {
                FloatRowSpace rs = {};
                rs.g_ent = grads[0]; rs.E = m.ent_total; rs.R = 0; rs.hub_base = m.ent_total; rs.hub_rows = 1; rs.rows = m.ent_total;
                if ((rc = float_records_reduce(rows_max, De, rs, stream))) return rc;
            }
            if (De > 192 && Dr > 192) {
                const int span = rows_max >= 256 * R ? SPAN2 : 1;
                const dim3 wg2((max_tiles + span - 1) / span, 2);
                hipLaunchKernelGGL(wgrad3_kernel, wg2, dim3(256), 0, stream, ga, grads[2], span);
            } else {
                const int tiles_i1 = (De + 31) / 32;
                const int wgt = WG_TILES * 32 / RW2;
                const unsigned groups1 = (max_tiles + wgt - 1) / wgt;
                hipLaunchKernelGGL(wgrad_kernel, dim3(groups1 * tiles_i1, (Dr + TN - 1) / TN), dim3(256), 0, stream, ga, grads[2], tiles_i1, RW2, wgt);
            }
            return hip_check(hipGetLastError(), "transr launch (group layout)");
        }
    }
    // dgrad's entity-gradient rows as float records + segmented sum instead of atomics: from 32 768 slots on (below that the
    // reduction's four launches cost more than the atomics do), entity row spaces the record sort handles
    float *drec = nullptr;
    int32_t *ddst = nullptr;
    const bool dgrad_records = v2 && engine().transr_dgrad_records && slots >= engine().transr_dgrad_records_min;
    if (dgrad_records && (rc = float_records_workspace(slots, De, drec, ddst))) return rc;
    hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, stream, d_h, d_t, d_r, (long long)n_pos, (long long)n_neg,
                       (long long)stride, (int)m.negative_rel, (int)R, g_w.keys, g_w.vals, g_w.job_ent, ddst);
    if ((R + 1) * kRelSub <= kRelBins && !engine().counts_force_sort) {
        // two-launch counting sort by relation; bucket starts and the tile map come with it
        if (!g_w.rel_hist) {
            if ((rc = grow(g_w.rel_hist, 4 * (size_t)kRelBins, "transr relation histogram"))) return rc;
            if ((rc = hip_check(hipMemset(g_w.rel_hist, 0, sizeof(int32_t) * 4 * kRelBins), "zero relation histogram"))) return rc;
        }
        int32_t *pair = g_w.rel_hist + (g_w.rel_parity ? 2 * kRelBins : 0), *other = g_w.rel_hist + (g_w.rel_parity ? 0 : 2 * kRelBins);
        g_w.rel_parity ^= 1;
        const unsigned tiles = (unsigned)((slots + kRelTile - 1) / kRelTile);
        hipLaunchKernelGGL(rel_count_kernel, dim3(tiles), dim3(256), 0, stream, g_w.keys, (int)slots, ((int)R + 1) * kRelSub, pair, other);
        SamplerArgs ride = {};
        unsigned n_ride = 0;
        if (upload_jump_table() == KGE_OK && !take_attached_sampler(ride, n_ride)) n_ride = 0;
        hipLaunchKernelGGL(rel_scatter_kernel, dim3(tiles + n_ride), dim3(256), 0, stream, g_w.keys, g_w.vals, (int)slots, (int)R, pair,
                           pair + kRelBins, g_w.vals2, g_w.bucket_start, g_w.tile_rel, g_w.tile_row0, g_w.n_tiles, v2 ? 7 : 5, ride,
                           (int)tiles, (int)n_ride, 0, (int32_t *)nullptr);
    } else {
        size_t tmp = g_w.sort_tmp_bytes;
        rc = hip_check(rocprim::radix_sort_pairs(g_w.sort_tmp, tmp, g_w.keys, g_w.keys2, g_w.vals, g_w.vals2, (size_t)slots, 0,
                                                 bits_for(R), stream), "transr bucket sort");
        if (rc) return rc;
        hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(1024), 0, stream, g_w.keys2, (int)slots, (int)R, g_w.bucket_start,
                           g_w.tile_rel, g_w.tile_row0, g_w.n_tiles, v2 ? 7 : 5);
    }
    GemmArgs ga = {};
    ga.ent = tables[0]; ga.mat = tables[2]; ga.GP = g_w.GP; ga.P = g_w.P; ga.g_ent = grads[0];
    ga.sorted_slots = g_w.vals2; ga.job_ent = g_w.job_ent; ga.bucket_start = g_w.bucket_start;
    ga.tile_rel = g_w.tile_rel; ga.tile_row0 = g_w.tile_row0; ga.n_tiles = g_w.n_tiles;
    ga.De = De; ga.Dr = Dr;
    ga.rec_out = drec; ga.rec_dst = ddst; ga.row_ent = g_w.row_ent;
    const unsigned max_tiles = (unsigned)(slots / (v2 ? RW2 : 32) + R + 1);
    // sparse buckets (config #4's auto batch: 46 rows per relation, ~one tile per relation, fewer tiles than CUs): two column
    // blocks of 7 tiles per row tile instead of one of 13
    const bool split_cols = v2 && slots < 256 * R && engine().transr_v1 == 0;
    // v3: the same tiles with the products on the bf16 matrix pipe (three-term split, rows_gemm3_kernel)
    const bool v3 = v2 && engine().transr_bf16x3 && engine().transr_v1 == 0;
    if (v3 && Dr <= 112) hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_PROJECT, 7>), dim3(max_tiles), dim3(256), 0, stream, ga);
    else if (v3) hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_PROJECT, 13>), dim3(max_tiles), dim3(256), 0, stream, ga);
    else if (v2 && Dr <= 112) hipLaunchKernelGGL((rows_gemm2_kernel<GEMM_PROJECT, 7>), dim3(max_tiles, 1), dim3(256), 0, stream, ga);
    else if (v2 && split_cols) hipLaunchKernelGGL((rows_gemm2_kernel<GEMM_PROJECT, 7>), dim3(max_tiles, (Dr + 111) / 112), dim3(256), 0, stream, ga);
    else if (v2) hipLaunchKernelGGL((rows_gemm2_kernel<GEMM_PROJECT, 13>), dim3(max_tiles, 1), dim3(256), 0, stream, ga);
    else hipLaunchKernelGGL((rows_gemm_kernel<GEMM_PROJECT>), dim3(max_tiles, (Dr + TN - 1) / TN), dim3(256), 0, stream, ga);
    const bool lean = transr_lean_vector_stage(Dr);   // the lean vector stage writes every GP row that dgrad / wgrad read
    if (!lean && (rc = hip_check(hipMemsetAsync(g_w.GP, 0, sizeof(float) * (size_t)slots * Dr, stream), "zero GP"))) return rc;
    rc = launch_transr_vector_stage(tables[1], grads[1], g_w.P, g_w.GP, d_h, d_t, d_r, n_pos, n_neg, stride, denom, Dr, m.margin,
                                    m.negative_rel, d_loss, stream, lean, sampler_shaped, R);
    if (rc) return rc;
    if (v2) {
        if (v3 && De <= 112) hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_DGRAD, 7>), dim3(max_tiles), dim3(256), 0, stream, ga);
        else if (v3) hipLaunchKernelGGL((rows_gemm3_kernel<GEMM_DGRAD, 13>), dim3(max_tiles), dim3(256), 0, stream, ga);
        else if (De <= 112) hipLaunchKernelGGL((rows_gemm2_kernel<GEMM_DGRAD, 7>), dim3(max_tiles, 1), dim3(256), 0, stream, ga);
        else if (split_cols) hipLaunchKernelGGL((rows_gemm2_kernel<GEMM_DGRAD, 7>), dim3(max_tiles, (De + 111) / 112), dim3(256), 0, stream, ga);
        else hipLaunchKernelGGL((rows_gemm2_kernel<GEMM_DGRAD, 13>), dim3(max_tiles, 1), dim3(256), 0, stream, ga);
        if (dgrad_records) {
            FloatRowSpace rs = {};
            rs.g_ent = grads[0]; rs.E = m.ent_total; rs.R = 0; rs.hub_base = m.ent_total; rs.hub_rows = 1; rs.rows = m.ent_total;
            if ((rc = float_records_reduce(slots, De, rs, stream))) return rc;
        }
        // the all-output-tiles wgrad (full 13 x 13 tile grid only) pays one 160 kB flush per relation change: only with
        // well-filled buckets (measured: 316 vs 400 us at 574 rows per relation, 131 vs 77 us at 46)
        const bool full_grid = De > 192 && Dr > 192;
        const int opt = engine().transr_v1;
        if (full_grid && opt != 3) {
            // well-filled buckets: 512 rows between flushes; sparse buckets (a tile = one relation's whole bucket, stored by its
            // single owner): one tile per workgroup so that every tile is in flight
            const int span = (slots >= 256 * R || opt == 2) ? SPAN2 : 1;
            const dim3 wg2((max_tiles + span - 1) / span, 2);
            if (v3) hipLaunchKernelGGL(wgrad3_kernel, wg2, dim3(256), 0, stream, ga, grads[2], span);
            else hipLaunchKernelGGL(wgrad2_kernel, wg2, dim3(256), 0, stream, ga, grads[2], span);
        } else {
            const int tiles_i1 = (De + 31) / 32;
            const int wgt = WG_TILES * 32 / RW2;   // the same 256 rows between flushes as with 32-row tiles
            const unsigned groups1 = (max_tiles + wgt - 1) / wgt;
            hipLaunchKernelGGL(wgrad_kernel, dim3(groups1 * tiles_i1, (Dr + TN - 1) / TN), dim3(256), 0, stream, ga, grads[2], tiles_i1, RW2, wgt);
        }
        return hip_check(hipGetLastError(), "transr launch");
    }
    hipLaunchKernelGGL((rows_gemm_kernel<GEMM_DGRAD>), dim3(max_tiles, (De + TN - 1) / TN), dim3(256), 0, stream, ga);
    const int tiles_i = (De + 31) / 32;
    const unsigned groups = (max_tiles + WG_TILES - 1) / WG_TILES;
    hipLaunchKernelGGL(wgrad_kernel, dim3(groups * tiles_i, (Dr + TN - 1) / TN), dim3(256), 0, stream, ga, grads[2], tiles_i, 32, WG_TILES);
    return hip_check(hipGetLastError(), "transr launch");
}

int launch_predict_transr(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                          const int32_t *d_r, int64_t n, float *d_out, hipStream_t stream) {
    const int De = m.ent_dim, Dr = m.rel_dim;
    const int64_t R = m.rel_total, slots = 2 * n;
    int rc = ensure_work(slots, Dr, R);
    if (rc) return rc;
    int blocks = (int)((slots + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(predict_prep_kernel, dim3(blocks), dim3(256), 0, stream, d_h, d_t, d_r, (long long)n, (int)R, g_w.vals2,
                       g_w.job_ent, g_w.bucket_start, g_w.tile_rel, g_w.tile_row0, g_w.n_tiles);
    GemmArgs ga = {};
    ga.ent = tables[0]; ga.mat = tables[2]; ga.P = g_w.P;
    ga.sorted_slots = g_w.vals2; ga.job_ent = g_w.job_ent; ga.bucket_start = g_w.bucket_start;
    ga.tile_rel = g_w.tile_rel; ga.tile_row0 = g_w.tile_row0; ga.n_tiles = g_w.n_tiles;
    ga.De = De; ga.Dr = Dr;
    hipLaunchKernelGGL((rows_gemm_kernel<GEMM_PROJECT>), dim3((unsigned)((slots + 31) / 32), (Dr + TN - 1) / TN), dim3(256), 0,
                       stream, ga);
    return launch_transr_predict_stage(tables[1], g_w.P, d_r, n, Dr, d_out, stream);
}

// link prediction: P_out[j] = ent[j] . M_r for EVERY entity j (one relation bucket holding all E rows)
namespace {
__global__ void project_all_prep_kernel(long long E, int r0, int R, int32_t *__restrict__ vals, int32_t *__restrict__ job_ent,
                                        int32_t *__restrict__ bucket_start, int32_t *__restrict__ tile_rel,
                                        int32_t *__restrict__ tile_row0, int32_t *__restrict__ n_tiles) {
    for (long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x; slot < E; slot += (long long)gridDim.x * blockDim.x) {
        vals[slot] = (int32_t)slot;
        job_ent[slot] = (int32_t)slot;
        if ((slot & 31) == 0) { tile_rel[slot >> 5] = r0; tile_row0[slot >> 5] = (int32_t)slot; }
    }
    if (blockIdx.x == 0) {
        for (int r = threadIdx.x; r <= R + 1; r += blockDim.x) bucket_start[r] = r <= r0 ? 0 : (int32_t)E;
        if (threadIdx.x == 0) n_tiles[0] = (int)((E + 31) >> 5);
    }
}
}  // namespace

int transr_project_all(const kge_model_desc &m, const float *const tables[4], int64_t r, float *P_out, hipStream_t stream) {
    const int De = m.ent_dim, Dr = m.rel_dim;
    const int64_t R = m.rel_total, E = m.ent_total;
    int rc = ensure_work(E, Dr, R);
    if (rc) return rc;
    int blocks = (int)((E + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(project_all_prep_kernel, dim3(blocks), dim3(256), 0, stream, (long long)E, (int)r, (int)R, g_w.vals2, g_w.job_ent,
                       g_w.bucket_start, g_w.tile_rel, g_w.tile_row0, g_w.n_tiles);
    GemmArgs ga = {};
    ga.ent = tables[0]; ga.mat = tables[2]; ga.P = P_out;
    ga.sorted_slots = g_w.vals2; ga.job_ent = g_w.job_ent; ga.bucket_start = g_w.bucket_start;
    ga.tile_rel = g_w.tile_rel; ga.tile_row0 = g_w.tile_row0; ga.n_tiles = g_w.n_tiles;
    ga.De = De; ga.Dr = Dr;
    hipLaunchKernelGGL((rows_gemm_kernel<GEMM_PROJECT>), dim3((unsigned)((E + 31) / 32), (Dr + TN - 1) / TN), dim3(256), 0, stream, ga);
    return hip_check(hipGetLastError(), "transr project-all launch");
}

}  // namespace kge
