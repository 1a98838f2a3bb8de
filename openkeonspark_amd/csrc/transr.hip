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
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "engine.hpp"

namespace kge {

int launch_transr_vector_stage(const float *rel, float *g_rel, const float *P, float *GP, const int32_t *d_h,
                               const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                               int64_t denom, int rel_dim, float margin, int negative_rel, float *d_loss,
                               hipStream_t stream);
int launch_transr_predict_stage(const float *rel, const float *P, const int32_t *d_r, int64_t n, int rel_dim, float *d_out,
                                hipStream_t stream);

namespace {

struct TrWork {
    float *P = nullptr, *GP = nullptr;
    int32_t *keys = nullptr, *keys2 = nullptr, *vals = nullptr, *vals2 = nullptr, *job_ent = nullptr;
    int32_t *bucket_start = nullptr;  // [R+2]
    int32_t *tile_rel = nullptr, *tile_row0 = nullptr, *n_tiles = nullptr;
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
        size_t bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, bytes, g_w.keys, g_w.keys2, g_w.vals, g_w.vals2, (size_t)s, 0, 32, nullptr);
        if (bytes > g_w.sort_tmp_bytes) {
            if (g_w.sort_tmp) (void)hipFree(g_w.sort_tmp);
            g_w.sort_tmp = nullptr;
            if ((rc = hip_check(hipMalloc(&g_w.sort_tmp, bytes), "transr sort temp"))) return rc;
            g_w.sort_tmp_bytes = bytes;
        }
        g_w.cap_slots = s; g_w.cap_dim = d;
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
                            int32_t *__restrict__ keys, int32_t *__restrict__ vals, int32_t *__restrict__ job_ent) {
    const long long total = 2 * n_pos * (1 + n_neg);
    for (long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += (long long)gridDim.x * blockDim.x) {
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

// bucket_start[r] = first sorted position with key >= r (r = 0..R+1), then the (relation, row tile) map
__global__ __launch_bounds__(1024) void bounds_kernel(const int32_t *__restrict__ sorted_keys, int J, int R,
                                                      int32_t *__restrict__ bucket_start, int32_t *__restrict__ tile_rel,
                                                      int32_t *__restrict__ tile_row0, int32_t *__restrict__ n_tiles) {
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
    for (int r = r0; r < r1; r++) mine += (bucket_start[r + 1] - bucket_start[r] + 31) >> 5;
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
        for (int row = bucket_start[r]; row < bucket_start[r + 1]; row += 32) { tile_rel[t] = r; tile_row0[t] = row; t++; }
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

__device__ __forceinline__ void wgrad_flush(const GemmArgs &a, float *__restrict__ g_mat, int r, int i0, int jg, int lane,
                                            const f32x16 &acc) {
    if (jg >= a.Dr) return;
    float *G = g_mat + (long long)r * a.De * a.Dr;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int ig = i0 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (ig < a.De)
            __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(G + (long long)ig * a.Dr + jg), acc[reg]);
    }
}

__global__ __launch_bounds__(256) void wgrad_kernel(GemmArgs a, float *__restrict__ g_mat, int tiles_i) {
    const int grp = blockIdx.x / tiles_i, it = blockIdx.x - grp * tiles_i;
    const int n_tiles = a.n_tiles[0];
    const int t0 = grp * WG_TILES;
    if (t0 >= n_tiles) return;
    const int t1 = min(t0 + WG_TILES, n_tiles);
    __shared__ float As[KC][32 + 1];
    __shared__ float Bs[KC][TN + 1];
    __shared__ int s_slot[KC], s_ent[KC];
    const int i0 = it * 32, j0 = blockIdx.y * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jg = j0 + wave * 32 + (lane & 31);
    f32x16 acc = {0};
    int r_cur = a.tile_rel[t0];
    for (int t = t0; t < t1; t++) {
        const int r = a.tile_rel[t];
        if (r != r_cur) {
            wgrad_flush(a, g_mat, r_cur, i0, jg, lane, acc);
            acc = f32x16{0};
            r_cur = r;
        }
        const int row0 = a.tile_row0[t];
        const int rows = min(32, a.bucket_start[r + 1] - row0);
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
    wgrad_flush(a, g_mat, r_cur, i0, jg, lane, acc);
}

int bits_for(int64_t v) { int b = 1; while ((int64_t(1) << b) <= v) b++; return b; }

}  // namespace

int launch_forward_backward_transr(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h,
                                   const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                                   int64_t denom, float *const grads[4], float *d_loss, hipStream_t stream) {
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
    hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, stream, d_h, d_t, d_r, (long long)n_pos, (long long)n_neg,
                       (long long)stride, (int)m.negative_rel, (int)R, g_w.keys, g_w.vals, g_w.job_ent);
    size_t tmp = g_w.sort_tmp_bytes;
    rc = hip_check(rocprim::radix_sort_pairs(g_w.sort_tmp, tmp, g_w.keys, g_w.keys2, g_w.vals, g_w.vals2, (size_t)slots, 0,
                                             bits_for(R), stream), "transr bucket sort");
    if (rc) return rc;
    hipLaunchKernelGGL(bounds_kernel, dim3(1), dim3(1024), 0, stream, g_w.keys2, (int)slots, (int)R, g_w.bucket_start,
                       g_w.tile_rel, g_w.tile_row0, g_w.n_tiles);
    GemmArgs ga;
    ga.ent = tables[0]; ga.mat = tables[2]; ga.GP = g_w.GP; ga.P = g_w.P; ga.g_ent = grads[0];
    ga.sorted_slots = g_w.vals2; ga.job_ent = g_w.job_ent; ga.bucket_start = g_w.bucket_start;
    ga.tile_rel = g_w.tile_rel; ga.tile_row0 = g_w.tile_row0; ga.n_tiles = g_w.n_tiles;
    ga.De = De; ga.Dr = Dr;
    const unsigned max_tiles = (unsigned)(slots / 32 + R + 1);
    hipLaunchKernelGGL((rows_gemm_kernel<GEMM_PROJECT>), dim3(max_tiles, (Dr + TN - 1) / TN), dim3(256), 0, stream, ga);
    rc = hip_check(hipMemsetAsync(g_w.GP, 0, sizeof(float) * (size_t)slots * Dr, stream), "zero GP");
    if (rc) return rc;
    rc = launch_transr_vector_stage(tables[1], grads[1], g_w.P, g_w.GP, d_h, d_t, d_r, n_pos, n_neg, stride, denom, Dr, m.margin,
                                    m.negative_rel, d_loss, stream);
    if (rc) return rc;
    hipLaunchKernelGGL((rows_gemm_kernel<GEMM_DGRAD>), dim3(max_tiles, (De + TN - 1) / TN), dim3(256), 0, stream, ga);
    const int tiles_i = (De + 31) / 32;
    const unsigned groups = (max_tiles + WG_TILES - 1) / WG_TILES;
    hipLaunchKernelGGL(wgrad_kernel, dim3(groups * tiles_i, (Dr + TN - 1) / TN), dim3(256), 0, stream, ga, grads[2], tiles_i);
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

}  // namespace kge
