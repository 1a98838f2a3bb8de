// TransE sign-count gradient path, stages 2 and 3 (stage 1, the emit kernel, lives in models.hip).
//
// Why: the generic path adds one fp32 gradient row per touched row with memory-side atomics, and
// those run at ~1.1 TB/s on MI355X whatever the access pattern (experiments/atomic_bench.hip), 6-7x
// below plain stores -- the fused kernel was pinned to that floor (profiles/r01_a_*).  For TransE the
// upstream gradient of every NORMALISED vector is an integer multiple of 1/denom (sign vectors), so
// stage 1 emits 4x smaller int8 records with plain coalesced stores, and the reduction is exact
// integer arithmetic: order-independent, bit-reproducible, and identical on every data-parallel
// replica.
//
//   stage 2a  stable radix sort of (destination row, record id) pairs                    (rocPRIM)
//   stage 2b  segsum_kernel: one team per CHUNK of 64 sorted records; runs of equal destination are
//             summed in registers; a run that lies inside the chunk is STORED, only the first/last
//             run of a chunk (which may continue in the neighbour) is added with int32 atomics -- hub
//             rows are thus spread over many teams and ~1/64 of the records' bytes touch an atomic.
//   stage 3   apply_counts_kernel: per ROW, g = (1/denom) * inv * (S - x^ <x^,S>) + residual, then SGD
//             or TF1-semantics Adam in place, counts re-zeroed.  The normalise-backward is linear in
//             the upstream gradient, so applying it once to the summed counts equals the sum of the
//             per-use backward passes of the reference graph (TransE.py:12-15).
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "engine.hpp"
#include "team.hpp"
#include "sampler_dev.hpp"

namespace kge {

void transe_team_shape(int D, int &L, int &C);
int launch_transe_emit(const kge_model_desc &m, const float *ent, const float *rel, float *resid_ent, float *resid_rel,
                       const int32_t *d_h, const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                       int64_t denom, uint32_t *rec, int32_t *dst, int krel, float *d_loss, hipStream_t stream, bool track_deferred,
                       uint8_t *rec2 = nullptr);

int transe_deferred_groups(int32_t *out);

namespace {

int bits_for_rows(int64_t v) { int b = 1; while ((int64_t(1) << b) <= v) b++; return b; }

struct CtWork {
    uint32_t *rec = nullptr;
    float *hub_sums = nullptr;         // deterministic float-record reduce: per-copy sums of the relation-side rows
    size_t hub_sums_cap = 0;
    int32_t *dst = nullptr, *dst_sorted = nullptr, *ids = nullptr, *ids_sorted = nullptr;
    int32_t *n_valid = nullptr;
    int32_t *tile_hist = nullptr, *bucket_start = nullptr;   // LDS-bucket path: bucket totals + cursors, bucket starts
    int32_t *pairs = nullptr;                                  // [2*cap] (record id, destination) grouped by bucket
    float2 *pair_aux = nullptr;                                // [cap] pair-count path: (a, +-1/|projected|) per record
    int64_t pair_aux_cap = 0;
    int64_t cap_hist = 0;
    void *sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0, rec_elems = 0;
    int64_t cap_idx = 0;
    int32_t *bflag = nullptr;
    int64_t bflag_cap = 0;
    // fused single-process step: the plan of segapply_kernel
    int4 *pieces = nullptr;
    int64_t pieces_cap = 0;
    int32_t *n_pieces = nullptr;
    int2 *row_span = nullptr;
    int64_t row_span_cap = 0;
};
CtWork g_c;

template <typename T>
int regrow(T *&p, size_t count, const char *what) {
    if (p) (void)hipFree(p);
    p = nullptr;
    return hip_check(hipMalloc(&p, sizeof(T) * (count ? count : 1)), what);
}

// index workspace for M records (keys, ids, sort scratch) and -- rec_dwords > 0 -- an engine-owned record buffer
// (callers of the stage-level ABI bring their own records and pass 0)
int ensure_counts_work(int64_t M, size_t rec_dwords) {
    int rc;
    if (M > g_c.cap_idx) {
        const int64_t cap = M;
        if ((rc = regrow(g_c.dst, (size_t)cap, "counts dst"))) return rc;
        if ((rc = regrow(g_c.dst_sorted, (size_t)cap, "counts dst_sorted"))) return rc;
        if ((rc = regrow(g_c.ids, (size_t)cap, "counts ids"))) return rc;
        if ((rc = regrow(g_c.ids_sorted, (size_t)cap, "counts ids_sorted"))) return rc;
        if ((rc = regrow(g_c.pairs, (size_t)cap * 2, "counts pairs"))) return rc;
        if (!g_c.n_valid && (rc = regrow(g_c.n_valid, 1, "counts n_valid"))) return rc;
        size_t bytes = 0;
        (void)rocprim::radix_sort_pairs(nullptr, bytes, g_c.dst, g_c.dst_sorted, g_c.ids, g_c.ids_sorted, (size_t)cap, 0, 32, nullptr);
        if (bytes > g_c.sort_tmp_bytes) {
            if (g_c.sort_tmp) (void)hipFree(g_c.sort_tmp);
            g_c.sort_tmp = nullptr;
            if ((rc = hip_check(hipMalloc(&g_c.sort_tmp, bytes), "counts sort temp"))) return rc;
            g_c.sort_tmp_bytes = bytes;
        }
        // record ids 0..cap-1 never change
        std::vector<int32_t> iota((size_t)cap);
        for (int64_t i = 0; i < cap; i++) iota[(size_t)i] = (int32_t)i;
        if ((rc = hip_check(hipMemcpy(g_c.ids, iota.data(), sizeof(int32_t) * (size_t)cap, hipMemcpyHostToDevice), "counts iota"))) return rc;
        g_c.cap_idx = cap;
    }
    if (rec_dwords > 0 && (size_t)M * rec_dwords > g_c.rec_elems) {
        if ((rc = regrow(g_c.rec, (size_t)M * rec_dwords, "counts records"))) return rc;
        g_c.rec_elems = (size_t)M * rec_dwords;
    }
    return KGE_OK;
}

// destination -1 (inactive / no record) is remapped to the sentinel key `rows` so it sorts last
__global__ void fix_keys_kernel(int32_t *dst, long long M, int sentinel) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long long)gridDim.x * blockDim.x)
        if (dst[i] < 0) dst[i] = sentinel;
}

__global__ void count_valid_kernel(const int32_t *__restrict__ sorted, int M, int sentinel, int32_t *n_valid) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int lo = 0, hi = M;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (sorted[mid] < sentinel) lo = mid + 1; else hi = mid; }
        n_valid[0] = lo;
    }
}

constexpr int CHUNK = 64;

// value held by lane `i` of the calling lane's L-lane team (i a compile-time constant where it matters: a full wave reads it
// into an SGPR with v_readlane, narrower teams go through ds_bpermute)
template <int L>
__device__ __forceinline__ int team_pick(int v, int i) {
    if constexpr (L == 64) return __builtin_amdgcn_readlane(v, i);
    else return __shfl(v, i, L);
}
// entry i of a CHUNK-long list held as a[j] = entry (lane + L*j); never indexes `a` dynamically (an unrolled select), so the
// array stays in registers even where the caller's loop is not unrolled
template <int L, int J>
__device__ __forceinline__ int chunk_entry(const int (&a)[J], int i) {
    int v = a[0];
#pragma unroll
    for (int j = 1; j < J; j++) v = (i / L == j) ? a[j] : v;
    return team_pick<L>(v, i % L);
}

// element held in accumulator c of `lane`: strided layout (dword-per-lane kernels) or natural layout
// (vectorised kernels: record dword w = lane + L*(c/4) holds elements 4w .. 4w+3)
template <int L, bool NAT>
__device__ __forceinline__ int elem_of(int lane, int c) { return NAT ? 4 * (lane + L * (c / 4)) + (c % 4) : lane + L * c; }

// One run's sum into row `row` of S.  Natural layout (NAT: lane l holds elements 4(l + L q) .. +3): an exclusive run is one
// 16-byte store per lane; an ATOMIC run is first turned through LDS into the strided layout (lane l: elements l, l+L, ...),
// because memory-side atomics are served per 64-byte line an instruction touches -- adding element 4l+j from lane l (16-byte
// stride) makes every one of the four instructions touch all the row's lines, four times the line operations of contiguous
// adds (measured on the pair-count emit kernel: 452 -> 324 us).  `stage` = this TEAM's L*C ints; the traffic stays inside the
// wave, whose LDS operations complete in order (no barrier).
template <int L, int C, bool NAT>
__device__ __forceinline__ void flush_run(int32_t *__restrict__ S, int D, int lane, long long row, const int (&acc)[C], bool atomic,
                                          int32_t *stage) {
    int32_t *p = S + row * D;
    if constexpr (NAT && C % 4 == 0) {
        constexpr int Q = C / 4;
        if (!atomic) {
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int e0 = 4 * (lane + L * q);
                if (e0 < D) *reinterpret_cast<int4 *>(p + e0) = make_int4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < Q; q++)
            *reinterpret_cast<int4 *>(stage + 4 * (lane + L * q)) = make_int4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int e = lane + L * c;
            const int v = stage[e];
            if (e < D && v != 0) atomicAdd(p + e, v);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int e = elem_of<L, NAT>(lane, c);
            if (e < D) {
                if (atomic) { if (acc[c] != 0) atomicAdd(p + e, acc[c]); }
                else p[e] = acc[c];
            }
        }
    }
}

// d/dx of the normalised row applied to the integer sign sum: unit * (1/|x|) * (S - x^ <x^,S>), with every
// operation individually rounded so that all apply kernels agree bit for bit given the same reduction order
__device__ __forceinline__ float count_grad(float unit, float inv, float s, float d, float xn) {
#pragma clang fp contract(off)
    return mul_rn(mul_rn(unit, inv), sub_rn(s, mul_rn(d, xn)));
}

// Sparse-row SGD on ONE row from its summed integer counts held in the NATURAL layout of the vectorised kernels
// (accumulator c of lane l = element 4*(l + L*(c/4)) + c%4; D % 4 == 0): the single arithmetic used by the fused
// segmented-sum-and-apply kernel and by the row-list apply kernel, so a row gets the same bits whichever of them
// handles it (which one does depends on where chunk boundaries fall, i.e. on the number of ranks).
template <int L, int C>
__device__ __forceinline__ void apply_row_nat(const int (&acc)[C], float *__restrict__ p, int D, int lane, float unit, float lr) {
#pragma clang fp contract(off)   // (the same bits from both kernels that inline this body: see apply_row_update)
    constexpr int Q = (C + 3) / 4;
    float x[4 * Q], sv[4 * Q];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e0 = 4 * (lane + L * q);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e0 < D) v = *reinterpret_cast<const float4 *>(p + e0);
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            sv[4 * q + j] = (4 * q + j < C) ? (float)acc[(4 * q + j < C) ? 4 * q + j : 0] : 0.f;
            ss += x[4 * q + j] * x[4 * q + j];
        }
    }
    ss = team_sum<L>(ss);
    const bool uc = ss >= 1e-12f;
    const float inv = 1.0f / sqrtf(uc ? ss : 1e-12f);
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 4 * Q; c++) d += (x[c] * inv) * sv[c];
    d = team_sum<L>(d);
    if (!uc) d = 0.f;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e0 = 4 * (lane + L * q);
        if (e0 < D) {
            float o[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = 4 * q + j;
                const float g = add_rn(count_grad(unit, inv, sv[c], d, x[c] * inv), 0.f);
                o[j] = g != 0.f ? sub_rn(x[c], mul_rn(lr, g)) : x[c];
                any = any || g != 0.f;
            }
            if (any) *reinterpret_cast<float4 *>(p + e0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// fused segmented sum + apply (sparse rows, SGD): rows whose records all lie inside one chunk are updated straight from
// the registers; only chunk-boundary rows go through the compact image (flagged in bflag) and a second, small pass
struct FuseArgs {
    float *ent, *rel;
    long long E;
    float unit, lr;
    // dense path with virtual relation-row copies (emit kernel's krel > 1): a key >= fold_E is copy (key - fold_E) / fold_R of
    // relation (key - fold_E) % fold_R; its runs land in row fold_E + that relation, always with atomics (copies alias one row)
    int fold_E, fold_R;
};

template <int L, int C, bool NAT, bool FUSE = false>
__global__ __launch_bounds__(256) void segsum_kernel(const uint32_t *__restrict__ rec, const int32_t *__restrict__ keys,
                                                     const int32_t *__restrict__ ids, const int32_t *__restrict__ n_valid_p,
                                                     const int32_t *__restrict__ uidx, int32_t *__restrict__ S, int D, FuseArgs fz = FuseArgs()) {
    // uidx == nullptr: S is the dense [rows, D] table and a run lands in row `key`;
    // uidx != nullptr: S is compact, a run lands in row uidx[position of the run] (unique-row index)
    constexpr int TEAMS = 256 / L;
    constexpr int Q = (C + 3) / 4;
    constexpr int RD = L * Q;
    __shared__ int32_t stage_all[256 * C];                  // flush_run's turn-around buffer: L*C ints per team
    int32_t *stage = stage_all + (threadIdx.x / L) * (L * C);
    const int lane = threadIdx.x % L;
    const int n_valid = n_valid_p[0];
    const long long chunk = (long long)blockIdx.x * TEAMS + threadIdx.x / L;
    const long long start = chunk * CHUNK;
    if (start >= n_valid) return;
    const int n = (int)min((long long)CHUNK, n_valid - start);
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0;
    int cur = keys[start];
    const bool fold = !FUSE && fz.fold_R > 0;
    auto row_of = [&](int key, int u) { return u >= 0 ? u : ((fold && key >= fz.fold_E) ? fz.fold_E + (key - fz.fold_E) % fz.fold_R : key); };
    int cur_row = row_of(cur, uidx ? uidx[start] : -1);
    bool first_run = true;
    constexpr int U = 16;  // records in flight per team
    // the chunk's keys / record ids / unique-row indices: ONE coalesced load each (lane l holds entries l, l+L, ...),
    // handed to the team record by record with a lane broadcast -- not one dependent scalar load per group of U
    constexpr int J = CHUNK / L;
    int kl[J], idl[J], url[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int i = min(lane + L * j, n - 1);
        kl[j] = keys[start + i];
        idl[j] = ids[start + i];
        url[j] = uidx ? uidx[start + i] : kl[j];
    }
#pragma unroll
    for (int i0 = 0; i0 < CHUNK; i0 += U) {
        if (i0 >= n) break;
        int k[U], id[U], ur[U];
        uint32_t w[U][Q];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u;   // compile-time after unrolling; entries past n hold the clamped last record
            k[u] = chunk_entry<L, J>(kl, i);
            id[u] = chunk_entry<L, J>(idl, i);
            ur[u] = chunk_entry<L, J>(url, i);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t *p = rec + (long long)id[u] * RD;
#pragma unroll
            for (int q = 0; q < Q; q++) w[u][q] = p[lane + L * q];
        }
#pragma unroll
        for (int u = 0; u < U; u++) if (i0 + u >= n) k[u] = -1;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (k[u] < 0) break;
            if (k[u] != cur) {
                if (FUSE && !first_run) {   // an interior run: this team holds the row's whole sum
                    float *prow = cur < fz.E ? fz.ent + (long long)cur * D : fz.rel + ((long long)cur - fz.E) * D;
                    apply_row_nat<L, C>(acc, prow, D, lane, fz.unit, fz.lr);
                } else {
                    flush_run<L, C, NAT>(S, D, lane, cur_row, acc, first_run || (fold && cur >= fz.fold_E), stage);
                }
#pragma unroll
                for (int c = 0; c < C; c++) acc[c] = 0;
                cur = k[u];
                cur_row = row_of(k[u], uidx ? ur[u] : -1);
                first_run = false;
            }
#pragma unroll
            for (int q = 0; q < Q; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int c = 4 * q + j;
                    if (c < C) acc[c] += (int)(int8_t)(w[u][q] >> (8 * j));
                }
        }
    }
    flush_run<L, C, NAT>(S, D, lane, cur_row, acc, true, stage);
}


// ---------------------------------------------------------------------------------------------
// Small row spaces (FB15k-237-sized tables): bucket the records by destination RANGE with one
// counting pass (NB buckets of `rpb` consecutive rows), then one workgroup per bucket accumulates
// its records into an int32 LDS image of its rows (ds_add, conflict-free: consecutive lanes hit
// consecutive dwords) and writes the rows out with plain coalesced stores.  No global atomics at
// all, no comparison sort; the order inside a bucket is irrelevant because the sums are integers.
// ---------------------------------------------------------------------------------------------
constexpr int NB = 512;          // buckets (+1 trash bucket for inactive records)
constexpr int BTILE = 4096;      // records per histogram / scatter tile

__device__ __forceinline__ int bucket_of(int d, int rpb) { return d < 0 ? NB : d / rpb; }

// bucket_total[b] += number of records of this tile that fall into bucket b
__global__ __launch_bounds__(256) void bkt_hist_kernel(const int32_t *__restrict__ dst, int M, int rpb, int32_t *__restrict__ bucket_total,
                                                       int32_t *__restrict__ plan_counter, SamplerArgs ride, int n_tiles) {
    if ((int)blockIdx.x >= n_tiles) {      // a part of the next batch's sampler riding along (see bkt_scatter_kernel)
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - n_tiles, bern_lds);
        return;
    }
    __shared__ int hist[NB + 1];
    if (plan_counter && blockIdx.x == 0 && threadIdx.x == 0) plan_counter[0] = 0;   // fused step: the piece counter bkt_sort_kernel<true> adds to
    for (int i = threadIdx.x; i <= NB; i += 256) hist[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * BTILE;
    int d[BTILE / 256];
#pragma unroll
    for (int k = 0; k < BTILE / 256; k++) {   // all of the tile's loads in flight before the first LDS atomic
        const int m = base + threadIdx.x + 256 * k;
        d[k] = dst[m < M ? m : M - 1];
    }
#pragma unroll
    for (int k = 0; k < BTILE / 256; k++)
        if (base + (int)threadIdx.x + 256 * k < M) atomicAdd(&hist[bucket_of(d[k], rpb)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i <= NB; i += 256)
        if (hist[i]) atomicAdd(&bucket_total[i], hist[i]);
}

// (record id, destination row) pairs grouped by bucket; the order inside a bucket is arbitrary
// Every block first scans the NB+1 bucket totals itself (an exclusive scan of 513 ints in LDS is cheaper than the launch
// of a scan kernel); block 0 publishes bucket_start[0..NB+1] for the kernels that follow.  Cursors count from 0 inside
// each bucket; cursors and totals are re-zeroed by bkt_sort_kernel.
// Workgroups beyond the scatter's own tiles (n_ride of them) run the NEXT batch's sampler (kge_sampling_attach): this launch
// has one 256-thread workgroup per CU and waits on LDS / global atomics most of the time, the sampler is a latency-bound
// pointer chase with no dependence on anything in the step -- together they take about as long as the longer of the two.
__global__ __launch_bounds__(256) void bkt_scatter_kernel(const int32_t *__restrict__ dst, int M, int rpb, const int32_t *__restrict__ bucket_total,
                                                          int32_t *__restrict__ bucket_start, int32_t *__restrict__ cursor,
                                                          int2 *__restrict__ pairs, SamplerArgs ride, int n_tiles, int n_ride) {
    if ((int)blockIdx.x >= n_tiles) {
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - n_tiles, bern_lds);
        return;
    }
    __shared__ int hist[NB + 1];
    __shared__ int base_of[NB + 1];
    __shared__ int scan[1024];
    const int base = blockIdx.x * BTILE;
    int d[BTILE / 256];
#pragma unroll
    for (int k = 0; k < BTILE / 256; k++) {   // the tile's destinations travel while the totals are scanned
        const int m = base + threadIdx.x + 256 * k;
        d[k] = m < M ? dst[m] : -2;
    }
    for (int i = threadIdx.x; i <= NB; i += 256) hist[i] = 0;
    {   // inclusive scan of the NB+1 totals (padded to 1024): 4 consecutive entries per thread, a shuffle scan per wave, the
        // four wave totals through LDS -- two barriers (a Hillis-Steele scan over LDS took twenty)
        __shared__ int wave_total[4];
        int v[4], run = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = 4 * threadIdx.x + k; v[k] = i <= NB ? bucket_total[i] : 0; run += v[k]; v[k] = run; }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        int base = incl - run;
        for (int w = 0; w < wave; w++) base += wave_total[w];
#pragma unroll
        for (int k = 0; k < 4; k++) scan[4 * threadIdx.x + k] = base + v[k];
        __syncthreads();
    }
    // exclusive start of bucket i = scan[i - 1]
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i <= NB + 1; i += 256) bucket_start[i] = i ? scan[i - 1] : 0;
    }
#pragma unroll
    for (int k = 0; k < BTILE / 256; k++)
        if (base + (int)threadIdx.x + 256 * k < M) atomicAdd(&hist[bucket_of(d[k], rpb)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i <= NB; i += 256) {
        const int c = hist[i];
        base_of[i] = c ? (i ? scan[i - 1] : 0) + atomicAdd(&cursor[i], c) : 0;   // reserve this tile's range in bucket i
        hist[i] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BTILE / 256; k++) {
        const int m = base + threadIdx.x + 256 * k;
        if (m < M && d[k] >= 0) {
            const int bk = bucket_of(d[k], rpb);
            const int pos = base_of[bk] + atomicAdd(&hist[bk], 1);
            pairs[pos] = make_int2(m, d[k]);
        }
    }
}


// The three bucketing launches (and the launch that ends the step) can each take a PART of an armed sampler along: engine option
// "ride_shares", percent of the sampler's workgroups for hist / scatter / sort / closing launch.  Default: all of it in the
// scatter launch (measured: spreading it made the step slower, see engine.hpp).
static unsigned take_ride(SamplerArgs &ride, int which) {
    unsigned n_ride = 0;
    const int pct = (engine().ride_shares >> (8 * which)) & 0xFF;
    // this translation unit's copy of the jump table first: an armed sampler is only taken along once it can run here
    if (pct == 0 || upload_jump_table() != KGE_OK || !take_attached_sampler(ride, n_ride, pct >= 100 ? 1.0f : 0.01f * (float)pct)) n_ride = 0;
    return n_ride;
}
static void launch_bkt_hist(int n_tiles, int M, int rpb, int32_t *totals, int32_t *plan_counter, hipStream_t stream) {
    SamplerArgs ride = {};
    const unsigned n_ride = take_ride(ride, 0);
    hipLaunchKernelGGL(bkt_hist_kernel, dim3((unsigned)n_tiles + n_ride), dim3(256), 0, stream, g_c.dst, M, rpb, totals, plan_counter, ride, n_tiles);
}
static void launch_bkt_scatter(int n_tiles, int M, int rpb, int32_t *totals, int32_t *cursor, int2 *pairs, hipStream_t stream) {
    SamplerArgs ride = {};
    const unsigned n_ride = take_ride(ride, 1);
    hipLaunchKernelGGL(bkt_scatter_kernel, dim3((unsigned)n_tiles + n_ride), dim3(256), 0, stream, g_c.dst, M, rpb, totals, g_c.bucket_start,
                       cursor, pairs, ride, n_tiles, (int)n_ride);
}

// second level: counting sort of one bucket's pairs by destination row (rpb rows, LDS histogram), so
// that the whole list is row-sorted and the register-accumulating segsum_kernel can consume it.
// (An LDS-image reduction with ds_add per element was tried first: correct, but bound by the LDS
// atomic rate -- 3.8 M wave-level ds_add per step -- at 230 us; registers + sorted runs take ~60.)
// PLAN (the fused single-process step; keys are 2 * row + kind, FbArgs::rec2): the bucket also writes where each of its keys
// lies in the sorted list -- row_span[key] = (first position, number of records), for EVERY key of the bucket, empty ones
// included -- which is all segapply_kernel needs to give every row a team of its own (a row = two adjacent keys = two
// homogeneous record lists).  A row with a list of more than `cap` records is cut into pieces of at most `cap` records of one
// list (slots from one global counter, zeroed by bkt_hist_kernel), each piece a team of its own that adds into the count image.
// rpb is even, so the two keys of a row lie in the same bucket.
struct SegPlan {
    int2 *row_span;        // [2 * virtual rows]
    int4 *pieces;          // [max_pieces]  (key, first position, records, 0)
    int32_t *n_pieces;
    int cap;
};

template <bool PLAN>
__global__ __launch_bounds__(256) void bkt_sort_kernel(const int2 *__restrict__ pairs, const int32_t *__restrict__ bucket_start,
                                                       int rpb, int rows, int32_t *__restrict__ out_keys,
                                                       int32_t *__restrict__ out_ids, int32_t *__restrict__ bucket_total,
                                                       int32_t *__restrict__ cursor, SegPlan plan, SamplerArgs ride) {
    extern __shared__ int lds_h[];   // [rpb] histogram, then running offsets
    if ((int)blockIdx.x >= NB) {           // a part of the next batch's sampler riding along (see bkt_scatter_kernel)
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - NB, bern_lds);
        return;
    }
    const int b = blockIdx.x;
    const int row0 = b * rpb;
    if (threadIdx.x == 0) {   // totals and cursors back to zero for the next step (block 0 also the trash bucket's)
        bucket_total[b] = 0; cursor[b] = 0;
        if (b == 0) { bucket_total[NB] = 0; cursor[NB] = 0; }
    }
    if (row0 >= rows) return;
    const int start = bucket_start[b], end = bucket_start[b + 1];
    for (int i = threadIdx.x; i < rpb; i += 256) lds_h[i] = 0;
    if (end == start) {
        if constexpr (PLAN)
            for (int i = threadIdx.x; i < rpb; i += 256) if (row0 + i < rows) plan.row_span[row0 + i] = make_int2(start, 0);
        return;
    }
    // the first 256*PC pairs of the bucket (all of it, normally) are read ONCE, into registers, for both passes
    constexpr int PC = 8;
    int2 pr[PC];
#pragma unroll
    for (int k = 0; k < PC; k++) pr[k] = pairs[min(start + (int)threadIdx.x + 256 * k, end - 1)];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PC; k++)
        if (start + (int)threadIdx.x + 256 * k < end) atomicAdd(&lds_h[pr[k].y - row0], 1);
    for (int i = start + 256 * PC + threadIdx.x; i < end; i += 256) atomicAdd(&lds_h[pairs[i].y - row0], 1);
    __syncthreads();
    {   // exclusive scan of the rpb counters: every thread sums its own contiguous span, one wave scans the 256 span sums,
        // every thread then rewrites its span (a single wave walking all counters 64 at a time took 105 serial rounds at
        // rpb = 6 730 -- the pair keys of a 237-relation graph -- and 39 us)
        __shared__ int span_sum[256];
        const int per = (rpb + 255) / 256;
        const int lo = min((int)threadIdx.x * per, rpb), hi = min(lo + per, rpb);
        int sum = 0;
        for (int i = lo; i < hi; i++) sum += lds_h[i];
        span_sum[threadIdx.x] = sum;
        __syncthreads();
        if (threadIdx.x < 64) {
            int v[4], tot = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = span_sum[4 * threadIdx.x + k]; tot += v[k]; }
            int incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if ((int)threadIdx.x >= off) incl += t; }
            int run = incl - tot;
#pragma unroll
            for (int k = 0; k < 4; k++) { span_sum[4 * threadIdx.x + k] = run; run += v[k]; }
        }
        __syncthreads();
        int run = span_sum[threadIdx.x];
        for (int i = lo; i < hi; i++) { const int c = lds_h[i]; lds_h[i] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PC; k++)
        if (start + (int)threadIdx.x + 256 * k < end) {
            const int pos = start + atomicAdd(&lds_h[pr[k].y - row0], 1);
            out_keys[pos] = pr[k].y;
            out_ids[pos] = pr[k].x;
        }
    for (int i = start + 256 * PC + threadIdx.x; i < end; i += 256) {
        const int2 q = pairs[i];
        const int pos = start + atomicAdd(&lds_h[q.y - row0], 1);
        out_keys[pos] = q.y;
        out_ids[pos] = q.x;
    }
    if constexpr (PLAN) {
        __syncthreads();     // lds_h[i] is now the END offset of key i inside the bucket (= the start of key i + 1)
        const int cap = plan.cap;
        for (int i = threadIdx.x; i < rpb; i += 256) {
            const int key = row0 + i;
            if (key >= rows) break;
            const int lo = i ? lds_h[i - 1] : 0, c = lds_h[i] - lo;
            plan.row_span[key] = make_int2(start + lo, c);
            const int io = i ^ 1;                                         // the row's other list (same bucket: rpb is even)
            const int co = lds_h[io] - (io ? lds_h[io - 1] : 0);
            if (c > 0 && (c > cap || co > cap)) {                         // a long row: BOTH its lists go to the image, in pieces
                const int np = (c + cap - 1) / cap;
                const int base = atomicAdd(plan.n_pieces, np);
                for (int q = 0; q < np; q++) plan.pieces[base + q] = make_int4(key, start + lo + q * cap, min(cap, c - q * cap), 0);
            }
        }
    }
}


template <bool PLAN>
static void launch_bkt_sort(int2 *pairs, int rpb, int rows, int32_t *totals, int32_t *cursor, const SegPlan &plan, hipStream_t stream) {
    SamplerArgs ride = {};
    const unsigned n_ride = take_ride(ride, 2);
    hipLaunchKernelGGL(bkt_sort_kernel<PLAN>, dim3(NB + n_ride), dim3(256), sizeof(int) * (size_t)rpb, stream, pairs, g_c.bucket_start, rpb, rows,
                       g_c.dst_sorted, g_c.ids_sorted, totals, cursor, plan, ride);
}

// ---------------------------------------------------------------------------------------------
// Sparse form (tables too large for a dense [rows, D] count image, and the multi-GPU record exchange):
// the sorted records are summed into a COMPACT [n_unique, D] image, one row per touched table row.
// ---------------------------------------------------------------------------------------------
__global__ void run_flags_kernel(const int32_t *__restrict__ keys, const int32_t *__restrict__ n_valid_p, int M, int32_t *__restrict__ flags) {
    const int n = n_valid_p[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x)
        flags[i] = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}

// uidx1 = inclusive scan of the run-start flags: record i belongs to unique row uidx1[i]-1
__global__ void unique_rows_kernel(const int32_t *__restrict__ keys, const int32_t *__restrict__ n_valid_p, int32_t *__restrict__ uidx1,
                                   int32_t *__restrict__ rows_out, int32_t *__restrict__ n_rows_out, int32_t *__restrict__ S, int D,
                                   int32_t *__restrict__ bflag) {
    // bflag (fused path, zeroed beforehand): 1 for the rows that go through the compact image
    const int n = n_valid_p[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int u = uidx1[i] - 1;
        uidx1[i] = u;
        if (i == 0 || keys[i] != keys[i - 1]) rows_out[u] = keys[i];
        if (i == n - 1) n_rows_out[0] = u + 1;
        // rows that a chunk's first / last run touches are accumulated with atomics: clear them first
        if (bflag && ((i % CHUNK) == 0 || (i % CHUNK) == CHUNK - 1 || i == n - 1)) bflag[u] = 1;
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) n_rows_out[0] = 0;
}

// rows that a chunk's first / last run touches are accumulated with atomics: clear them first (64 lanes per row)
__global__ __launch_bounds__(256) void zero_boundary_rows_kernel(const int32_t *__restrict__ uidx, const int32_t *__restrict__ n_valid_p,
                                                                 int32_t *__restrict__ S, int D) {
    const int n = n_valid_p[0];
    const int lane = threadIdx.x & 63;
    const long long n_chunks = ((long long)n + CHUNK - 1) / CHUNK;
    for (long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); c < n_chunks; c += (long long)gridDim.x * 4) {
        const long long first = c * CHUNK, last = min(first + CHUNK, (long long)n) - 1;
        const int u0 = uidx[first], u1 = uidx[last];
        for (int e = lane; e < D; e += 64) {
            S[(long long)u0 * D + e] = 0;
            S[(long long)u1 * D + e] = 0;
        }
    }
}



// ---------------------------------------------------------------------------------------------
// Float records (TransH / TransD / TransE without counts): the same ordering machinery, fp32 payload.
// fwdbwd_kernel<..., REC> stores each gradient row as a D-float record; here the records are ordered by
// destination row and summed by segments INTO the dense accumulators (which may already hold the few
// rows the atomic standalone path added): interior runs of a chunk are read-modify-written by their one
// owner, the first and last run of a chunk use atomics.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float *float_row_ptr(const FloatRowSpace &rs, long long row, int D) {
    if (row < rs.E) return rs.g_ent + row * D;
    if (row < rs.hub_base) return rs.g_auxe + (row - rs.E) * D;
    const long long q = (row - rs.hub_base) % rs.hub_rows;   // fold the hub copies back
    return q < rs.R ? rs.g_rel + q * D : rs.g_auxr + (q - rs.R) * D;
}

template <int L, int C>
__device__ __forceinline__ void flush_run_f32(const FloatRowSpace &rs, int D, int lane, long long row, const float (&acc)[C], bool atomic) {
    float *p = float_row_ptr(rs, row, D);
    atomic = atomic || row >= rs.hub_base;   // several virtual hub rows fold onto one real row: never exclusive
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int e = lane + L * c;
        if (e < D) {
            const float v = mul_rn(rs.scale, acc[c]);
            if (atomic) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(p + e), v);
            else p[e] += v;
        }
    }
}

template <int L, int C>
__global__ __launch_bounds__(256) void segsum_f32_kernel(const float *__restrict__ rec, const int32_t *__restrict__ keys,
                                                         const int32_t *__restrict__ ids, const int32_t *__restrict__ n_valid_p,
                                                         FloatRowSpace rs, int D, int chunk_len) {
    constexpr int TEAMS = 256 / L;
    constexpr int U = C <= 4 ? 8 : 4;   // records in flight per team
    const int lane = threadIdx.x % L;
    const int n_valid = n_valid_p[0];
    const long long chunk = (long long)blockIdx.x * TEAMS + threadIdx.x / L;
    const long long start = chunk * chunk_len;
    if (start >= n_valid) return;
    const int n = (int)min((long long)chunk_len, n_valid - start);
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0.f;
    int cur = keys[start];
    bool first_run = true;
    // the chunk's keys and record ids in one coalesced load each (chunk_len <= CHUNK), broadcast per record (see segsum_kernel)
    constexpr int J = CHUNK / L;
    int kl[J], idl[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int i = min(lane + L * j, n - 1);
        kl[j] = keys[start + i];
        idl[j] = ids[start + i];
    }
#pragma unroll
    for (int i0 = 0; i0 < CHUNK; i0 += U) {
        if (i0 >= n) break;
        int k[U];
        float x[U][C];
#pragma unroll
        for (int u = 0; u < U; u++) {   // unconditional loads (entries past n repeat the last record): all in flight together
            const int i = i0 + u;
            k[u] = chunk_entry<L, J>(kl, i);
            const float *p = rec + (long long)chunk_entry<L, J>(idl, i) * D;
#pragma unroll
            for (int c = 0; c < C; c++) { const int e = lane + L * c; x[u][c] = p[e < D ? e : 0]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (i0 + u >= n) break;
            if (k[u] != cur) {
                flush_run_f32<L, C>(rs, D, lane, cur, acc, first_run);
#pragma unroll
                for (int c = 0; c < C; c++) acc[c] = 0.f;
                cur = k[u];
                first_run = false;
            }
#pragma unroll
            for (int c = 0; c < C; c++) acc[c] += x[u][c];
        }
    }
    flush_run_f32<L, C>(rs, D, lane, cur, acc, true);
}


// Deterministic form of the float-record sum, for the data-parallel row-wise SGD (every rank reduces the same gathered records and
// the replicas must come out bit-identical): the records are ordered by a STABLE sort (rocPRIM radix sort: equal keys keep their
// record order), ONE team sums a whole run of equal keys in that order, an entity-side row has a single run and is updated by its
// team alone, and the hub copies of a relation-side row are first summed per copy into hub_sums [copy][hub row][D] and then folded in
// copy order by one thread per element (hub_fold_det_kernel).  No atomics, no order that depends on scheduling.
template <int L, int C>
__global__ __launch_bounds__(256) void segsum_f32_runs_kernel(const float *__restrict__ rec, const int32_t *__restrict__ keys,
                                                              const int32_t *__restrict__ ids, const int32_t *__restrict__ n_valid_p,
                                                              FloatRowSpace rs, int D, float *__restrict__ hub_sums) {
    constexpr int TEAMS = 256 / L;
    const int lane = threadIdx.x % L;
    const int n_valid = n_valid_p[0];
    for (long long i = (long long)blockIdx.x * TEAMS + threadIdx.x / L; i < n_valid; i += (long long)gridDim.x * TEAMS) {
        const int key = keys[i];
        if (i > 0 && keys[i - 1] == key) continue;           // not the first record of its run
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] = 0.f;
        long long j = i;
        do {
            const float *p = rec + (long long)ids[j] * D;
#pragma unroll
            for (int c = 0; c < C; c++) { const int e = lane + L * c; if (e < D) acc[c] += p[e]; }
            j++;
        } while (j < n_valid && keys[j] == key);
        if (key < rs.hub_base) {
            float *p = float_row_ptr(rs, key, D);
#pragma unroll
            for (int c = 0; c < C; c++) { const int e = lane + L * c; if (e < D) p[e] += mul_rn(rs.scale, acc[c]); }
        } else {
            float *p = hub_sums + (long long)(key - rs.hub_base) * D;
#pragma unroll
            for (int c = 0; c < C; c++) { const int e = lane + L * c; if (e < D) p[e] = acc[c]; }
        }
    }
}

// relation-side rows: the copies' sums in copy order (and the buffer re-zeroed for the next step)
__global__ __launch_bounds__(256) void hub_fold_det_kernel(float *__restrict__ hub_sums, FloatRowSpace rs, int D, int K) {
    const long long n = rs.hub_rows * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        bool any = false;
        for (int k = 0; k < K; k++) {
            const float v = hub_sums[(long long)k * n + i];
            if (v != 0.f) { s += v; any = true; hub_sums[(long long)k * n + i] = 0.f; }
        }
        if (any) {
            const long long q = i / D, e = i - q * D;
            float *p = q < rs.R ? rs.g_rel + q * D : rs.g_auxr + (q - rs.R) * D;
            p[e] += mul_rn(rs.scale, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pair-count path (TransH / TransD; stage 1 is pairs.hip's pair_emit_kernel): int8 sign records keyed by
// (entity x, relation r) as x*R + r.  The records of a key are summed as integers; at the end of a run the
// pair's entity-row gradient is computed ONCE in fp32 -- the backward of normalise + projection is linear in the
// upstream gradient for a fixed (x, r) -- and accumulated per ENTITY ROW in registers (the keys of one entity are
// adjacent); interior rows of a chunk are read-modify-written by their one owner, the first and last row with atomics.
// One wave per chunk of 64 sorted records, lane l holding elements 4l..4l+3 (D <= 256).
// ---------------------------------------------------------------------------------------------
struct PairRed {
    const float *ent, *ctx, *auxe;    // ent_embeddings; per relation: TransH the NORMALISED normal vector (ctx_normalize_kernel), TransD rel_transfer; ent_transfer (TransD)
    float *g_ent, *g_auxe;
    const float2 *aux;                // per record: (a, +-1/|projected|) of its pair, from the emit kernel's forward
    int D, R, RD;                     // embedding width, relations, dwords per int8 record
    long long n_int8;                 // records [0, n_int8) are int8 sums (RD dwords each); the rest are 2-bit sign records of 16 dwords
    unsigned magic;                   // ceil(2^32 / R): key / R by multiply-high (+ fix-up)
    float unit;
};

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// l2_normalize of every normal vector, once per step (TransH): the per-pair backward then needs no norm of w
__global__ __launch_bounds__(256) void ctx_normalize_kernel(const float *__restrict__ w, int R, int D, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
        float ss = 0.f;
        for (int e = lane; e < D; e += 64) ss += w[(long long)r * D + e] * w[(long long)r * D + e];
        ss = team_sum<64>(ss);
        const float inv = 1.0f / sqrtf(ss >= 1e-12f ? ss : 1e-12f);
        for (int e = lane; e < D; e += 64) out[(long long)r * D + e] = w[(long long)r * D + e] * inv;
    }
}

template <int MODEL>
__global__ __launch_bounds__(256) void segsum_pairs_kernel(const uint32_t *__restrict__ rec, const int32_t *__restrict__ keys,
                                                           const int32_t *__restrict__ ids, const int32_t *__restrict__ n_valid_p, PairRed pr) {
    constexpr int L = 64;
    constexpr int W = 4;   // records of a pair requested together with its rows (longer runs: a follow-up loop)
    __shared__ float stage_all[4][256];   // atomic flushes go out in the strided layout (see flush_run): one turn-around buffer per wave
    float *stage = stage_all[threadIdx.x / L];
    const int lane = threadIdx.x % L;
    const int n_valid = n_valid_p[0];
    const long long chunk = (long long)blockIdx.x * (256 / L) + threadIdx.x / L;
    const long long start = chunk * CHUNK;
    if (start >= n_valid) return;
    const int n = (int)min((long long)CHUNK, n_valid - start);
    const bool valid = 4 * lane < pr.D;
    const int D = pr.D, R = pr.R;
    const int lane4 = valid ? 4 * lane : 0;   // clamped element offset: loads stay unconditional, invalid lanes are zeroed by selects
    const int kl = keys[start + min(lane, n - 1)], idl = ids[start + min(lane, n - 1)];
    // the runs (pairs) of this chunk are known up front: bit i of `starts` = record i opens a run
    const int prev = __shfl_up(kl, 1);
    unsigned long long starts = __ballot(lane < n && (lane == 0 || kl != prev));
    float4 racc = make_float4(0.f, 0.f, 0.f, 0.f), racc2 = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur_row = -1;
    bool first_row = true;
    auto flush_one = [&](float *tab, const float4 &v, bool atomic) {
        float *pr_ = tab + (long long)cur_row * D;
        if (atomic) {
            *reinterpret_cast<float4 *>(stage + 4 * lane) = v;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int e = lane + L * c;
                const float x = stage[e];
                if (e < D) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(pr_ + e), x);
            }
        } else if (valid) {
            float4 o = *reinterpret_cast<float4 *>(pr_ + 4 * lane);
            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            *reinterpret_cast<float4 *>(pr_ + 4 * lane) = o;
        }
    };
    auto flush_row = [&](bool atomic) {
        flush_one(pr.g_ent, racc, atomic);
        if constexpr (MODEL == KGE_TRANSD) flush_one(pr.g_auxe, racc2, atomic);
        racc = make_float4(0.f, 0.f, 0.f, 0.f); racc2 = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // one pair = one run of equal keys: [lo, hi) inside the chunk.  Its rows and first W records are REQUESTED one pair ahead
    // of the arithmetic (the dependent gathers, not the arithmetic, bounded the first version of this kernel).
    struct Pair { int lo, hi, row, rel, signs; float4 x, cw, xa; float2 ai; uint32_t w[W]; };   // signs: bit u = record u is a 2-bit sign record
    // record `id`: an int8 record gives this lane the dword of its four elements; a 2-bit sign record gives it the dword that
    // holds them as the four fields of byte lane/16 (the emit kernel's 16-lane layout); which kind goes into bit u of `kinds`
    const long long base2 = pr.n_int8 * pr.RD;
    const int sh2 = 8 * (lane >> 4);
    auto load_record = [&](long long id, int &kinds, int u) -> uint32_t {
        if (id < pr.n_int8) return rec[id * pr.RD + (valid ? lane : 0)];
        kinds |= 1 << u;
        return rec[base2 + (id - pr.n_int8) * 16 + (lane & 15)];
    };
    // element j of this lane from a record dword of either kind
    auto elem = [&](uint32_t w, bool sign2, int j) -> int {
        return sign2 ? __builtin_amdgcn_sbfe((int)w, sh2 + 2 * j, 2) : (int)(int8_t)(w >> (8 * j));
    };
    int req_row = -1;                      // the entity row most recently fetched by request(), and its data
    float4 req_x = make_float4(0.f, 0.f, 0.f, 0.f), req_xa = make_float4(0.f, 0.f, 0.f, 0.f);
    auto request = [&](Pair &q, int lo, int hi) {
        q.lo = lo; q.hi = hi;
        const unsigned key = (unsigned)__builtin_amdgcn_readlane(kl, lo);
        int row = (int)__umulhi(key, pr.magic), rel = (int)key - row * R;
        if (rel < 0) { row--; rel += R; }
        if (rel >= R) { row++; rel -= R; }
        q.row = row; q.rel = rel;
        // the pairs of one entity are adjacent in key order (a dozen per entity at 25 negatives): its row(s) are fetched once, not
        // once per pair -- the kernel is bound by the bytes it gathers (records + rows), and the rows were over half of them
        if (row != req_row) {
            req_x = *reinterpret_cast<const float4 *>(pr.ent + (long long)row * D + lane4);
            if constexpr (MODEL == KGE_TRANSD) req_xa = *reinterpret_cast<const float4 *>(pr.auxe + (long long)row * D + lane4);
            req_row = row;
        }
        q.x = req_x;
        if constexpr (MODEL == KGE_TRANSD) q.xa = req_xa;
        q.cw = *reinterpret_cast<const float4 *>(pr.ctx + (long long)rel * D + lane4);
        q.signs = 0;
#pragma unroll
        for (int u = 0; u < W; u++) {
            const long long id = __builtin_amdgcn_readlane(idl, min(lo + u, hi - 1));
            q.w[u] = load_record(id, q.signs, u);
            if (u == 0) q.ai = pr.aux[id];   // every record of a pair carries the same two numbers
        }
    };
    auto next_run = [&](int &lo, int &hi) {
        lo = __ffsll((long long)starts) - 1;
        starts &= starts - 1;
        hi = starts ? __ffsll((long long)starts) - 1 : n;
    };
    Pair cur;
    { int lo, hi; next_run(lo, hi); request(cur, lo, hi); }
    for (;;) {
        const bool more = starts != 0;
        Pair nxt;
        if (more) { int lo, hi; next_run(lo, hi); request(nxt, lo, hi); }
        // ---- integer sum of the run ----
        int acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = elem(cur.w[0], cur.signs & 1, j);
        if (cur.hi - cur.lo > 1) {   // (most runs are one record long: the wave-uniform branch skips the other three unpack-and-adds)
#pragma unroll
            for (int u = 1; u < W; u++) {
                if (cur.lo + u < cur.hi) {
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] += elem(cur.w[u], (cur.signs >> u) & 1, j);
                }
            }
        }
        for (int i = cur.lo + W; i < cur.hi; i++) {   // long runs (hub pairs): the rest, one record at a time
            const long long id = __builtin_amdgcn_readlane(idl, i);
            int kind = 0;
            const uint32_t w = load_record(id, kind, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] += elem(w, kind & 1, j);
        }
        if (cur.row != cur_row) {
            if (cur_row >= 0) { flush_row(first_row); first_row = false; }
            cur_row = cur.row;
        }
        // ---- the pair's entity-row gradient (side_backward of models_dev.hpp applied once to the summed signs) ----
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 x = valid ? cur.x : z, cw = valid ? cur.cw : z;
        // the projection coefficient and 1/|projected| of this (entity, relation) pair come from the emit kernel's forward
        // (two reductions and a rsqrt per pair less; they are the very numbers the scores were computed with)
        float4 xa = z, xp;
        const float a = cur.ai.x;
        const bool uc = cur.ai.y > 0.f;
        const float inv = fabsf(cur.ai.y);
        if constexpr (MODEL == KGE_TRANSH) {
            xp = make_float4(x.x - a * cw.x, x.y - a * cw.y, x.z - a * cw.z, x.w - a * cw.w);
        } else {
            xa = valid ? cur.xa : z;
            xp = make_float4(x.x + a * cw.x, x.y + a * cw.y, x.z + a * cw.z, x.w + a * cw.w);
        }
        const float4 S = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
        const float dp = team_sum<L>(dot4(xp, S));
        const float al = uc ? inv * pr.unit * dp : 0.f;       // <nrm, G>,  G = unit S
        const float cg = inv * pr.unit, cx = -inv * inv * al;   // gxp = inv (G - al nrm) = cg S + cx xp
        const float4 gxp = make_float4(cg * S.x + cx * xp.x, cg * S.y + cx * xp.y, cg * S.z + cx * xp.z, cg * S.w + cx * xp.w);
        const float d = team_sum<L>(dot4(gxp, cw));
        if constexpr (MODEL == KGE_TRANSH) {
            racc.x += gxp.x - d * cw.x; racc.y += gxp.y - d * cw.y; racc.z += gxp.z - d * cw.z; racc.w += gxp.w - d * cw.w;
        } else {
            racc.x += gxp.x + d * xa.x; racc.y += gxp.y + d * xa.y; racc.z += gxp.z + d * xa.z; racc.w += gxp.w + d * xa.w;
            racc2.x += d * x.x; racc2.y += d * x.y; racc2.z += d * x.z; racc2.w += d * x.w;
        }
        if (!more) break;
        cur = nxt;
    }
    flush_row(true);
}

}  // namespace

int float_records_workspace(int64_t M, int D, float *&rec, int32_t *&dst) {
    int rc = ensure_counts_work(M, (size_t)D);
    if (rc) return rc;
    rec = reinterpret_cast<float *>(g_c.rec);
    dst = g_c.dst;
    return KGE_OK;
}

int float_records_reduce(int64_t M, int D, const FloatRowSpace &rs, hipStream_t stream, const float *rec_ext, int32_t *dst_ext, bool deterministic) {
    int rc;
    // records held by the caller: size the sort's work buffers for M, then let the launches below read the caller's arrays (the
    // pointers are passed by value at launch, so the workspace's own are put back before returning)
    struct Swap {
        uint32_t *rec; int32_t *dst; bool on;
        ~Swap() { if (on) { g_c.rec = rec; g_c.dst = dst; } }
    } swap = {g_c.rec, g_c.dst, false};
    if (rec_ext && dst_ext) {
        if ((rc = ensure_counts_work(M, (size_t)D))) return rc;
        swap.rec = g_c.rec; swap.dst = g_c.dst; swap.on = true;
        g_c.rec = reinterpret_cast<uint32_t *>(const_cast<float *>(rec_ext));
        g_c.dst = dst_ext;
    }
    const int rows = (int)rs.rows;
    const int32_t *n_valid_p = nullptr;
    const int rpb = (rows + NB - 1) / NB;
    if (rpb <= 8192 && !engine().counts_force_sort && !deterministic) {
        const int n_tiles = (int)((M + BTILE - 1) / BTILE);
        if (!g_c.bucket_start) {
            if ((rc = regrow(g_c.bucket_start, NB + 2, "counts bucket_start"))) return rc;
            if ((rc = regrow(g_c.tile_hist, 2 * (NB + 2), "counts bucket totals/cursors"))) return rc;
            if ((rc = hip_check(hipMemset(g_c.tile_hist, 0, sizeof(int32_t) * 2 * (NB + 2)), "zero bucket totals"))) return rc;
        }
        int32_t *totals = g_c.tile_hist, *cursor = g_c.tile_hist + (NB + 2);
        int2 *pairs = reinterpret_cast<int2 *>(g_c.pairs);
        launch_bkt_hist(n_tiles, (int)M, rpb, totals, nullptr, stream);
        launch_bkt_scatter(n_tiles, (int)M, rpb, totals, cursor, pairs, stream);
        launch_bkt_sort<false>(pairs, rpb, rows, totals, cursor, SegPlan(), stream);
        n_valid_p = g_c.bucket_start + NB;
    } else {
        int blocks = (int)((M + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(fix_keys_kernel, dim3(blocks), dim3(256), 0, stream, g_c.dst, (long long)M, rows);
        size_t tmp = g_c.sort_tmp_bytes;
        rc = hip_check(rocprim::radix_sort_pairs(g_c.sort_tmp, tmp, g_c.dst, g_c.dst_sorted, g_c.ids, g_c.ids_sorted, (size_t)M, 0,
                                                 bits_for_rows(rows), stream), "float records sort");
        if (rc) return rc;
        hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(64), 0, stream, g_c.dst_sorted, (int)M, rows, g_c.n_valid);
        n_valid_p = g_c.n_valid;
    }
    const float *rec = reinterpret_cast<const float *>(g_c.rec);
    if (deterministic) {
        const int64_t hub_virtual = rs.rows - rs.hub_base;          // copies x hub rows
        const int K = rs.hub_rows > 0 ? (int)(hub_virtual / rs.hub_rows) : 0;
        const size_t need = (size_t)hub_virtual * D;
        if (need > g_c.hub_sums_cap) {
            if ((rc = regrow(g_c.hub_sums, need, "hub copy sums"))) return rc;
            if ((rc = hip_check(hipMemsetAsync(g_c.hub_sums, 0, sizeof(float) * need, stream), "zero hub copy sums"))) return rc;
            g_c.hub_sums_cap = need;
        }
#define KGE_SEGR(LL, CC)                                                                                              \
    {                                                                                                                 \
        long long nb = (M + (256 / LL) - 1) / (256 / LL);                                                             \
        if (nb > 16384) nb = 16384;                                                                                   \
        hipLaunchKernelGGL((segsum_f32_runs_kernel<LL, CC>), dim3((unsigned)nb), dim3(256), 0, stream, rec, g_c.dst_sorted, \
                           g_c.ids_sorted, n_valid_p, rs, D, g_c.hub_sums);                                           \
    }
        if (D <= 16) KGE_SEGR(16, 1) else if (D <= 32) KGE_SEGR(16, 2) else if (D <= 64) KGE_SEGR(16, 4)
        else if (D <= 128) KGE_SEGR(32, 4) else if (D <= 256) KGE_SEGR(64, 4) else if (D <= 512) KGE_SEGR(64, 8) else KGE_SEGR(64, 16)
#undef KGE_SEGR
        if (K > 0) {
            long long nb = (rs.hub_rows * D + 255) / 256;
            if (nb > 4096) nb = 4096;
            hipLaunchKernelGGL(hub_fold_det_kernel, dim3((unsigned)nb), dim3(256), 0, stream, g_c.hub_sums, rs, D, K);
        }
        return hip_check(hipGetLastError(), "float records deterministic reduce launch");
    }
    // shorter chunks while the step is small: the per-team record loop is a chain of dependent loads
    const int chunk_len = M >= (int64_t(1) << 20) ? 64 : (M >= (int64_t(1) << 18) ? 32 : 16);
#define KGE_SEGF(LL, CC)                                                                                              \
    {                                                                                                                 \
        const long long chunks = (M + chunk_len - 1) / chunk_len;                                                     \
        const long long nb = (chunks + (256 / LL) - 1) / (256 / LL);                                                  \
        hipLaunchKernelGGL((segsum_f32_kernel<LL, CC>), dim3((unsigned)nb), dim3(256), 0, stream, rec, g_c.dst_sorted, \
                           g_c.ids_sorted, n_valid_p, rs, D, chunk_len);                                              \
    }
    if (D <= 16) KGE_SEGF(16, 1) else if (D <= 32) KGE_SEGF(16, 2) else if (D <= 64) KGE_SEGF(16, 4)
    else if (D <= 128) KGE_SEGF(32, 4) else if (D <= 256) KGE_SEGF(64, 4) else if (D <= 512) KGE_SEGF(64, 8) else KGE_SEGF(64, 16)
#undef KGE_SEGF
    return hip_check(hipGetLastError(), "float records reduce launch");
}

// ---- pair-count path, host side (see segsum_pairs_kernel) ----
bool pair_keys_sortable(int64_t ent_total, int64_t rel_total) {
    const int64_t rows = ent_total * rel_total;   // key = x*R + r in an int32 (the sentinel for "no record" is `rows`)
    return rows > 0 && rows < (int64_t(1) << 31) - 1;
}

int pair_records_workspace(int64_t M, int rd, uint32_t *&rec, int32_t *&dst, float2 *&aux) {
    int rc = ensure_counts_work(M, (size_t)rd);
    if (rc) return rc;
    if (M > g_c.pair_aux_cap) {
        if ((rc = regrow(g_c.pair_aux, (size_t)M, "pair records aux"))) return rc;
        g_c.pair_aux_cap = M;
    }
    rec = g_c.rec;
    dst = g_c.dst;
    aux = g_c.pair_aux;
    return KGE_OK;
}

int pair_records_reduce(int model, int64_t M, int64_t n_int8, int D, int rd, int64_t ent_total, int64_t rel_total, const float *const tables[4],
                        float *const grads[4], float unit, hipStream_t stream) {
    int rc;
    const int rows = (int)(ent_total * rel_total);
    const int rpb = (rows + NB - 1) / NB;
    const int32_t *n_valid_p;
    if (rpb <= 8192 && !engine().counts_force_sort) {
        // two-level counting sort: key spaces up to NB * 8192 = 4.2 M (entity, relation) pairs
        const int n_tiles = (int)((M + BTILE - 1) / BTILE);
        if (!g_c.bucket_start) {
            if ((rc = regrow(g_c.bucket_start, NB + 2, "counts bucket_start"))) return rc;
            if ((rc = regrow(g_c.tile_hist, 2 * (NB + 2), "counts bucket totals/cursors"))) return rc;
            if ((rc = hip_check(hipMemset(g_c.tile_hist, 0, sizeof(int32_t) * 2 * (NB + 2)), "zero bucket totals"))) return rc;
        }
        int32_t *totals = g_c.tile_hist, *cursor = g_c.tile_hist + (NB + 2);
        int2 *pairs = reinterpret_cast<int2 *>(g_c.pairs);
        launch_bkt_hist(n_tiles, (int)M, rpb, totals, nullptr, stream);
        launch_bkt_scatter(n_tiles, (int)M, rpb, totals, cursor, pairs, stream);
        launch_bkt_sort<false>(pairs, rpb, rows, totals, cursor, SegPlan(), stream);
        n_valid_p = g_c.bucket_start + NB;   // start of the trash bucket == number of live records
    } else {
        // larger key spaces (e.g. FB15k: 14 951 entities x 1 345 relations): rocPRIM's radix sort on the same keys
        int blocks = (int)((M + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(fix_keys_kernel, dim3(blocks), dim3(256), 0, stream, g_c.dst, (long long)M, rows);
        size_t tmp = g_c.sort_tmp_bytes;
        rc = hip_check(rocprim::radix_sort_pairs(g_c.sort_tmp, tmp, g_c.dst, g_c.dst_sorted, g_c.ids, g_c.ids_sorted, (size_t)M, 0,
                                                 bits_for_rows(rows), stream), "pair records sort");
        if (rc) return rc;
        hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(64), 0, stream, g_c.dst_sorted, (int)M, rows, g_c.n_valid);
        n_valid_p = g_c.n_valid;
    }
    PairRed pr;
    pr.ent = tables[0]; pr.ctx = tables[2]; pr.auxe = tables[3];
    pr.g_ent = grads[0]; pr.g_auxe = grads[3];
    pr.D = D; pr.R = (int)rel_total; pr.RD = rd; pr.unit = unit; pr.aux = g_c.pair_aux; pr.n_int8 = n_int8;
    pr.magic = rel_total == 1 ? 0xFFFFFFFFu : (unsigned)(((uint64_t(1) << 32) + (uint64_t)rel_total - 1) / (uint64_t)rel_total);
    if (model == KGE_TRANSH) {   // normalised normal vectors, once per step
        static float *ctxn = nullptr;
        static int64_t ctxn_cap = 0;
        if (rel_total * D > ctxn_cap) {
            if ((rc = regrow(ctxn, (size_t)(rel_total * D), "normalised normal vectors"))) return rc;
            ctxn_cap = rel_total * D;
        }
        int nb = (int)((rel_total + 3) / 4);
        if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(ctx_normalize_kernel, dim3(nb), dim3(256), 0, stream, tables[2], (int)rel_total, D, ctxn);
        pr.ctx = ctxn;
    }
    const long long chunks = (M + CHUNK - 1) / CHUNK;
    const long long nb = (chunks + 3) / 4;
    if (model == KGE_TRANSH)
        hipLaunchKernelGGL((segsum_pairs_kernel<KGE_TRANSH>), dim3((unsigned)nb), dim3(256), 0, stream, g_c.rec, g_c.dst_sorted, g_c.ids_sorted,
                           n_valid_p, pr);
    else
        hipLaunchKernelGGL((segsum_pairs_kernel<KGE_TRANSD>), dim3((unsigned)nb), dim3(256), 0, stream, g_c.rec, g_c.dst_sorted, g_c.ids_sorted,
                           n_valid_p, pr);
    return hip_check(hipGetLastError(), "pair records reduce launch");
}

namespace {

// SGD on listed rows from the compact image, natural layout (D % 4 == 0); bflag != nullptr: only flagged rows
template <int L, int C>
__global__ __launch_bounds__(256) void apply_rows_nat_kernel(FuseArgs fz, const int32_t *__restrict__ row_list, const int32_t *__restrict__ S,
                                                             const int32_t *__restrict__ n_rows_p, const int32_t *__restrict__ bflag, int D) {
    constexpr int TEAMS = 256 / L;
    constexpr int Q = (C + 3) / 4;
    const int lane = threadIdx.x % L;
    const int n = n_rows_p[0];
    for (long long i = (long long)blockIdx.x * TEAMS + threadIdx.x / L; i < n; i += (long long)gridDim.x * TEAMS) {
        if (bflag && !bflag[i]) continue;
        const long long row = row_list[i];
        int acc[C];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int e0 = 4 * (lane + L * q);
            int4 v = make_int4(0, 0, 0, 0);
            if (e0 < D) v = *reinterpret_cast<const int4 *>(S + i * D + e0);
            if (4 * q < C) acc[4 * q] = v.x;
            if (4 * q + 1 < C) acc[4 * q + 1] = v.y;
            if (4 * q + 2 < C) acc[4 * q + 2] = v.z;
            if (4 * q + 3 < C) acc[4 * q + 3] = v.w;
        }
        float *prow = row < fz.E ? fz.ent + row * D : fz.rel + (row - fz.E) * D;
        apply_row_nat<L, C>(acc, prow, D, lane, fz.unit, fz.lr);
    }
}

// stage 3.  optimizer: 0 = SGD (lr), 1 = Adam (lr = lr_t)
struct ApplyArgs {
    float *p, *m, *v;
    int32_t *S;
    float *resid;
    long long rows;
    int D;
    float unit, lr, b1, b2, eps;
    int adam;
    // two tables in one launch: rows [0,E) live in p / m / v / resid, rows [E, rows) in p2 / m2 / v2 / resid2
    // sparse-row form: S is the compact [n_rows, D] image of the rows listed in row_list (same row space);
    // SGD only, no residuals
    const int32_t *row_list, *n_rows;
    float *p2, *m2, *v2, *resid2;
    long long E;
    long long row_lo;   // dense form on a row RANGE [row_lo, rows): S points at the image of row_lo (a rank's reduce-scattered chunk)
    const int32_t *row_live;    // row-list form: listed row i is processed only where row_live[i] != 0 (null: every listed row).  The sharded
                                // lazy-Adam step lists ALL relation rows with their all-reduced counts: a relation no rank had a record for must not move
    const int2 *row_span;       // fused step: (first position, records) per key 2 * row + kind; entity rows whose two lists hold at most
    int span_cap;               // span_cap records each were updated by segapply_kernel itself and are skipped here; null otherwise
    float *inv_out;     // dense full-table form: the emit kernel's 1/|row| table ([E + R], row-space index), refreshed for every row rewritten
                        // here so that the next step needs no pre-pass
};

// One row's update from its summed counts s[] (team layout: lane l holds elements l, l+L, ...), residual gradient rs[] and --
// `have` -- its parameter / moment rows already in registers.  THE arithmetic of the count path's optimizer step: the dense apply
// kernel, the row-list (sparse / lazy-Adam) kernel and the fused segmented-sum-and-apply kernel all call it, so a row gets the
// same bits whichever of them handles it.  Returns false where SGD leaves an untouched row alone.
//   i : the row's index in the [(E + R), D] row space (image row, 1/|row| table entry);  Sp / rp : its image / residual rows to
//   re-zero (null: nothing to re-zero)
template <int L, int C, bool SPARSE>
__device__ __forceinline__ bool apply_row_update(const Team<L, C> &tm, const ApplyArgs &a, float *table, float *mt, float *vt, long long row,
                                                 long long i, const float (&s)[C], const float (&rs)[C], float (&x)[C], float (&m_old)[C],
                                                 float (&v_old)[C], bool have, int32_t *Sp, float *rp) {
    // HIP's __fmul_rn / __fadd_rn are plain `*` / `+` (__clang_hip_math.h): left to the optimizer they are contracted into fma
    // differently in each kernel this body is inlined into (seen: v of the fused kernel one ulp off the apply kernel's).  With
    // contraction off for this body every caller computes the same bits.
#pragma clang fp contract(off)
    float g[C];
    float touched = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) touched += (s[c] != 0.f || rs[c] != 0.f) ? 1.f : 0.f;
    touched = team_sum<L>(touched);
    if (touched == 0.f && !a.adam) return false;  // SGD leaves untouched rows alone; TF1 Adam moves every row
    if (!have) {
        tm.load(table, row, x);
        if (a.adam) { tm.load(mt, row, m_old); tm.load(vt, row, v_old); }
    }
    if (touched != 0.f) {
        float xn[C], inv; bool uc;
        tm.normalize(x, xn, inv, uc);
        float d = tm.dot(xn, s);
        if (!uc) d = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) g[c] = add_rn(count_grad(a.unit, inv, s[c], d, xn[c]), rs[c]);
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) g[c] = 0.f;
    }
    float *pp = table + row * a.D;
    float xnew[C];      // the row as it stands after this update
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int e = tm.lane + L * c;
        xnew[c] = 0.f;
        if (e >= a.D) continue;
        float pn = x[c];
        if (a.adam) {
            float *mp = mt + row * a.D + e, *vp = vt + row * a.D + e;
            float mi = mul_rn(m_old[c], a.b1), vi = mul_rn(v_old[c], a.b2);
            if (g[c] != 0.f) {
                mi = add_rn(mi, mul_rn(g[c], 1.0f - a.b1));
                vi = add_rn(vi, mul_rn(mul_rn(g[c], g[c]), 1.0f - a.b2));
            }
            *mp = mi; *vp = vi;
            pn = sub_rn(x[c], __fdiv_rn(mul_rn(a.lr, mi), add_rn(__fsqrt_rn(vi), a.eps)));
            pp[e] = pn;
        } else if (g[c] != 0.f) {
            pn = sub_rn(x[c], mul_rn(a.lr, g[c]));
            pp[e] = pn;
        }
        xnew[c] = pn;
        if (!SPARSE && touched != 0.f) { if (Sp) Sp[e] = 0; if (rp && rs[c] != 0.f) rp[e] = 0.f; }
    }
    if constexpr (!SPARSE) {
        if (a.inv_out) {       // (wave-uniform: a kernel argument)
            const float inv_new = row_inv_norm<L, C>(xnew);
            if (tm.lane == 0) a.inv_out[i] = inv_new;
        }
    }
    return true;
}

// n_own > 0: workgroups from n_own on run a part of the next batch's sampler (the launch that ends a single-process step takes
// what the bucketing launches left of it, see launch_bkt_hist)
template <int L, int C, bool SPARSE>
__global__ __launch_bounds__(256) void apply_counts_kernel(ApplyArgs a, SamplerArgs ride = SamplerArgs(), int n_own = 0) {
    constexpr int TEAMS = 256 / L;
    if (n_own > 0 && (int)blockIdx.x >= n_own) {
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - n_own, bern_lds);
        return;
    }
    const long long n_blocks = n_own > 0 ? n_own : gridDim.x;
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    const long long n_rows = SPARSE ? (long long)a.n_rows[0] : a.rows;
    for (long long i = a.row_lo + (long long)blockIdx.x * TEAMS + threadIdx.x / L; i < n_rows; i += n_blocks * TEAMS) {
        long long row = SPARSE ? (long long)a.row_list[i] : i;
        if (SPARSE && a.row_live && a.row_live[i] == 0) continue;
        // behind the fused segmented-sum-and-apply kernel: the entity rows it has already updated from registers are skipped
        if (!SPARSE && a.row_span && row < a.E && a.row_span[2 * row].y <= a.span_cap && a.row_span[2 * row + 1].y <= a.span_cap) continue;
        float *table = a.p, *mt = a.m, *vt = a.v, *rt = a.resid;
        if (row >= a.E) { row -= a.E; table = a.p2; mt = a.m2; vt = a.v2; rt = a.resid2; }
        int32_t *Sp = a.S + (i - a.row_lo) * a.D;
        float *rp = (SPARSE || !rt) ? nullptr : rt + row * a.D;
        float s[C], rs[C];
        float x[C], m_old[C], v_old[C];
        // TF1 Adam moves EVERY row: its parameter and moment rows are requested together with the counts, not after the
        // "touched?" reduction (one memory round trip per row instead of two)
        const bool every_row = !SPARSE && a.adam;
        if (every_row) {
            tm.load(table, row, x); tm.load(mt, row, m_old); tm.load(vt, row, v_old);
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int e = tm.lane + L * c;
            const int si = e < a.D ? Sp[e] : 0;
            rs[c] = (rp && e < a.D) ? rp[e] : 0.f;
            s[c] = (float)si;
        }
        apply_row_update<L, C, SPARSE>(tm, a, table, mt, vt, row, i, s, rs, x, m_old, v_old, every_row, SPARSE ? nullptr : Sp, rp);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused segmented sum + optimizer step (single process, dense tables): stages 2b and 3 in one kernel, without the count image
// for the rows whose records one team can hold.
//
// ONE TEAM PER ROW of the virtual row space (entities, then the virtual copies of the relation rows), told by bkt_sort_kernel<true>
// where the row's records lie in the sorted list (row_span).  The team requests the row's parameter / moment rows, sums the
// row's records in registers (natural layout: lane l holds elements 4l..4l+3) -- int8 records through v_dot4 with a one-hot
// selector (one instruction per element), the negatives' 2-bit records (FbArgs::rec2) by spreading the lane's byte into four
// byte-wide fields and adding them packed (five instructions per record, unpacked every <= 127 records) -- turns the sums into
// the team layout through LDS and calls apply_row_update: exactly what the apply kernel would have done with the same sums
// (bit-identical to the two-kernel path, which the data-parallel step keeps).  TF1 Adam moves rows without records too: their
// teams do that.  With a full wave per team (widths 132..1024) everything about WHICH records is wave-uniform: positions, record
// ids and base addresses live in scalar registers (scalar loads, scalar address arithmetic), the vector unit only adds.
//   * A row of more than `cap` records (hub entities) is cut into pieces by the planner; the pieces are further teams of this
//     launch that add into the int32 count image with atomics, and so does every relation row (virtual copies fold onto one row;
//     a hub relation has thousands of records).  apply_counts_kernel then runs over those rows only (row_span tells it which).
// What this saves against segsum_kernel + apply_counts_kernel: the image traffic of nearly all rows (written, read, re-zeroed:
// 35 MB of a bench step), the boundary atomics (a chunk of 64 sorted records there ends inside a row almost always), the key
// comparisons per record, and three quarters of the record bytes.
// ---------------------------------------------------------------------------------------------
struct SegApplyArgs {
    const uint32_t *rec;     // int8 records [n8][RD dwords]; the 2-bit records [*][RD bytes] follow at byte offset rec2_off
    const int32_t *ids;      // record ids in key order
    const int2 *row_span;    // [2 * n_rows_v] (first position, records) per key 2 * row + kind
    const int4 *pieces;      // (key, first position, records, 0)
    const int32_t *n_pieces;
    unsigned rec2_off;       // (all record bytes together stay below 4 GB: offsets are 32-bit, the host checks)
    int n8;                  // ids below this are int8 records
    int m_last;              // last valid position of `ids`
    int n_rows_v;            // virtual rows: E + krel * R
    int RD, cap;
    int32_t *S;              // count image
    int fold_R;              // relations: row >= ap.E is copy (row - E) / fold_R of relation (row - E) % fold_R
    int diag;                // measurement hook (option "counts_fused_diag"): 1 = no record loops, 2 = no row update (WRONG results)
    ApplyArgs ap;
};

template <int L, int C>
__global__ __launch_bounds__(256) void segapply_kernel(SegApplyArgs sa) {
    constexpr int TEAMS = 256 / L;
    constexpr int Q = C / 4;
    constexpr int U = 16;             // records in flight per team
    constexpr int J = 128 / L > 0 ? (128 / L > 4 ? 4 : 128 / L) : 1;   // id registers per list: a list holds at most J * L records (the planner's cap)
    static_assert(C % 4 == 0, "natural record layout: four elements per lane and chunk");
    __shared__ int32_t stage_all[256 * C];
    int32_t *stage = stage_all + (threadIdx.x / L) * (L * C);
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = sa.ap.D;
    const int lane = tm.lane;
    const ApplyArgs &ap = sa.ap;
    // a full wave per team: the team index is wave-uniform (told to the compiler, so that what follows is scalar code)
    auto uni = [](int v) { if constexpr (L == 64) return __builtin_amdgcn_readfirstlane(v); else return v; };
    const int team = uni((int)(blockIdx.x * TEAMS + threadIdx.x / L));
    int row, start8, n8, start2, n2;
    bool piece = false;
    // the longest-running teams first: the pieces of long rows (cap records each), then the relation rows (hundreds of records on a
    // graph with few relations), then the entity rows -- with the pieces at the END of the grid the launch had a tail of teams that
    // started last and ran longest (measured: 2 waves per SIMD resident on average)
    const int np = uni(sa.n_pieces[0]);
    if (team < np) {
        const int4 pc = sa.pieces[team];
        const int key = uni(pc.x);
        row = key >> 1;
        start8 = start2 = uni(pc.y);
        n8 = (key & 1) ? 0 : uni(pc.z);
        n2 = (key & 1) ? uni(pc.z) : 0;
        piece = true;
    } else {
        const int r = team - np, n_rel_v = sa.n_rows_v - (int)ap.E;
        if (r >= sa.n_rows_v) return;
        row = r < n_rel_v ? (int)ap.E + r : r - n_rel_v;
        const int4 sp = *reinterpret_cast<const int4 *>(sa.row_span + 2 * row);     // both lists of the row
        start8 = uni(sp.x); n8 = uni(sp.y); start2 = uni(sp.z); n2 = uni(sp.w);
        if (n8 > sa.cap || n2 > sa.cap) return;                                      // a long row: its pieces are teams of their own
        if (n8 + n2 == 0 && (row >= ap.E || (!ap.adam && !ap.resid))) return;        // nothing to do (TF1 Adam moves record-less rows too)
    }
    const bool direct = !piece && row < ap.E;
    float x[C], mo[C], vo[C];
    if (direct) {                      // requested now, needed after the records
        tm.load(ap.p, row, x);
        if (ap.adam) { tm.load(ap.m, row, mo); tm.load(ap.v, row, vo); }
    }
    if (sa.diag & 1) n8 = n2 = 0;
    // The record ids of both lists, requested at once with the parameter rows (lane l holds ids l, l + L, ... of a list), and
    // turned into 32-bit BYTE OFFSETS of the records right here, in the lanes: per record the loops below then spend one
    // v_readlane (the offset, into a scalar register), one add and the load -- no scalar multiply, no test, no branch.  (The
    // first version computed every record's address with ~14 scalar instructions and two branches: the kernel was bound by the
    // scalar unit, 75 us of instruction issue with the loads removed.)
    unsigned o8[J], o2[J];
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int i8 = sa.ids[min(start8 + min(lane + L * j, max(n8 - 1, 0)), sa.m_last)];     // (an empty list at the very end: stay inside the array)
        const int i2 = sa.ids[min(start2 + min(lane + L * j, max(n2 - 1, 0)), sa.m_last)];
        o8[j] = (unsigned)i8 * (unsigned)(4 * sa.RD);
        o2[j] = sa.rec2_off + (unsigned)(i2 - sa.n8) * (unsigned)sa.RD;
    }
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0;
    const char *recb = reinterpret_cast<const char *>(sa.rec);
    auto offset_at = [&](const unsigned (&o)[J], int i) -> unsigned {     // entry i of a list of offsets (i is uniform across the team)
        unsigned v = o[0];
#pragma unroll
        for (int j = 1; j < J; j++) v = (i / L == j) ? o[j] : v;
        if constexpr (L == 64) return (unsigned)__builtin_amdgcn_readlane((int)v, i % L); else return (unsigned)__shfl((int)v, i % L, L);
    };
    // ---- list 1: int8 records (the positives' h / t / r slots), lane l reads dword l (+ L q); v_dot4 with a one-hot selector adds
    // one element per instruction
    for (int i0 = 0; i0 < n8; i0 += U) {
        uint32_t w[U][Q];
        const int nb = min(U, n8 - i0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < nb) {
                const char *p = recb + offset_at(o8, i0 + u);
#pragma unroll
                for (int q = 0; q < Q; q++) w[u][q] = *reinterpret_cast<const uint32_t *>(p + 4 * (lane + L * q));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < nb) {
#pragma unroll
                for (int q = 0; q < Q; q++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[4 * q + j] = __builtin_amdgcn_sdot4((int)w[u][q], 1 << (8 * j), acc[4 * q + j], false);
                }
            }
        }
    }
    // ---- list 2: the negatives' 2-bit records, lane l reads BYTE l (+ L q): four fields (sign + 1); spread into four byte-wide
    // fields and added packed; a list holds at most 128 records (cap), so a field stays below 256 and is unpacked once
    uint32_t pk[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) pk[q] = 0u;
    for (int i0 = 0; i0 < n2; i0 += U) {
        uint32_t w[U][Q];
        const int nb = min(U, n2 - i0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < nb) {
                const uint8_t *p = reinterpret_cast<const uint8_t *>(recb) + offset_at(o2, i0 + u);
#pragma unroll
                for (int q = 0; q < Q; q++) w[u][q] = p[lane + L * q];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u < nb) {
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const uint32_t b = w[u][q];
                    pk[q] += (b | (b << 6) | (b << 12) | (b << 18)) & 0x03030303u;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
#pragma unroll
        for (int j = 0; j < 4; j++) acc[4 * q + j] += (int)((pk[q] >> (8 * j)) & 0xFFu) - n2;      // fields hold sum (sign + 1)
    }
    // natural -> team layout through this team's LDS patch (one wave: its LDS operations complete in order)
#pragma unroll
    for (int q = 0; q < Q; q++)
        *reinterpret_cast<int4 *>(stage + 4 * (lane + L * q)) = make_int4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    int si[C];
#pragma unroll
    for (int c = 0; c < C; c++) { const int e = lane + L * c; si[c] = e < ap.D ? stage[e] : 0; }
    if (sa.diag & 2) return;
    if (direct) {
        float sf[C], rs[C];
        float *rp = ap.resid ? ap.resid + (long long)row * ap.D : nullptr;
#pragma unroll
        for (int c = 0; c < C; c++) { const int e = lane + L * c; sf[c] = (float)si[c]; rs[c] = (rp && e < ap.D) ? rp[e] : 0.f; }
        apply_row_update<L, C, false>(tm, ap, ap.p, ap.m, ap.v, row, row, sf, rs, x, mo, vo, true, nullptr, rp);
    } else {
        const long long irow = row < ap.E ? row : ap.E + (row - ap.E) % sa.fold_R;
        int32_t *p = sa.S + irow * ap.D;
#pragma unroll
        for (int c = 0; c < C; c++) { const int e = lane + L * c; if (e < ap.D && si[c] != 0) atomicAdd(p + e, si[c]); }
    }
}

#define KGE_SHAPE_DISPATCH(D, CALL)                                      \
    if (D % 4 == 0 && D <= 64) { CALL(16, 4); }                          \
    else if (D <= 16) { CALL(16, 1); } else if (D <= 32) { CALL(16, 2); } \
    else if (D <= 64) { CALL(16, 4); } else if (D <= 128) { CALL(32, 4); } \
    else if (D <= 256) { CALL(64, 4); } else if (D <= 512) { CALL(64, 8); } \
    else { CALL(64, 16); }


}  // namespace

}  // namespace kge

using namespace kge;

extern "C" {

int kge_transe_counts_supported(const kge_model_desc *m, INT n_neg) {
    if (!m) return 0;
    return m->model == KGE_TRANSE && m->ent_dim == m->rel_dim && m->ent_dim <= 1024 && n_neg >= 1 && n_neg <= 63 &&
           m->ent_total + m->rel_total < (int64_t(1) << 30);
}

}  // extern "C"

namespace kge { namespace {
// optimizer half of the fused single-process step (kge_transe_train_step_counts); null = forward only (kge_transe_forward_counts)
struct FusedOpt {
    float *p[2], *m[2], *v[2];
    int adam;
    float lr, b1, b2, eps;
    bool done = false;     // set where the fused kernels ran; false on return = the caller applies the image itself
};
} }

static int forward_counts_impl(const kge_model_desc *m, const float *d_ent, const float *d_rel, const int32_t *d_h,
                               const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom,
                               int32_t *d_counts, float *d_resid_ent, float *d_resid_rel, float *d_loss, void *stream_, FusedOpt *fo) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_forward_counts: no usable HIP device");
    if (!m || !kge_transe_counts_supported(m, n_neg)) return fail(KGE_ERR_UNSUPPORTED, "sign-count path: TransE, dim <= 1024, 1..63 negatives");
    if (n_pos < 0 || stride < n_pos || denom <= 0) return fail(KGE_ERR_BAD_ARG, "kge_transe_forward_counts: bad sizes");
    if (n_pos == 0) return hip_check(hipMemsetAsync(d_loss, 0, sizeof(float), stream), "zero loss");
    int L, C;
    transe_team_shape(m->ent_dim, L, C);
    const size_t rd = (size_t)L * ((C + 3) / 4);
    const int64_t M = n_pos * (3 + n_neg);
    if (M >= (int64_t(1) << 31)) return fail(KGE_ERR_UNSUPPORTED, "batch too large for the sign-count path");
    int rc = ensure_counts_work(M, rd);
    if (rc) return rc;
    // Relation rows are hubs: every group sends one record to one of R rows (the Zipf head relation of the FB15k-237-shaped graph takes
    // a sixth of them), and in the bucketing kernels all those records count on a handful of LDS bins -- same-address LDS atomics
    // serialise, which set the tail of bkt_hist / bkt_scatter / bkt_sort (SQ_LDS_BANK_CONFLICT 64-92 % of their LDS cycles).  Group b
    // therefore addresses copy b % krel of the relation rows (virtual rows E + c R + r); the segmented sum folds them back.
    int krel = 1;
    if (n_pos >= 4096 && m->rel_total > 0 && engine().counts_krel > 1) {
        krel = engine().counts_krel;
        while (krel > 1 && (int64_t)krel * m->rel_total > 2 * (m->ent_total + m->rel_total)) krel >>= 1;   // copies stay a minor part of the row space
    }
    const int rows = (int)(m->ent_total + (int64_t)krel * m->rel_total);
    const int D = m->ent_dim;
    const bool nat = D % 4 == 0;   // the vectorised emit kernel writes records in natural element order
    const int rpb = (rows + NB - 1) / NB;
    // the fused segmented-sum-and-apply path: natural-layout records, the bucket sort (over the doubled key space 2 * row + kind,
    // whose buckets tell every row where its two record lists lie), all record bytes below 4 GB (32-bit offsets in the kernel)
    const int rows2 = 2 * rows, rpb2 = 2 * ((rows + NB - 1) / NB);
    const bool fused = fo && nat && rpb2 <= 4096 && !engine().counts_force_sort && engine().counts_fused &&
                       (uint64_t)M * rd * 4 < (uint64_t(1) << 32) && rows2 < (1 << 30);
    uint8_t *rec2 = fused ? reinterpret_cast<uint8_t *>(g_c.rec + (size_t)3 * (size_t)n_pos * rd) : nullptr;
    rc = launch_transe_emit(*m, d_ent, d_rel, d_resid_ent, d_resid_rel, d_h, d_t, d_r, n_pos, n_neg, stride, denom, g_c.rec, g_c.dst,
                            krel, d_loss, stream, d_resid_ent != nullptr && d_resid_rel != nullptr, rec2);
    if (rc) return rc;
    FuseArgs fold = FuseArgs();
    if (krel > 1) { fold.fold_E = (int)m->ent_total; fold.fold_R = (int)m->rel_total; }
    if (fused) {
        int L_, C_;
        transe_team_shape(D, L_, C_);
        const int cap = std::min(engine().counts_fused_cap > 0 ? engine().counts_fused_cap : 64, std::min(127, 4 * L_));   // (the kernel holds min(128, 4 L) ids per list; 127: the packed 2-bit sums stay below 256)
        const int64_t max_pieces = 3 * (M / cap + 1) + 16;     // sum of ceil(c / cap) over both lists of the rows with a list of more than cap records
        if (max_pieces > g_c.pieces_cap) {
            if ((rc = regrow(g_c.pieces, (size_t)max_pieces, "fused step piece table"))) return rc;
            g_c.pieces_cap = max_pieces;
        }
        if (!g_c.n_pieces) {
            if ((rc = regrow(g_c.n_pieces, 1, "fused step piece counter"))) return rc;
            if ((rc = hip_check(hipMemset(g_c.n_pieces, 0, sizeof(int32_t)), "zero piece counter"))) return rc;
        }
        if (rows2 > g_c.row_span_cap) {
            if ((rc = regrow(g_c.row_span, (size_t)rows2, "fused step row spans"))) return rc;
            g_c.row_span_cap = rows2;
        }
        const int n_tiles = (int)((M + BTILE - 1) / BTILE);
        if (!g_c.bucket_start) {
            if ((rc = regrow(g_c.bucket_start, NB + 2, "counts bucket_start"))) return rc;
            if ((rc = regrow(g_c.tile_hist, 2 * (NB + 2), "counts bucket totals/cursors"))) return rc;
            if ((rc = hip_check(hipMemset(g_c.tile_hist, 0, sizeof(int32_t) * 2 * (NB + 2)), "zero bucket totals"))) return rc;
        }
        int32_t *totals = g_c.tile_hist, *cursor = g_c.tile_hist + (NB + 2);
        int2 *pairs = reinterpret_cast<int2 *>(g_c.pairs);
        SegPlan plan;
        plan.row_span = g_c.row_span; plan.pieces = g_c.pieces; plan.n_pieces = g_c.n_pieces; plan.cap = cap;
        launch_bkt_hist(n_tiles, (int)M, rpb2, totals, g_c.n_pieces, stream);
        launch_bkt_scatter(n_tiles, (int)M, rpb2, totals, cursor, pairs, stream);
        launch_bkt_sort<true>(pairs, rpb2, rows2, totals, cursor, plan, stream);
        const long long all_rows = m->ent_total + m->rel_total;
        SegApplyArgs sa = {};
        sa.rec = g_c.rec; sa.rec2_off = (unsigned)((size_t)3 * (size_t)n_pos * rd * 4); sa.ids = g_c.ids_sorted; sa.row_span = g_c.row_span;
        sa.pieces = g_c.pieces; sa.n_pieces = g_c.n_pieces;
        sa.n8 = (int)(3 * n_pos); sa.m_last = (int)M - 1; sa.n_rows_v = rows; sa.RD = (int)rd; sa.cap = cap; sa.S = d_counts; sa.fold_R = (int)m->rel_total; sa.diag = engine().counts_fused_diag;
        ApplyArgs &a = sa.ap;
        a.p = fo->p[0]; a.p2 = fo->p[1]; a.resid = d_resid_ent; a.resid2 = d_resid_rel;
        if (fo->adam) { a.m = fo->m[0]; a.m2 = fo->m[1]; a.v = fo->v[0]; a.v2 = fo->v[1]; }
        a.S = d_counts; a.rows = all_rows; a.row_lo = 0; a.E = m->ent_total; a.D = D;
        a.unit = 1.0f / (float)denom; a.lr = fo->lr; a.b1 = fo->b1; a.b2 = fo->b2; a.eps = fo->eps; a.adam = fo->adam;
        a.row_span = g_c.row_span; a.span_cap = cap;
        {
            Engine &e = engine();     // (as kge_transe_apply_counts_range: the two kernels together rewrite every row's 1/|row| entry)
            const bool keeps = e.inv_carry && e.inv_valid && e.inv_norm && e.inv_for_ent == fo->p[0] && e.inv_for_rel == fo->p[1] && e.inv_cap >= all_rows;
            if (keeps) a.inv_out = e.inv_norm;
            else tables_written();
        }
#define KGE_SEGAPPLY(LL, CC)                                                                                          \
    {                                                                                                                 \
        const long long nb = ((long long)rows + max_pieces + (256 / LL) - 1) / (256 / LL);                            \
        hipLaunchKernelGGL((segapply_kernel<LL, CC>), dim3((unsigned)nb), dim3(256), 0, stream, sa);                   \
        long long nb2 = (all_rows + (256 / LL) - 1) / (256 / LL);                                                     \
        if (nb2 > 8192) nb2 = 8192;                                                                                   \
        SamplerArgs ride = {};                                                                                        \
        const unsigned n_ride = take_ride(ride, 3);                                                                   \
        hipLaunchKernelGGL((apply_counts_kernel<LL, CC, false>), dim3((unsigned)nb2 + n_ride), dim3(256), 0, stream, a, ride, (int)nb2); \
    }
        if (D <= 64) KGE_SEGAPPLY(16, 4) else if (D <= 128) KGE_SEGAPPLY(32, 4) else if (D <= 256) KGE_SEGAPPLY(64, 4)
        else if (D <= 512) KGE_SEGAPPLY(64, 8) else KGE_SEGAPPLY(64, 16)
#undef KGE_SEGAPPLY
        fo->done = true;
        return hip_check(hipGetLastError(), "fused counts step launch");
    }
    if (rpb <= 8192 && !engine().counts_force_sort) {
        // ---- two-level counting sort (hand-written) + segmented sum: row spaces up to NB*8192 rows ----
        const int n_tiles = (int)((M + BTILE - 1) / BTILE);
        if (!g_c.bucket_start) {
            if ((rc = regrow(g_c.bucket_start, NB + 2, "counts bucket_start"))) return rc;
            if ((rc = regrow(g_c.tile_hist, 2 * (NB + 2), "counts bucket totals/cursors"))) return rc;
            if ((rc = hip_check(hipMemset(g_c.tile_hist, 0, sizeof(int32_t) * 2 * (NB + 2)), "zero bucket totals"))) return rc;
        }
        int32_t *totals = g_c.tile_hist, *cursor = g_c.tile_hist + (NB + 2);
        int2 *pairs = reinterpret_cast<int2 *>(g_c.pairs);
        launch_bkt_hist(n_tiles, (int)M, rpb, totals, nullptr, stream);
        launch_bkt_scatter(n_tiles, (int)M, rpb, totals, cursor, pairs, stream);
        launch_bkt_sort<false>(pairs, rpb, rows, totals, cursor, SegPlan(), stream);
        const int32_t *n_valid_p = g_c.bucket_start + NB;   // start of the trash bucket == number of live records
#define KGE_SEG2(LL, CC)                                                                                              \
    {                                                                                                                 \
        const long long chunks = (M + CHUNK - 1) / CHUNK;                                                             \
        const long long nb = (chunks + (256 / LL) - 1) / (256 / LL);                                                  \
        if (nat) hipLaunchKernelGGL((segsum_kernel<LL, CC, true>), dim3((unsigned)nb), dim3(256), 0, stream, g_c.rec,  \
                                    g_c.dst_sorted, g_c.ids_sorted, n_valid_p, nullptr, d_counts, D, fold);           \
        else hipLaunchKernelGGL((segsum_kernel<LL, CC, false>), dim3((unsigned)nb), dim3(256), 0, stream, g_c.rec,     \
                                g_c.dst_sorted, g_c.ids_sorted, n_valid_p, nullptr, d_counts, D, fold);               \
    }
        KGE_SHAPE_DISPATCH(D, KGE_SEG2)
#undef KGE_SEG2
        return hip_check(hipGetLastError(), "counts bucket reduce launch");
    }
    // ---- general path: sort + segmented sum (large row spaces) ----
    int blocks = (int)((M + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fix_keys_kernel, dim3(blocks), dim3(256), 0, stream, g_c.dst, (long long)M, rows);
    size_t tmp = g_c.sort_tmp_bytes;
    rc = hip_check(rocprim::radix_sort_pairs(g_c.sort_tmp, tmp, g_c.dst, g_c.dst_sorted, g_c.ids, g_c.ids_sorted, (size_t)M, 0,
                                             bits_for_rows(rows), stream), "counts sort");
    if (rc) return rc;
    hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(64), 0, stream, g_c.dst_sorted, (int)M, rows, g_c.n_valid);
#define KGE_SEG(LL, CC)                                                                                               \
    {                                                                                                                 \
        const long long chunks = (M + CHUNK - 1) / CHUNK;                                                             \
        const long long nb = (chunks + (256 / LL) - 1) / (256 / LL);                                                  \
        if (nat) hipLaunchKernelGGL((segsum_kernel<LL, CC, true>), dim3((unsigned)nb), dim3(256), 0, stream, g_c.rec,  \
                                    g_c.dst_sorted, g_c.ids_sorted, g_c.n_valid, nullptr, d_counts, D, fold);         \
        else hipLaunchKernelGGL((segsum_kernel<LL, CC, false>), dim3((unsigned)nb), dim3(256), 0, stream, g_c.rec,     \
                                g_c.dst_sorted, g_c.ids_sorted, g_c.n_valid, nullptr, d_counts, D, fold);             \
    }
    KGE_SHAPE_DISPATCH(D, KGE_SEG)
#undef KGE_SEG
    return hip_check(hipGetLastError(), "counts reduce launch");
}

extern "C" {

int kge_transe_forward_counts(const kge_model_desc *m, const float *d_ent, const float *d_rel, const int32_t *d_h,
                              const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom,
                              int32_t *d_counts, float *d_resid_ent, float *d_resid_rel, float *d_loss, void *stream_) {
    return forward_counts_impl(m, d_ent, d_rel, d_h, d_t, d_r, n_pos, n_neg, stride, denom, d_counts, d_resid_ent, d_resid_rel, d_loss,
                               stream_, nullptr);
}

int kge_transe_train_step_counts(const kge_model_desc *m, float *const d_p[2], float *const d_m[2], float *const d_v[2], const int32_t *d_h,
                                 const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, int32_t *d_counts,
                                 float *const d_resid[2], int32_t sampler_shaped, int32_t adam, float lr, float beta1, float beta2, float eps,
                                 float *d_loss, void *stream_) {
    if (!m || !d_p || !d_p[0] || !d_p[1] || !d_counts || !d_resid || !d_resid[0] || !d_resid[1] || denom <= 0)
        return fail(KGE_ERR_BAD_ARG, "kge_transe_train_step_counts: bad arguments");
    if (adam && (!d_m || !d_v || !d_m[0] || !d_m[1] || !d_v[0] || !d_v[1]))
        return fail(KGE_ERR_BAD_ARG, "kge_transe_train_step_counts: Adam needs the moment tables");
    FusedOpt fo;
    fo.p[0] = d_p[0]; fo.p[1] = d_p[1];
    fo.m[0] = adam ? d_m[0] : nullptr; fo.m[1] = adam ? d_m[1] : nullptr; fo.v[0] = adam ? d_v[0] : nullptr; fo.v[1] = adam ? d_v[1] : nullptr;
    fo.adam = adam; fo.lr = lr; fo.b1 = beta1; fo.b2 = beta2; fo.eps = eps;
    const bool fed = !sampler_shaped;     // a hand-made batch: deferral bookkeeping + the exact fp32 pass into the residual tables
    int rc = forward_counts_impl(m, d_p[0], d_p[1], d_h, d_t, d_r, n_pos, n_neg, stride, denom, d_counts, fed ? d_resid[0] : nullptr,
                                 fed ? d_resid[1] : nullptr, d_loss, stream_, &fo);
    if (rc || fo.done) return rc;
    // (this width / table size has no fused kernels: the image is complete, the apply kernel finishes the step)
    return kge_transe_apply_counts_range(m, d_p, d_m, d_v, d_counts, d_resid, 0, m->ent_total + m->rel_total, denom, adam, lr, beta1, beta2,
                                         eps, stream_);
}

INT kge_transe_record_dwords(const kge_model_desc *m) {
    if (!m) return 0;
    int L, C;
    transe_team_shape(m->ent_dim, L, C);
    return (INT)L * ((C + 3) / 4);
}

int kge_transe_emit_records(const kge_model_desc *m, const float *d_ent, const float *d_rel, const int32_t *d_h, const int32_t *d_t,
                            const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, uint32_t *d_rec, int32_t *d_dst,
                            float *d_resid_ent, float *d_resid_rel, float *d_loss, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_emit_records: no usable HIP device");
    if (!m || !kge_transe_counts_supported(m, n_neg)) return fail(KGE_ERR_UNSUPPORTED, "sign-count path: TransE, dim <= 1024, 1..63 negatives");
    if (n_pos < 0 || stride < n_pos || denom <= 0 || !d_rec || !d_dst) return fail(KGE_ERR_BAD_ARG, "kge_transe_emit_records: bad arguments");
    if (n_pos == 0) return hip_check(hipMemsetAsync(d_loss, 0, sizeof(float), stream), "zero loss");
    return launch_transe_emit(*m, d_ent, d_rel, d_resid_ent, d_resid_rel, d_h, d_t, d_r, n_pos, n_neg, stride, denom, d_rec, d_dst, 1,
                              d_loss, stream, true);
}

int kge_transe_deferred_groups(int32_t *n_groups) {
    if (!n_groups) return fail(KGE_ERR_BAD_ARG, "kge_transe_deferred_groups: null output");
    return transe_deferred_groups(n_groups);
}

static int reduce_records_impl(const kge_model_desc *m, const uint32_t *d_rec, int32_t *d_dst, INT n_records, int32_t *d_rows,
                               int32_t *d_row_counts, int32_t *d_n_rows, const FuseArgs *fuse, hipStream_t stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_reduce_records: no usable HIP device");
    if (!m || !d_rec || !d_dst || !d_rows || !d_row_counts || !d_n_rows || n_records < 0 || n_records >= (INT(1) << 31))
        return fail(KGE_ERR_BAD_ARG, "kge_transe_reduce_records: bad arguments");
    if (n_records == 0) return hip_check(hipMemsetAsync(d_n_rows, 0, sizeof(int32_t), stream), "zero n_rows");
    int L, C;
    transe_team_shape(m->ent_dim, L, C);
    const int64_t M = n_records;
    int rc = ensure_counts_work(M, 0);   // the records are the caller's
    if (rc) return rc;
    const int rows = (int)(m->ent_total + m->rel_total);
    const int D = m->ent_dim;
    const bool nat = D % 4 == 0;
    int blocks = (int)((M + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    // order (row, record id) pairs by row: rocPRIM radix sort handles any row space
    hipLaunchKernelGGL(fix_keys_kernel, dim3(blocks), dim3(256), 0, stream, d_dst, (long long)M, rows);
    size_t tmp = g_c.sort_tmp_bytes;
    rc = hip_check(rocprim::radix_sort_pairs(g_c.sort_tmp, tmp, d_dst, g_c.dst_sorted, g_c.ids, g_c.ids_sorted, (size_t)M, 0,
                                             bits_for_rows(rows), stream), "records sort");
    if (rc) return rc;
    hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(64), 0, stream, g_c.dst_sorted, (int)M, rows, g_c.n_valid);
    // unique-row index of every record (flags -> inclusive scan), row list, boundary rows cleared
    int32_t *uidx = g_c.pairs;   // [M] scratch
    hipLaunchKernelGGL(run_flags_kernel, dim3(blocks), dim3(256), 0, stream, g_c.dst_sorted, g_c.n_valid, (int)M, uidx);
    size_t scan_bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, uidx, uidx, (size_t)M, rocprim::plus<int32_t>(), stream);
    if (scan_bytes > g_c.sort_tmp_bytes) {
        if (g_c.sort_tmp) (void)hipFree(g_c.sort_tmp);
        g_c.sort_tmp = nullptr;
        if ((rc = hip_check(hipMalloc(&g_c.sort_tmp, scan_bytes), "scan temp"))) return rc;
        g_c.sort_tmp_bytes = scan_bytes;
    }
    rc = hip_check(rocprim::inclusive_scan(g_c.sort_tmp, scan_bytes, uidx, uidx, (size_t)M, rocprim::plus<int32_t>(), stream), "run scan");
    if (rc) return rc;
    int32_t *bflag = nullptr;
    if (fuse) {
        if (!nat) return fail(KGE_ERR_UNSUPPORTED, "fused reduce+apply needs an embedding width that is a multiple of 4");
        if ((int64_t)g_c.bflag_cap < M) {
            if ((rc = regrow(g_c.bflag, (size_t)M, "boundary flags"))) return rc;
            g_c.bflag_cap = M;
        }
        bflag = g_c.bflag;
        if ((rc = hip_check(hipMemsetAsync(bflag, 0, sizeof(int32_t) * (size_t)M, stream), "zero boundary flags"))) return rc;
    }
    hipLaunchKernelGGL(unique_rows_kernel, dim3(blocks), dim3(256), 0, stream, g_c.dst_sorted, g_c.n_valid, uidx, d_rows, d_n_rows,
                       d_row_counts, D, bflag);
    {
        long long zb = ((M + CHUNK - 1) / CHUNK + 3) / 4;
        if (zb > 8192) zb = 8192;
        hipLaunchKernelGGL(zero_boundary_rows_kernel, dim3((unsigned)zb), dim3(256), 0, stream, uidx, g_c.n_valid, d_row_counts, D);
    }
    if (fuse) {
#define KGE_SEGFUSE(LL, CC)                                                                                           \
    {                                                                                                                 \
        const long long chunks = (M + CHUNK - 1) / CHUNK;                                                             \
        const long long nb = (chunks + (256 / LL) - 1) / (256 / LL);                                                  \
        hipLaunchKernelGGL((segsum_kernel<LL, CC, true, true>), dim3((unsigned)nb), dim3(256), 0, stream, d_rec,      \
                           g_c.dst_sorted, g_c.ids_sorted, g_c.n_valid, uidx, d_row_counts, D, *fuse);                \
        long long nb2 = (M + (256 / LL) - 1) / (256 / LL);                                                            \
        if (nb2 > 8192) nb2 = 8192;                                                                                   \
        hipLaunchKernelGGL((apply_rows_nat_kernel<LL, CC>), dim3((unsigned)nb2), dim3(256), 0, stream, *fuse, d_rows,  \
                           d_row_counts, d_n_rows, bflag, D);                                                         \
    }
        KGE_SHAPE_DISPATCH(D, KGE_SEGFUSE)
#undef KGE_SEGFUSE
        return hip_check(hipGetLastError(), "records reduce+apply launch");
    }
#define KGE_SEGC(LL, CC)                                                                                              \
    {                                                                                                                 \
        const long long chunks = (M + CHUNK - 1) / CHUNK;                                                             \
        const long long nb = (chunks + (256 / LL) - 1) / (256 / LL);                                                  \
        if (nat) hipLaunchKernelGGL((segsum_kernel<LL, CC, true>), dim3((unsigned)nb), dim3(256), 0, stream, d_rec,    \
                                    g_c.dst_sorted, g_c.ids_sorted, g_c.n_valid, uidx, d_row_counts, D);              \
        else hipLaunchKernelGGL((segsum_kernel<LL, CC, false>), dim3((unsigned)nb), dim3(256), 0, stream, d_rec,       \
                                g_c.dst_sorted, g_c.ids_sorted, g_c.n_valid, uidx, d_row_counts, D);                  \
    }
    KGE_SHAPE_DISPATCH(D, KGE_SEGC)
#undef KGE_SEGC
    return hip_check(hipGetLastError(), "records reduce launch");
}

int kge_transe_reduce_records(const kge_model_desc *m, const uint32_t *d_rec, int32_t *d_dst, INT n_records, int32_t *d_rows,
                              int32_t *d_row_counts, int32_t *d_n_rows, void *stream_) {
    return reduce_records_impl(m, d_rec, d_dst, n_records, d_rows, d_row_counts, d_n_rows, nullptr, (hipStream_t)stream_);
}

int kge_transe_reduce_apply_records_sgd(const kge_model_desc *m, const uint32_t *d_rec, int32_t *d_dst, INT n_records, float *d_ent,
                                        float *d_rel, int32_t *d_rows, int32_t *d_row_counts, int32_t *d_n_rows, INT denom, float lr,
                                        void *stream_) {
    if (!m || !d_ent || !d_rel || denom <= 0) return fail(KGE_ERR_BAD_ARG, "kge_transe_reduce_apply_records_sgd: bad arguments");
    tables_written();
    FuseArgs fz;
    fz.ent = d_ent; fz.rel = d_rel; fz.E = m->ent_total; fz.unit = 1.0f / (float)denom; fz.lr = lr;
    return reduce_records_impl(m, d_rec, d_dst, n_records, d_rows, d_row_counts, d_n_rows, &fz, (hipStream_t)stream_);
}

int kge_transe_apply_rows_sgd(const kge_model_desc *m, float *d_ent, float *d_rel, const int32_t *d_rows, const int32_t *d_row_counts,
                              const int32_t *d_n_rows, INT max_rows, INT denom, float lr, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_apply_rows_sgd: no usable HIP device");
    if (!m || !d_rows || !d_row_counts || !d_n_rows || denom <= 0) return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_rows_sgd: bad arguments");
    tables_written();
    if (max_rows <= 0) return KGE_OK;
    const int D = m->ent_dim;
    if (D % 4 == 0) {   // the arithmetic of the fused kernel: a row gets the same bits whichever kernel handles it
        FuseArgs fz;
        fz.ent = d_ent; fz.rel = d_rel; fz.E = m->ent_total; fz.unit = 1.0f / (float)denom; fz.lr = lr;
#define KGE_RNAT(LL, CC)                                                                                          \
    {                                                                                                             \
        long long nb = (max_rows + (256 / LL) - 1) / (256 / LL);                                                  \
        if (nb > 8192) nb = 8192;                                                                                 \
        hipLaunchKernelGGL((apply_rows_nat_kernel<LL, CC>), dim3((unsigned)nb), dim3(256), 0, stream, fz, d_rows, \
                           d_row_counts, d_n_rows, (const int32_t *)nullptr, D);                                  \
    }
        KGE_SHAPE_DISPATCH(D, KGE_RNAT)
#undef KGE_RNAT
        return hip_check(hipGetLastError(), "apply rows launch");
    }
    ApplyArgs a = {};
    a.p = d_ent; a.p2 = d_rel; a.row_list = d_rows; a.S = const_cast<int32_t *>(d_row_counts); a.n_rows = d_n_rows;
    a.E = m->ent_total; a.D = m->ent_dim; a.unit = 1.0f / (float)denom; a.lr = lr; a.adam = 0;
#define KGE_RAPPLY(LL, CC)                                                                                  \
    {                                                                                                       \
        long long nb = (max_rows + (256 / LL) - 1) / (256 / LL);                                            \
        if (nb > 8192) nb = 8192;                                                                           \
        hipLaunchKernelGGL((apply_counts_kernel<LL, CC, true>), dim3((unsigned)nb), dim3(256), 0, stream, a); \
    }
    KGE_SHAPE_DISPATCH(D, KGE_RAPPLY)
#undef KGE_RAPPLY
    return hip_check(hipGetLastError(), "apply rows launch");
}

static const int32_t *g_lazy_row_live = nullptr;     // kge_transe_lazy_row_live: per-listed-row switch for the NEXT lazy-Adam row-list call

int kge_transe_lazy_row_live(const int32_t *d_row_live) { g_lazy_row_live = d_row_live; return KGE_OK; }

int kge_transe_apply_rows_adam_lazy(const kge_model_desc *m, float *d_ent, float *d_rel, float *d_m_ent, float *d_m_rel, float *d_v_ent,
                                    float *d_v_rel, const int32_t *d_rows, const int32_t *d_row_counts, const int32_t *d_n_rows,
                                    INT max_rows, INT denom, float lr_t, float beta1, float beta2, float eps, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_apply_rows_adam_lazy: no usable HIP device");
    if (!m || !d_ent || !d_rel || !d_m_ent || !d_m_rel || !d_v_ent || !d_v_rel || !d_rows || !d_row_counts || !d_n_rows || denom <= 0)
        return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_rows_adam_lazy: bad arguments");
    tables_written();
    if (max_rows <= 0) return KGE_OK;
    ApplyArgs a = {};
    a.p = d_ent; a.p2 = d_rel; a.m = d_m_ent; a.m2 = d_m_rel; a.v = d_v_ent; a.v2 = d_v_rel;
    a.row_list = d_rows; a.S = const_cast<int32_t *>(d_row_counts); a.n_rows = d_n_rows;
    a.E = m->ent_total; a.D = m->ent_dim; a.unit = 1.0f / (float)denom; a.lr = lr_t; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.adam = 1;
    a.row_live = g_lazy_row_live; g_lazy_row_live = nullptr;      // (one call only)
    const int D = m->ent_dim;
#define KGE_RLAZY(LL, CC)                                                                                   \
    {                                                                                                       \
        long long nb = (max_rows + (256 / LL) - 1) / (256 / LL);                                            \
        if (nb > 8192) nb = 8192;                                                                           \
        hipLaunchKernelGGL((apply_counts_kernel<LL, CC, true>), dim3((unsigned)nb), dim3(256), 0, stream, a); \
    }
    KGE_SHAPE_DISPATCH(D, KGE_RLAZY)
#undef KGE_RLAZY
    return hip_check(hipGetLastError(), "lazy adam rows launch");
}

int kge_transe_apply_counts(float *d_p, float *d_m, float *d_v, int32_t *d_counts, float *d_resid, int64_t rows, int32_t dim,
                            INT denom, int32_t adam, float lr, float beta1, float beta2, float eps, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_apply_counts: no usable HIP device");
    if (rows <= 0) return KGE_OK;
    if (dim > 1024 || denom <= 0) return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_counts: bad sizes");
    tables_written();
    ApplyArgs a = {};
    a.p = d_p; a.m = d_m; a.v = d_v; a.S = d_counts; a.resid = d_resid; a.rows = rows; a.D = dim; a.E = rows;
    a.unit = 1.0f / (float)denom; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.adam = adam;
    const int D = dim;
#define KGE_APPLY(LL, CC)                                                                                   \
    {                                                                                                       \
        long long nb = (rows + (256 / LL) - 1) / (256 / LL);                                                \
        if (nb > 4096) nb = 4096;                                                                           \
        hipLaunchKernelGGL((apply_counts_kernel<LL, CC, false>), dim3((unsigned)nb), dim3(256), 0, stream, a); \
    }
    KGE_SHAPE_DISPATCH(D, KGE_APPLY)
#undef KGE_APPLY
    return hip_check(hipGetLastError(), "apply counts launch");
}

int kge_transe_apply_counts_range(const kge_model_desc *m, float *const d_p[2], float *const d_m[2], float *const d_v[2],
                                  int32_t *d_counts_chunk, float *const d_resid[2], INT row_lo, INT row_hi, INT denom, int32_t adam,
                                  float lr, float beta1, float beta2, float eps, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_transe_apply_counts: no usable HIP device");
    if (!m || !d_p || !d_p[0] || !d_p[1] || !d_counts_chunk || !d_resid || !d_resid[0] || !d_resid[1] || denom <= 0)
        return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_counts_tables/_range: bad arguments");
    if (adam && (!d_m || !d_v || !d_m[0] || !d_m[1] || !d_v[0] || !d_v[1]))
        return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_counts_tables/_range: Adam needs the moment tables");
    const long long all_rows = m->ent_total + m->rel_total;
    if (row_lo < 0 || row_hi > all_rows) return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_counts_range: row range outside the tables");
    const long long rows = row_hi - row_lo;
    if (rows <= 0) return KGE_OK;
    ApplyArgs a = {};
    a.p = d_p[0]; a.p2 = d_p[1]; a.resid = d_resid[0]; a.resid2 = d_resid[1];
    if (adam) { a.m = d_m[0]; a.m2 = d_m[1]; a.v = d_v[0]; a.v2 = d_v[1]; }
    a.S = d_counts_chunk; a.rows = row_hi; a.row_lo = row_lo; a.E = m->ent_total; a.D = m->ent_dim;
    a.unit = 1.0f / (float)denom; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.adam = adam;
    {
        // the emit kernel's 1/|row| table stays current only when THIS launch rewrites its entries: the whole row space of exactly
        // the tables it was computed for; any other update of those tables makes it stale
        Engine &e = engine();
        const bool keeps = e.inv_carry && e.inv_valid && e.inv_norm && row_lo == 0 && row_hi == all_rows && e.inv_for_ent == d_p[0] &&
                           e.inv_for_rel == d_p[1] && e.inv_cap >= all_rows;
        if (keeps) a.inv_out = e.inv_norm;
        else tables_written();
    }
    const int D = m->ent_dim;
#define KGE_APPLY2(LL, CC)                                                                                  \
    {                                                                                                       \
        long long nb = (rows + (256 / LL) - 1) / (256 / LL);                                                \
        if (nb > 8192) nb = 8192;                                                                           \
        SamplerArgs ride = {};                                                                              \
        const unsigned n_ride = take_ride(ride, 3);                                                         \
        hipLaunchKernelGGL((apply_counts_kernel<LL, CC, false>), dim3((unsigned)nb + n_ride), dim3(256), 0, stream, a, ride, (int)nb); \
    }
    KGE_SHAPE_DISPATCH(D, KGE_APPLY2)
#undef KGE_APPLY2
    return hip_check(hipGetLastError(), "apply counts launch");
}

int kge_transe_apply_counts_tables(const kge_model_desc *m, float *const d_p[2], float *const d_m[2], float *const d_v[2],
                                   int32_t *d_counts, float *const d_resid[2], INT denom, int32_t adam, float lr, float beta1,
                                   float beta2, float eps, void *stream_) {
    if (!m) return fail(KGE_ERR_BAD_ARG, "kge_transe_apply_counts_tables: bad arguments");
    return kge_transe_apply_counts_range(m, d_p, d_m, d_v, d_counts, d_resid, 0, m->ent_total + m->rel_total, denom, adam, lr, beta1,
                                         beta2, eps, stream_);
}

}  // extern "C"
