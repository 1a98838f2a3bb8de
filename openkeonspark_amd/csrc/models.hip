// Fused gather -> project -> L2-normalise -> L1 score -> margin-ranking hinge -> backward for the
// vector models (TransE / TransH / TransD).  Replaces the ~60 TensorFlow ops one
// `sess.run(train_op)` executes for TransE.py:26-51, TransH.py:33-69, TransD.py:46-84.
//
// Work decomposition (CDNA4, wave64): a TEAM of L lanes (16/32/64, chosen from the embedding
// width) owns one positive and all of its negatives.  Lane l of the team holds elements
// l, l+L, l+2L, ... of every vector, so a row is read and its gradient is added with consecutive
// lanes on consecutive dwords (64..256 contiguous bytes per wave instruction -- the shape float
// atomics need, MI355X_MICROARCH.md "Global float atomics").  Reductions over the embedding
// dimension are DPP / swizzle butterflies inside the team, no LDS traffic.
//
// What is NOT re-read or re-added: a negative produced by the sampler differs from its positive
// in exactly one slot (Base.cpp:118-139), so the positive's h, t, r (and the relation-side
// projection vectors) stay in registers for all n negatives and their gradient is accumulated in
// registers as a small-integer combination of sign vectors; only the one new row per negative
// is gathered and only its gradient row is scattered.  Per positive that is (3+n) rows read and
// (3+n) rows added for TransE instead of the reference's 3(1+n) + 3(1+n).
// Arbitrary (non sampler-shaped) batches are still handled exactly, through the standalone path.
//
// Gradients are ADDED into dense per-table accumulators (the deduplicated IndexedSlices sum that
// TF1 forms before the optimizer) with hardware fp32 atomics.
#include "models_dev.hpp"
#include "sampler_dev.hpp"

namespace kge {

// fixed-order sum of the per-block partial hinge sums -> loss = sum / denom  (TransE.py:51)
__global__ void loss_finalize_kernel(const float *partials, int n, float unit, float *out) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0] * unit;
}

// ------------------------------------------------------------------------------------------------
// TransE sign-count path, stage 1: same forward as fwdbwd_kernel<TRANSE>, but no atomics.
//
// For the L1 score every gradient w.r.t. a NORMALISED vector is (1/denom) x a small integer vector:
// -/+ sign(e_k) for the new row of an active negative, and cnt*sign(e_p) -/+ sum_k sign(e_k) for the
// positive's shared h, t, r (|value| <= 2N).  The kernel therefore emits one int8 RECORD per touched
// row (L*4*ceil(C/4) bytes, one coalesced dword store per lane) plus its destination row id; the sums
// per destination are exact integers whatever the order (stage 2), and the normalise-backward, which
// is linear in the upstream gradient, is applied once per ROW on the summed counts (stage 3).
// Record m = slot*n_pos + b, slot 0/1/2 = the positive's h/t/r, slot 3+k = negative k.
// Destination row space: entities [0,E), then relation rows spread over `krel` virtual copies
// E + (b % krel)*R + r so that a hub relation's records do not all land in one reducer.
// Negatives that are not sampler-shaped take the exact float standalone path into `g_ent/g_rel`
// (residual accumulators that stay all-zero otherwise).
// ------------------------------------------------------------------------------------------------
// --- four int8 lanes per 32-bit word: the integer gradient vectors live packed in registers -----
__device__ __forceinline__ uint32_t padd8(uint32_t x, uint32_t y) {  // per-byte add, no carry across bytes
    return ((x & 0x7F7F7F7Fu) + (y & 0x7F7F7F7Fu)) ^ ((x ^ y) & 0x80808080u);
}
__device__ __forceinline__ uint32_t pneg8(uint32_t x) { return padd8(~x, 0x01010101u); }  // per-byte two's complement
// bytes of s are in {-1,0,+1}: per-byte s * cnt, 0 <= cnt <= 127
__device__ __forceinline__ uint32_t pmul_sign8(uint32_t s, uint32_t cnt) {
    const uint32_t nz = s & 0x01010101u, neg = (s >> 7) & 0x01010101u;
    return (nz & ~neg) * cnt + neg * ((256u - cnt) & 0xFFu);
}

template <int L, int Q>
__device__ __forceinline__ void store_record(const FbArgs &a, int lane, long long m, const uint32_t (&w)[Q]) {
    uint32_t *p = a.rec + m * (long long)(L * Q);
#pragma unroll
    for (int q = 0; q < Q; q++) p[lane + L * q] = w[q];
}

template <int L, int C>
__global__ __launch_bounds__(256) void transe_emit_kernel(FbArgs a) {
    constexpr int TEAMS = 256 / L;
    constexpr int Q = (C + 3) / 4;
    constexpr int K = 4;  // negatives processed side by side: 4 row gathers in flight, 4 interleaved reductions
    __shared__ float red[TEAMS];
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    const int team_in_block = threadIdx.x / L;
    float lsum = 0.f;
    for (long long b = (long long)blockIdx.x * TEAMS + team_in_block; b < a.n_pos; b += (long long)gridDim.x * TEAMS) {
        const int h = a.bh[b], t = a.bt[b], r = a.br[b];
        {   // a group with ANY negative that is not sampler-shaped goes, whole, to the exact fp32 kernel
            float bad = 0.f;
            for (int k = tm.lane; k < (int)a.n_neg; k += L) {
                const long long j = b + (long long)(k + 1) * a.stride;
                if (!classify_negative<KGE_TRANSE>(h, t, r, a.bh[j], a.bt[j], a.br[j], a.negative_rel).fast) bad = 1.f;
            }
            if (team_sum<L>(bad) != 0.f) {
                if (tm.lane == 0) {
                    for (long long sl = 0; sl < 3 + a.n_neg; sl++) a.dst[sl * a.n_pos + b] = -1;
                    if (a.group_list) a.group_list[atomicAdd(a.group_count, 1)] = (int32_t)b;
                }
                continue;
            }
        }
        float hn[C], tn[C], rn[C];
        {
            float raw[C], inv; bool uc;
            tm.load(a.ent, h, raw); tm.normalize(raw, hn, inv, uc);
            tm.load(a.ent, t, raw); tm.normalize(raw, tn, inv, uc);
            tm.load(a.rel, r, raw); tm.normalize(raw, rn, inv, uc);
        }
        float p;
        uint32_t sp[Q];  // packed sign(e_p)
        {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < Q; q++) sp[q] = 0;
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float e = hn[c] + rn[c] - tn[c];
                acc += fabsf(e);
                sp[c / 4] |= (uint32_t)(((e > 0.f) - (e < 0.f)) & 0xFF) << (8 * (c % 4));
            }
            p = team_sum<L>(acc);
        }
        uint32_t Ah[Q], At[Q], Ar[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) { Ah[q] = 0; At[q] = 0; Ar[q] = 0; }
        int cnt = 0;
        for (int k0 = 0; k0 < (int)a.n_neg; k0 += L) {
            // ids of up to L negatives, one per lane, classified once (no per-negative scalar loads later)
            // code: 0 new head, 1 new tail, 2 new relation vector, 3 not sampler-shaped
            const int my_k = k0 + tm.lane;
            int my_code = 3, my_row = 0;
            if (my_k < (int)a.n_neg) {
                const long long j = b + (long long)(my_k + 1) * a.stride;
                const int nh = a.bh[j], nt = a.bt[j], nr = a.br[j];
                const NegClass nc = classify_negative<KGE_TRANSE>(h, t, r, nh, nt, nr, a.negative_rel);
                if (nc.fast) { my_code = !nc.same_h ? 0 : (!nc.same_t ? 1 : 2); my_row = !nc.same_h ? nh : (!nc.same_t ? nt : nr); }
            }
            const int in_round = min(L, (int)a.n_neg - k0);
            for (int kk = 0; kk < in_round; kk += K) {
                int code[K], row[K];
                float x[K][C];
#pragma unroll
                for (int u = 0; u < K; u++) {
                    const int src = min(kk + u, in_round - 1);
                    code[u] = team_bcast<L>(my_code, src);
                    row[u] = team_bcast<L>(my_row, src);
                }
#pragma unroll
                for (int u = 0; u < K; u++)   // unconditional gathers (a padded slot re-reads a valid row): all K in flight
                    tm.load(code[u] == 2 ? a.rel : a.ent, row[u], x[u]);
#pragma unroll
                for (int u = 0; u < K; u++) if (kk + u >= in_round) code[u] = -1;
                float ss[K];
#pragma unroll
                for (int u = 0; u < K; u++) {
                    float sq = 0.f;
                    if (code[u] >= 0) {
#pragma unroll
                        for (int c = 0; c < C; c++) sq += x[u][c] * x[u][c];
                    }
                    ss[u] = sq;
                }
#pragma unroll
                for (int u = 0; u < K; u++) ss[u] = team_sum<L>(ss[u]);
                float sc[K];
                uint32_t sg[K][Q];  // packed sign(e_k)
#pragma unroll
                for (int u = 0; u < K; u++) {
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < Q; q++) sg[u][q] = 0;
                    if (code[u] >= 0) {
                        const float inv = 1.0f / sqrtf(ss[u] >= 1e-12f ? ss[u] : 1e-12f);
#pragma unroll
                        for (int c = 0; c < C; c++) {
                            const float xn = x[u][c] * inv;
                            const float e = code[u] == 0 ? xn + rn[c] - tn[c] : (code[u] == 1 ? hn[c] + rn[c] - xn : hn[c] + xn - tn[c]);
                            acc += fabsf(e);
                            sg[u][c / 4] |= (uint32_t)(((e > 0.f) - (e < 0.f)) & 0xFF) << (8 * (c % 4));
                        }
                    }
                    sc[u] = acc;
                }
#pragma unroll
                for (int u = 0; u < K; u++) sc[u] = team_sum<L>(sc[u]);
#pragma unroll
                for (int u = 0; u < K; u++) {
                    if (code[u] < 0) continue;
                    const long long m = (3 + k0 + kk + u) * a.n_pos + b;
                    const float v = p - sc[u] + a.margin;
                    long long dest = -1;
                    if (v >= 0.f) {
                        cnt++; lsum += v;
                        uint32_t rec[Q];
                        if (code[u] == 0) {         // new head: dL/dx^ = -unit*s ; kept t gets +s, r gets -s
#pragma unroll
                            for (int q = 0; q < Q; q++) { const uint32_t ns = pneg8(sg[u][q]); rec[q] = ns; At[q] = padd8(At[q], sg[u][q]); Ar[q] = padd8(Ar[q], ns); }
                            dest = row[u];
                        } else if (code[u] == 1) {  // new tail: +unit*s ; kept h gets -s, r gets -s
#pragma unroll
                            for (int q = 0; q < Q; q++) { const uint32_t ns = pneg8(sg[u][q]); rec[q] = sg[u][q]; Ah[q] = padd8(Ah[q], ns); Ar[q] = padd8(Ar[q], ns); }
                            dest = row[u];
                        } else {                    // new relation vector: -unit*s ; h gets -s, t gets +s
#pragma unroll
                            for (int q = 0; q < Q; q++) { const uint32_t ns = pneg8(sg[u][q]); rec[q] = ns; Ah[q] = padd8(Ah[q], ns); At[q] = padd8(At[q], sg[u][q]); }
                            dest = (long long)a.ent_total + (long long)((int)b & (a.krel - 1)) * a.rel_total + row[u];
                        }
                        store_record<L, Q>(a, tm.lane, m, rec);
                    }
                    if (tm.lane == 0) a.dst[m] = (int32_t)dest;
                }
            }
        }
        if (cnt > 0) {
            uint32_t rh[Q], rt[Q], rr[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const uint32_t sv = pmul_sign8(sp[q], (uint32_t)cnt);
                rh[q] = padd8(Ah[q], sv); rt[q] = padd8(At[q], pneg8(sv)); rr[q] = padd8(Ar[q], sv);
            }
            store_record<L, Q>(a, tm.lane, b, rh);
            store_record<L, Q>(a, tm.lane, a.n_pos + b, rt);
            store_record<L, Q>(a, tm.lane, 2 * a.n_pos + b, rr);
        }
        if (tm.lane == 0) {
            a.dst[b] = cnt > 0 ? (int32_t)h : -1;
            a.dst[a.n_pos + b] = cnt > 0 ? (int32_t)t : -1;
            a.dst[2 * a.n_pos + b] = cnt > 0 ? (int32_t)(a.ent_total + ((int)b & (a.krel - 1)) * a.rel_total + r) : -1;
        }
    }
    finish_loss<TEAMS>(a, red, lsum, tm.lane, team_in_block);
}

// ---- vectorised, instruction-lean variant (D % 4 == 0) ----------------------------------------------
// Profiling the first version showed the emit kernel ISSUE-bound (~300 VALU instructions per negative:
// three-way selects per element, compare/select sign extraction, IEEE sqrt+divide, byte-wise adds),
// not memory-bound: skipping its stores or gathering only hot rows did not change its time.  This
// version keeps the per-negative work to ~50 instructions:
//   * 1/|row| comes from a per-row table refreshed once per step (row_inv_norm_kernel): no sum of
//     squares, no rsqrt, and only ONE team reduction per negative (the L1 score);
//   * e = f*x + B with B = (r^-t^), (h^+r^) or (h^-t^) precomputed per group and f = +-1/|x|:
//     one fma per element;
//   * sign(e) = med3(bits(e), -1, 1) on the integer pattern of the float (one instruction/element;
//     -0.0 would read as -1, so the per-group constants are cleared of negative zeros, after which
//     e = fma(f, x, B) can never be -0.0: tests/test_gpu_models.py::test_negative_zero_has_sign_zero);
//   * the integer gradient vectors are packed int16 pairs (v_pk_add_i16), bytes only when stored;
//   * one global_load_dwordx4 per row chunk.  Record dword w = lane + L*q holds elements 4w..4w+3
//     ("natural" layout, flagged to the reducers).
template <int L, int Q, int K, int WPE, bool INV_TAB, bool REC2 = false>
__global__ __launch_bounds__(256, WPE) void transe_emit_vec_kernel(FbArgs a) {
    constexpr int TEAMS = 256 / L;
    __shared__ float red[TEAMS];
    const int lane = threadIdx.x % L;
    const int team_in_block = threadIdx.x / L;
    const int D = a.D;
    const float *inv_ent = a.inv_norm, *inv_rel = a.inv_norm + a.ent_total;
    bool valid[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) valid[q] = 4 * (lane + L * q) < D;
    auto load4 = [&](const float *__restrict__ tab, long long row, float4 (&x)[Q]) {
        const float *p = tab + row * D;
#pragma unroll
        for (int q = 0; q < Q; q++)
            x[q] = valid[q] ? *reinterpret_cast<const float4 *>(p + 4 * (lane + L * q)) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float lsum = 0.f;
    for (long long b = (long long)blockIdx.x * TEAMS + team_in_block; b < a.n_pos; b += (long long)gridDim.x * TEAMS) {
        const int h = a.bh[b], t = a.bt[b], r = a.br[b];
        const int rel_row0 = a.ent_total + ((int)b & (a.krel - 1)) * a.rel_total;   // this group's virtual copy of the relation rows (krel: a power of two)
        // ---- negatives' ids: one per lane (first round), classified once ----
        int my_code = 0, my_row = 0;
        float my_f = 0.f;
        float bad = 0.f;
        for (int k = lane; k < (int)a.n_neg; k += L) {
            const long long j = b + (long long)(k + 1) * a.stride;
            const int nh = a.bh[j], nt = a.bt[j], nr = a.br[j];
            const NegClass nc = classify_negative<KGE_TRANSE>(h, t, r, nh, nt, nr, a.negative_rel);
            if (!nc.fast) bad = 1.f;
            if (k < L) {   // keep round 0 (the only round when n_neg <= L)
                my_code = !nc.same_h ? 0 : (!nc.same_t ? 1 : 2);
                my_row = !nc.same_h ? nh : (!nc.same_t ? nt : nr);
            }
        }
        if (team_sum<L>(bad) != 0.f) {   // not sampler-shaped: the whole group goes to the exact fp32 kernel
            if (lane == 0) {
                for (long long sl = 0; sl < 3 + a.n_neg; sl++) a.dst[sl * a.n_pos + b] = -1;
                if (a.group_list) a.group_list[atomicAdd(a.group_count, 1)] = (int32_t)b;
            }
            continue;
        }
        // f = +1/|x| for a new head or relation vector, -1/|x| for a new tail (e = f*x + B).  With the
        // per-row table (small tables) it is known before the row arrives; without it (tables too large to
        // sweep every step) it is computed from the gathered row, one extra team reduction per negative.
        if constexpr (INV_TAB) {
            if (lane < (int)a.n_neg) { const float iv = my_code == 2 ? inv_rel[my_row] : inv_ent[my_row]; my_f = my_code == 1 ? -iv : iv; }
        }
        // ---- the positive: B0 = r^-t^, B1 = h^+r^, B2 = h^-t^ ----
        float4 B0[Q], B1[Q], B2[Q];
        float p;
        s16x2 sp_lo[Q], sp_hi[Q];
        {
            float4 hn[Q], tn[Q], rn[Q];
            load4(a.ent, h, hn); load4(a.ent, t, tn); load4(a.rel, r, rn);
            float ih, it, ir;
            if constexpr (INV_TAB) { ih = inv_ent[h]; it = inv_ent[t]; ir = inv_rel[r]; }
            else {
                auto ssq = [&](const float4 (&x)[Q]) { float q2 = 0.f;
#pragma unroll
                    for (int q = 0; q < Q; q++) q2 += x[q].x * x[q].x + x[q].y * x[q].y + x[q].z * x[q].z + x[q].w * x[q].w;
                    return q2; };
                float sh = team_sum<L>(ssq(hn)), st = team_sum<L>(ssq(tn)), sr = team_sum<L>(ssq(rn));
                ih = 1.0f / sqrtf(sh >= 1e-12f ? sh : 1e-12f); it = 1.0f / sqrtf(st >= 1e-12f ? st : 1e-12f);
                ir = 1.0f / sqrtf(sr >= 1e-12f ? sr : 1e-12f);
            }
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const float4 H = make_float4(hn[q].x * ih, hn[q].y * ih, hn[q].z * ih, hn[q].w * ih);
                const float4 T = make_float4(tn[q].x * it, tn[q].y * it, tn[q].z * it, tn[q].w * it);
                const float4 R = make_float4(rn[q].x * ir, rn[q].y * ir, rn[q].z * ir, rn[q].w * ir);
                // "+ 0.0f" turns a -0.0 into +0.0 (sign_of_bits reads the sign BIT; tf.sign(-0.0) is 0).  With B free of
                // negative zeros, e = fma(f, x, B) cannot be -0.0 either: an exact cancellation rounds to +0.0, and
                // (-0.0) + (+0.0) = +0.0.  Once per group, not per negative.
                B0[q] = make_float4(R.x - T.x + 0.0f, R.y - T.y + 0.0f, R.z - T.z + 0.0f, R.w - T.w + 0.0f);
                B1[q] = make_float4(H.x + R.x + 0.0f, H.y + R.y + 0.0f, H.z + R.z + 0.0f, H.w + R.w + 0.0f);
                B2[q] = make_float4(H.x - T.x + 0.0f, H.y - T.y + 0.0f, H.z - T.z + 0.0f, H.w - T.w + 0.0f);
                const float e0 = H.x + R.x - T.x + 0.0f, e1 = H.y + R.y - T.y + 0.0f, e2 = H.z + R.z - T.z + 0.0f, e3 = H.w + R.w - T.w + 0.0f;
                acc += fabsf(e0) + fabsf(e1) + fabsf(e2) + fabsf(e3);
                sp_lo[q] = pack16(sign_of_bits(e0), sign_of_bits(e1));
                sp_hi[q] = pack16(sign_of_bits(e2), sign_of_bits(e3));
            }
            p = team_sum<L>(acc);
        }
        s16x2 Ah_lo[Q], Ah_hi[Q], At_lo[Q], At_hi[Q], Ar_lo[Q], Ar_hi[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) { Ah_lo[q] = 0; Ah_hi[q] = 0; At_lo[q] = 0; At_hi[q] = 0; Ar_lo[q] = 0; Ar_hi[q] = 0; }
        int cnt = 0;
        for (int k0 = 0; k0 < (int)a.n_neg; k0 += L) {
            if (k0 > 0) {   // later rounds (n_neg > L): fetch and classify this round's ids
                const int my_k = k0 + lane;
                my_code = 0; my_row = 0; my_f = 0.f;
                if (my_k < (int)a.n_neg) {
                    const long long j = b + (long long)(my_k + 1) * a.stride;
                    const int nh = a.bh[j], nt = a.bt[j], nr = a.br[j];
                    const NegClass nc = classify_negative<KGE_TRANSE>(h, t, r, nh, nt, nr, a.negative_rel);
                    my_code = !nc.same_h ? 0 : (!nc.same_t ? 1 : 2);
                    my_row = !nc.same_h ? nh : (!nc.same_t ? nt : nr);
                    if constexpr (INV_TAB) {
                        const float iv = my_code == 2 ? inv_rel[my_row] : inv_ent[my_row];
                        my_f = my_code == 1 ? -iv : iv;
                    }
                }
            }
            const int in_round = min(L, (int)a.n_neg - k0);
            int my_dst = -1;
            for (int kk = 0; kk < in_round; kk += K) {
                int code[K], row[K];
                float f[K];
                float4 x[K][Q];
#pragma unroll
                for (int u = 0; u < K; u++) {
                    const int src = min(kk + u, in_round - 1);
                    code[u] = team_bcast<L>(my_code, src);
                    row[u] = team_bcast<L>(my_row, src);
                    f[u] = __builtin_bit_cast(float, team_bcast<L>(__builtin_bit_cast(int, my_f), src));
                }
#pragma unroll
                for (int u = 0; u < K; u++) load4(code[u] == 2 ? a.rel : a.ent, row[u], x[u]);  // K gathers in flight
                if constexpr (!INV_TAB) {
                    float ss[K];
#pragma unroll
                    for (int u = 0; u < K; u++) {
                        float q2 = 0.f;
#pragma unroll
                        for (int q = 0; q < Q; q++) q2 += x[u][q].x * x[u][q].x + x[u][q].y * x[u][q].y + x[u][q].z * x[u][q].z + x[u][q].w * x[u][q].w;
                        ss[u] = q2;
                    }
#pragma unroll
                    for (int u = 0; u < K; u++) {
                        ss[u] = team_sum<L>(ss[u]);
                        const float iv = 1.0f / sqrtf(ss[u] >= 1e-12f ? ss[u] : 1e-12f);
                        f[u] = code[u] == 1 ? -iv : iv;
                    }
                }
                float sc[K];
#pragma unroll
                for (int u = 0; u < K; u++) {
                    float acc = 0.f;
                    auto apply = [&](const float4 (&Bs)[Q]) {
#pragma unroll
                        for (int q = 0; q < Q; q++) {
                            x[u][q].x = fmaf(f[u], x[u][q].x, Bs[q].x); x[u][q].y = fmaf(f[u], x[u][q].y, Bs[q].y);
                            x[u][q].z = fmaf(f[u], x[u][q].z, Bs[q].z); x[u][q].w = fmaf(f[u], x[u][q].w, Bs[q].w);
                            acc += fabsf(x[u][q].x) + fabsf(x[u][q].y) + fabsf(x[u][q].z) + fabsf(x[u][q].w);
                        }
                    };
                    if (code[u] == 0) apply(B0); else if (code[u] == 1) apply(B1); else apply(B2);
                    sc[u] = acc;
                }
#pragma unroll
                for (int u = 0; u < K; u++) sc[u] = team_sum<L>(sc[u]);
#pragma unroll
                for (int u = 0; u < K; u++) {
                    if (kk + u >= in_round) continue;
                    const float v = p - sc[u] + a.margin;
                    if (v >= 0.f) {
                        cnt++; lsum += v;
                        const long long m = (long long)(3 + k0 + kk + u) * a.n_pos + b;
                        uint32_t rec[Q];
                        // new head (0): dL/dx^ = -s, kept t gets +s, r gets -s;  new tail (1): +s, kept h gets -s, r gets -s;
                        // new relation vector (2): -s, h gets -s, t gets +s.  Branch-free: the three cases differ only in small
                        // integer multipliers (uniform per negative), so the accumulators take packed multiply-adds instead of
                        // three code paths whose merges cost a register move per accumulator
                        const int c = code[u];
                        const s16x2 kh = pack16(c == 0 ? 0 : -1, c == 0 ? 0 : -1), kt = pack16(c == 1 ? 0 : 1, c == 1 ? 0 : 1);
                        const s16x2 kr = pack16(c == 2 ? 0 : -1, c == 2 ? 0 : -1), kx = pack16(c == 1 ? 1 : -1, c == 1 ? 1 : -1);
#pragma unroll
                        for (int q = 0; q < Q; q++) {
                            const s16x2 s_lo = pack16(sign_of_bits(x[u][q].x), sign_of_bits(x[u][q].y));
                            const s16x2 s_hi = pack16(sign_of_bits(x[u][q].z), sign_of_bits(x[u][q].w));
                            rec[q] = bytes_of(s_lo * kx, s_hi * kx);
                            Ah_lo[q] += s_lo * kh; Ah_hi[q] += s_hi * kh;
                            At_lo[q] += s_lo * kt; At_hi[q] += s_hi * kt;
                            Ar_lo[q] += s_lo * kr; Ar_hi[q] += s_hi * kr;
                        }
                        if constexpr (REC2) {
                            // bytes {-1, 0, +1} -> 2-bit fields (value + 1) of ONE byte per lane: low two bits of every byte, + 1,
                            // then the four fields gathered into the top byte by a carry-free multiply (2-bit fields at bits 0, 8, 16,
                            // 24 times 2^24 + 2^18 + 2^12 + 2^6 land at bits 24, 26, 28, 30; every other partial product falls on a
                            // bit position of its own below 24 or beyond 31)
                            uint8_t *p2 = a.rec2 + ((long long)(k0 + kk + u) * a.n_pos + b) * (long long)(L * Q);
#pragma unroll
                            for (int q = 0; q < Q; q++) {
                                const uint32_t f = ((rec[q] & 0x03030303u) + 0x01010101u) & 0x03030303u;
                                if (valid[q]) p2[lane + L * q] = (uint8_t)((f * 0x01041040u) >> 24);
                            }
                        } else {
                            store_record<L, Q>(a, lane, m, rec);
                        }
                        // (REC2: the destination key is 2 * row + kind -- a row's int8 records and its 2-bit records form two lists)
                        if (lane == kk + u) my_dst = REC2 ? 2 * (code[u] == 2 ? rel_row0 + row[u] : row[u]) + 1 : (code[u] == 2 ? rel_row0 + row[u] : row[u]);
                    }
                }
            }
            // destinations of this round's negatives: one store instruction for the whole round
            if (k0 + lane < (int)a.n_neg) a.dst[(long long)(3 + k0 + lane) * a.n_pos + b] = my_dst;
        }
        if (cnt > 0) {
            uint32_t rh[Q], rt[Q], rr[Q];
            const s16x2 c2 = pack16(cnt, cnt);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const s16x2 v_lo = sp_lo[q] * c2, v_hi = sp_hi[q] * c2;
                rh[q] = bytes_of(Ah_lo[q] + v_lo, Ah_hi[q] + v_hi);
                rt[q] = bytes_of(At_lo[q] - v_lo, At_hi[q] - v_hi);
                rr[q] = bytes_of(Ar_lo[q] + v_lo, Ar_hi[q] + v_hi);
            }
            store_record<L, Q>(a, lane, b, rh);
            store_record<L, Q>(a, lane, a.n_pos + b, rt);
            store_record<L, Q>(a, lane, 2 * a.n_pos + b, rr);
        }
        if (lane == 0) {
            constexpr int KM = REC2 ? 2 : 1;
            a.dst[b] = cnt > 0 ? (int32_t)(KM * h) : -1;
            a.dst[a.n_pos + b] = cnt > 0 ? (int32_t)(KM * t) : -1;
            a.dst[2 * a.n_pos + b] = cnt > 0 ? (int32_t)(KM * (rel_row0 + r)) : -1;
        }
    }
    finish_loss<TEAMS>(a, red, lsum, lane, team_in_block);
}

// 1/max(|row|,1e-6): tf.nn.l2_normalize's rsqrt(max(sum x^2, 1e-12)) for every row of the two tables
// The table's pre-pass, in the team shape of the apply kernel for this width (transe_team_shape) so that both produce the same bits
template <int L, int C>
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float *__restrict__ ent_, const float *__restrict__ rel_, long long E, long long R,
                                                           int D, float *__restrict__ out) {
    constexpr int TEAMS = 256 / L;
    const int lane = threadIdx.x % L;
    for (long long row = (long long)blockIdx.x * TEAMS + threadIdx.x / L; row < E + R; row += (long long)gridDim.x * TEAMS) {
        float x[C];
        const float *p = row < E ? ent_ + row * D : rel_ + (row - E) * D;
#pragma unroll
        for (int c = 0; c < C; c++) { const int e = lane + L * c; x[c] = e < D ? p[e] : 0.f; }
        const float inv = row_inv_norm<L, C>(x);
        if (lane == 0) out[row] = inv;
    }
}

// lets another stream start behind the emit kernel just launched (the next batch's sampler, Config.prefetch_sampling)
static void record_emit_done(hipStream_t stream) {
    Engine &eng = engine();
    if (!eng.record_emit_event) return;
    if (!eng.emit_done) (void)hipEventCreateWithFlags(&eng.emit_done, hipEventDisableTiming);
    (void)hipEventRecord(eng.emit_done, stream);
    eng.emit_seq++;
}

constexpr int kDeferBlocks = 128;

template <int L, int C>
static void launch_emit(const FbArgs &a_in, float *d_loss, hipStream_t stream) {
    constexpr int TEAMS = 256 / L;
    FbArgs a = a_in;
    const bool defer_pass = a.g_ent && a.g_rel;   // residual accumulators given: deferred groups get the fp32 pass
    guard_loss_stream(stream);
    if (!defer_pass) { a.loss_out = d_loss; a.loss_ticket = engine().dev.loss_ticket; }   // the emit kernel's last block writes the loss
    long long blocks = (a.n_pos + TEAMS - 1) / TEAMS;
    if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
    if (blocks < 1) blocks = 1;
    if (a.D % 4 == 0) {
        // the per-row inverse-norm table costs one sweep of both tables per step: only while they are
        // cache-sized (FB15k-237 x 200: 11.8 MB); beyond 256 MB the norms are computed from the gathered rows
        const bool inv_tab = a.inv_norm != nullptr;
        Engine &eng0 = engine();
        // the table is carried over from the previous step when the full-table apply kernel kept it current (transe_counts.hip)
        const bool carried = eng0.inv_carry && eng0.inv_valid && eng0.inv_for_ent == a.ent && eng0.inv_for_rel == a.rel;
        if (inv_tab && !carried) {
            long long nb = (a.ent_total + a.rel_total + TEAMS - 1) / TEAMS;
            if (nb > 2048) nb = 2048;
            hipLaunchKernelGGL((row_inv_norm_kernel<L, C>), dim3((unsigned)nb), dim3(256), 0, stream, a.ent, a.rel,
                               (long long)a.ent_total, (long long)a.rel_total, a.D, const_cast<float *>(a.inv_norm));
            eng0.inv_for_ent = a.ent; eng0.inv_for_rel = a.rel; eng0.inv_valid = 1;
        }
        Engine &eng = engine();
        const int slot = (int)(eng.emit_launches % Engine::kEmitRing);
        const bool timed = eng.time_emit > 0 && (eng.emit_seen++ % eng.time_emit) == 0;   // every time_emit-th launch
        if (timed) {   // HIP events around THE kernel, on its launch stream (bench.py roofline)
            if (!eng.ev_emit0[slot]) { (void)hipEventCreate(&eng.ev_emit0[slot]); (void)hipEventCreate(&eng.ev_emit1[slot]); }
            (void)hipEventRecord(eng.ev_emit0[slot], stream);
        }
        if (a.rec2) {
            if (inv_tab) hipLaunchKernelGGL((transe_emit_vec_kernel<L, (C + 3) / 4, 4, 1, true, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((transe_emit_vec_kernel<L, (C + 3) / 4, 4, 1, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
        } else if (inv_tab) hipLaunchKernelGGL((transe_emit_vec_kernel<L, (C + 3) / 4, 4, 1, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((transe_emit_vec_kernel<L, (C + 3) / 4, 4, 1, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
        if (timed) { (void)hipEventRecord(eng.ev_emit1[slot], stream); eng.emit_launches++; }
        record_emit_done(stream);
    } else {
        hipLaunchKernelGGL((transe_emit_kernel<L, C>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
        record_emit_done(stream);   // widths that are not multiples of 4 (dim 50): the prefetched sampler waits on this one too
    }
    // groups with non sampler-shaped negatives: exact fp32 path into the residual accumulators
    // (no residual accumulators = record-only caller: the groups stay listed, kge_transe_deferred_groups reports them)
    if (!defer_pass) return;
    FbArgs d = a;
    d.loss_partials = a.loss_partials + blocks;
    hipLaunchKernelGGL((fwdbwd_kernel<KGE_TRANSE, L, C>), dim3(kDeferBlocks), dim3(256), 0, stream, d);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, a.loss_partials, (int)blocks + kDeferBlocks, a.unit, d_loss);
}

// The per-workgroup loss partials and the "last workgroup" tickets are ONE process-global set of buffers: two loss-producing
// kernels in flight on different streams would interleave tickets and partials.  A launch on another stream than the previous
// one therefore waits for that stream first (the training loop uses one stream: this never triggers there).
void guard_loss_stream(hipStream_t stream) {
    static hipStream_t last = nullptr;
    static bool have = false;
    if (have && last != stream) (void)hipStreamSynchronize(last);
    last = stream; have = true;
}

int ensure_loss_buffers() {
    Engine &e = engine();
    if (!e.dev.loss_partials) {
        int rc = hip_check(hipMalloc(&e.dev.loss_partials, sizeof(float) * (kMaxLossBlocks + 256)), "alloc loss partials");
        if (rc) return rc;
    }
    if (!e.dev.loss_ticket) {
        const size_t n = 1 + (kMaxLossBlocks + 31) / 32;   // master + one sub-counter per 32 blocks
        int rc = hip_check(hipMalloc(&e.dev.loss_ticket, sizeof(unsigned) * n), "alloc loss ticket");
        if (rc) return rc;
        if ((rc = hip_check(hipMemset(e.dev.loss_ticket, 0, sizeof(unsigned) * n), "zero loss ticket"))) return rc;
    }
    return KGE_OK;
}

// team shape used for dimension D (shared with transe_counts.hip through these two helpers)
void transe_team_shape(int D, int &L, int &C) {
    if (D % 4 == 0 && D <= 64) { L = 16; C = 4; }  // vectorised kernel: one float4 per lane
    else if (D <= 16) { L = 16; C = 1; } else if (D <= 32) { L = 16; C = 2; } else if (D <= 64) { L = 16; C = 4; }
    else if (D <= 128) { L = 32; C = 4; } else if (D <= 256) { L = 64; C = 4; } else if (D <= 512) { L = 64; C = 8; }
    else { L = 64; C = 16; }
}

static int32_t *g_defer_list = nullptr, *g_defer_count = nullptr;
static int64_t g_defer_cap = 0;

// groups of the LAST emit launch that were not sampler-shaped (synchronous read of the device counter)
static int32_t *g_skipped = nullptr;     // negatives the last in-place SGD step skipped (FbArgs::skipped)
int sgd_rows_skipped(int32_t *out) {
    *out = 0;
    if (!g_skipped) return KGE_OK;
    return hip_check(hipMemcpy(out, g_skipped, sizeof(int32_t), hipMemcpyDeviceToHost), "read skipped-negatives counter");
}

int transe_deferred_groups(int32_t *out) {
    *out = 0;
    if (!g_defer_count) return KGE_OK;
    return hip_check(hipMemcpy(out, g_defer_count, sizeof(int32_t), hipMemcpyDeviceToHost), "read deferred count");
}

int launch_transe_emit(const kge_model_desc &m, const float *ent, const float *rel, float *resid_ent, float *resid_rel,
                       const int32_t *d_h, const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                       int64_t denom, uint32_t *rec, int32_t *dst, int krel, float *d_loss, hipStream_t stream, bool track_deferred,
                       uint8_t *rec2) {
    // rec2: 2-bit records for the negatives (FbArgs::rec2; widths that are multiples of 4 only -- the caller checks); null = int8
    // track_deferred = false: the caller guarantees sampler-shaped negatives (a device-sampled batch): no
    // deferral list, no counter reset, no fp32 pass
    Engine &e = engine();
    { int rc = ensure_loss_buffers(); if (rc) return rc; }
    int32_t *&defer_list = g_defer_list, *&defer_count = g_defer_count;
    int64_t &defer_cap = g_defer_cap;
    if (n_pos > defer_cap) {
        if (defer_list) (void)hipFree(defer_list);
        defer_list = nullptr;
        int rc = hip_check(hipMalloc(&defer_list, sizeof(int32_t) * (size_t)n_pos), "alloc deferred groups");
        if (rc) return rc;
        if (!defer_count && (rc = hip_check(hipMalloc(&defer_count, sizeof(int32_t)), "alloc deferred count"))) return rc;
        defer_cap = n_pos;
    }
    if (track_deferred) {
        int rc = hip_check(hipMemsetAsync(defer_count, 0, sizeof(int32_t), stream), "zero deferred count");
        if (rc) return rc;
    }
    const bool use_inv_table = (m.ent_total + m.rel_total) * (int64_t)m.ent_dim * 4 <= e.inv_table_max_bytes;
    if (use_inv_table && m.ent_total + m.rel_total > e.inv_cap) {
        if (e.inv_norm) (void)hipFree(e.inv_norm);
        e.inv_norm = nullptr; e.inv_valid = 0;
        int rc = hip_check(hipMalloc(&e.inv_norm, sizeof(float) * (size_t)(m.ent_total + m.rel_total)), "alloc row inverse norms");
        if (rc) return rc;
        e.inv_cap = m.ent_total + m.rel_total;
    }
    FbArgs a = {};
    a.inv_norm = use_inv_table ? e.inv_norm : nullptr;
    a.group_list = track_deferred ? defer_list : nullptr; a.group_count = defer_count;
    a.ent = ent; a.rel = rel; a.g_ent = resid_ent; a.g_rel = resid_rel;
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.n_pos = n_pos; a.n_neg = n_neg; a.stride = stride;
    a.D = m.ent_dim; a.margin = m.margin; a.unit = 1.0f / (float)denom;
    a.loss_partials = e.dev.loss_partials;
    a.negative_rel = m.negative_rel;
    a.rec = rec; a.rec2 = (m.ent_dim % 4 == 0) ? rec2 : nullptr; a.dst = dst; a.ent_total = (int)m.ent_total; a.rel_total = (int)m.rel_total; a.krel = 1;
    a.loss_limbs = track_deferred ? nullptr : e.loss_limbs;   // (with a deferred pass the loss is finalised by loss_finalize_kernel: the caller converts it)
    while (a.krel * 2 <= krel) a.krel *= 2;      // a power of two: the kernels take b & (krel - 1)
    const int D = a.D;
    if (D % 4 == 0 && D <= 64) launch_emit<16, 4>(a, d_loss, stream);
    else if (D <= 16) launch_emit<16, 1>(a, d_loss, stream);
    else if (D <= 32) launch_emit<16, 2>(a, d_loss, stream);
    else if (D <= 64) launch_emit<16, 4>(a, d_loss, stream);
    else if (D <= 128) launch_emit<32, 4>(a, d_loss, stream);
    else if (D <= 256) launch_emit<64, 4>(a, d_loss, stream);
    else if (D <= 512) launch_emit<64, 8>(a, d_loss, stream);
    else if (D <= 1024) launch_emit<64, 16>(a, d_loss, stream);
    else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024");
    return hip_check(hipGetLastError(), "transe emit launch");
}

// The atomic-path kernel with the NEXT batch's sampler riding in its launch (kge_sampling_attach): at the reference's batch
// sizes this launch is a few hundred latency-bound workgroups on 256 CUs, and the sampler -- a pointer chase that depends on
// nothing in the step -- took a quarter of the step as a launch of its own (config #1: 9.7 of 38 us).  Workgroups beyond
// a.loss_blocks are the sampler's.
template <int MODEL, int L, int C>
__global__ __launch_bounds__(256) void fwdbwd_ride_kernel(FbArgs a, SamplerArgs ride, int n_ride) {
    if ((int)blockIdx.x >= a.loss_blocks) {
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - a.loss_blocks, bern_lds);
        return;
    }
    fwdbwd_body<MODEL, L, C, false>(a);
}
template <int MODEL, int L, int C>
__global__ __launch_bounds__(256, 4) void fwdbwd_ride_kernel_occ4(FbArgs a, SamplerArgs ride, int n_ride) {   // (see fwdbwd_kernel_occ4)
    if ((int)blockIdx.x >= a.loss_blocks) {
        __shared__ float bern_lds[kBernLds];
        sample_block_ride(ride, (long long)blockIdx.x - a.loss_blocks, bern_lds);
        return;
    }
    fwdbwd_body<MODEL, L, C, false>(a);
}

template <int MODEL, int L, int C>
static void launch_fb(const FbArgs &a, float *d_loss, hipStream_t stream) {
    constexpr int TEAMS = 256 / L;
    long long blocks = (a.n_pos + TEAMS - 1) / TEAMS;
    if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
    if (blocks < 1) blocks = 1;
    FbArgs f = a;
    guard_loss_stream(stream);
    f.loss_out = d_loss; f.loss_ticket = engine().dev.loss_ticket;   // the last block writes the loss
    if constexpr (MODEL != KGE_TRANSR) {     // (TransR's armed sampler has ridden in its relation scatter by now)
        SamplerArgs ride = {};
        unsigned n_ride = 0;
        if (upload_jump_table() == KGE_OK && take_attached_sampler(ride, n_ride) && n_ride > 0) {
            f.loss_blocks = (int)blocks;
            if constexpr (MODEL != KGE_TRANSE && C <= 4) {
                if (engine().fb_occ4) {
                    hipLaunchKernelGGL((fwdbwd_ride_kernel_occ4<MODEL, L, C>), dim3((unsigned)blocks + n_ride), dim3(256), 0, stream, f, ride, (int)n_ride);
                    return;
                }
            }
            hipLaunchKernelGGL((fwdbwd_ride_kernel<MODEL, L, C>), dim3((unsigned)blocks + n_ride), dim3(256), 0, stream, f, ride, (int)n_ride);
            return;
        }
    }
    if constexpr (MODEL != KGE_TRANSE && C <= 4) {
        if (engine().fb_occ4) { hipLaunchKernelGGL((fwdbwd_kernel_occ4<MODEL, L, C>), dim3((unsigned)blocks), dim3(256), 0, stream, f); return; }
    }
    hipLaunchKernelGGL((fwdbwd_kernel<MODEL, L, C>), dim3((unsigned)blocks), dim3(256), 0, stream, f);
}

template <int MODEL, int L, int C>
static void launch_fb_records(const FbArgs &a, float *d_loss, hipStream_t stream) {
    constexpr int TEAMS = 256 / L;
    long long blocks = (a.n_pos + TEAMS - 1) / TEAMS;
    if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
    if (blocks < 1) blocks = 1;
    FbArgs f = a;
    guard_loss_stream(stream);
    f.loss_out = d_loss; f.loss_ticket = engine().dev.loss_ticket;
    if constexpr ((MODEL == KGE_TRANSH || MODEL == KGE_TRANSD) && C <= 4) {
        if (engine().fb_occ4) { hipLaunchKernelGGL((fwdbwd_kernel_occ4<MODEL, L, C, true>), dim3((unsigned)blocks), dim3(256), 0, stream, f); return; }
    }
    hipLaunchKernelGGL((fwdbwd_kernel<MODEL, L, C, true>), dim3((unsigned)blocks), dim3(256), 0, stream, f);
}

template <int MODEL>
static int dispatch_fb_records(const FbArgs &a, float *d_loss, hipStream_t stream) {
    const int D = a.D;
    if (D <= 16) launch_fb_records<MODEL, 16, 1>(a, d_loss, stream);
    else if (D <= 32) launch_fb_records<MODEL, 16, 2>(a, d_loss, stream);
    else if (D <= 64) launch_fb_records<MODEL, 16, 4>(a, d_loss, stream);
    else if (D <= 128) launch_fb_records<MODEL, 32, 4>(a, d_loss, stream);
    else if (D <= 256) launch_fb_records<MODEL, 64, 4>(a, d_loss, stream);
    else if (D <= 512) launch_fb_records<MODEL, 64, 8>(a, d_loss, stream);
    else if (D <= 1024) launch_fb_records<MODEL, 64, 16>(a, d_loss, stream);
    else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024 is not supported by the vector-model kernels");
    return KGE_OK;
}

// sum the hub copies into the accumulators and re-zero them
__global__ __launch_bounds__(256) void hub_fold_kernel(float *__restrict__ copies_rel, float *__restrict__ copies_auxr,
                                                       float *__restrict__ g_rel, float *__restrict__ g_auxr, int K, long long RD) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * RD; i += (long long)gridDim.x * blockDim.x) {
        float *c = i < RD ? copies_rel : copies_auxr;
        float *g = i < RD ? g_rel : g_auxr;
        if (!c) continue;
        const long long j = i < RD ? i : i - RD;
        float s = 0.f;
        for (int k0 = 0; k0 < K; k0 += 8) {   // eight copies in flight (a chain of dependent loads made this small kernel take 27 us)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = c[(long long)min(k0 + u, K - 1) * RD + j];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k0 + u < K && v[u] != 0.f) { s += v[u]; c[(long long)(k0 + u) * RD + j] = 0.f; }
        }
        if (s != 0.f) g[j] += s;
    }
}

template <int MODEL>
static int dispatch_fb(const FbArgs &a, float *d_loss, hipStream_t stream) {
    const int D = a.D;
    if (D <= 16) launch_fb<MODEL, 16, 1>(a, d_loss, stream);
    else if (D <= 32) launch_fb<MODEL, 16, 2>(a, d_loss, stream);
    else if (D <= 64) launch_fb<MODEL, 16, 4>(a, d_loss, stream);
    else if (D <= 128) launch_fb<MODEL, 32, 4>(a, d_loss, stream);
    else if (D <= 256) launch_fb<MODEL, 64, 4>(a, d_loss, stream);
    else if (D <= 512) launch_fb<MODEL, 64, 8>(a, d_loss, stream);
    else if (D <= 1024) launch_fb<MODEL, 64, 16>(a, d_loss, stream);
    else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024 is not supported by the vector-model kernels");
    return KGE_OK;
}

// Atomic relation-side adds on a KG with few relations: group b adds into copy b % hub_k of [hub_k][R][D] buffers (zero between
// steps: hub_fold_kernel re-zeroes what it folds) when a relation-side row would take >= 128 adds per step.
static int attach_hub_copies(const kge_model_desc &m, int64_t n_pos, FbArgs &a) {
    Engine &e = engine();
    const int64_t hub_rows = (m.model == KGE_TRANSE ? 1 : 2) * m.rel_total;
    const int64_t group_rel = m.model == KGE_TRANSE ? 1 : 2;
    const int64_t per_row = hub_rows > 0 ? (group_rel * n_pos) / hub_rows : 0;
    if (m.model == KGE_TRANSE || per_row < 128 || !e.hub_copies) return KGE_OK;
    int64_t copies = per_row / 16;
    if (copies > 64) copies = 64;
    const int64_t per_copy = m.rel_total * (int64_t)a.D;
    while (copies > 1 && copies * per_copy * 4 > (int64_t(32) << 20)) copies >>= 1;
    if (copies <= 1) return KGE_OK;
    static float *buf_rel = nullptr, *buf_auxr = nullptr;
    static int64_t buf_elems = 0;
    int rc;
    if (copies * per_copy > buf_elems) {
        if (buf_rel) (void)hipFree(buf_rel);
        if (buf_auxr) (void)hipFree(buf_auxr);
        buf_rel = buf_auxr = nullptr;
        buf_elems = copies * per_copy;
        if ((rc = hip_check(hipMalloc(&buf_rel, sizeof(float) * (size_t)buf_elems), "alloc hub copies"))) return rc;
        if ((rc = hip_check(hipMalloc(&buf_auxr, sizeof(float) * (size_t)buf_elems), "alloc hub copies"))) return rc;
        if ((rc = hip_check(hipMemset(buf_rel, 0, sizeof(float) * (size_t)buf_elems), "zero hub copies"))) return rc;
        if ((rc = hip_check(hipMemset(buf_auxr, 0, sizeof(float) * (size_t)buf_elems), "zero hub copies"))) return rc;
    }
    a.copies_rel = buf_rel; a.copies_auxr = buf_auxr; a.hub_k = (int)copies;
    a.rel_total = (int)m.rel_total;
    return KGE_OK;
}

// pair-count path (pairs.hip, transe_counts.hip)
bool pair_counts_shape_ok(int model, int D, int64_t n_neg);
bool pair_keys_sortable(int64_t ent_total, int64_t rel_total);
long long pair_emit_blocks(int64_t n_pos);
int pair_record_dwords(int D);
void launch_pair_emit(int model, const FbArgs &a, hipStream_t stream);
int pair_records_workspace(int64_t M, int rd, uint32_t *&rec, int32_t *&dst, float2 *&aux);
int pair_records_reduce(int model, int64_t M, int64_t n_int8, int D, int rd, int64_t ent_total, int64_t rel_total, const float *const tables[4],
                        float *const grads[4], float unit, hipStream_t stream);

// does a step of this shape take the pair-count path?  (also behind kge_pair_path_active: Config places its sampler prefetch by it)
bool pair_path_active(const kge_model_desc &m, int64_t n_pos, int64_t n_neg) {
    Engine &e = engine();
    // measured cross-over (tools/pair_threshold_sweep.py, profiles/r02_h_pair_threshold_sweep.jsonl): the per-group work of the pair path
    // is amortised from 5 negatives per positive on for TransH, from 3 for TransD (two rows per entity side on the float path)
    const int64_t min_neg = e.pair_counts_min_neg > 0 ? e.pair_counts_min_neg : (m.model == KGE_TRANSD ? 3 : 5);
    return e.pair_counts && m.ent_dim == m.rel_dim && n_neg >= min_neg && pair_counts_shape_ok(m.model, m.ent_dim, n_neg) &&
           pair_keys_sortable(m.ent_total, m.rel_total) && n_pos * (2 + n_neg) >= e.float_records_min &&
           n_pos * (2 + n_neg) < (int64_t(1) << 31);
}

// the exact fp32 kernel over the groups an emit kernel deferred (a.group_list), partial losses behind the emit kernel's
template <int MODEL>
static int dispatch_fb_deferred(const FbArgs &a, hipStream_t stream) {
    const int D = a.D;
#define KGE_DEFER(LL, CC) hipLaunchKernelGGL((fwdbwd_kernel<MODEL, LL, CC>), dim3(kDeferBlocks), dim3(256), 0, stream, a)
    if (D <= 16) KGE_DEFER(16, 1); else if (D <= 32) KGE_DEFER(16, 2); else if (D <= 64) KGE_DEFER(16, 4);
    else if (D <= 128) KGE_DEFER(32, 4); else if (D <= 256) KGE_DEFER(64, 4); else if (D <= 512) KGE_DEFER(64, 8);
    else if (D <= 1024) KGE_DEFER(64, 16);
    else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024 is not supported by the vector-model kernels");
#undef KGE_DEFER
    return KGE_OK;
}

// ---- TransR vector stage, lean form (rel_dim a multiple of 4 up to 1024; sampler-shaped groups) ------------------------------
// One wave per positive and its negatives over the PROJECTED rows P (written by the project GEMM, TransR.py:16-17): normalise,
// L1 score, hinge, and the normalise-backward of +-unit*sign(e) into GP (read by dgrad / wgrad).  Lane l holds the float4 chunks
// l, l+64, ... of a row (one 16-byte load or store per row chunk; the generic fwdbwd_kernel<TRANSR> issues four strided
// conditional loads per row, each compiled into its own branch + wait).  Every GP row dgrad / wgrad will read is WRITTEN here --
// zeros when the hinge is inactive -- so GP needs no memset.  Groups with a negative that is not entity-corrupted with the
// positive's matrix go to the generic kernel (a.group_list), after their GP rows have been zeroed here.
template <int Q>
__global__ __launch_bounds__(256) void transr_vec_kernel(FbArgs a) {
    constexpr int L = 64, TEAMS = 4, E = 4 * Q;
    __shared__ float red[TEAMS];
    __shared__ float add_stage[TEAMS][L * E];
    const int lane = threadIdx.x % L, team_in_block = threadIdx.x / L;
    const int D = a.D;
    auto load = [&](const float *__restrict__ tab, long long row, float (&x)[E]) {
        const float *p = tab + row * D;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int e0 = 4 * (lane + L * q);
            const bool ok = e0 < D;
            const float4 v = *reinterpret_cast<const float4 *>(p + (ok ? e0 : 0));   // clamped + select: no branch around the load
            x[4 * q] = ok ? v.x : 0.f; x[4 * q + 1] = ok ? v.y : 0.f; x[4 * q + 2] = ok ? v.z : 0.f; x[4 * q + 3] = ok ? v.w : 0.f;
        }
    };
    auto store = [&](float *__restrict__ tab, long long row, const float (&x)[E]) {
        float *p = tab + row * D;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int e0 = 4 * (lane + L * q);
            if (e0 < D) *reinterpret_cast<float4 *>(p + e0) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        }
    };
    auto dot = [&](const float (&x)[E], const float (&y)[E]) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) s += x[e] * y[e];
        return team_sum<L>(s);
    };
    auto normalize = [&](float (&x)[E], float &inv, bool &uc) {   // in place
        const float ss = dot(x, x);
        uc = ss >= 1e-12f;
        inv = 1.0f / sqrtf(uc ? ss : 1e-12f);
#pragma unroll
        for (int e = 0; e < E; e++) x[e] *= inv;
    };
    // gx = inv (G - y <y, G>) with G = unit * g
    auto normalize_bwd = [&](const float (&y)[E], const float (&g)[E], float inv, bool uc, float (&gx)[E]) {
        float d = dot(y, g) * a.unit;
        if (!uc) d = 0.f;
#pragma unroll
        for (int e = 0; e < E; e++) gx[e] = inv * (a.unit * g[e] - d * y[e]);
    };
    float zero[E];
#pragma unroll
    for (int e = 0; e < E; e++) zero[e] = 0.f;
    float lsum = 0.f;
    for (long long b = (long long)blockIdx.x * TEAMS + team_in_block; b < a.n_pos; b += (long long)gridDim.x * TEAMS) {
        const int h = a.bh[b], t = a.bt[b], r = a.br[b];
        // ---- every negative entity-corrupted, scored with the positive's matrix? ----
        float bad = 0.f;
        int my_new_head = 0;      // lane k: does negative k (k < 64) replace the head?  Handed out by lane broadcast below, so that the
                                  // negatives' row loads do not each wait for an id load of their own
        for (int k = lane; k < (int)a.n_neg; k += L) {
            const long long j = b + (long long)(k + 1) * a.stride;
            const int nh = a.bh[j];
            const NegClass nc = classify_negative<KGE_TRANSR>(h, t, r, nh, a.bt[j], a.br[j], a.negative_rel);
            if (!nc.fast || nc.same_h == nc.same_t) bad = 1.f;
            if (k < L) my_new_head = nh != h ? 1 : 0;
        }
        if (team_sum<L>(bad) != 0.f) {
            // the generic kernel writes the rows of its active hinges; everything dgrad / wgrad may read of this group is zeroed first
            for (long long k = 0; k <= a.n_neg; k++) {
                const long long sl = 2 * (k * a.n_pos + b);
                store(a.GP, sl, zero); store(a.GP, sl + 1, zero);
            }
            if (lane == 0 && a.group_list) a.group_list[atomicAdd(a.group_count, 1)] = (int32_t)b;
            continue;
        }
        float rn[E], hn[E], tn[E];
        float inv_r, inv_h, inv_t; bool uc_r, uc_h, uc_t;
        // negative k's projected row: requested one negative ahead (the first one together with the positive's three rows), so a
        // row's memory round trip overlaps the reductions of the row before it
        auto head_is_new = [&](long long k) { return k < L ? team_bcast<L>(my_new_head, (int)k) != 0 : a.bh[b + (k + 1) * a.stride] != h; };
        float x_ahead[E];
        bool nh_ahead = head_is_new(0);
        load(a.rel, r, rn); load(a.P, 2 * b, hn); load(a.P, 2 * b + 1, tn);
        load(a.P, 2 * (a.n_pos + b) + (nh_ahead ? 0 : 1), x_ahead);
        normalize(rn, inv_r, uc_r); normalize(hn, inv_h, uc_h); normalize(tn, inv_t, uc_t);
        float sp[E], Ah[E], At[E], Ar[E];
        float p;
        {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) { const float ev = hn[e] + rn[e] - tn[e]; acc += fabsf(ev); sp[e] = sgn(ev); Ah[e] = 0.f; At[e] = 0.f; Ar[e] = 0.f; }
            p = team_sum<L>(acc);
        }
        int cnt = 0;
        for (long long k = 0; k < a.n_neg; k++) {
            const bool new_head = nh_ahead;
            const long long slot = 2 * ((k + 1) * a.n_pos + b) + (new_head ? 0 : 1);   // the corrupted side's projected row
            float xn[E], sg[E];
            float inv; bool uc;
#pragma unroll
            for (int e = 0; e < E; e++) xn[e] = x_ahead[e];
            if (k + 1 < a.n_neg) {
                nh_ahead = head_is_new(k + 1);
                load(a.P, 2 * ((k + 2) * a.n_pos + b) + (nh_ahead ? 0 : 1), x_ahead);
            }
            normalize(xn, inv, uc);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float ev = new_head ? xn[e] + rn[e] - tn[e] : hn[e] + rn[e] - xn[e];
                acc += fabsf(ev); sg[e] = sgn(ev);
            }
            const float nk = team_sum<L>(acc);
            const float v = p - nk + a.margin;
            if (v >= 0.f) {
                cnt++; lsum += v;
                float g[E], gx[E];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    // new head: dL/dx^ = -s, the kept t gets +s ; new tail: +s, the kept h gets -s ; r^ gets -s either way
                    g[e] = new_head ? -sg[e] : sg[e];
                    if (new_head) At[e] += sg[e]; else Ah[e] -= sg[e];
                    Ar[e] -= sg[e];
                }
                normalize_bwd(xn, g, inv, uc, gx);
                store(a.GP, slot, gx);
            } else {
                store(a.GP, slot, zero);
            }
        }
        if (cnt == 0) { store(a.GP, 2 * b, zero); store(a.GP, 2 * b + 1, zero); continue; }
        const float fc = (float)cnt;
        float g[E], gx[E];
#pragma unroll
        for (int e = 0; e < E; e++) g[e] = Ah[e] + fc * sp[e];
        normalize_bwd(hn, g, inv_h, uc_h, gx);
        store(a.GP, 2 * b, gx);
#pragma unroll
        for (int e = 0; e < E; e++) g[e] = At[e] - fc * sp[e];
        normalize_bwd(tn, g, inv_t, uc_t, gx);
        store(a.GP, 2 * b + 1, gx);
#pragma unroll
        for (int e = 0; e < E; e++) g[e] = Ar[e] + fc * sp[e];
        normalize_bwd(rn, g, inv_r, uc_r, gx);
        {   // g_rel[r] += gx, contiguous per instruction (memory-side atomics are served per line touched: see flush_run)
            float *stage = add_stage[team_in_block];
#pragma unroll
            for (int q = 0; q < Q; q++)
                *reinterpret_cast<float4 *>(stage + 4 * (lane + L * q)) = make_float4(gx[4 * q], gx[4 * q + 1], gx[4 * q + 2], gx[4 * q + 3]);
            // relation rows are hubs (a Zipf head relation takes a sixth of the batch): same-address atomics serialise, so group b
            // adds into copy b % hub_k, folded afterwards (hub_fold_kernel)
            float *pr = (a.copies_rel ? a.copies_rel + (b % a.hub_k) * (long long)a.rel_total * D : a.g_rel) + (long long)r * D;
#pragma unroll
            for (int c = 0; c < E; c++) {
                const int e = lane + L * c;
                const float v = stage[e];
                if (e < D) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(pr + e), v);
            }
        }
    }
    finish_loss<TEAMS>(a, red, lsum, lane, team_in_block);
}

// TransR: the score / hinge / backward over the projected vectors (transr.hip runs the GEMMs around it)
int launch_transr_vector_stage(const float *rel, float *g_rel, const float *P, float *GP, const int32_t *d_h,
                               const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                               int64_t denom, int rel_dim, float margin, int negative_rel, float *d_loss,
                               hipStream_t stream, bool lean, bool sampler_shaped, int64_t rel_total) {
    Engine &e = engine();
    FbArgs a = {};
    a.rel = rel; a.g_rel = g_rel; a.P = P; a.GP = GP;
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.n_pos = n_pos; a.n_neg = n_neg; a.stride = stride;
    a.D = rel_dim; a.margin = margin; a.unit = 1.0f / (float)denom;
    a.loss_partials = e.dev.loss_partials;
    a.negative_rel = negative_rel;
    int rc;
    if (lean) {   // every GP row read later is written by the kernel itself (the caller skipped the memset)
        constexpr int TEAMS = 4;
        {
            kge_model_desc md = {};
            md.model = KGE_TRANSR; md.rel_total = rel_total;
            if ((rc = attach_hub_copies(md, n_pos, a))) return rc;   // a.D = rel_dim: [hub_k][R][rel_dim] copies of g_rel
        }
        long long blocks = (n_pos + TEAMS - 1) / TEAMS;
        if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
        if (blocks < 1) blocks = 1;
        guard_loss_stream(stream);
        const int Q = (rel_dim / 4 + 63) / 64;
        auto launch = [&](const FbArgs &f) {
            if (Q == 1) hipLaunchKernelGGL((transr_vec_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, stream, f);
            else if (Q == 2) hipLaunchKernelGGL((transr_vec_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, stream, f);
            else hipLaunchKernelGGL((transr_vec_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, stream, f);
        };
        if (sampler_shaped) {   // the caller vouches for the batch: the kernel's last workgroup writes the loss
            a.loss_out = d_loss; a.loss_ticket = e.dev.loss_ticket;
            launch(a);
        } else {
            if (n_pos > g_defer_cap) {
                if (g_defer_list) (void)hipFree(g_defer_list);
                g_defer_list = nullptr;
                if ((rc = hip_check(hipMalloc(&g_defer_list, sizeof(int32_t) * (size_t)n_pos), "alloc deferred groups"))) return rc;
                if (!g_defer_count && (rc = hip_check(hipMalloc(&g_defer_count, sizeof(int32_t)), "alloc deferred count"))) return rc;
                g_defer_cap = n_pos;
            }
            if ((rc = hip_check(hipMemsetAsync(g_defer_count, 0, sizeof(int32_t), stream), "zero deferred count"))) return rc;
            a.group_list = g_defer_list; a.group_count = g_defer_count;
            launch(a);
            FbArgs d = a;
            d.loss_partials = a.loss_partials + blocks;
            if ((rc = dispatch_fb_deferred<KGE_TRANSR>(d, stream))) return rc;
            hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, a.loss_partials, (int)blocks + kDeferBlocks, a.unit, d_loss);
        }
        if (a.copies_rel) {
            const long long RD = (long long)rel_total * rel_dim;
            long long nb = (RD + 255) / 256;
            if (nb > 1024) nb = 1024;
            hipLaunchKernelGGL(hub_fold_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.copies_rel, (float *)nullptr, g_rel, (float *)nullptr,
                               a.hub_k, RD);
        }
        return hip_check(hipGetLastError(), "transr vector stage launch");
    }
    rc = dispatch_fb<KGE_TRANSR>(a, d_loss, stream);
    if (rc) return rc;
    return hip_check(hipGetLastError(), "transr vector stage launch");
}

// the lean vector stage writes every GP row that is read afterwards (no memset needed): rel_dim a multiple of 4 up to 1024
bool transr_lean_vector_stage(int rel_dim) { return engine().transr_lean && rel_dim % 4 == 0 && rel_dim >= 4 && rel_dim <= 1024; }

int launch_forward_backward_transr(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h,
                                   const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                                   int64_t denom, float *const grads[4], float *d_loss, hipStream_t stream, bool sampler_shaped);

// virtual row space of a step's float gradient records: entity rows, then hub_k copies of the relation-side rows
struct RecordSpace { int64_t slots, ent_rows, hub_rows, hub_k, rows; };
static RecordSpace record_space(const kge_model_desc &m, int64_t n_pos, int64_t n_neg) {
    RecordSpace s;
    s.slots = m.model == KGE_TRANSE ? 3 + n_neg : (m.model == KGE_TRANSH ? 4 + n_neg : 6 + 2 * n_neg);
    // relation-side rows are hubs: every group writes 1 (TransE) or 2 of them, onto only R or 2R rows.  Group b writes
    // into virtual copy b mod hub_k, with hub_k chosen so that a copy of a row still collects ~64 records: the sort
    // buckets stay bounded AND the segmented sum folds a run of ~64 records into ONE atomic row add (one copy per
    // record would put every record's 4*D bytes through same-address atomics again: measured 82 us for 43 k records).
    s.ent_rows = (m.model == KGE_TRANSD ? 2 : 1) * m.ent_total;
    s.hub_rows = (m.model == KGE_TRANSE ? 1 : 2) * m.rel_total;
    const int64_t group_rel = m.model == KGE_TRANSE ? 1 : 2;
    s.hub_k = s.hub_rows > 0 ? (group_rel * n_pos) / (s.hub_rows * 64) : 1;
    if (s.hub_k < 1) s.hub_k = 1;
    if (s.hub_k > 4096) s.hub_k = 4096;
    s.rows = s.ent_rows + s.hub_k * s.hub_rows;
    return s;
}

// Data-parallel form of the row-wise SGD in place (kge_forward_backward_sgd_rows cut in two): every rank stores the gradient rows
// of ITS slice of the batch as float records into its slice [rec_offset, rec_offset + rec_slice) of a buffer all ranks then
// all-gather, and every rank applies ALL records to its replica (launch_float_records_apply) -- the touched-row exchange of
// north_star; the replicas stay identical because every rank sums the same records in the same order.  The row space (hub copies)
// is that of the GLOBAL batch, so all ranks key their records alike.
int launch_forward_backward_records(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                                    const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride, int64_t denom, int64_t n_pos_total,
                                    float *d_rec, int32_t *d_dst, int64_t rec_offset, int64_t rec_slice, float *d_loss, hipStream_t stream) {
    Engine &e = engine();
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_forward_backward_records: no usable HIP device");
    if (n_pos < 0 || n_neg < 1 || stride < n_pos || denom <= 0 || n_pos_total < n_pos || !d_rec || !d_dst || rec_offset < 0)
        return fail(KGE_ERR_BAD_ARG, "kge_forward_backward_records: bad arguments");
    if (m.model != KGE_TRANSE && m.model != KGE_TRANSH && m.model != KGE_TRANSD)
        return fail(KGE_ERR_UNSUPPORTED, "gradient rows as float records: TransE / TransH / TransD only");
    if (m.ent_dim != m.rel_dim) return fail(KGE_ERR_BAD_ARG, "TransE/H/D need ent_dim == rel_dim (hidden_size)");
    int rc;
    if ((rc = ensure_loss_buffers())) return rc;
    const RecordSpace sp = record_space(m, n_pos_total, n_neg);
    const int64_t M = n_pos * sp.slots;
    if (M > rec_slice) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward_records: the slice is smaller than this rank's records");
    if (sp.rows >= (int64_t(1) << 31) - 1 || m.ent_dim > 1024) return fail(KGE_ERR_UNSUPPORTED, "gradient rows as float records: row space too large");
    // the part of the slice this rank does not fill carries no record
    if (rec_slice > M && (rc = hip_check(hipMemsetAsync(d_dst + rec_offset + M, 0xff, sizeof(int32_t) * (size_t)(rec_slice - M), stream), "blank record keys"))) return rc;
    if (n_pos == 0) return hip_check(hipMemsetAsync(d_loss, 0, sizeof(float), stream), "zero loss");
    FbArgs a = {};
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3];
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.n_pos = n_pos; a.n_neg = n_neg; a.stride = stride;
    a.D = m.ent_dim; a.margin = m.margin; a.unit = 1.0f / (float)denom;
    a.loss_partials = e.dev.loss_partials;
    a.negative_rel = m.negative_rel;
    a.frec = d_rec + (size_t)rec_offset * a.D; a.fdst = d_dst + rec_offset;
    a.ent_total = (int)m.ent_total; a.rel_total = (int)m.rel_total;
    a.hub_base = sp.ent_rows; a.hub_k = (int)sp.hub_k; a.hub_rows = (int)sp.hub_rows;
    if (!g_skipped && (rc = hip_check(hipMalloc(&g_skipped, sizeof(int32_t)), "alloc skipped-negatives counter"))) return rc;
    if ((rc = hip_check(hipMemsetAsync(g_skipped, 0, sizeof(int32_t), stream), "zero skipped-negatives counter"))) return rc;
    a.skipped = g_skipped;
    switch (m.model) {
        case KGE_TRANSE: rc = dispatch_fb_records<KGE_TRANSE>(a, d_loss, stream); break;
        case KGE_TRANSH: rc = dispatch_fb_records<KGE_TRANSH>(a, d_loss, stream); break;
        default: rc = dispatch_fb_records<KGE_TRANSD>(a, d_loss, stream); break;
    }
    if (rc) return rc;
    return hip_check(hipGetLastError(), "forward_backward records launch");
}

int launch_float_records_apply(const kge_model_desc &m, float *const tables[4], const float *d_rec, int32_t *d_dst, int64_t M_total,
                               int64_t n_pos_total, int64_t n_neg, float lr, hipStream_t stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_float_records_apply: no usable HIP device");
    if (!d_rec || !d_dst || M_total < 0 || n_pos_total < 0 || !(lr > 0.f)) return fail(KGE_ERR_BAD_ARG, "kge_float_records_apply: bad arguments");
    if (m.model != KGE_TRANSE && m.model != KGE_TRANSH && m.model != KGE_TRANSD)
        return fail(KGE_ERR_UNSUPPORTED, "gradient rows as float records: TransE / TransH / TransD only");
    if (M_total == 0) return KGE_OK;
    if (M_total >= (int64_t(1) << 31)) return fail(KGE_ERR_UNSUPPORTED, "kge_float_records_apply: too many records for the record sort");
    const RecordSpace sp = record_space(m, n_pos_total, n_neg);
    FloatRowSpace rs;
    rs.g_ent = tables[0]; rs.g_rel = tables[1]; rs.g_auxr = tables[2]; rs.g_auxe = tables[3];
    rs.E = m.ent_total; rs.R = m.rel_total; rs.hub_base = sp.ent_rows; rs.hub_rows = sp.hub_rows; rs.rows = sp.rows;
    rs.scale = -lr;
    tables_written();
    return float_records_reduce(M_total, m.ent_dim, rs, stream, d_rec, d_dst, true);
}

int launch_forward_backward(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                            const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride, int64_t denom,
                            float *const grads[4], float *d_loss, hipStream_t stream, bool sampler_shaped, float inplace_lr) {
    // inplace_lr != 0 (kge_forward_backward_sgd_rows): `grads` ARE the parameter tables; the step's gradient rows go through the
    // float-record path whatever its size and every summed run is added to its row as -lr * sum -- SGD on the touched rows, no
    // gradient tables, no sweep.  The forward has finished reading the tables when the segmented sum starts (one stream).
    Engine &e = engine();
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_forward_backward: no usable HIP device");
    if (n_pos < 0 || n_neg < 1 || stride < n_pos || denom <= 0) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward: bad sizes");
    { int rc = ensure_loss_buffers(); if (rc) return rc; }
    if (m.model == KGE_TRANSR) {
        if (inplace_lr != 0.f) return fail(KGE_ERR_UNSUPPORTED, "row-wise SGD in place: TransE / TransH / TransD only");
        return launch_forward_backward_transr(m, tables, d_h, d_t, d_r, n_pos, n_neg, stride, denom, grads, d_loss, stream, sampler_shaped);
    }
    if (m.ent_dim != m.rel_dim) return fail(KGE_ERR_BAD_ARG, "TransE/H/D need ent_dim == rel_dim (hidden_size)");
    FbArgs a = {};
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3];
    a.g_ent = grads[0]; a.g_rel = grads[1]; a.g_auxr = grads[2]; a.g_auxe = grads[3];
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.n_pos = n_pos; a.n_neg = n_neg; a.stride = stride;
    a.D = m.ent_dim; a.margin = m.margin; a.unit = 1.0f / (float)denom;
    a.loss_partials = e.dev.loss_partials;
    a.P = nullptr; a.GP = nullptr; a.negative_rel = m.negative_rel;
    int rc;
    // Pair-count path (TransH / TransD): int8 sign records keyed by (entity, relation), the backward applied once per pair
    if (inplace_lr == 0.f && pair_path_active(m, n_pos, n_neg)) {     // (its per-pair backward re-reads the rows: not for in-place updates)
        const int64_t M = n_pos * (2 + n_neg);
        const int rd = pair_record_dwords(a.D);
        if ((rc = pair_records_workspace(M, rd, a.rec, a.dst, a.pair_aux))) return rc;
        a.ent_total = (int)m.ent_total; a.rel_total = (int)m.rel_total;
        if ((rc = attach_hub_copies(m, n_pos, a))) return rc;
        guard_loss_stream(stream);
        if (sampler_shaped) {
            // the caller vouches for the batch: no deferral list, the emit kernel's last workgroup writes the loss
            a.loss_out = d_loss; a.loss_ticket = e.dev.loss_ticket;
            launch_pair_emit(m.model, a, stream);
            record_emit_done(stream);
        } else {
            if (n_pos > g_defer_cap) {
                if (g_defer_list) (void)hipFree(g_defer_list);
                g_defer_list = nullptr;
                if ((rc = hip_check(hipMalloc(&g_defer_list, sizeof(int32_t) * (size_t)n_pos), "alloc deferred groups"))) return rc;
                if (!g_defer_count && (rc = hip_check(hipMalloc(&g_defer_count, sizeof(int32_t)), "alloc deferred count"))) return rc;
                g_defer_cap = n_pos;
            }
            if ((rc = hip_check(hipMemsetAsync(g_defer_count, 0, sizeof(int32_t), stream), "zero deferred count"))) return rc;
            a.group_list = g_defer_list; a.group_count = g_defer_count;
            launch_pair_emit(m.model, a, stream);
            record_emit_done(stream);
            // groups with negatives that are not sampler-shaped: the exact fp32 kernel (atomic adds), its partial losses behind the emit's
            FbArgs d = a;
            d.loss_partials = a.loss_partials + pair_emit_blocks(n_pos);
            rc = m.model == KGE_TRANSH ? dispatch_fb_deferred<KGE_TRANSH>(d, stream) : dispatch_fb_deferred<KGE_TRANSD>(d, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, a.loss_partials, (int)pair_emit_blocks(n_pos) + kDeferBlocks,
                               a.unit, d_loss);
        }
        if ((rc = pair_records_reduce(m.model, M, 2 * n_pos, a.D, rd, m.ent_total, m.rel_total, tables, grads, a.unit, stream))) return rc;
        if (a.copies_rel) {
            const long long RD = (long long)m.rel_total * a.D;
            long long nb = (2 * RD + 255) / 256;
            if (nb > 1024) nb = 1024;
            hipLaunchKernelGGL(hub_fold_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.copies_rel, a.copies_auxr, grads[1], grads[2],
                               a.hub_k, RD);
        }
        return hip_check(hipGetLastError(), "pair-count forward_backward launch");
    }
    // Float-record path: gradient rows are stored as records, ordered by destination row and summed by
    // segments into the accumulators (plain stores + one sorted pass instead of memory-side fp32 atomics,
    // which cap at ~1.1 TB/s).  Worth it once a step has enough rows to fill the chip.
    const RecordSpace sp = record_space(m, n_pos, n_neg);
    const int64_t slots = sp.slots, ent_rows = sp.ent_rows, hub_rows = sp.hub_rows, hub_k = sp.hub_k, rows = sp.rows;
    const int64_t M = n_pos * slots;
    const bool records_fit = M < (int64_t(1) << 31) && rows < (int64_t(1) << 31) - 1 && a.D <= 1024;
    if (inplace_lr != 0.f && !records_fit) return fail(KGE_ERR_UNSUPPORTED, "row-wise SGD in place: step or row space too large for the record sort");
    if (n_pos == 0 && inplace_lr != 0.f) return hip_check(hipMemsetAsync(d_loss, 0, sizeof(float), stream), "zero loss");
    if (records_fit && (inplace_lr != 0.f || (e.float_records && M >= e.float_records_min))) {
        float *frec = nullptr;
        int32_t *fdst = nullptr;
        if ((rc = float_records_workspace(M, a.D, frec, fdst))) return rc;
        a.frec = frec; a.fdst = fdst;
        a.ent_total = (int)m.ent_total; a.rel_total = (int)m.rel_total;
        a.hub_base = ent_rows; a.hub_k = (int)hub_k; a.hub_rows = (int)hub_rows;
        if (inplace_lr != 0.f) {
            if (!g_skipped && (rc = hip_check(hipMalloc(&g_skipped, sizeof(int32_t)), "alloc skipped-negatives counter"))) return rc;
            if ((rc = hip_check(hipMemsetAsync(g_skipped, 0, sizeof(int32_t), stream), "zero skipped-negatives counter"))) return rc;
            a.skipped = g_skipped;
        }
        switch (m.model) {
            case KGE_TRANSE: rc = dispatch_fb_records<KGE_TRANSE>(a, d_loss, stream); break;
            case KGE_TRANSH: rc = dispatch_fb_records<KGE_TRANSH>(a, d_loss, stream); break;
            case KGE_TRANSD: rc = dispatch_fb_records<KGE_TRANSD>(a, d_loss, stream); break;
            default: return fail(KGE_ERR_BAD_ARG, "unknown model id");
        }
        if (rc) return rc;
        FloatRowSpace rs;
        rs.g_ent = grads[0]; rs.g_rel = grads[1]; rs.g_auxr = grads[2]; rs.g_auxe = grads[3];
        rs.E = m.ent_total; rs.R = m.rel_total; rs.hub_base = ent_rows; rs.hub_rows = hub_rows; rs.rows = rows;
        if (inplace_lr != 0.f) { rs.scale = -inplace_lr; tables_written(); }
        return float_records_reduce(M, a.D, rs, stream);
    }
    // atomic path: hub copies for the relation-side rows when a row would take hundreds of adds per step
    if ((rc = attach_hub_copies(m, n_pos, a))) return rc;
    switch (m.model) {
        case KGE_TRANSE: rc = dispatch_fb<KGE_TRANSE>(a, d_loss, stream); break;
        case KGE_TRANSH: rc = dispatch_fb<KGE_TRANSH>(a, d_loss, stream); break;
        case KGE_TRANSD: rc = dispatch_fb<KGE_TRANSD>(a, d_loss, stream); break;
        default: return fail(KGE_ERR_BAD_ARG, "unknown model id");
    }
    if (rc) return rc;
    if (a.copies_rel) {
        const long long RD = (long long)m.rel_total * a.D;
        long long nb = (2 * RD + 255) / 256;
        if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(hub_fold_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a.copies_rel, a.copies_auxr, grads[1], grads[2],
                           a.hub_k, RD);
    }
    if (rc) return rc;
    return hip_check(hipGetLastError(), "forward_backward launch");
}

// ------------------------------------------------------------------------------------------------
// predict: one team per triple.  TransE: mean over D (TransE.py:58); TransH/D: sum (TransH.py:82,
// TransD.py:98).
// ------------------------------------------------------------------------------------------------
template <int MODEL, int L, int C>
__global__ __launch_bounds__(256) void predict_kernel(FbArgs a, long long n, float *out) {
    constexpr int TEAMS = 256 / L;
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    for (long long i = (long long)blockIdx.x * TEAMS + threadIdx.x / L; i < n; i += (long long)gridDim.x * TEAMS) {
        Ctx<C> cx;
        ctx_forward<MODEL, L, C>(tm, a, a.br[i], cx);
        Side<C> sh, st;
        side_forward<MODEL, L, C>(tm, a, MODEL == KGE_TRANSR ? 2 * i : (long long)a.bh[i], cx.cw, sh);
        side_forward<MODEL, L, C>(tm, a, MODEL == KGE_TRANSR ? 2 * i + 1 : (long long)a.bt[i], cx.cw, st);
        float sg[C];
        float s = l1_score<L, C>(tm, sh.nrm, cx.rn, st.nrm, sg);
        if (tm.lane == 0) out[i] = MODEL == KGE_TRANSE ? s / (float)a.D : s;
    }
}

template <int MODEL>
static int dispatch_predict(const FbArgs &a, long long n, float *out, hipStream_t stream) {
    const int D = a.D;
#define KGE_PRED(LL, CC)                                                                                     \
    {                                                                                                        \
        long long blocks = (n + (256 / LL) - 1) / (256 / LL);                                                \
        if (blocks > 8192) blocks = 8192;                                                                    \
        if (blocks < 1) blocks = 1;                                                                          \
        hipLaunchKernelGGL((predict_kernel<MODEL, LL, CC>), dim3((unsigned)blocks), dim3(256), 0, stream, a, n, out); \
    }
    if (D <= 16) KGE_PRED(16, 1)
    else if (D <= 32) KGE_PRED(16, 2)
    else if (D <= 64) KGE_PRED(16, 4)
    else if (D <= 128) KGE_PRED(32, 4)
    else if (D <= 256) KGE_PRED(64, 4)
    else if (D <= 512) KGE_PRED(64, 8)
    else if (D <= 1024) KGE_PRED(64, 16)
    else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024 is not supported by the vector-model kernels");
#undef KGE_PRED
    return KGE_OK;
}

int launch_predict_transr(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                          const int32_t *d_r, int64_t n, float *d_out, hipStream_t stream);

// TransR predict: sum |l2n(P[2i]) + l2n(rel[r_i]) - l2n(P[2i+1])| over the projected vectors (TransR.py:77-87)
int launch_transr_predict_stage(const float *rel, const float *P, const int32_t *d_r, int64_t n, int rel_dim, float *d_out,
                                hipStream_t stream) {
    FbArgs a = {};
    a.rel = rel; a.P = P; a.br = d_r; a.D = rel_dim;
    int rc = dispatch_predict<KGE_TRANSR>(a, n, d_out, stream);
    if (rc) return rc;
    return hip_check(hipGetLastError(), "transr predict launch");
}

int launch_predict(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                   const int32_t *d_r, int64_t n, float *d_out, hipStream_t stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_predict: no usable HIP device");
    if (n <= 0) return KGE_OK;
    if (m.model == KGE_TRANSR) return launch_predict_transr(m, tables, d_h, d_t, d_r, n, d_out, stream);
    FbArgs a = {};
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3];
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.D = m.ent_dim;
    int rc;
    switch (m.model) {
        case KGE_TRANSE: rc = dispatch_predict<KGE_TRANSE>(a, n, d_out, stream); break;
        case KGE_TRANSH: rc = dispatch_predict<KGE_TRANSH>(a, n, d_out, stream); break;
        case KGE_TRANSD: rc = dispatch_predict<KGE_TRANSD>(a, n, d_out, stream); break;
        default: return fail(KGE_ERR_BAD_ARG, "unknown model id");
    }
    if (rc) return rc;
    return hip_check(hipGetLastError(), "predict launch");
}

// ------------------------------------------------------------------------------------------------
// Link prediction, relation-grouped (eval.hip kge_link_prediction): the test triples of one relation r all rank the
// same E candidates under the same relation context, so the candidates' projected + normalised vectors are built ONCE
// per relation into a table T_r [E, D] (for TransE: once per call), and a request only streams T_r against its fixed
// side -- one row read and one reduction per candidate instead of three rows, two projections and three normalisations.
// The arithmetic is the predict kernel's, function for function (ctx_forward, side_forward, l1_score), so the scores
// are bit-identical to kge_predict on getHeadBatch / getTailBatch.
// ------------------------------------------------------------------------------------------------
template <int MODEL, int L, int C>
__global__ __launch_bounds__(256) void lp_table_kernel(FbArgs a, long long r, long long E, float *__restrict__ T) {
    constexpr int TEAMS = 256 / L;
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    Ctx<C> cx;
    ctx_forward<MODEL, L, C>(tm, a, r, cx);
    for (long long j = (long long)blockIdx.x * TEAMS + threadIdx.x / L; j < E; j += (long long)gridDim.x * TEAMS) {
        Side<C> s;
        side_forward<MODEL, L, C>(tm, a, j, cx.cw, s);   // TransR: row j of a.P (all entities projected by M_r)
        tm.store(T, j, s.nrm);
    }
}

template <int MODEL, int L, int C>
__global__ __launch_bounds__(256) void lp_score_kernel(FbArgs a, const float *__restrict__ T, long long r, const int32_t *__restrict__ req_fixed,
                                                       const int32_t *__restrict__ req_head, long long E, float *__restrict__ scores) {
    constexpr int TEAMS = 256 / L;
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    const long long q = blockIdx.y;
    Ctx<C> cx;
    ctx_forward<MODEL, L, C>(tm, a, r, cx);
    float fixed[C], x[C], sg[C];
    tm.load(T, req_fixed[q], fixed);
    const bool head = req_head[q] != 0;
    float *out = scores + q * E;
    for (long long j = (long long)blockIdx.x * TEAMS + threadIdx.x / L; j < E; j += (long long)gridDim.x * TEAMS) {
        tm.load(T, j, x);
        const float s = head ? l1_score<L, C>(tm, x, cx.rn, fixed, sg) : l1_score<L, C>(tm, fixed, cx.rn, x, sg);
        if (tm.lane == 0) out[j] = MODEL == KGE_TRANSE ? s / (float)a.D : s;
    }
}

// T_r for every entity.  TransR: P = all entities projected by M_r (transr.hip) must be in `P_all`.
int launch_lp_table(const kge_model_desc &m, const float *const tables[4], const float *P_all, int64_t r, float *T, hipStream_t stream) {
    FbArgs a = {};
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3]; a.P = P_all;
    a.D = m.model == KGE_TRANSR ? m.rel_dim : m.ent_dim;
    const long long E = m.ent_total;
    const int D = a.D;
#define KGE_LPT(MODEL, LL, CC)                                                                                         \
    {                                                                                                                  \
        long long blocks = (E + (256 / LL) - 1) / (256 / LL);                                                          \
        if (blocks > 4096) blocks = 4096;                                                                              \
        hipLaunchKernelGGL((lp_table_kernel<MODEL, LL, CC>), dim3((unsigned)blocks), dim3(256), 0, stream, a, (long long)r, E, T); \
    }
#define KGE_LPT_D(MODEL)                                                                                               \
    if (D <= 16) KGE_LPT(MODEL, 16, 1) else if (D <= 32) KGE_LPT(MODEL, 16, 2) else if (D <= 64) KGE_LPT(MODEL, 16, 4)  \
    else if (D <= 128) KGE_LPT(MODEL, 32, 4) else if (D <= 256) KGE_LPT(MODEL, 64, 4) else if (D <= 512) KGE_LPT(MODEL, 64, 8) \
    else if (D <= 1024) KGE_LPT(MODEL, 64, 16) else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024");
    switch (m.model) {
        case KGE_TRANSE: KGE_LPT_D(KGE_TRANSE) break;
        case KGE_TRANSH: KGE_LPT_D(KGE_TRANSH) break;
        case KGE_TRANSR: KGE_LPT_D(KGE_TRANSR) break;
        case KGE_TRANSD: KGE_LPT_D(KGE_TRANSD) break;
        default: return fail(KGE_ERR_BAD_ARG, "unknown model id");
    }
#undef KGE_LPT_D
#undef KGE_LPT
    return hip_check(hipGetLastError(), "lp table launch");
}

int launch_lp_scores(const kge_model_desc &m, const float *const tables[4], const float *T, int64_t r, const int32_t *d_req_fixed,
                     const int32_t *d_req_head, int64_t n_req, float *d_scores, hipStream_t stream) {
    if (n_req <= 0) return KGE_OK;
    FbArgs a = {};
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3];
    a.D = m.model == KGE_TRANSR ? m.rel_dim : m.ent_dim;
    const long long E = m.ent_total;
    const int D = a.D;
#define KGE_LPS(MODEL, LL, CC)                                                                                         \
    {                                                                                                                  \
        long long bx = (E + (256 / LL) * 16 - 1) / ((256 / LL) * 16);   /* ~16 candidates per team */                  \
        if (bx > 1024) bx = 1024;                                                                                      \
        if (bx < 1) bx = 1;                                                                                            \
        hipLaunchKernelGGL((lp_score_kernel<MODEL, LL, CC>), dim3((unsigned)bx, (unsigned)n_req), dim3(256), 0, stream, a, T, \
                           (long long)r, d_req_fixed, d_req_head, E, d_scores);                                        \
    }
#define KGE_LPS_D(MODEL)                                                                                               \
    if (D <= 16) KGE_LPS(MODEL, 16, 1) else if (D <= 32) KGE_LPS(MODEL, 16, 2) else if (D <= 64) KGE_LPS(MODEL, 16, 4)  \
    else if (D <= 128) KGE_LPS(MODEL, 32, 4) else if (D <= 256) KGE_LPS(MODEL, 64, 4) else if (D <= 512) KGE_LPS(MODEL, 64, 8) \
    else if (D <= 1024) KGE_LPS(MODEL, 64, 16) else return fail(KGE_ERR_UNSUPPORTED, "embedding dimension > 1024");
    switch (m.model) {   // only ctx_forward's relation vector and the TransE mean differ between models here
        case KGE_TRANSE: KGE_LPS_D(KGE_TRANSE) break;
        case KGE_TRANSH: KGE_LPS_D(KGE_TRANSH) break;
        case KGE_TRANSR: KGE_LPS_D(KGE_TRANSR) break;
        case KGE_TRANSD: KGE_LPS_D(KGE_TRANSD) break;
        default: return fail(KGE_ERR_BAD_ARG, "unknown model id");
    }
#undef KGE_LPS_D
#undef KGE_LPS
    return hip_check(hipGetLastError(), "lp score launch");
}

}  // namespace kge
