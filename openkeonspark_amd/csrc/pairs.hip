// Pair-count path for the projecting models (TransH, TransD), stage 1: the emit kernel.
//
// Why: with the float-record path every gradient row of a step travels through HBM twice as 4*D bytes
// (record store, then the segmented sum's gather), and the per-negative backward keeps the vector kernel
// issue-bound (five 64-lane reductions, ~200 wave instructions per negative).  But for a
// FIXED (entity x, relation r) pair the whole backward of an entity side -- normalise, projection
// (TransH.py:12-14 / TransD.py:23-25) -- is LINEAR in the upstream gradient G = dL/d(normalised projected
// vector), and G is +-unit * sign(e) per scored pair (L1 score, margin ranking loss: TransH.py:52-69).
// So the signs of all uses of (x, r) in a step can be summed as INTEGERS first and pushed through the
// backward once per pair:
//
//   stage 1  pair_emit_kernel (this file): forward + hinge; per entity side one sign record keyed by
//            x*R + r (a negative's: 2 bits per element, 64 bytes; the positive's two: int8 sums, D bytes); the relation-side gradients (r^: an
//            integer sum; w^ / r_p: bilinear in (x, sign), accumulated in registers per group) are
//            finished per GROUP and added to the hub copies of the relation tables.
//   stage 2  pair_records_reduce (transe_counts.hip): the two-level counting sort by key, then
//            segsum_pairs_kernel: integer run sums per pair -> the pair's entity-row gradient in fp32,
//            accumulated per entity row in registers and written once per row.
//
// Team shape: 16 lanes own one positive and its negatives, lane l holding the float4 chunks l, l+16, ...
// of a row ("natural" record layout: record dword j = elements 4j..4j+3).  A 16-lane all-reduce is four
// DPP adds and one wave instruction serves four groups, which is what makes the five reductions per
// negative affordable; the state per group is kept small (two hinge constants, the projection vector,
// ONE float accumulator for the relation-context gradient, packed int16 sign sums) and the positive's
// own rows are gathered again at the end of the group instead of being held.
#include "models_dev.hpp"

namespace kge {

namespace {

constexpr int PT = 16;   // lanes per team

// Row gather in the natural layout.  Q = ceil(D / 64), so every lane's chunks q < Q-1 exist; only the last chunk can lie
// past the row's end.  It is loaded from a clamped address and zeroed by a select: a conditional load would compile into a
// branch around it with its own wait, i.e. one memory latency per chunk instead of one per row.
template <int Q>
__device__ __forceinline__ void load_row(const float *__restrict__ tab, long long row, int D, int lane, float (&x)[4 * Q]) {
    const float *p = tab + row * D;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int e0 = 4 * (lane + PT * q);
        const bool ok = q < Q - 1 || e0 < D;
        const float4 v = *reinterpret_cast<const float4 *>(p + (ok ? e0 : 0));
        x[4 * q] = ok ? v.x : 0.f; x[4 * q + 1] = ok ? v.y : 0.f; x[4 * q + 2] = ok ? v.z : 0.f; x[4 * q + 3] = ok ? v.w : 0.f;
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// dot product over the team: two running sums (even / odd elements) so that the products go through v_pk_fma_f32 -- half the
// instructions of a scalar fma chain, and a third of the packed-multiply + scalar-add form the compiler chose for one sum
template <int E>
__device__ __forceinline__ float dot16(const float (&x)[E], const float (&y)[E]) {
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < E; e += 2) {
        const f32x2 xv = {x[e], x[e + 1]}, yv = {y[e], y[e + 1]};
        s = __builtin_elementwise_fma(xv, yv, s);
    }
    return team_sum<PT>(s.x + s.y);
}

// Atomic add of a row held in the natural layout (lane l: float4 chunks l, l+16, ...).  Memory-side atomics are served per
// 64-byte line touched by an instruction, so adding element 4l+j from lane l (a 16-byte stride: every instruction touches all
// the row's lines) costs four times the line operations of a contiguous add.  The row is therefore turned through LDS into the
// strided layout (lane l: elements l, l+16, ...): each instruction then covers 64 contiguous bytes per team.  `stage` is this
// TEAM's 16*4*Q floats; all traffic stays inside the wave (LDS operations of a wave complete in order: no barrier).
template <int Q>
__device__ __forceinline__ void atomic_add_row(float *__restrict__ tab, long long row, int D, int lane, const float (&g)[4 * Q], float *stage) {
    float *p = tab + row * D;
#pragma unroll
    for (int q = 0; q < Q; q++)
        *reinterpret_cast<float4 *>(stage + 4 * (lane + PT * q)) = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
#pragma unroll
    for (int c = 0; c < 4 * Q; c++) {
        const int e = lane + PT * c;
        const float v = stage[e];
        if (e < D) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(p + e), v);
    }
}

// Forward of one entity side against the relation context: projected vector xp, its 1/|xp| and the projection
// coefficient.  TransH (cw = w^): a = x.w^, xp = x - a w^.  TransD (cw = r_p, xa = the entity's transfer row):
// a = x.x_p, xp = x + a r_p.
template <int MODEL, int E>
__device__ __forceinline__ void project(const float (&x)[E], const float (&xa)[E], const float (&cw)[E], float (&xp)[E], float &a,
                                        float &inv, bool &uc) {
    if constexpr (MODEL == KGE_TRANSH) {
        a = dot16<E>(x, cw);
#pragma unroll
        for (int e = 0; e < E; e++) xp[e] = x[e] - a * cw[e];
    } else {
        a = dot16<E>(x, xa);
#pragma unroll
        for (int e = 0; e < E; e++) xp[e] = x[e] + a * cw[e];
    }
    const float ss = dot16<E>(xp, xp);
    uc = ss >= 1e-12f;
    inv = fast_rsqrt(uc ? ss : 1e-12f);
}

}  // namespace

// One positive and its negatives per 16-lane team.  Records: slot 0 = h, slot 1 = t, slot 2+k = negative k
// (record m = slot*n_pos + b); a.dst[m] = x*R + r, or -1 for no record.  The two records of the positive hold integer sums
// (int8, RD = 16*Q dwords each, natural layout) at rec[m*RD]; the negatives' records hold signs only: 16 dwords each (lane l's
// 4*Q elements as 2-bit fields of dword l, element 4q+j of the lane at bits 8q+2j) behind them, at rec[2*n_pos*RD + (m - 2*n_pos)*16].
template <int MODEL, int Q>
__global__ __launch_bounds__(256, 2) void pair_emit_kernel(FbArgs a) {
    constexpr int TEAMS = 256 / PT;
    constexpr int E = 4 * Q;
    __shared__ float red[TEAMS];
    // the two hinge constants of a group live in LDS, lane-private (a lane reads back only what it wrote: no barrier):
    // 2*E registers freed for a second row in flight per team, and the new-head / new-tail choice becomes an address
    __shared__ float4 hinge_const[2][Q][256];
    __shared__ float add_stage[TEAMS][PT * 4 * Q];   // atomic_add_row's turn-around buffer, one per team
    const int lane = threadIdx.x % PT;
    const int team_in_block = threadIdx.x / PT;
    const int D = a.D;
    const long long R = a.rel_total;
    float lsum = 0.f;
    for (long long b = (long long)blockIdx.x * TEAMS + team_in_block; b < a.n_pos; b += (long long)gridDim.x * TEAMS) {
        const int h = a.bh[b], t = a.bt[b], r = a.br[b];
        // ---- sampler-shaped?  (exactly one entity slot differs, same relation) ----
        float bad = 0.f;
        for (int k = lane; k < (int)a.n_neg; k += PT) {
            const long long j = b + (long long)(k + 1) * a.stride;
            const NegClass nc = classify_negative<MODEL>(h, t, r, a.bh[j], a.bt[j], a.br[j], a.negative_rel);
            if (!nc.fast || nc.same_h == nc.same_t) bad = 1.f;
        }
        if (team_sum<PT>(bad) != 0.f) {   // the whole group goes to the exact fp32 kernel
            for (long long sl = lane; sl < 2 + a.n_neg; sl += PT) a.dst[sl * a.n_pos + b] = -1;
            if (lane == 0 && a.group_list) a.group_list[atomicAdd(a.group_count, 1)] = (int32_t)b;   // (no list: the caller vouched for the batch)
            continue;
        }
        // ---- relation context ----
        float cw[E];   // TransH: w^ ; TransD: r_p
        float inv_w = 1.f, ww = 1.f;
        bool uc_w = true;
        load_row<Q>(a.auxr, r, D, lane, cw);
        if constexpr (MODEL == KGE_TRANSH) {
            const float ssw = dot16<E>(cw, cw);
            uc_w = ssw >= 1e-12f;
            inv_w = 1.0f / sqrtf(uc_w ? ssw : 1e-12f);
#pragma unroll
            for (int e = 0; e < E; e++) cw[e] *= inv_w;
            ww = dot16<E>(cw, cw);   // |w^|^2 (1 up to rounding): xp.w^ = a (1 - ww)
        }
        // ---- the positive: p, and the two hinge constants B0 = r^ - t^n (new head), B1 = h^n + r^ (new tail) ----
        float p;
        {
            float B0[E], B1[E];
            float rn[E], x[E], xa[E], xp[E];
            load_row<Q>(a.rel, r, D, lane, rn);
            {
                const float ssr = dot16<E>(rn, rn);
                const float ir = 1.0f / sqrtf(ssr >= 1e-12f ? ssr : 1e-12f);
#pragma unroll
                for (int e = 0; e < E; e++) rn[e] *= ir;
            }
            float aa, inv; bool uc;
            load_row<Q>(a.ent, h, D, lane, x);
            if constexpr (MODEL == KGE_TRANSD) load_row<Q>(a.auxe, h, D, lane, xa);
            project<MODEL, E>(x, xa, cw, xp, aa, inv, uc);
#pragma unroll
            for (int e = 0; e < E; e++) B1[e] = xp[e] * inv + rn[e];
            load_row<Q>(a.ent, t, D, lane, x);
            if constexpr (MODEL == KGE_TRANSD) load_row<Q>(a.auxe, t, D, lane, xa);
            project<MODEL, E>(x, xa, cw, xp, aa, inv, uc);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float tn = xp[e] * inv;
                acc += fabsf(B1[e] - tn);
                // "+ 0.0f": no negative zeros in the constants, so that fma(f, xp, B) is never -0.0 (sign_of_bits reads the sign bit)
                B0[e] = rn[e] - tn + 0.0f;
                B1[e] = B1[e] + 0.0f;
            }
            p = team_sum<PT>(acc);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                hinge_const[0][q][threadIdx.x] = make_float4(B0[4 * q], B0[4 * q + 1], B0[4 * q + 2], B0[4 * q + 3]);
                hinge_const[1][q][threadIdx.x] = make_float4(B1[4 * q], B1[4 * q + 1], B1[4 * q + 2], B1[4 * q + 3]);
            }
        }
        // ---- negatives ----
        // integer sums of the signs reaching the positive's h^n (-s per active new-tail pair) and t^n (+s per active new-head pair);
        // the sum reaching r^ (-s per active pair of either kind) is their difference Ah - At and is not kept
        s16x2 Ah_lo[Q], Ah_hi[Q], At_lo[Q], At_hi[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) { Ah_lo[q] = 0; Ah_hi[q] = 0; At_lo[q] = 0; At_hi[q] = 0; }
        float acw[E];       // relation-context gradient, vector part: sum of c1*x + c2*g
        float acw_s = 0.f;  // TransH: coefficient of w^ ; TransD: coefficient of r_p
#pragma unroll
        for (int e = 0; e < E; e++) acw[e] = 0.f;
        int cnt = 0;
        // the relation-context gradient of one entity side with upstream gradient unit*g (g integer valued):
        //   TransH: acw -= d x + a gxp     TransD: acw += a gxp     with gxp = inv (G - nrm <nrm, G>), nrm = inv xp
        // written on x, g and the context vector so that neither gxp nor nrm is materialised
        // (the upstream gradient is unit * kf * gf: gf the sign vector as floats, kf = -1 / +1 / 0 for a negative, 1 for the sums)
        auto context_grad = [&](const float (&x)[E], const float (&xp)[E], const float (&gf)[E], float kf, float aa, float inv, bool uc) {
            const float uk = a.unit * kf;
            float al = dot16<E>(xp, gf);
            al = uc ? inv * uk * al : 0.f;                // <nrm, G>
            const float i2a = inv * inv * al;             // gxp = (inv unit) g - i2a xp
            if constexpr (MODEL == KGE_TRANSH) {
                const float qq = dot16<E>(cw, gf);
                const float d = inv * uk * qq - i2a * (aa * (1.0f - ww));       // gxp . w^   (xp . w^ = a (1 - |w^|^2))
                const float c1 = d - aa * i2a, c2 = aa * inv * uk;              // d x + a gxp = c1 x + c2 g + (a^2 i2a) w^
#pragma unroll
                for (int e = 0; e < E; e++) acw[e] += c1 * x[e] + c2 * gf[e];
                acw_s += aa * aa * i2a;
            } else {
                const float c1 = -aa * i2a, c2 = aa * inv * uk;                 // a gxp = c2 g - a i2a (x + a r_p)
#pragma unroll
                for (int e = 0; e < E; e++) acw[e] += c1 * x[e] + c2 * gf[e];
                acw_s += -aa * aa * i2a;
            }
        };
        for (int k0 = 0; k0 < (int)a.n_neg; k0 += PT) {
            // this round's ids: negative k0 + lane in lane `lane`
            int my_code = 0, my_row = 0, my_dst = -1;
            if (k0 + lane < (int)a.n_neg) {
                const long long j = b + (long long)(k0 + lane + 1) * a.stride;
                const int nh = a.bh[j], nt = a.bt[j];
                my_code = nh != h ? 0 : 1;
                my_row = nh != h ? nh : nt;
            }
            const int in_round = min(PT, (int)a.n_neg - k0);
            // one negative: forward, hinge, sign record, integer sums, relation-context gradient
            auto score_negative = [&](const float (&x)[E], const float (&xa)[E], int row, int code, int kk) {
                float xp[E];
                float aa, inv; bool uc;
                project<MODEL, E>(x, xa, cw, xp, aa, inv, uc);
                // new head (0): e = x^n + (r^ - t^n) ; new tail (1): e = (h^n + r^) - x^n
                const float f = code == 1 ? -inv : inv;
                float ev[E];
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const float4 Bq = hinge_const[code][q][threadIdx.x];
                    ev[4 * q] = fmaf(f, xp[4 * q], Bq.x); ev[4 * q + 1] = fmaf(f, xp[4 * q + 1], Bq.y);
                    ev[4 * q + 2] = fmaf(f, xp[4 * q + 2], Bq.z); ev[4 * q + 3] = fmaf(f, xp[4 * q + 3], Bq.w);
                    acc += fabsf(ev[4 * q]) + fabsf(ev[4 * q + 1]) + fabsf(ev[4 * q + 2]) + fabsf(ev[4 * q + 3]);
                }
                const float nk = team_sum<PT>(acc);
                const float v = p - nk + a.margin;
                const bool act = v >= 0.f;
                if (act) { cnt++; lsum += v; }
                // g = d(loss)/d(x^n) / unit : -s for a new head, +s for a new tail; 0 when the hinge is inactive
                const int kx = act ? (code == 1 ? 1 : -1) : 0;
                float gf[E];
                {
                    const s16x2 kh = pack16(code == 1 && act ? -1 : 0, code == 1 && act ? -1 : 0);   // kept h of a new-tail pair gets -s
                    const s16x2 kt = pack16(code == 0 && act ? 1 : 0, code == 0 && act ? 1 : 0);     // kept t of a new-head pair gets +s
                    // a negative's record holds signs only: 2 bits per element (00 / 01 / 11 = 0 / +1 / -1), the lane's 4*Q elements
                    // in ONE dword -- a 64-byte record instead of 16*Q*4 bytes (the pair segsum is bound by the record bytes it gathers)
                    uint32_t word = 0u;
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const int s0 = sign_of_bits(ev[4 * q]), s1 = sign_of_bits(ev[4 * q + 1]);
                        const int s2 = sign_of_bits(ev[4 * q + 2]), s3 = sign_of_bits(ev[4 * q + 3]);
                        gf[4 * q] = (float)s0; gf[4 * q + 1] = (float)s1; gf[4 * q + 2] = (float)s2; gf[4 * q + 3] = (float)s3;
                        const s16x2 s_lo = pack16(s0, s1), s_hi = pack16(s2, s3);
                        word |= ((uint32_t)(s0 & 3) << (8 * q)) | ((uint32_t)(s1 & 3) << (8 * q + 2)) | ((uint32_t)(s2 & 3) << (8 * q + 4)) |
                                ((uint32_t)(s3 & 3) << (8 * q + 6));
                        Ah_lo[q] += s_lo * kh; Ah_hi[q] += s_hi * kh;
                        At_lo[q] += s_lo * kt; At_hi[q] += s_hi * kt;
                    }
                    if (act) {
                        if (kx < 0) word ^= (word & 0x55555555u) << 1;   // negate every 2-bit field: 01 <-> 11, 00 stays
                        const long long m = (long long)(2 + k0 + kk) * a.n_pos + b;
                        a.rec[2 * a.n_pos * (long long)(PT * Q) + ((long long)(k0 + kk) * a.n_pos + b) * PT + lane] = word;
                        if (lane == 0) a.pair_aux[m] = make_float2(aa, uc ? inv : -inv);
                    }
                }
                context_grad(x, xp, gf, (float)kx, aa, inv, uc);   // (all zero when inactive: kx = 0)
                if (lane == kk) my_dst = act ? (int)((long long)row * R + r) : -1;
            };
            // Two row buffers used alternately (the loop is unrolled by two, so no register moves): while a negative is scored
            // the next one's row(s) are on their way.  A deeper queue did not pay (measured): the kernel is bound by issue, not
            // by the number of gathers in flight.
            float xA[E], xaA[E], xB[E], xaB[E];
            auto fetch = [&](int kk, float (&x)[E], float (&xa)[E], int &row, int &code) {
                const int src = min(kk, in_round - 1);
                row = __shfl(my_row, src, PT); code = __shfl(my_code, src, PT);
                load_row<Q>(a.ent, row, D, lane, x);
                if constexpr (MODEL == KGE_TRANSD) load_row<Q>(a.auxe, row, D, lane, xa);
            };
            int rowA, codeA, rowB, codeB;
            fetch(0, xA, xaA, rowA, codeA);
            for (int kk = 0; kk < in_round; kk += 2) {
                fetch(kk + 1, xB, xaB, rowB, codeB);
                score_negative(xA, xaA, rowA, codeA, kk);
                if (kk + 1 < in_round) {
                    fetch(kk + 2, xA, xaA, rowA, codeA);
                    score_negative(xB, xaB, rowB, codeB, kk + 1);
                }
            }
            if (k0 + lane < (int)a.n_neg) a.dst[(long long)(2 + k0 + lane) * a.n_pos + b] = my_dst;
        }
        // ---- the positive's own sides and the relation-side rows ----
        if (cnt == 0) {
            if (lane < 2) a.dst[(long long)lane * a.n_pos + b] = -1;
            continue;
        }
        {
            float rn[E];
            load_row<Q>(a.rel, r, D, lane, rn);
            const float ssr = dot16<E>(rn, rn);
            const bool uc_r = ssr >= 1e-12f;
            const float inv_r = 1.0f / sqrtf(uc_r ? ssr : 1e-12f);
#pragma unroll
            for (int e = 0; e < E; e++) rn[e] *= inv_r;
            float xh[E], xah[E], xph[E], xt[E], xat[E], xpt[E];
            float ah, ih, at, it; bool uch, uct;
            load_row<Q>(a.ent, h, D, lane, xh);
            if constexpr (MODEL == KGE_TRANSD) load_row<Q>(a.auxe, h, D, lane, xah);
            project<MODEL, E>(xh, xah, cw, xph, ah, ih, uch);
            load_row<Q>(a.ent, t, D, lane, xt);
            if constexpr (MODEL == KGE_TRANSD) load_row<Q>(a.auxe, t, D, lane, xat);
            project<MODEL, E>(xt, xat, cw, xpt, at, it, uct);
            // sign of the positive's e, then the three integer upstream sums
            float gh[E], gt[E], gr[E];
            uint32_t rech[Q], rect[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                int sp[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = 4 * q + j;
                    sp[j] = sign_of_bits(xph[e] * ih + rn[e] - xpt[e] * it + 0.0f);
                }
                const s16x2 c2 = pack16(cnt, cnt);
                const s16x2 v_lo = pack16(sp[0], sp[1]) * c2, v_hi = pack16(sp[2], sp[3]) * c2;
                const s16x2 h_lo = Ah_lo[q] + v_lo, h_hi = Ah_hi[q] + v_hi;
                const s16x2 t_lo = At_lo[q] - v_lo, t_hi = At_hi[q] - v_hi;
                const s16x2 r_lo = Ah_lo[q] - At_lo[q] + v_lo, r_hi = Ah_hi[q] - At_hi[q] + v_hi;
                rech[q] = bytes_of(h_lo, h_hi); rect[q] = bytes_of(t_lo, t_hi);
                gh[4 * q] = (float)h_lo.x; gh[4 * q + 1] = (float)h_lo.y; gh[4 * q + 2] = (float)h_hi.x; gh[4 * q + 3] = (float)h_hi.y;
                gt[4 * q] = (float)t_lo.x; gt[4 * q + 1] = (float)t_lo.y; gt[4 * q + 2] = (float)t_hi.x; gt[4 * q + 3] = (float)t_hi.y;
                gr[4 * q] = (float)r_lo.x; gr[4 * q + 1] = (float)r_lo.y; gr[4 * q + 2] = (float)r_hi.x; gr[4 * q + 3] = (float)r_hi.y;
            }
            {
                uint32_t *ph = a.rec + b * (long long)(PT * Q), *pt = a.rec + (a.n_pos + b) * (long long)(PT * Q);
#pragma unroll
                for (int q = 0; q < Q; q++) { ph[lane + PT * q] = rech[q]; pt[lane + PT * q] = rect[q]; }
                if (lane == 0) { a.dst[b] = (int)((long long)h * R + r); a.pair_aux[b] = make_float2(ah, uch ? ih : -ih); }
                if (lane == 1) { a.dst[a.n_pos + b] = (int)((long long)t * R + r); a.pair_aux[a.n_pos + b] = make_float2(at, uct ? it : -it); }
            }
            context_grad(xh, xph, gh, 1.0f, ah, ih, uch);
            context_grad(xt, xpt, gt, 1.0f, at, it, uct);
            // hub copy of the relation-side tables this group adds into
            const long long hub = a.copies_rel ? b % a.hub_k : 0;
            float *grel = a.copies_rel ? a.copies_rel + hub * R * D : a.g_rel;
            float *gaux = a.copies_auxr ? a.copies_auxr + hub * R * D : a.g_auxr;
            float g[E];
            {   // r^: normalise backward of unit * gr
                float d = dot16<E>(rn, gr) * a.unit;
                if (!uc_r) d = 0.f;
#pragma unroll
                for (int e = 0; e < E; e++) g[e] = inv_r * (a.unit * gr[e] - d * rn[e]);
                atomic_add_row<Q>(grel, r, D, lane, g, add_stage[team_in_block]);
            }
            if constexpr (MODEL == KGE_TRANSH) {   // w^: acw = -(vector part + acw_s w^), then the normalise backward of w
                float tot[E];
#pragma unroll
                for (int e = 0; e < E; e++) tot[e] = -(acw[e] + acw_s * cw[e]);
                float d = dot16<E>(cw, tot);
                if (!uc_w) d = 0.f;
#pragma unroll
                for (int e = 0; e < E; e++) g[e] = inv_w * (tot[e] - d * cw[e]);
            } else {                               // r_p: acw = vector part + acw_s r_p, no normalisation
#pragma unroll
                for (int e = 0; e < E; e++) g[e] = acw[e] + acw_s * cw[e];
            }
            atomic_add_row<Q>(gaux, r, D, lane, g, add_stage[team_in_block]);
        }
    }
    finish_loss<TEAMS>(a, red, lsum, lane, team_in_block);
}

bool pair_counts_shape_ok(int model, int D, int64_t n_neg) {
    return (model == KGE_TRANSH || model == KGE_TRANSD) && D % 4 == 0 && D >= 4 && D <= 256 && n_neg >= 1 && n_neg <= 63;
}

template <int MODEL>
static void launch_pair_emit_q(const FbArgs &a, long long blocks, hipStream_t stream) {
    const int Q = (a.D / 4 + PT - 1) / PT;
    switch (Q) {
        case 1: hipLaunchKernelGGL((pair_emit_kernel<MODEL, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((pair_emit_kernel<MODEL, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL((pair_emit_kernel<MODEL, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL((pair_emit_kernel<MODEL, 4>), dim3((unsigned)blocks), dim3(256), 0, stream, a); break;
    }
}

// blocks of the emit launch (its loss partials occupy loss_partials[0 .. blocks))
long long pair_emit_blocks(int64_t n_pos) {
    long long blocks = (n_pos + (256 / PT) - 1) / (256 / PT);
    if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
    if (blocks < 1) blocks = 1;
    return blocks;
}

int pair_record_dwords(int D) { return PT * ((D / 4 + PT - 1) / PT); }

void launch_pair_emit(int model, const FbArgs &a, hipStream_t stream) {
    const long long blocks = pair_emit_blocks(a.n_pos);
    if (model == KGE_TRANSH) launch_pair_emit_q<KGE_TRANSH>(a, blocks, stream);
    else launch_pair_emit_q<KGE_TRANSD>(a, blocks, stream);
}

}  // namespace kge
