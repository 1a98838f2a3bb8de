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
#include "engine.hpp"

namespace kge {

// ------------------------------------------------------------------------------------------------
// team reductions
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// Sum over the L lanes of the caller's team; every lane of the team receives the total.
template <int L>
__device__ __forceinline__ float team_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2] : lane ^ 1
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1] : lane ^ 2
    v += dpp_f<0x124>(v);  // row_ror:4  (rotations keep the 16-lane row sum uniform)
    v += dpp_f<0x128>(v);  // row_ror:8
    if constexpr (L >= 32) {
        // ds_swizzle bit mode: and=0x1F, or=0, xor=0x10 -> lane ^ 16 inside each 32-lane half
        v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (0x10 << 10) | 0x1F));
    }
    if constexpr (L == 64) {
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    }
    return v;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f ? 1.f : 0.f) - (x < 0.f ? 1.f : 0.f); }

struct FbArgs {
    const float *ent, *rel, *auxr, *auxe;  // tables
    float *g_ent, *g_rel, *g_auxr, *g_auxe;  // dense gradient accumulators
    const int32_t *bh, *bt, *br;
    long long n_pos, n_neg, stride;
    int D;
    float margin, unit;
    float *loss_partials;
    // TransR vector stage (transr.hip): entity sides are rows of the projected buffer P (one row per
    // canonical (scored triple, side) slot) and their gradients are STORED to GP, not added to g_ent
    const float *P;
    float *GP;
    int negative_rel;
};

template <int L, int C>
struct Team {
    int lane;  // lane inside the team
    int D;
    __device__ __forceinline__ void load(const float *__restrict__ tab, long long row, float (&x)[C]) const {
        const float *p = tab + row * D;
#pragma unroll
        for (int c = 0; c < C; c++) { int e = lane + L * c; x[c] = e < D ? p[e] : 0.f; }
    }
    __device__ __forceinline__ void add(float *__restrict__ tab, long long row, const float (&v)[C]) const {
        float *p = tab + row * D;
#pragma unroll
        for (int c = 0; c < C; c++) {
            int e = lane + L * c;
            if (e < D) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(p + e), v[c]);
        }
    }
    __device__ __forceinline__ void store(float *__restrict__ tab, long long row, const float (&v)[C]) const {
        float *p = tab + row * D;
#pragma unroll
        for (int c = 0; c < C; c++) { int e = lane + L * c; if (e < D) p[e] = v[c]; }
    }
    __device__ __forceinline__ float dot(const float (&a)[C], const float (&b)[C]) const {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) s += a[c] * b[c];
        return team_sum<L>(s);
    }
    // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))  (TransE.py:12-14)
    __device__ __forceinline__ void normalize(const float (&x)[C], float (&y)[C], float &inv, bool &unclipped) const {
        float ss = dot(x, x);
        unclipped = ss >= 1e-12f;
        inv = 1.0f / sqrtf(unclipped ? ss : 1e-12f);
#pragma unroll
        for (int c = 0; c < C; c++) y[c] = x[c] * inv;
    }
    // backward of normalize: gx = inv * (gy - [unclipped] y <y,gy>)
    __device__ __forceinline__ void normalize_bwd(const float (&y)[C], const float (&gy)[C], float inv, bool unclipped,
                                                  float (&gx)[C]) const {
        float d = dot(y, gy);
        if (!unclipped) d = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) gx[c] = inv * (gy[c] - d * y[c]);
    }
};

// One entity side (h or t slot) of a scored triple: raw row(s), projected+normalised vector.
template <int C>
struct Side {
    float raw[C];  // ent_embeddings row          (unused after projection for TransE)
    float aux[C];  // ent_transfer row             (TransD only)
    float nrm[C];  // l2_normalize(projected)
    float inv, a;  // 1/|projected| , projection coefficient (e.w  or  e.e_p)
    bool uc;
};

// Projection onto the relation context.  TransH.py:12-14: e - (e.w^)w^ ; TransD.py:23-25: e + (e.e_p) r_p
template <int MODEL, int L, int C>
__device__ __forceinline__ void side_forward(const Team<L, C> &tm, const FbArgs &a, long long row, const float (&cw)[C],
                                             Side<C> &s) {
    if constexpr (MODEL == KGE_TRANSR) tm.load(a.P, row, s.raw);  // row = slot of the projected buffer
    else tm.load(a.ent, row, s.raw);
    float xp[C];
    if constexpr (MODEL == KGE_TRANSE || MODEL == KGE_TRANSR) {
#pragma unroll
        for (int c = 0; c < C; c++) xp[c] = s.raw[c];
        s.a = 0.f;
    } else if constexpr (MODEL == KGE_TRANSH) {
        s.a = tm.dot(s.raw, cw);
#pragma unroll
        for (int c = 0; c < C; c++) xp[c] = s.raw[c] - s.a * cw[c];
    } else {
        tm.load(a.auxe, row, s.aux);
        s.a = tm.dot(s.raw, s.aux);
#pragma unroll
        for (int c = 0; c < C; c++) xp[c] = s.raw[c] + s.a * cw[c];
    }
    tm.normalize(xp, s.nrm, s.inv, s.uc);
}

// Backward of one entity side given G = dL/d(normalised projected vector).  Adds the row gradient(s)
// and accumulates the relation-context gradient into acw (TransH: d/dw^, TransD: d/dr_p).
template <int MODEL, int L, int C>
__device__ __forceinline__ void side_backward(const Team<L, C> &tm, const FbArgs &a, long long row, const Side<C> &s,
                                              const float (&G)[C], const float (&cw)[C], float (&acw)[C]) {
    float gxp[C];
    tm.normalize_bwd(s.nrm, G, s.inv, s.uc, gxp);
    if constexpr (MODEL == KGE_TRANSE) {
        tm.add(a.g_ent, row, gxp);
    } else if constexpr (MODEL == KGE_TRANSR) {
        tm.store(a.GP, row, gxp);  // each canonical slot is written by exactly one team
    } else if constexpr (MODEL == KGE_TRANSH) {
        float d = tm.dot(gxp, cw);
        float gx[C];
#pragma unroll
        for (int c = 0; c < C; c++) { gx[c] = gxp[c] - d * cw[c]; acw[c] -= d * s.raw[c] + s.a * gxp[c]; }
        tm.add(a.g_ent, row, gx);
    } else {
        float d = tm.dot(gxp, cw);
        float gx[C], gv[C];
#pragma unroll
        for (int c = 0; c < C; c++) { gx[c] = gxp[c] + d * s.aux[c]; gv[c] = d * s.raw[c]; acw[c] += s.a * gxp[c]; }
        tm.add(a.g_ent, row, gx);
        tm.add(a.g_auxe, row, gv);
    }
}

// Relation context of a triple: normalised relation vector + projection vector.
template <int C>
struct Ctx {
    float rn[C];  // l2_normalize(rel_embeddings[r])
    float cw[C];  // TransH: l2_normalize(normal_vectors[r]); TransD: rel_transfer[r]
    float inv_r, inv_w;
    bool uc_r, uc_w;
};

template <int MODEL, int L, int C>
__device__ __forceinline__ void ctx_forward(const Team<L, C> &tm, const FbArgs &a, long long r, Ctx<C> &cx) {
    float raw[C];
    tm.load(a.rel, r, raw);
    tm.normalize(raw, cx.rn, cx.inv_r, cx.uc_r);
    cx.inv_w = 1.f; cx.uc_w = true;
    if constexpr (MODEL == KGE_TRANSH) {
        tm.load(a.auxr, r, raw);
        tm.normalize(raw, cx.cw, cx.inv_w, cx.uc_w);
    } else if constexpr (MODEL == KGE_TRANSD) {
        tm.load(a.auxr, r, cx.cw);
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) cx.cw[c] = 0.f;
    }
}

// Adds the relation-side gradients: Gr = dL/d rn, acw = accumulated dL/d cw.
template <int MODEL, int L, int C>
__device__ __forceinline__ void ctx_backward(const Team<L, C> &tm, const FbArgs &a, long long r, const Ctx<C> &cx,
                                             const float (&Gr)[C], const float (&acw)[C]) {
    float g[C];
    tm.normalize_bwd(cx.rn, Gr, cx.inv_r, cx.uc_r, g);
    tm.add(a.g_rel, r, g);
    if constexpr (MODEL == KGE_TRANSH) {
        tm.normalize_bwd(cx.cw, acw, cx.inv_w, cx.uc_w, g);
        tm.add(a.g_auxr, r, g);
    } else if constexpr (MODEL == KGE_TRANSD) {
        tm.add(a.g_auxr, r, acw);
    }
}

template <int L, int C>
__device__ __forceinline__ float l1_score(const Team<L, C> &tm, const float (&hn)[C], const float (&rn)[C],
                                          const float (&tn)[C], float (&sg)[C]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) { float e = hn[c] + rn[c] - tn[c]; s += fabsf(e); sg[c] = sgn(e); }
    return team_sum<L>(s);
}

// A negative that the fast path cannot use (more than one slot differs, or the relation differs
// for a projecting model): score it from scratch and, if its hinge is active, add all its gradients
// separately (gs = -unit).  Kept frugal in registers -- the entity sides are re-gathered for the
// backward instead of being held -- because this rare path must not cost the common one occupancy.
template <int MODEL, int L, int C>
__device__ __forceinline__ bool standalone_negative(const Team<L, C> &tm, const FbArgs &a, long long nh, long long nt,
                                                    long long nr, float p, float &hinge) {
    // nh / nt are ROW HANDLES (entity ids, or projected-buffer slots for TransR); nr the relation id
    Ctx<C> cx;
    ctx_forward<MODEL, L, C>(tm, a, nr, cx);
    float sg[C];
    float nk;
    {
        float hn[C];
        {
            Side<C> sx;
            side_forward<MODEL, L, C>(tm, a, nh, cx.cw, sx);
#pragma unroll
            for (int c = 0; c < C; c++) hn[c] = sx.nrm[c];
        }
        Side<C> sx;
        side_forward<MODEL, L, C>(tm, a, nt, cx.cw, sx);
        nk = l1_score<L, C>(tm, hn, cx.rn, sx.nrm, sg);
    }
    float v = p - nk + a.margin;
    if (!(v >= 0.f)) { hinge = 0.f; return false; }
    hinge = v;
    float G[C], acw[C];
#pragma unroll
    for (int c = 0; c < C; c++) { G[c] = -a.unit * sg[c]; acw[c] = 0.f; }
    {
        Side<C> sx;
        side_forward<MODEL, L, C>(tm, a, nh, cx.cw, sx);
        side_backward<MODEL, L, C>(tm, a, nh, sx, G, cx.cw, acw);
    }
    {
        Side<C> sx;
        side_forward<MODEL, L, C>(tm, a, nt, cx.cw, sx);
        float Gt[C];
#pragma unroll
        for (int c = 0; c < C; c++) Gt[c] = -G[c];
        side_backward<MODEL, L, C>(tm, a, nt, sx, Gt, cx.cw, acw);
    }
    ctx_backward<MODEL, L, C>(tm, a, nr, cx, G, acw);
    return true;
}

template <int MODEL, int L, int C>
__global__ __launch_bounds__(256) void fwdbwd_kernel(FbArgs a) {
    constexpr int TEAMS = 256 / L;
    __shared__ float red[TEAMS];
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = a.D;
    const int team_in_block = threadIdx.x / L;
    float lsum = 0.f;
    for (long long b = (long long)blockIdx.x * TEAMS + team_in_block; b < a.n_pos; b += (long long)gridDim.x * TEAMS) {
        const long long h = a.bh[b], t = a.bt[b], r = a.br[b];
        Ctx<C> cx;
        ctx_forward<MODEL, L, C>(tm, a, r, cx);
        // row handles of the two entity sides: the entity id, or (TransR) the slot of the projected
        // vector of (scored triple s, side): 2*s + side with s = k*n_pos + b
        const long long row_h = MODEL == KGE_TRANSR ? 2 * b : h;
        const long long row_t = MODEL == KGE_TRANSR ? 2 * b + 1 : t;
        Side<C> sh, st;
        side_forward<MODEL, L, C>(tm, a, row_h, cx.cw, sh);
        side_forward<MODEL, L, C>(tm, a, row_t, cx.cw, st);
        float sp[C];
        const float p = l1_score<L, C>(tm, sh.nrm, cx.rn, st.nrm, sp);
        // gradients w.r.t. the three shared normalised vectors, in units of `unit`
        float Ah[C], At[C], Ar[C], acw[C];
#pragma unroll
        for (int c = 0; c < C; c++) { Ah[c] = 0.f; At[c] = 0.f; Ar[c] = 0.f; acw[c] = 0.f; }
        int cnt = 0;
        for (long long k = 0; k < a.n_neg; k++) {
            const long long j = b + (k + 1) * a.stride;
            const long long nh = a.bh[j], nt = a.bt[j], nr = a.br[j];
            const NegClass nc = classify_negative<MODEL>(h, t, r, nh, nt, nr, a.negative_rel);
            const long long s_neg = (k + 1) * a.n_pos + b;
            const long long nrow_h = MODEL == KGE_TRANSR ? 2 * s_neg : nh;
            const long long nrow_t = MODEL == KGE_TRANSR ? 2 * s_neg + 1 : nt;
            if (!nc.fast) {
                float hinge;
                if (standalone_negative<MODEL, L, C>(tm, a, nrow_h, nrow_t, nr, p, hinge)) { cnt++; lsum += hinge; }
                continue;
            }
            float sg[C];
            if (!nc.same_h) {  // head corrupted (corrupt_tail keeps t, Base.cpp:123-126)
                Side<C> sx;
                side_forward<MODEL, L, C>(tm, a, nrow_h, cx.cw, sx);
                float nk = l1_score<L, C>(tm, sx.nrm, cx.rn, st.nrm, sg);
                float v = p - nk + a.margin;
                if (v >= 0.f) {
                    cnt++; lsum += v;
                    float G[C];
#pragma unroll
                    for (int c = 0; c < C; c++) { G[c] = -a.unit * sg[c]; At[c] += sg[c]; Ar[c] -= sg[c]; }
                    side_backward<MODEL, L, C>(tm, a, nrow_h, sx, G, cx.cw, acw);
                }
            } else if (!nc.same_t) {  // tail corrupted (corrupt_head keeps h, Base.cpp:119-121)
                Side<C> sx;
                side_forward<MODEL, L, C>(tm, a, nrow_t, cx.cw, sx);
                float nk = l1_score<L, C>(tm, sh.nrm, cx.rn, sx.nrm, sg);
                float v = p - nk + a.margin;
                if (v >= 0.f) {
                    cnt++; lsum += v;
                    float G[C];
#pragma unroll
                    for (int c = 0; c < C; c++) { G[c] = a.unit * sg[c]; Ah[c] -= sg[c]; Ar[c] -= sg[c]; }
                    side_backward<MODEL, L, C>(tm, a, nrow_t, sx, G, cx.cw, acw);
                }
            } else {  // relation vector corrupted while both projected entities are shared (TransE; TransR with negative_rel == 0)
                float raw[C], xn[C], inv; bool uc;
                tm.load(a.rel, nr, raw);
                tm.normalize(raw, xn, inv, uc);
                float nk = l1_score<L, C>(tm, sh.nrm, xn, st.nrm, sg);
                float v = p - nk + a.margin;
                if (v >= 0.f) {
                    cnt++; lsum += v;
                    float G[C], g[C];
#pragma unroll
                    for (int c = 0; c < C; c++) { G[c] = -a.unit * sg[c]; Ah[c] -= sg[c]; At[c] += sg[c]; }
                    tm.normalize_bwd(xn, G, inv, uc, g);
                    tm.add(a.g_rel, nr, g);
                }
            }
        }
        if (cnt > 0) {
            const float fc = (float)cnt;
            float Gh[C], Gt[C], Gr[C];
#pragma unroll
            for (int c = 0; c < C; c++) {
                Gh[c] = a.unit * (Ah[c] + fc * sp[c]);
                Gt[c] = a.unit * (At[c] - fc * sp[c]);
                Gr[c] = a.unit * (Ar[c] + fc * sp[c]);
            }
            side_backward<MODEL, L, C>(tm, a, row_h, sh, Gh, cx.cw, acw);
            side_backward<MODEL, L, C>(tm, a, row_t, st, Gt, cx.cw, acw);
            ctx_backward<MODEL, L, C>(tm, a, r, cx, Gr, acw);
        }
    }
    if (tm.lane == 0) red[team_in_block] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TEAMS; i++) s += red[i];
        a.loss_partials[blockIdx.x] = s;
    }
}

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

template <int MODEL, int L, int C>
static void launch_fb(const FbArgs &a, float *d_loss, hipStream_t stream) {
    constexpr int TEAMS = 256 / L;
    long long blocks = (a.n_pos + TEAMS - 1) / TEAMS;
    if (blocks > kMaxLossBlocks) blocks = kMaxLossBlocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fwdbwd_kernel<MODEL, L, C>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, a.loss_partials, (int)blocks, a.unit, d_loss);
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

// TransR: the score / hinge / backward over the projected vectors (transr.hip runs the GEMMs around it)
int launch_transr_vector_stage(const float *rel, float *g_rel, const float *P, float *GP, const int32_t *d_h,
                               const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                               int64_t denom, int rel_dim, float margin, int negative_rel, float *d_loss,
                               hipStream_t stream) {
    Engine &e = engine();
    FbArgs a = {};
    a.rel = rel; a.g_rel = g_rel; a.P = P; a.GP = GP;
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.n_pos = n_pos; a.n_neg = n_neg; a.stride = stride;
    a.D = rel_dim; a.margin = margin; a.unit = 1.0f / (float)denom;
    a.loss_partials = e.dev.loss_partials;
    a.negative_rel = negative_rel;
    int rc = dispatch_fb<KGE_TRANSR>(a, d_loss, stream);
    if (rc) return rc;
    return hip_check(hipGetLastError(), "transr vector stage launch");
}

int launch_forward_backward_transr(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h,
                                   const int32_t *d_t, const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride,
                                   int64_t denom, float *const grads[4], float *d_loss, hipStream_t stream);

int launch_forward_backward(const kge_model_desc &m, const float *const tables[4], const int32_t *d_h, const int32_t *d_t,
                            const int32_t *d_r, int64_t n_pos, int64_t n_neg, int64_t stride, int64_t denom,
                            float *const grads[4], float *d_loss, hipStream_t stream) {
    Engine &e = engine();
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_forward_backward: no usable HIP device");
    if (n_pos < 0 || n_neg < 1 || stride < n_pos || denom <= 0) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward: bad sizes");
    if (!e.dev.loss_partials) {
        int rc = hip_check(hipMalloc(&e.dev.loss_partials, sizeof(float) * kMaxLossBlocks), "alloc loss partials");
        if (rc) return rc;
    }
    if (m.model == KGE_TRANSR)
        return launch_forward_backward_transr(m, tables, d_h, d_t, d_r, n_pos, n_neg, stride, denom, grads, d_loss, stream);
    if (m.ent_dim != m.rel_dim) return fail(KGE_ERR_BAD_ARG, "TransE/H/D need ent_dim == rel_dim (hidden_size)");
    FbArgs a;
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3];
    a.g_ent = grads[0]; a.g_rel = grads[1]; a.g_auxr = grads[2]; a.g_auxe = grads[3];
    a.bh = d_h; a.bt = d_t; a.br = d_r;
    a.n_pos = n_pos; a.n_neg = n_neg; a.stride = stride;
    a.D = m.ent_dim; a.margin = m.margin; a.unit = 1.0f / (float)denom;
    a.loss_partials = e.dev.loss_partials;
    a.P = nullptr; a.GP = nullptr; a.negative_rel = m.negative_rel;
    int rc;
    switch (m.model) {
        case KGE_TRANSE: rc = dispatch_fb<KGE_TRANSE>(a, d_loss, stream); break;
        case KGE_TRANSH: rc = dispatch_fb<KGE_TRANSH>(a, d_loss, stream); break;
        case KGE_TRANSD: rc = dispatch_fb<KGE_TRANSD>(a, d_loss, stream); break;
        default: return fail(KGE_ERR_BAD_ARG, "unknown model id");
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

}  // namespace kge
