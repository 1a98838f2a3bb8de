// Optimiser sweeps as device functions (same arithmetic for the one-launch-per-step kernels of optim.hip and for the
// persistent multi-step launch of persist.hip): SGD p -= lr*g and TF1-semantics Adam (_apply_sparse_shared op order).
#pragma once
#include "engine.hpp"
#include "team.hpp"

namespace kge {

// up to four tables per launch: blockIdx.y selects the table (one launch per optimizer step instead of one per table)
struct SweepTables {
    float *p[4], *g[4], *m[4], *v[4];
    long long n[4];
};

// p -= lr * g; g = 0 over one table, threads tid, tid + stride, ... (16 B per lane; untouched 16-byte groups are skipped).
// U groups per thread are loaded before any is processed (U x 16 B in flight per lane): the one-launch kernel has enough
// threads to cover the memory latency with U = 1; the persistent launch (one workgroup per CU) needs the bytes in flight per
// thread instead.  Same arithmetic per element whatever U.
template <int U = 1>
__device__ __forceinline__ void sgd_sweep(float *__restrict__ p, float *__restrict__ g, long long n, float lr, long long tid, long long stride) {
    const long long n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p);
    float4 *g4 = reinterpret_cast<float4 *>(g);
    for (long long i0 = tid; i0 < n4; i0 += stride * U) {
        float4 gv[U], pv[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long i = i0 + stride * u;
            gv[u] = g4[i < n4 ? i : i0];
            live[u] = i < n4 && (gv[u].x != 0.f || gv[u].y != 0.f || gv[u].z != 0.f || gv[u].w != 0.f);  // untouched rows: p - lr*0 == p, skip the stores
        }
#pragma unroll
        for (int u = 0; u < U; u++) pv[u] = p4[live[u] ? i0 + stride * u : i0];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (live[u]) {
                const long long i = i0 + stride * u;
                pv[u].x -= lr * gv[u].x; pv[u].y -= lr * gv[u].y; pv[u].z -= lr * gv[u].z; pv[u].w -= lr * gv[u].w;
                p4[i] = pv[u];
                g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) {
        p[i] -= lr * g[i];
        g[i] = 0.f;
    }
}

__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, float lr_t, float b1, float b2, float eps) {
    // TF1 op order, each product rounded before the add.  HIP's __fmul_rn / __fadd_rn are plain `*` / `+` (__clang_hip_math.h) and
    // would be contracted into fma where the optimizer sees fit -- differently from one inlining context to the next: contraction is
    // switched off for this body, so every caller gets the same bits
#pragma clang fp contract(off)
    float mi = mul_rn(m, b1);
    float vi = mul_rn(v, b2);
    if (g != 0.f) {
        mi = add_rn(mi, mul_rn(g, 1.0f - b1));
        vi = add_rn(vi, mul_rn(mul_rn(g, g), 1.0f - b2));
    }
    m = mi; v = vi;
    p = sub_rn(p, __fdiv_rn(mul_rn(lr_t, mi), add_rn(__fsqrt_rn(vi), eps)));
}

__device__ __forceinline__ void adam_sweep(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, float *__restrict__ g, long long n,
                                           float lr_t, float b1, float b2, float eps, long long tid, long long stride) {
    const long long n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m);
    float4 *v4 = reinterpret_cast<float4 *>(v), *g4 = reinterpret_cast<float4 *>(g);
    for (long long i = tid; i < n4; i += stride) {
        float4 pv = p4[i], mv = m4[i], vv = v4[i], gv = g4[i];
        adam_one(pv.x, mv.x, vv.x, gv.x, lr_t, b1, b2, eps);
        adam_one(pv.y, mv.y, vv.y, gv.y, lr_t, b1, b2, eps);
        adam_one(pv.z, mv.z, vv.z, gv.z, lr_t, b1, b2, eps);
        adam_one(pv.w, mv.w, vv.w, gv.w, lr_t, b1, b2, eps);
        p4[i] = pv; m4[i] = mv; v4[i] = vv;
        if (gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) {
        adam_one(p[i], m[i], v[i], g[i], lr_t, b1, b2, eps);
        g[i] = 0.f;
    }
}

}  // namespace kge
