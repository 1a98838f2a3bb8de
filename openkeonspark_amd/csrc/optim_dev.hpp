// Optimiser sweeps as device functions (same arithmetic for the one-launch-per-step kernels of optim.hip and for the
// persistent multi-step launch of persist.hip): SGD p -= lr*g and TF1-semantics Adam (_apply_sparse_shared op order).
#pragma once
#include "engine.hpp"

namespace kge {

// up to four tables per launch: blockIdx.y selects the table (one launch per optimizer step instead of one per table)
struct SweepTables {
    float *p[4], *g[4], *m[4], *v[4];
    long long n[4];
};

// p -= lr * g; g = 0 over one table, threads tid, tid + stride, ... (16 B per lane; untouched 16-byte groups are skipped)
__device__ __forceinline__ void sgd_sweep(float *__restrict__ p, float *__restrict__ g, long long n, float lr, long long tid, long long stride) {
    const long long n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p);
    float4 *g4 = reinterpret_cast<float4 *>(g);
    for (long long i = tid; i < n4; i += stride) {
        float4 gv = g4[i];
        if (gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f) {  // untouched rows: p - lr*0 == p, skip the stores
            float4 pv = p4[i];
            pv.x -= lr * gv.x; pv.y -= lr * gv.y; pv.z -= lr * gv.z; pv.w -= lr * gv.w;
            p4[i] = pv;
            g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (long long i = (n4 << 2) + tid; i < n; i += stride) {
        p[i] -= lr * g[i];
        g[i] = 0.f;
    }
}

__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, float lr_t, float b1, float b2, float eps) {
    // TF1 op order, each product rounded before the add (no fma contraction across the ops)
    float mi = __fmul_rn(m, b1);
    float vi = __fmul_rn(v, b2);
    if (g != 0.f) {
        mi = __fadd_rn(mi, __fmul_rn(g, 1.0f - b1));
        vi = __fadd_rn(vi, __fmul_rn(__fmul_rn(g, g), 1.0f - b2));
    }
    m = mi; v = vi;
    p = __fsub_rn(p, __fdiv_rn(__fmul_rn(lr_t, mi), __fadd_rn(__fsqrt_rn(vi), eps)));
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
