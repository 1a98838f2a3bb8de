// Device-side team primitives shared by the vector-model kernels (models.hip) and the TransE
// sign-count path (transe_counts.hip): a TEAM of L lanes (16/32/64) owns one embedding vector, lane l
// holding elements l, l+L, l+2L, ...; reductions over the embedding dimension are DPP / swizzle
// butterflies, no LDS traffic.
#pragma once
#include <hip/hip_runtime.h>

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
    if constexpr (L == 32) {
        // ds_swizzle bit mode: and=0x1F, or=0, xor=0x10 -> lane ^ 16 inside each 32-lane half
        v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (0x10 << 10) | 0x1F));
    }
    if constexpr (L == 64) {
        // wave-wide: row_bcast:15 adds row 0's sum into row 1 and row 2's into row 3, row_bcast:31 adds lane 31 (rows 0+1)
        // into rows 2,3; lane 63 then holds the total.  All DPP: no ds_swizzle (an LDS-path op with an lgkmcnt wait in
        // the middle of every reduction), one v_readlane instead of two.
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    }
    return v;
}

// 1/sqrt(x) for x >= 1e-12: the hardware estimate (1 ulp) refined by one Newton step -- 5 instructions where the IEEE
// sqrt + divide sequence takes ~25 (it sits in the per-negative path of the pair-count kernels; results agree with
// 1.0f / sqrtf(x) to the last bit or two, far inside the 1e-5 parity tolerance)
__device__ __forceinline__ float fast_rsqrt(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    return y * (1.5f - 0.5f * x * y * y);
}

// 1/|row| as the TransE emit kernel's per-row table holds it (tf.nn.l2_normalize's rsqrt(max(sum x^2, 1e-12)), TransE.py:12-14),
// from a row held in the team layout (lane l: elements l, l+L, ...; padding elements are 0).  ONE definition for the table's
// pre-pass (row_inv_norm_kernel) and for the apply kernel that refreshes the entry of every row it rewrites: a step gets the
// same bits whichever of the two produced its table (a resumed run recomputes it, an uninterrupted run carries it over).
template <int L, int C>
__device__ __forceinline__ float row_inv_norm(const float (&x)[C]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) s = __fmaf_rn(x[c], x[c], s);
    s = team_sum<L>(s);
    return 1.0f / sqrtf(s >= 1e-12f ? s : 1e-12f);
}

// Single operations rounded on their own, which the optimizer may NOT contract into fma whatever they are inlined next to.
// (HIP's __fmul_rn / __fadd_rn / __fsub_rn are plain `*` `+` `-` carrying the translation unit's contraction licence --
// __clang_hip_math.h -- so a chain of them comes out fused in one kernel and unfused in the next: seen as one-ulp differences
// between two kernels sharing one update function.  The pragma must sit where the operator is, not in the caller.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f ? 1.f : 0.f) - (x < 0.f ? 1.f : 0.f); }

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
        // explicit fma chain: `s += a*b` is contracted or not as the optimizer sees fit in each inlining context, and kernels that
        // must agree bit for bit (apply / fused segmented-sum-and-apply, transe_counts.hip) share this function
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) s = __builtin_fmaf(a[c], b[c], s);
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


}  // namespace kge
