// Per-row update of the TransE sign-count path from a row's summed integer counts -- shared by the reducers
// (transe_counts.hip) and by the emit kernel's in-place update of rows that a step touches exactly once (models.hip), so that a
// row gets the same bits whichever of them handles it.
#pragma once
#include "team.hpp"

namespace kge {

// d/dx of the normalised row applied to the integer sign sum: unit * (1/|x|) * (S - x^ <x^,S>), with every
// operation individually rounded so that all apply kernels agree bit for bit given the same reduction order
__device__ __forceinline__ float count_grad(float unit, float inv, float s, float d, float xn) {
    return __fmul_rn(__fmul_rn(unit, inv), __fsub_rn(s, __fmul_rn(d, xn)));
}

// Sparse-row SGD on ONE row from its summed integer counts held in the NATURAL layout of the vectorised kernels
// (accumulator c of lane l = element 4*(l + L*(c/4)) + c%4; D % 4 == 0): the single arithmetic used by the fused
// segmented-sum-and-apply kernel and by the row-list apply kernel, so a row gets the same bits whichever of them
// handles it (which one does depends on where chunk boundaries fall, i.e. on the number of ranks).
template <int L, int C>
__device__ __forceinline__ void apply_row_nat(const int (&acc)[C], float *__restrict__ p, int D, int lane, float unit, float lr) {
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
                const float g = __fadd_rn(count_grad(unit, inv, sv[c], d, x[c] * inv), 0.f);
                o[j] = g != 0.f ? __fsub_rn(x[c], __fmul_rn(lr, g)) : x[c];
                any = any || g != 0.f;
            }
            if (any) *reinterpret_cast<float4 *>(p + e0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}


}  // namespace kge
