// Micro-benchmark (not part of the product): how fast can gradient rows be scatter-added?
// Pattern = the fwd/bwd kernel's: one wave adds one 800-byte row (D=200 fp32) into a random row of a
// [E=14541, 200] table; M rows in total.  Variants: agent-scope fp32 atomics (baseline), XCD-private
// copies with workgroup/wavefront-scope atomics, int32 atomics, plain stores (upper bound).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

template <int MODE>
__global__ __launch_bounds__(256) void scatter_kernel(float *tab, const int *rows, long long M, int D, long long copy_stride) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    float *base = tab;
    if (MODE == 1 || MODE == 2) base = tab + (long long)xcc_id() * copy_stride;
    for (long long m = wave; m < M; m += nwaves) {
        const int row = rows[m];
        float *p = base + (long long)row * D;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int e = lane + 64 * c;
            if (e < D) {
                float v = 1.0f;
                if (MODE == 0) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(p + e), v);
                else if (MODE == 1) __hip_atomic_fetch_add(p + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 2) __hip_atomic_fetch_add(p + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else if (MODE == 3) atomicAdd((int *)(p + e), 1);
                else if (MODE == 4) p[e] = v;
                else if (MODE == 5) __hip_atomic_fetch_add((int *)(p + e), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

int main(int argc, char **argv) {
    const int E = 14541, D = 200;
    const long long M = 34014LL * 28;
    std::vector<int> rows(M);
    srand(1);
    for (auto &r : rows) r = rand() % E;
    int *d_rows; float *d_tab;
    const long long copy_stride = (long long)E * D;
    CK(hipMalloc(&d_rows, M * sizeof(int)));
    CK(hipMalloc(&d_tab, 8 * copy_stride * sizeof(float)));
    CK(hipMemcpy(d_rows, rows.data(), M * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"f32 agent-scope atomic (baseline)", "f32 workgroup-scope, XCD-private copy", "f32 wavefront-scope, XCD-private copy",
                           "i32 agent-scope atomic", "plain store (racy upper bound)", "i32 workgroup-scope, XCD-private copy"};
    std::vector<float> host(8 * copy_stride);
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e9, sum_ms = 0;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemset(d_tab, 0, 8 * copy_stride * sizeof(float)));
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: scatter_kernel<0><<<2048, 256>>>(d_tab, d_rows, M, D, copy_stride); break;
                case 1: scatter_kernel<1><<<2048, 256>>>(d_tab, d_rows, M, D, copy_stride); break;
                case 2: scatter_kernel<2><<<2048, 256>>>(d_tab, d_rows, M, D, copy_stride); break;
                case 3: scatter_kernel<3><<<2048, 256>>>(d_tab, d_rows, M, D, copy_stride); break;
                case 4: scatter_kernel<4><<<2048, 256>>>(d_tab, d_rows, M, D, copy_stride); break;
                case 5: scatter_kernel<5><<<2048, 256>>>(d_tab, d_rows, M, D, copy_stride); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) { best = std::min(best, ms); sum_ms += ms; }
        }
        // correctness: total over all copies must equal M*D
        CK(hipMemcpy(host.data(), d_tab, 8 * copy_stride * sizeof(float), hipMemcpyDeviceToHost));
        double total = 0; int used_copies = 0;
        for (int c = 0; c < 8; c++) {
            double t = 0;
            for (long long i = 0; i < copy_stride; i++) t += (mode == 3 || mode == 5) ? (double)((int *)host.data())[c * copy_stride + i] : (double)host[c * copy_stride + i];
            if (t != 0) used_copies++;
            total += t;
        }
        double bytes = (double)M * D * 4;
        printf("%-45s best %.3f ms  avg %.3f ms  -> %.2f TB/s added   sum %s (%.0f vs %.0f) copies_used=%d\n", names[mode], best, sum_ms / 5,
               bytes / (best * 1e-3) / 1e12, (mode == 4 || total == (double)M * D) ? "OK" : "MISMATCH", total, (double)M * D, used_copies);
    }
    return 0;
}
