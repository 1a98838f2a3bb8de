// Optimiser sweeps over a dense table and its dense summed-gradient accumulator.
//
// The reference applies `GradientDescentOptimizer` / `AdamOptimizer` to IndexedSlices
// (distribute_training.py:95-101).  SGD: scatter_sub of lr*g, duplicates accumulating -- i.e.
// p -= lr * (summed g).  Adam (TF1 _apply_sparse_shared): m and v decay for EVERY row, touched rows
// receive (1-beta)*g, then EVERY row moves -- a dense sweep of three tables per step (SURVEY.md
// A13).  Both are single streaming passes here: 16 B per lane, grid-stride, the accumulator is
// re-zeroed in the same pass so no separate memset is needed.  HBM-bound by construction:
// SGD 16 B/element (read p,g; write p,g), Adam 32 B/element.
#include "optim_dev.hpp"

namespace kge {

__global__ __launch_bounds__(256) void sgd_kernel(SweepTables tb, float lr) {
    sgd_sweep(tb.p[blockIdx.y], tb.g[blockIdx.y], tb.n[blockIdx.y], lr, (long long)blockIdx.x * blockDim.x + threadIdx.x,
              (long long)gridDim.x * blockDim.x);
}

__global__ __launch_bounds__(256) void adam_kernel(SweepTables tb, float lr_t, float b1, float b2, float eps) {
    adam_sweep(tb.p[blockIdx.y], tb.m[blockIdx.y], tb.v[blockIdx.y], tb.g[blockIdx.y], tb.n[blockIdx.y], lr_t, b1, b2, eps,
               (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

static unsigned sweep_blocks(long long n) {
    long long b = ((n >> 2) + 255) / 256;
    if (b > 2048) b = 2048;  // 8 blocks per CU, grid-stride beyond that
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int check_tables(int n_tables, float *const *p, float *const *g, float *const *m, float *const *v, const int64_t *numel,
                        SweepTables &tb, long long &n_max, const char *who) {
    if (n_tables < 1 || n_tables > 4 || !p || !g || !numel) return fail(KGE_ERR_BAD_ARG, std::string(who) + ": 1..4 tables");
    n_max = 0;
    for (int i = 0; i < 4; i++) { tb.p[i] = tb.g[i] = tb.m[i] = tb.v[i] = nullptr; tb.n[i] = 0; }
    for (int i = 0; i < n_tables; i++) {
        tb.p[i] = p[i]; tb.g[i] = g[i]; tb.m[i] = m ? m[i] : nullptr; tb.v[i] = v ? v[i] : nullptr;
        tb.n[i] = numel[i] > 0 ? numel[i] : 0;
        uintptr_t bits = reinterpret_cast<uintptr_t>(p[i]) | reinterpret_cast<uintptr_t>(g[i]);
        if (m) bits |= reinterpret_cast<uintptr_t>(m[i]) | reinterpret_cast<uintptr_t>(v[i]);
        if (bits & 15) return fail(KGE_ERR_BAD_ARG, "tables must be 16-byte aligned");
        if (tb.n[i] > n_max) n_max = tb.n[i];
    }
    return KGE_OK;
}

int launch_sgd_tables(int n_tables, float *const *p, float *const *g, const int64_t *numel, float lr, hipStream_t stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_sgd_update: no usable HIP device");
    SweepTables tb;
    long long n_max;
    int rc = check_tables(n_tables, p, g, nullptr, nullptr, numel, tb, n_max, "kge_sgd_update_tables");
    if (rc) return rc;
    if (n_max <= 0) return KGE_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(sweep_blocks(n_max), (unsigned)n_tables), dim3(256), 0, stream, tb, lr);
    return hip_check(hipGetLastError(), "sgd launch");
}

int launch_adam_tables(int n_tables, float *const *p, float *const *m, float *const *v, float *const *g, const int64_t *numel,
                       float lr_t, float b1, float b2, float eps, hipStream_t stream) {
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_adam_update: no usable HIP device");
    if (!m || !v) return fail(KGE_ERR_BAD_ARG, "kge_adam_update_tables: null moment tables");
    SweepTables tb;
    long long n_max;
    int rc = check_tables(n_tables, p, g, m, v, numel, tb, n_max, "kge_adam_update_tables");
    if (rc) return rc;
    if (n_max <= 0) return KGE_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(sweep_blocks(n_max), (unsigned)n_tables), dim3(256), 0, stream, tb, lr_t, b1, b2, eps);
    return hip_check(hipGetLastError(), "adam launch");
}

int launch_sgd(float *p, float *g, int64_t n, float lr, hipStream_t stream) {
    return launch_sgd_tables(1, &p, &g, &n, lr, stream);
}

int launch_adam(float *p, float *m, float *v, float *g, int64_t n, float lr_t, float b1, float b2, float eps,
                hipStream_t stream) {
    return launch_adam_tables(1, &p, &m, &v, &g, &n, lr_t, b1, b2, eps, stream);
}

}  // namespace kge

// ---- the loss riding in the exchanges of a data-parallel step (Config.train_step) ------------------------------------------
// A TransE step exchanges an INT32 count image, so the rank's loss travels in it as four 16-bit limbs of llrint(loss * 2^32):
// the SUM reduce-scatter then adds the ranks' losses exactly (integers; up to 32 768 ranks before a limb overflows), and the
// result does not depend on the number of ranks or the reduction order.  Losses are < 2^20 (a mean hinge is a few units).
namespace kge {
__global__ void loss_to_limbs_kernel(const float *__restrict__ loss, int32_t *__restrict__ limbs) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double x = (double)loss[0] * 4294967296.0;
        long long v = x >= 0.0 ? (long long)(x + 0.5) : 0;       // the loss is a mean of hinges: never negative
        for (int i = 0; i < 4; i++) limbs[i] = (int32_t)((v >> (16 * i)) & 0xFFFF);
    }
}
__global__ void limbs_to_loss_kernel(const int32_t *__restrict__ limbs, float *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long v = 0;
        for (int i = 0; i < 4; i++) v += (long long)limbs[i] << (16 * i);
        out[0] = (float)((double)v / 4294967296.0);
    }
}
}  // namespace kge

extern "C" int kge_loss_to_limbs(const float *d_loss, int32_t *d_limbs4, void *stream) {
    if (!d_loss || !d_limbs4) return kge::fail(KGE_ERR_BAD_ARG, "kge_loss_to_limbs: null argument");
    hipLaunchKernelGGL(kge::loss_to_limbs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_loss, d_limbs4);
    return kge::hip_check(hipGetLastError(), "loss limbs launch");
}
extern "C" int kge_limbs_to_loss(const int32_t *d_limbs4, float *d_out, void *stream) {
    if (!d_limbs4 || !d_out) return kge::fail(KGE_ERR_BAD_ARG, "kge_limbs_to_loss: null argument");
    hipLaunchKernelGGL(kge::limbs_to_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_limbs4, d_out);
    return kge::hip_check(hipGetLastError(), "loss limbs launch");
}
