// Many training steps inside ONE persistent launch: the reference's loop body
//     con.sampling(); sess.run([train_op, loss, global_step], feed_dict)      (distribute_training.py:267-283)
// at the reference's own batch sizes (Config.py:189-210: 2 721 positives for FB15k-237, 8 683 for WN18RR) moves a few
// MB per step -- microseconds of memory time -- so a step made of separate launches is bound by launch latency
// (4-6 launches x ~5 us of device-side dispatch + their fill / drain).  Here one grid of one workgroup per CU stays
// resident and walks S steps, separated by two grid-wide barriers per step:
//
//   phase A(s):  optimizer sweep for step s-1 (p -= lr*g, g = 0 / TF1 Adam; the fold of the relation hub copies is
//                fused into it)  +  sampling of batch s (independent of the parameters, so it shares the phase)
//   ---- grid barrier ----
//   phase B(s):  forward / hinge / backward of batch s, one team per positive group (the fwdbwd_group body of the
//                one-launch-per-stage kernels), gradient rows added with fp32 atomics into the dense accumulators
//   ---- grid barrier ----
//
// The barrier is XCD-hierarchical (MI355X_MICROARCH.md, "barrier-xcd"): workgroups arrive at their XCD's counter; the
// last arriver of each XCD writes that L2's dirty lines back once (agent release), arrives at the top counter, waits for
// the other XCDs, and releases its XCD through a generation word; every workgroup then invalidates its CU's L1 (agent
// acquire).  Per-XCD L2s are not coherent with each other, so parameters written by one XCD's sweep are only visible to
// another XCD's gathers through that release / acquire pair.  Every spin is bounded: a barrier that cannot complete sets
// an abort word that every later barrier and phase checks, so the grid always drains.
//
// Sampling inside the launch draws step s from the rng states the launch STARTED with, jumped ahead by s whole batches
// (the fixed draw budget per positive makes the state in front of any draw a pure function of its index), so the batches
// are bit-identical to S calls of the reference's `sampling`; the streams are advanced once after the launch.
#include "models_dev.hpp"
#include "optim_dev.hpp"
#include "sampler_dev.hpp"

namespace kge {
namespace {

constexpr int kLine = 32;              // counters sit on 128-byte lines of their own
constexpr unsigned kSpinLimit = 1u << 22;

struct GridBarrier {
    unsigned xcd_count[8 * kLine];     // arrivals per XCD (monotonic)
    unsigned xcd_gen[8 * kLine];       // generation released per XCD
    unsigned xcd_pop[8 * kLine];       // workgroups resident per XCD (census of this launch)
    unsigned top_count[kLine];         // XCD leaders arrived (monotonic)
    unsigned flat[kLine];              // census barrier
    unsigned abort_flag[kLine];
};

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct BarrierState {
    unsigned xcc, pop, n_xcd, gen;
};

// thread 0 of every workgroup; returns false when the launch must bail out
__device__ bool barrier_census(GridBarrier *gb, BarrierState &bs) {
    bs.xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;      // HW_REG_XCC_ID[3:0]
    // Two adds to different words from one thread are not ordered by themselves: a workgroup that sees flat == gridDim.x must also
    // see every xcd_pop add, or it latches a population that is too small and its XCD releases early.  So the census add is a
    // RETURNING atomic whose result is waited for (it has been performed once its value is back) before the flat add is issued.
    const unsigned before = __hip_atomic_fetch_add(&gb->xcd_pop[bs.xcc * kLine], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (before >= gridDim.x) {    // (keeps the returned value live; cannot be true)
        __hip_atomic_store(&gb->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    __hip_atomic_fetch_add(&gb->flat[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_agent(&gb->flat[0]) < gridDim.x) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinLimit || ld_agent(&gb->abort_flag[0])) {
            __hip_atomic_store(&gb->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // the populations are read only behind this
    bs.pop = ld_agent(&gb->xcd_pop[bs.xcc * kLine]);
    bs.n_xcd = 0;
    unsigned total = 0;
    for (int x = 0; x < 8; x++) { const unsigned c = ld_agent(&gb->xcd_pop[x * kLine]); bs.n_xcd += c ? 1u : 0u; total += c; }
    bs.gen = 0;
    if (total != gridDim.x) {     // an incomplete census would let an XCD release early: drain instead
        __hip_atomic_store(&gb->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}

// Grid-wide barrier with release / acquire of everything the workgroups stored before it.  Called by ALL threads.
__device__ bool grid_barrier(GridBarrier *gb, BarrierState &bs, int *ok_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores and atomics have reached L2 / memory
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = true;
        bs.gen++;
        unsigned spins = 0;
        const unsigned arrived = __hip_atomic_fetch_add(&gb->xcd_count[bs.xcc * kLine], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (arrived == bs.pop * bs.gen) {      // last workgroup of this XCD: one L2 write-back for all of them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&gb->top_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_agent(&gb->top_count[0]) < bs.n_xcd * bs.gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit || ld_agent(&gb->abort_flag[0])) { ok = false; break; }
            }
            if (!ok) __hip_atomic_store(&gb->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gb->xcd_gen[bs.xcc * kLine], bs.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_agent(&gb->xcd_gen[bs.xcc * kLine]) < bs.gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit || ld_agent(&gb->abort_flag[0])) { ok = false; break; }
            }
            if (!ok) __hip_atomic_store(&gb->abort_flag[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // this CU's L1 (and stale L2 copies) give way to the released data
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ld_agent(&gb->abort_flag[0])) ok = false;
        *ok_lds = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_lds != 0;
}

struct PersistArgs {
    FbArgs fb;
    SamplerArgs sm;
    SweepTables tb;
    int n_tables, adam, n_steps;
    float b1, b2, eps;
    const float *lr;          // [n_steps]
    float *losses;            // [n_steps]
    float *partials;          // [2][gridDim.x] per-workgroup hinge sums of the two steps in flight
    GridBarrier *bar;
    long long B;
    int W;
    int fold_k;               // relation hub copies folded by the sweep (0 = none)
    long long fold_elems;     // R * D
    int ahead;                // 1 = idle teams draw the next batch during the forward/backward phase
    int touch;                // 1 = every group's rows are touched up front (touch_group_rows)
    unsigned long long *trace;   // option "persist_trace": 100 MHz wall-clock stamps of workgroup 0 at the phase boundaries, 6 per step
};

constexpr int kTraceSteps = 256;
constexpr int kTraceBlocks = 1024;    // per-workgroup phase-B durations of step 16 follow the per-step stamps
__device__ __forceinline__ void stamp(const PersistArgs &pa, int step, int slot) {
    if (pa.trace && blockIdx.x == 0 && threadIdx.x == 0 && step < kTraceSteps) pa.trace[step * 6 + slot] = wall_clock64();
}

// Relation-side tables whose gradient rows were spread over K hub copies (fwdbwd_group adds a group's relation rows into
// copy b % K: same-address atomics serialise): the copies are folded and the element updated by one and the same thread
__device__ __forceinline__ void sweep_folded(const PersistArgs &pa, float lr, long long tid, long long stride) {
    for (long long i = tid; i < 2 * pa.fold_elems; i += stride) {
        const int t = i < pa.fold_elems ? 1 : 2;
        if (t >= pa.n_tables) continue;
        const long long j = i < pa.fold_elems ? i : i - pa.fold_elems;
        float *c = t == 1 ? pa.fb.copies_rel : pa.fb.copies_auxr;
        float sum = 0.f;                               // same order as hub_fold_kernel (models.hip): copies first, then the accumulator
        for (int k = 0; k < pa.fold_k; k++) {
            const float v = c[k * pa.fold_elems + j];
            if (v != 0.f) { sum += v; c[k * pa.fold_elems + j] = 0.f; }
        }
        float g = pa.tb.g[t][j];
        if (sum != 0.f) g += sum;
        pa.tb.g[t][j] = 0.f;
        if (pa.adam) adam_one(pa.tb.p[t][j], pa.tb.m[t][j], pa.tb.v[t][j], g, lr, pa.b1, pa.b2, pa.eps);
        else if (g != 0.f) pa.tb.p[t][j] -= lr * g;
    }
}

// Every row a group will gather, touched up front by ONE wave instruction (lane = (row slot, 128-byte line)): after a grid
// barrier's acquire the tables are cold in this CU's L1 and this XCD's L2, and fwdbwd_group reads them as a chain of dependent
// round trips (ids -> relation rows -> head -> tail -> negative's ids -> its row); with the lines requested together the chain
// runs against warm caches.  Returns a value that depends on the loaded words (so the loads exist) and is never acted upon.
template <int MODEL, int L>
__device__ __forceinline__ unsigned touch_group_rows(const FbArgs &a, long long b, int lane) {
    const int h = a.bh[b], t = a.bt[b], r = a.br[b];
    const long long j = b + a.stride;                       // the first negative (Base.cpp:109-139 layout)
    const int nh = a.bh[j], nt = a.bt[j];
    const int neg = nh != h ? nh : nt;
    const int lines = (a.D * 4 + 127) >> 7;                 // 128-byte lines per row (<= 8: widths up to 256)
    unsigned junk = 0;
#pragma unroll
    for (int s0 = 0; s0 < 8; s0 += L / 8) {
        const int slot = s0 + (lane >> 3), line = min(lane & 7, lines - 1);
        const float *tab = a.ent;
        long long row = h;
        switch (slot) {
            case 1: row = t; break;
            case 2: tab = a.rel; row = r; break;
            case 3: row = neg; break;
            case 4: tab = MODEL == KGE_TRANSE ? a.rel : a.auxr; row = r; break;
            case 5: tab = MODEL == KGE_TRANSD ? a.auxe : a.ent; row = h; break;
            case 6: tab = MODEL == KGE_TRANSD ? a.auxe : a.ent; row = t; break;
            case 7: tab = MODEL == KGE_TRANSD ? a.auxe : a.ent; row = neg; break;
            default: break;
        }
        junk ^= *reinterpret_cast<const unsigned *>(tab + row * a.D + min(line * 32, a.D - 1));
    }
    return junk;
}

template <int MODEL, int L, int C, int THREADS>
__global__ __launch_bounds__(THREADS) void persistent_steps_kernel(PersistArgs pa) {
    constexpr int TEAMS = THREADS / L;
    __shared__ float red[TEAMS];
    __shared__ int ok_lds;
    __shared__ BarrierState bs_lds;
    BarrierState bs = {};
    if (threadIdx.x == 0) {
        ok_lds = barrier_census(pa.bar, bs) ? 1 : 0;
        bs_lds = bs;
    }
    __syncthreads();
    if (!ok_lds) return;
    bs = bs_lds;
    Team<L, C> tm;
    tm.lane = threadIdx.x % L;
    tm.D = pa.fb.D;
    const int team_in_block = threadIdx.x / L;
    const long long tid = (long long)blockIdx.x * THREADS + threadIdx.x, stride = (long long)gridDim.x * THREADS;
    const long long batch_len = pa.B * (1 + pa.sm.neg + pa.sm.negrel);
    // Groups are dealt round-robin over the workgroups (group b -> workgroup b % grid, team slot b / grid): at the reference's
    // batch sizes a workgroup then has fewer groups than teams, and its idle teams draw the NEXT step's batch meanwhile (the
    // sampler never reads the parameters): sampling leaves the critical path.  Batches alternate between two buffers.
    const long long G = (pa.B + gridDim.x - 1) / gridDim.x;          // most groups any workgroup holds
    const bool ahead = G < TEAMS && pa.ahead;
    const long long s_threads = ahead ? (long long)(TEAMS - G) * L : THREADS;
    const long long s_first = ahead ? G * L : 0;
    auto sample_batch = [&](int step, long long stid, long long sstride) {
        const int kshift = pa.sm.kshift, kp = 1 + pa.sm.neg + pa.sm.negrel;
        const unsigned long long draws = 1ull + 2ull * pa.sm.neg + pa.sm.negrel;
        int32_t *oh_p = pa.sm.out_h + (long long)(step & 1) * 3 * batch_len;
        for (long long g = stid; (g >> kshift) < pa.B; g += sstride) {
            const long long b = g >> kshift, k = g & ((1 << kshift) - 1);
            if (k >= kp) continue;
            // virtual thread id's slice holds min(per_thread, B - id*per_thread) positions per batch: that many * draws per step
            const long long id = (long long)((unsigned)b / (unsigned)pa.sm.per_thread);
            long long len = pa.B - id * pa.sm.per_thread;
            len = len < 0 ? 0 : (len > pa.sm.per_thread ? pa.sm.per_thread : len);
            int oh, ot, orr;
            sample_slot(pa.sm, b, k, (unsigned long long)step * (unsigned long long)len * draws, oh, ot, orr);
            const long long o = b + k * pa.sm.out_stride;
            oh_p[o] = oh; oh_p[batch_len + o] = ot; oh_p[2 * batch_len + o] = orr;
        }
    };
    for (int step = 0; step <= pa.n_steps; step++) {
        // ---------------- phase A: update and loss of step-1 (and, when no team is idle in phase B, sampling of this step) ----------------
        stamp(pa, step, 0);
        if (step > 0) {
            const float lr = pa.lr[step - 1];
            if (pa.fold_k > 1) sweep_folded(pa, lr, tid, stride);
            for (int i = 0; i < pa.n_tables; i++) {
                if (pa.fold_k > 1 && (i == 1 || i == 2)) continue;
                if (pa.adam) adam_sweep(pa.tb.p[i], pa.tb.m[i], pa.tb.v[i], pa.tb.g[i], pa.tb.n[i], lr, pa.b1, pa.b2, pa.eps, tid, stride);
                else sgd_sweep<4>(pa.tb.p[i], pa.tb.g[i], pa.tb.n[i], lr, tid, stride);
            }
            if (blockIdx.x == 0 && threadIdx.x < 64) {     // fixed-order sum of the workgroups' hinge sums (TransE.py:51)
                const float *part = pa.partials + (long long)((step - 1) & 1) * gridDim.x;
                float s = 0.f;
                for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) s += part[i];
                s = team_sum<64>(s);
                if (threadIdx.x == 0) pa.losses[step - 1] = s * pa.fb.unit;
            }
        }
        if (step == pa.n_steps) break;
        stamp(pa, step, 1);
        if (!ahead || step == 0) sample_batch(step, tid, stride);
        stamp(pa, step, 2);
        if (!grid_barrier(pa.bar, bs, &ok_lds)) return;
        stamp(pa, step, 3);
        // ---------------- phase B: forward / hinge / backward of this step's batch; idle teams sample the next one ----------------
        const unsigned long long tb0 = (pa.trace && step == 16 && threadIdx.x == 0) ? wall_clock64() : 0ull;
        FbArgs fb = pa.fb;
        fb.bh = pa.fb.bh + (long long)(step & 1) * 3 * batch_len;
        fb.bt = fb.bh + batch_len;
        fb.br = fb.bh + 2 * batch_len;
        float lsum = 0.f;
        unsigned junk = 0;
        if (pa.touch)
            for (long long slot = team_in_block; slot < G; slot += TEAMS) {
                const long long b = slot * gridDim.x + blockIdx.x;
                if (b < pa.B) junk ^= touch_group_rows<MODEL, L>(fb, b, tm.lane);
            }
        for (long long slot = team_in_block; slot < G; slot += TEAMS) {
            const long long b = slot * gridDim.x + blockIdx.x;
            if (b < pa.B) fwdbwd_group<MODEL, L, C, false>(tm, fb, b, lsum);
        }
        if (junk == 0x9E3779B9u && pa.n_steps < 0) lsum += 1.f;      // (never true: keeps the touch loads alive)
        if (ahead && step + 1 < pa.n_steps && (long long)threadIdx.x >= s_first)
            sample_batch(step + 1, (long long)blockIdx.x * s_threads + (threadIdx.x - s_first), (long long)gridDim.x * s_threads);
        if (tm.lane == 0) red[team_in_block] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < TEAMS; i++) s += red[i];
            pa.partials[(long long)(step & 1) * gridDim.x + blockIdx.x] = s;
        }
        if (pa.trace && step == 16 && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pa.trace[6 * kTraceSteps + blockIdx.x] = wall_clock64() - tb0;
        }
        stamp(pa, step, 4);
        if (!grid_barrier(pa.bar, bs, &ok_lds)) return;
        stamp(pa, step, 5);
    }
}

GridBarrier *g_bar = nullptr;
float *g_partials = nullptr, *g_lr = nullptr;
int32_t *g_batch = nullptr;
int64_t g_batch_cap = 0, g_lr_cap = 0;
float *g_hub_rel = nullptr, *g_hub_auxr = nullptr;
int64_t g_hub_elems = 0;
unsigned long long *g_trace = nullptr;

}  // namespace

__global__ void advance_streams_by_kernel(uint64_t *streams, long long W, long long B, long long per_thread, unsigned long long draws) {
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= W) return;
    long long lef = id * per_thread, rig = lef + per_thread;
    if (rig > B) rig = B;
    if (lef > B) lef = B;
    streams[id] = lcg_skip(streams[id], (unsigned long long)(rig - lef) * draws);
}

}  // namespace kge

using namespace kge;

extern "C" int kge_train_steps_persistent(const kge_model_desc *m, float *const tables[KGE_MAX_TABLES], float *const grads[KGE_MAX_TABLES],
                                          float *const adam_m[KGE_MAX_TABLES], float *const adam_v[KGE_MAX_TABLES], INT batchSize,
                                          INT negRate, INT negRelRate, INT n_steps, int32_t adam, const float *h_lr, float beta1,
                                          float beta2, float eps, float *d_losses, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Engine &e = engine();
    tables_written();
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "kge_train_steps_persistent: no usable HIP device");
    if (!m || !tables || !grads || !h_lr || !d_losses || batchSize <= 0 || negRate < 0 || negRelRate < 0 || negRate + negRelRate < 1 || n_steps < 1)
        return fail(KGE_ERR_BAD_ARG, "kge_train_steps_persistent: bad arguments");
    if (m->model != KGE_TRANSE && m->model != KGE_TRANSH && m->model != KGE_TRANSD)
        return fail(KGE_ERR_UNSUPPORTED, "kge_train_steps_persistent: TransE / TransH / TransD (TransR's MFMA stages are separate launches)");
    if (m->ent_dim != m->rel_dim || m->ent_dim > 256) return fail(KGE_ERR_UNSUPPORTED, "kge_train_steps_persistent: embedding width <= 256");
    if (adam && (!adam_m || !adam_v)) return fail(KGE_ERR_BAD_ARG, "kge_train_steps_persistent: Adam needs the moment tables");
    int rc = ensure_device_index();
    if (rc) return rc;
    if ((rc = flush_attached_sampler(stream))) return rc;
    if (e.index.train_dup <= 0) return fail(KGE_ERR_NO_DATASET, "kge_train_steps_persistent: empty training set");
    if (m->ent_total != e.index.ent_total || m->rel_total != e.index.rel_total)
        return fail(KGE_ERR_BAD_ARG, "kge_train_steps_persistent: model and dataset sizes differ");
    if ((rc = upload_jump_table())) return rc;
    const int64_t W = e.work_threads, B = batchSize, n_neg = negRate + negRelRate;
    const int n_tables = m->model == KGE_TRANSE ? 2 : (m->model == KGE_TRANSH ? 3 : 4);
    int cus = 0;
    if ((rc = hip_check(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0), "device attribute"))) return rc;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) return fail(KGE_ERR_NO_DEVICE, "kge_train_steps_persistent: no compute units reported");
    const unsigned blocks = (unsigned)cus;        // one workgroup per CU: co-resident by construction
    // ---- workspace ----
    if (!g_bar && (rc = hip_check(hipMalloc(&g_bar, sizeof(GridBarrier)), "alloc grid barrier"))) return rc;
    if (!g_partials && (rc = hip_check(hipMalloc(&g_partials, sizeof(float) * 2 * 1024), "alloc loss partials"))) return rc;
    if (blocks > 1024) return fail(KGE_ERR_UNSUPPORTED, "kge_train_steps_persistent: more than 1024 compute units");
    const int64_t batch_len = B * (1 + n_neg);
    if (batch_len > g_batch_cap) {
        if (g_batch) (void)hipFree(g_batch);
        g_batch = nullptr;
        if ((rc = hip_check(hipMalloc(&g_batch, sizeof(int32_t) * 2 * 3 * (size_t)batch_len), "alloc persistent batch"))) return rc;   // two batches: this step's and the next
        g_batch_cap = batch_len;
    }
    if (n_steps > g_lr_cap) {
        if (g_lr) (void)hipFree(g_lr);
        g_lr = nullptr;
        if ((rc = hip_check(hipMalloc(&g_lr, sizeof(float) * (size_t)n_steps), "alloc learning rates"))) return rc;
        g_lr_cap = n_steps;
    }
    if ((rc = hip_check(hipMemcpyAsync(g_lr, h_lr, sizeof(float) * (size_t)n_steps, hipMemcpyHostToDevice, stream), "upload learning rates"))) return rc;
    if ((rc = hip_check(hipMemsetAsync(g_bar, 0, sizeof(GridBarrier), stream), "zero grid barrier"))) return rc;
    PersistArgs pa = {};
    FbArgs &a = pa.fb;
    a.ent = tables[0]; a.rel = tables[1]; a.auxr = tables[2]; a.auxe = tables[3];
    a.g_ent = grads[0]; a.g_rel = grads[1]; a.g_auxr = grads[2]; a.g_auxe = grads[3];
    a.bh = g_batch; a.bt = g_batch + batch_len; a.br = g_batch + 2 * batch_len;
    a.n_pos = B; a.n_neg = n_neg; a.stride = B;
    a.D = m->ent_dim; a.margin = m->margin; a.unit = 1.0f / (float)(B * n_neg);
    a.negative_rel = m->negative_rel; a.ent_total = (int)m->ent_total; a.rel_total = (int)m->rel_total;
    // Relation-side hub copies: same-address fp32 atomics serialise at the memory side (~8 ns each), and inside this launch
    // the forward/backward phase ends when the hottest row has taken its last add.  Group b adds its relation-side rows into copy
    // b % K; the sweep folds the copies.  K: enough that the busiest relation (a Zipf-skewed head can hold ~a fifth of the batch)
    // stays near a hundred adds per copy, few enough that the fold (K x R x D elements) stays small against the sweep --
    // measured at config #1 (TransE, B = 2 721, 237 relations): K = 0 / 4 / 8 / 16 / 64 -> 34.7 / 28.5 / 28.8 / 31.1 / 43.2 us per step.
    const int64_t hub_rows = (m->model == KGE_TRANSE ? 1 : 2) * m->rel_total;
    const int64_t per_row = hub_rows > 0 ? ((m->model == KGE_TRANSE ? 1 : 2) * B) / hub_rows : 0;
    int64_t want = per_row >= 128 ? per_row / 16 : B / 640;
    if (want * m->rel_total * (int64_t)a.D > (int64_t(4) << 20)) want = (int64_t(4) << 20) / (m->rel_total * (int64_t)a.D);   // fold <= 4 M elements
    if (want > 1 && e.hub_copies) {
        int64_t copies = want;
        if (copies > 64) copies = 64;
        const int64_t per_copy = m->rel_total * (int64_t)a.D;
        if (copies > 1) {
            if (copies * per_copy > g_hub_elems) {
                if (g_hub_rel) (void)hipFree(g_hub_rel);
                if (g_hub_auxr) (void)hipFree(g_hub_auxr);
                g_hub_rel = g_hub_auxr = nullptr;
                g_hub_elems = copies * per_copy;
                if ((rc = hip_check(hipMalloc(&g_hub_rel, sizeof(float) * (size_t)g_hub_elems), "alloc hub copies"))) return rc;
                if ((rc = hip_check(hipMalloc(&g_hub_auxr, sizeof(float) * (size_t)g_hub_elems), "alloc hub copies"))) return rc;
                if ((rc = hip_check(hipMemset(g_hub_rel, 0, sizeof(float) * (size_t)g_hub_elems), "zero hub copies"))) return rc;
                if ((rc = hip_check(hipMemset(g_hub_auxr, 0, sizeof(float) * (size_t)g_hub_elems), "zero hub copies"))) return rc;
            }
            a.copies_rel = g_hub_rel; a.copies_auxr = g_hub_auxr; a.hub_k = (int)copies;
            pa.fold_k = (int)copies; pa.fold_elems = per_copy;
        }
    }
    SamplerArgs &s = pa.sm;
    s.pos = e.dev.pos; s.grp = e.dev.grp; s.ht = e.dev.ht;
    s.tails_hr = e.dev.tails_hr; s.heads_tr = e.dev.heads_tr; s.rels_ht = e.dev.rels_ht;
    s.bern_prob = e.dev.bern_prob; s.streams = e.dev.streams; s.streams_next = nullptr; s.W = W; s.B = B;
    s.out_h = g_batch; s.out_t = g_batch + batch_len; s.out_r = g_batch + 2 * batch_len;
    s.per_thread = (B % W == 0) ? B / W : B / W + 1; s.pos_lo = 0; s.n_local = B; s.out_stride = B;
    s.train_dup = e.index.train_dup; s.new_batch = e.index.new_batch;
    s.ent_total = (int)e.index.ent_total; s.rel_total = (int)e.index.rel_total;
    s.neg = (int)negRate; s.negrel = (int)negRelRate; s.bern = e.bern ? 1 : 0;
    s.pick_div = (unsigned long long)(s.new_batch > 0 ? s.new_batch : s.train_dup);
    s.pick_magic = ~0ull / s.pick_div;
    int kshift = 0;
    while ((1 << kshift) < 1 + negRate + negRelRate) kshift++;
    s.kshift = kshift;
    for (int i = 0; i < 4; i++) { pa.tb.p[i] = pa.tb.g[i] = pa.tb.m[i] = pa.tb.v[i] = nullptr; pa.tb.n[i] = 0; }
    for (int i = 0; i < n_tables; i++) {
        int64_t rows = 0, cols = 0;
        if (kge_table_shape(m, i, &rows, &cols)) return KGE_ERR_BAD_ARG;
        pa.tb.p[i] = tables[i]; pa.tb.g[i] = grads[i]; pa.tb.n[i] = rows * cols;
        if (adam) { pa.tb.m[i] = adam_m[i]; pa.tb.v[i] = adam_v[i]; }
        uintptr_t bits = reinterpret_cast<uintptr_t>(tables[i]) | reinterpret_cast<uintptr_t>(grads[i]);
        if (adam) bits |= reinterpret_cast<uintptr_t>(adam_m[i]) | reinterpret_cast<uintptr_t>(adam_v[i]);
        if (!tables[i] || !grads[i] || (bits & 15)) return fail(KGE_ERR_BAD_ARG, "kge_train_steps_persistent: tables must be non-null and 16-byte aligned");
    }
    pa.n_tables = n_tables; pa.adam = adam; pa.n_steps = (int)n_steps;
    pa.b1 = beta1; pa.b2 = beta2; pa.eps = eps;
    pa.lr = g_lr; pa.losses = d_losses; pa.partials = g_partials; pa.bar = g_bar;
    pa.B = B; pa.W = (int)W;
    pa.touch = e.persist_touch; pa.ahead = e.persist_ahead;
    if (e.persist_trace) {
        if (!g_trace && (rc = hip_check(hipMalloc(&g_trace, sizeof(unsigned long long) * (6 * kTraceSteps + kTraceBlocks)), "alloc phase trace"))) return rc;
        if ((rc = hip_check(hipMemsetAsync(g_trace, 0, sizeof(unsigned long long) * (6 * kTraceSteps + kTraceBlocks), stream), "zero phase trace"))) return rc;
        pa.trace = g_trace;
    }
    // 1024 threads: 16 waves per CU behind ONE barrier participant (<= 128 VGPRs); 512: half the waves, twice the registers
    const int D = a.D;
    const bool wide = e.persist_threads != 512;
    // Cooperative launch: the runtime checks the grid against what can be co-resident and REFUSES one that cannot be
    // (hipErrorCooperativeLaunchTooLarge) instead of leaving it to the bounded spins to find out half way through a step; its
    // +15-19 us of host time is paid once per launch of n_steps steps.
    void *kargs[] = {&pa};
    hipError_t launch_err = hipSuccess;
#define KGE_PERSIST(MODEL, LL, CC)                                                                                                 \
    {                                                                                                                              \
        if (wide) launch_err = hipLaunchCooperativeKernel((const void *)persistent_steps_kernel<MODEL, LL, CC, 1024>, dim3(blocks), \
                                                          dim3(1024), kargs, 0, stream);                                           \
        else launch_err = hipLaunchCooperativeKernel((const void *)persistent_steps_kernel<MODEL, LL, CC, 512>, dim3(blocks),      \
                                                     dim3(512), kargs, 0, stream);                                                 \
    }
#define KGE_PERSIST_D(MODEL)                                                                   \
    if (D <= 16) KGE_PERSIST(MODEL, 16, 1) else if (D <= 32) KGE_PERSIST(MODEL, 16, 2)         \
    else if (D <= 64) KGE_PERSIST(MODEL, 16, 4) else if (D <= 128) KGE_PERSIST(MODEL, 32, 4)   \
    else KGE_PERSIST(MODEL, 64, 4)
    switch (m->model) {
        case KGE_TRANSE: KGE_PERSIST_D(KGE_TRANSE) break;
        case KGE_TRANSH: KGE_PERSIST_D(KGE_TRANSH) break;
        default: KGE_PERSIST_D(KGE_TRANSD) break;
    }
#undef KGE_PERSIST_D
#undef KGE_PERSIST
    if (launch_err == hipErrorCooperativeLaunchTooLarge)
        return fail(KGE_ERR_UNSUPPORTED, "kge_train_steps_persistent: one workgroup per compute unit cannot be co-resident on this device "
                                         "(compute units masked or held by another process?)");
    if ((rc = hip_check(launch_err, "persistent steps launch"))) return rc;
    if ((rc = hip_check(hipGetLastError(), "persistent steps launch"))) return rc;
    // every stream moves by n_steps batches (the launch sampled from the states it started with)
    hipLaunchKernelGGL(advance_streams_by_kernel, dim3((unsigned)((W + 63) / 64)), dim3(64), 0, stream, e.dev.streams, (long long)W,
                       (long long)B, (long long)s.per_thread, (unsigned long long)(1 + 2 * negRate + negRelRate) * (unsigned long long)n_steps);
    e.dev.streams_sync = 2;
    return hip_check(hipGetLastError(), "persistent steps launch");
}

// test / measurement hook: the phase-boundary stamps of the last traced launch (6 per step, 100 MHz ticks), up to n_steps steps
extern "C" int kge_persistent_trace(uint64_t *h_out, INT n_steps) {
    if (!h_out || n_steps < 0) return fail(KGE_ERR_BAD_ARG, "kge_persistent_trace: bad arguments");
    if (!g_trace) return fail(KGE_ERR_BAD_ARG, "kge_persistent_trace: set option persist_trace and run a launch first");
    if (n_steps > kTraceSteps + kTraceBlocks / 6) n_steps = kTraceSteps + kTraceBlocks / 6;   // (rows beyond 256: per-workgroup phase-B ticks of step 16)
    return hip_check(hipMemcpy(h_out, g_trace, sizeof(uint64_t) * 6 * (size_t)n_steps, hipMemcpyDeviceToHost), "read phase trace");
}

extern "C" int kge_persistent_aborted(int32_t *flag) {
    if (!flag) return fail(KGE_ERR_BAD_ARG, "kge_persistent_aborted: null output");
    *flag = 0;
    if (!g_bar) return KGE_OK;
    unsigned v = 0;
    int rc = hip_check(hipMemcpy(&v, &g_bar->abort_flag[0], sizeof(unsigned), hipMemcpyDeviceToHost), "read abort flag");
    *flag = (int32_t)v;
    return rc;
}
