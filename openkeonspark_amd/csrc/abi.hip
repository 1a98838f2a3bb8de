// extern "C" surface of libkge_mi355.so (declared in include/kge_mi355.h) and the process-global
// engine state behind it.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "engine.hpp"

namespace kge {

Engine &engine() {
    static Engine e;
    return e;
}

void set_error(const std::string &msg) {
    engine().last_error = msg;
    std::fprintf(stderr, "[kge_mi355] error: %s\n", msg.c_str());
}

int fail(int code, const std::string &msg) {
    set_error(msg);
    return code;
}

int hip_check(hipError_t e, const char *what) {
    if (e == hipSuccess) return KGE_OK;
    return fail(KGE_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

bool device_ok() {
    Engine &e = engine();
    if (e.device_state == 0) {
        int n = 0;
        hipError_t rc = hipGetDeviceCount(&n);
        e.device_state = (rc == hipSuccess && n > 0) ? 1 : -1;
        if (e.device_state < 0) (void)hipGetLastError();
    }
    return e.device_state > 0;
}

template <typename T>
static int upload(T *&dst, const void *src, size_t count, const char *what) {
    if (dst) { (void)hipFree(dst); dst = nullptr; }
    size_t bytes = sizeof(T) * (count ? count : 1);
    int rc = hip_check(hipMalloc(&dst, bytes), what);
    if (rc) return rc;
    if (count) rc = hip_check(hipMemcpy(dst, src, sizeof(T) * count, hipMemcpyHostToDevice), what);
    return rc;
}

int ensure_device_index() {
    Engine &e = engine();
    if (!e.index.loaded) return fail(KGE_ERR_NO_DATASET, "no training set imported (importTrainFiles / kge_import_train_arrays)");
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "no usable HIP device: the sampler has no CPU fallback");
    int rc = KGE_OK;
    if (!e.dev.uploaded) {
        const KgIndex &ix = e.index;
        if ((rc = upload(e.dev.pos, ix.pos.data(), ix.pos.size(), "upload pos"))) return rc;
        if ((rc = upload(e.dev.grp, ix.grp.data(), ix.grp.size(), "upload grp"))) return rc;
        if ((rc = upload(e.dev.ht, ix.ht.data(), ix.ht.size(), "upload ht"))) return rc;
        if ((rc = upload(e.dev.tails_hr, ix.tails_hr.data(), ix.tails_hr.size(), "upload tails"))) return rc;
        if ((rc = upload(e.dev.heads_tr, ix.heads_tr.data(), ix.heads_tr.size(), "upload heads"))) return rc;
        if ((rc = upload(e.dev.rels_ht, ix.rels_ht.data(), ix.rels_ht.size(), "upload rels"))) return rc;
        if ((rc = upload(e.dev.bern_prob, ix.bern_prob.data(), ix.bern_prob.size(), "upload bern"))) return rc;
        e.dev.uploaded = true;
    }
    if ((int64_t)e.streams.size() != e.work_threads) {
        // setWorkThreads without randReset: the reference would read unallocated memory; give zeros
        e.streams.assign((size_t)e.work_threads, 0);
        e.dev.streams_sync = 0;
    }
    if (e.dev.streams_sync == 0) {
        if (e.dev.streams_cap < e.work_threads) {
            uint64_t *base = e.dev.streams < e.dev.streams_next ? e.dev.streams : e.dev.streams_next;   // one allocation, two halves
            if (base) (void)hipFree(base);
            e.dev.streams = e.dev.streams_next = nullptr;
            if ((rc = hip_check(hipMalloc(&e.dev.streams, sizeof(uint64_t) * 2 * (size_t)e.work_threads), "alloc streams"))) return rc;
            e.dev.streams_next = e.dev.streams + e.work_threads;
            e.dev.streams_cap = e.work_threads;
        }
        if ((rc = hip_check(hipMemcpy(e.dev.streams, e.streams.data(), sizeof(uint64_t) * (size_t)e.work_threads,
                                      hipMemcpyHostToDevice), "upload streams"))) return rc;
        e.dev.streams_sync = 1;
    }
    return KGE_OK;
}

static int pull_streams() {
    Engine &e = engine();
    { int rc = flush_attached_sampler(nullptr); if (rc) return rc; }   // an armed sampler has yet to write the states being read
    if (e.dev.streams_sync == 2) {
        int rc = hip_check(hipMemcpy(e.streams.data(), e.dev.streams, sizeof(uint64_t) * e.streams.size(),
                                     hipMemcpyDeviceToHost), "download streams");
        if (rc) return rc;
        e.dev.streams_sync = 1;
    }
    return KGE_OK;
}

static void adopt_index(KgIndex &&ix) {
    Engine &e = engine();
    e.index = std::move(ix);
    e.dev.uploaded = false;
}

// Reader.h:102-177 on the host (kg_index.cpp) or, for large training sets, on the device (index_build.hip);
// both produce the same arrays bit for bit
static std::string build_and_adopt(int64_t E, int64_t R, int64_t nb, int64_t n, const int64_t *h, const int64_t *t, const int64_t *r) {
    Engine &e = engine();
    // an armed sampler (kge_sampling_attach) points into the index arrays that are about to be replaced: run it first
    if (device_ok()) { (void)flush_attached_sampler(nullptr); (void)hipDeviceSynchronize(); }
    if (e.index_device_min >= 0 && n >= e.index_device_min && device_ok() && device_index_build_supported(E, R, n)) {
        KgIndex ix;
        std::string err = build_index_device(ix, e.dev, E, R, nb, n, h, t, r);
        if (!err.empty()) { e.dev.uploaded = false; return err; }
        e.index = std::move(ix);   // e.dev already holds the arrays (uploaded = true)
        return "";
    }
    KgIndex ix;
    std::string err = build_index(ix, E, R, nb, n, h, t, r);
    if (!err.empty()) return err;
    adopt_index(std::move(ix));
    return "";
}

}  // namespace kge

using namespace kge;

extern "C" {

void setInPath(char *path) {
    engine().in_path = path ? path : "";
    std::printf("Input Files Path : %s\n", engine().in_path.c_str());
}

void setOutPath(char *path) {
    engine().out_path = path ? path : "";
    std::printf("Output Files Path : %s\n", engine().out_path.c_str());
}

void setWorkThreads(INT threads) { engine().work_threads = threads < 1 ? 1 : threads; }
INT getWorkThreads(void) { return engine().work_threads; }
void setBern(INT con) { engine().bern = con; }

void randReset(void) {
    Engine &e = engine();
    e.streams.resize((size_t)e.work_threads);
    for (auto &s : e.streams) s = (uint64_t)(int64_t)e.libc.next();
    e.dev.streams_sync = 0;
}

void importTrainFiles(void) {
    Engine &e = engine();
    std::printf("The toolkit is importing datasets.\n");
    int64_t E = 0, R = 0, nb = 0;
    std::vector<int64_t> h, t, r;
    std::string err = load_openke_dir(e.in_path, E, R, nb, h, t, r);
    if (!err.empty()) {
        std::cout << err << std::endl;  // the reference prints and returns (Reader.h:36-39)
        set_error(err);
        return;
    }
    std::printf("The total of relations is %ld.\n", (long)R);
    std::printf("The total of entities is %ld.\n", (long)E);
    if (nb > 0) std::printf("The total number of new batch triples is: %ld\n", (long)nb);
    std::printf("The total of train triples is %ld.\n", (long)h.size());
    err = build_and_adopt(E, R, nb, (int64_t)h.size(), h.data(), t.data(), r.data());
    if (!err.empty()) { set_error(err); return; }
    std::fflush(stdout);
}

INT getEntityTotal(void) { return engine().index.ent_total; }
INT getRelationTotal(void) { return engine().index.rel_total; }
INT kge_eval_test_total(void); INT kge_eval_valid_total(void); INT kge_eval_triple_total(void);
INT getTripleTotal(void) { return kge_eval_triple_total(); }  // set by importTestFiles (Reader.h:231)
INT getTrainTotal(void) { return engine().index.train_uniq; }
INT getTrainTotal_(void) { return engine().index.train_dup; }
INT getBatchTotal(void) { return engine().index.new_batch; }
INT getTestTotal(void) { return kge_eval_test_total(); }
INT getValidTotal(void) { return kge_eval_valid_total(); }

void sampling(INT *batch_h, INT *batch_t, INT *batch_r, REAL *batch_y, INT batchSize, INT negRate, INT negRelRate) {
    Engine &e = engine();
    if (ensure_device_index()) return;
    if (batchSize <= 0 || negRate < 0 || negRelRate < 0) { set_error("sampling: bad sizes"); return; }
    const int64_t total = batchSize * (1 + negRate + negRelRate);
    if (e.dev.stage_cap < total) {
        if (e.dev.stage_i32) (void)hipFree(e.dev.stage_i32);
        if (e.dev.stage_i64) (void)hipFree(e.dev.stage_i64);
        e.dev.stage_i32 = nullptr; e.dev.stage_i64 = nullptr; e.dev.stage_cap = 0;
        if (hip_check(hipMalloc(&e.dev.stage_i32, sizeof(int32_t) * 3 * (size_t)total), "alloc stage")) return;
        if (hip_check(hipMalloc(&e.dev.stage_i64, (sizeof(int64_t) * 3 + sizeof(float)) * (size_t)total), "alloc stage")) return;
        e.dev.stage_cap = total;
    }
    int32_t *s = e.dev.stage_i32;
    // stage arrays are packed at `total` so the widen kernel can address them as [3][total]
    if (launch_sampler(s, s + total, s + 2 * total, batchSize, negRate, negRelRate, 0, e.work_threads, batchSize, nullptr, nullptr)) return;
    if (launch_widen(s, e.dev.stage_i64, batchSize, total, nullptr)) return;
    const int64_t *w = e.dev.stage_i64;
    if (hip_check(hipMemcpy(batch_h, w, sizeof(int64_t) * total, hipMemcpyDeviceToHost), "copy batch_h")) return;
    if (hip_check(hipMemcpy(batch_t, w + total, sizeof(int64_t) * total, hipMemcpyDeviceToHost), "copy batch_t")) return;
    if (hip_check(hipMemcpy(batch_r, w + 2 * total, sizeof(int64_t) * total, hipMemcpyDeviceToHost), "copy batch_r")) return;
    hip_check(hipMemcpy(batch_y, w + 3 * total, sizeof(float) * total, hipMemcpyDeviceToHost), "copy batch_y");
}

size_t kge_last_error(char *buf, size_t n) {
    const std::string &s = engine().last_error;
    if (buf && n) {
        size_t k = s.size() < n - 1 ? s.size() : n - 1;
        std::memcpy(buf, s.data(), k);
        buf[k] = 0;
    }
    return s.size();
}

void kge_clear_error(void) { engine().last_error.clear(); }
int kge_device_available(void) { return device_ok() ? 1 : 0; }
const char *kge_version(void) { return "kge_mi355 0.1 (gfx950)"; }

int kge_set_option(const char *name, INT value) {
    std::string n = name ? name : "";
    if (n == "counts_force_sort") { engine().counts_force_sort = value != 0; return KGE_OK; }
    if (n == "ride_shares") { engine().ride_shares = (int)value; return KGE_OK; }
    if (n == "transr_bf16x3") { engine().transr_bf16x3 = value != 0; return KGE_OK; }
    if (n == "transr_groups") { engine().transr_groups = (int)value; return KGE_OK; }
    if (n == "transr_fuse_vec") { engine().transr_fuse_vec = value != 0; return KGE_OK; }
    if (n == "counts_fused") { engine().counts_fused = value != 0; return KGE_OK; }
    if (n == "counts_fused_diag") { engine().counts_fused_diag = (int)value; return KGE_OK; }
    if (n == "counts_fused_cap") { engine().counts_fused_cap = value > 0 ? (int)value : 0; return KGE_OK; }
    if (n == "inv_table_max_bytes") { engine().inv_table_max_bytes = value; return KGE_OK; }
    if (n == "tables_changed") { tables_written(); return KGE_OK; }     // the caller wrote device tables itself (value ignored)
    if (n == "counts_krel") { int k = 1; while (k * 2 <= value && k < 64) k *= 2; engine().counts_krel = k; return KGE_OK; }
    if (n == "inv_carry") { engine().inv_carry = value != 0; tables_written(); return KGE_OK; }
    if (n == "float_records") { engine().float_records = value != 0; return KGE_OK; }
    if (n == "float_records_min") { engine().float_records_min = value; return KGE_OK; }
    if (n == "index_device_min") { engine().index_device_min = value; return KGE_OK; }
    if (n == "hub_copies") { engine().hub_copies = value != 0; return KGE_OK; }
    if (n == "pair_counts") { engine().pair_counts = value != 0; return KGE_OK; }
    if (n == "record_emit_event") { engine().record_emit_event = value != 0; return KGE_OK; }
    if (n == "transr_dgrad_records") { engine().transr_dgrad_records = value != 0; return KGE_OK; }
    if (n == "transr_dgrad_records_min") { engine().transr_dgrad_records_min = value; return KGE_OK; }
    if (n == "transr_lean") { engine().transr_lean = value != 0; return KGE_OK; }
    if (n == "pair_counts_min_neg") { engine().pair_counts_min_neg = (int)value; return KGE_OK; }
    if (n == "lp_v1") { engine().lp_v1 = value != 0; return KGE_OK; }
    if (n == "transr_v1") { engine().transr_v1 = (int)value; return KGE_OK; }
    if (n == "time_emit") { engine().time_emit = value > 0 ? (int)value : 0; if (value > 0) { engine().emit_launches = 0; engine().emit_seen = 0; } return KGE_OK; }
    if (n == "fb_occ4") { engine().fb_occ4 = value != 0; return KGE_OK; }
    if (n == "persist_ahead") { engine().persist_ahead = value != 0; return KGE_OK; }
    if (n == "persist_touch") { engine().persist_touch = value != 0; return KGE_OK; }
    if (n == "persist_trace") { engine().persist_trace = value != 0; return KGE_OK; }
    if (n == "persist_threads") { engine().persist_threads = value == 1024 ? 1024 : 512; return KGE_OK; }
    if (n == "libc_rand_restart") { engine().libc = LibcRand(); return KGE_OK; }  // as in a fresh process
    return fail(KGE_ERR_BAD_ARG, "kge_set_option: unknown option " + n);
}

int kge_last_kernel_ms(const char *name, float *ms) {
    Engine &e = engine();
    std::string n = name ? name : "";
    if (n != "transe_emit" || !ms) return fail(KGE_ERR_BAD_ARG, "kge_last_kernel_ms: unknown kernel " + n);
    if (e.emit_launches <= 0) return fail(KGE_ERR_BAD_ARG, "kge_last_kernel_ms: enable option time_emit and run a step first");
    const int slot = (int)((e.emit_launches - 1) % Engine::kEmitRing);
    if (hip_check(hipEventSynchronize(e.ev_emit1[slot]), "event sync")) return KGE_ERR_NO_DEVICE;
    return hip_check(hipEventElapsedTime(ms, e.ev_emit0[slot], e.ev_emit1[slot]), "event elapsed");
}

int kge_kernel_ms_mean(const char *name, float *mean_ms, INT *launches) {
    Engine &e = engine();
    std::string n = name ? name : "";
    if (n != "transe_emit" || !mean_ms) return fail(KGE_ERR_BAD_ARG, "kge_kernel_ms_mean: unknown kernel " + n);
    const long long have = e.emit_launches < Engine::kEmitRing ? e.emit_launches : (long long)Engine::kEmitRing;
    if (have <= 0) return fail(KGE_ERR_BAD_ARG, "kge_kernel_ms_mean: enable option time_emit and run steps first");
    double sum = 0.0;
    for (long long i = 0; i < have; i++) {
        const int slot = (int)((e.emit_launches - 1 - i) % Engine::kEmitRing);
        float ms = 0.f;
        if (hip_check(hipEventSynchronize(e.ev_emit1[slot]), "event sync")) return KGE_ERR_NO_DEVICE;
        if (hip_check(hipEventElapsedTime(&ms, e.ev_emit0[slot], e.ev_emit1[slot]), "event elapsed")) return KGE_ERR_NO_DEVICE;
        sum += ms;
    }
    *mean_ms = (float)(sum / (double)have);
    if (launches) *launches = have;
    return KGE_OK;
}

int kge_import_train_arrays(INT ent_total, INT rel_total, INT n, const INT *h, const INT *t, const INT *r,
                            INT new_batch_total) {
    std::string err = build_and_adopt(ent_total, rel_total, new_batch_total, n, (const int64_t *)h, (const int64_t *)t,
                                      (const int64_t *)r);
    if (!err.empty()) return fail(KGE_ERR_BAD_ARG, err);
    return KGE_OK;
}

int64_t kge_index_copy(const char *what, void *dst, int64_t bytes) {
    const KgIndex &ix = engine().index;
    if (!ix.loaded) return fail(KGE_ERR_NO_DATASET, "kge_index_copy: no dataset");
    const void *src = nullptr;
    int64_t have = 0;
    std::string w = what ? what : "";
#define KGE_ARR(name, vec) if (w == name) { src = (vec).data(); have = (int64_t)((vec).size() * sizeof((vec)[0])); }
    KGE_ARR("tails_hr", ix.tails_hr) KGE_ARR("heads_tr", ix.heads_tr) KGE_ARR("rels_ht", ix.rels_ht)
    KGE_ARR("pos", ix.pos) KGE_ARR("grp", ix.grp) KGE_ARR("ht", ix.ht)
    KGE_ARR("left_mean", ix.left_mean) KGE_ARR("right_mean", ix.right_mean) KGE_ARR("bern_prob", ix.bern_prob)
#undef KGE_ARR
    if (!src && have == 0 && w != "tails_hr" && w != "heads_tr" && w != "rels_ht" && w != "pos" && w != "grp" && w != "ht" &&
        w != "left_mean" && w != "right_mean" && w != "bern_prob")
        return fail(KGE_ERR_BAD_ARG, "kge_index_copy: unknown array " + w);
    if (dst && bytes > 0 && have > 0) std::memcpy(dst, src, (size_t)(bytes < have ? bytes : have));
    return have;
}

int kge_get_stream_states(uint64_t *dst, INT n) {
    Engine &e = engine();
    int rc = pull_streams();
    if (rc) return rc;
    for (INT i = 0; i < n && i < (INT)e.streams.size(); i++) dst[i] = e.streams[(size_t)i];
    return KGE_OK;
}

int kge_set_stream_states(const uint64_t *src, INT n) {
    Engine &e = engine();
    if (n != e.work_threads) return fail(KGE_ERR_BAD_ARG, "kge_set_stream_states: n must equal workThreads");
    // a sampler launched on another (non-blocking) stream may still be writing the other half of the device state buffer, which
    // the next upload reuses; nothing orders a blocking copy on the null stream behind it, so drain the device first (rare,
    // control-path call: restore / tests)
    if (e.dev.streams && device_ok()) { (void)flush_attached_sampler(nullptr); (void)hipDeviceSynchronize(); }
    e.streams.assign(src, src + n);
    e.dev.streams_sync = 0;
    return KGE_OK;
}

INT kge_slice_positions(INT batchSize, INT thread_lo, INT thread_hi, INT *first_position) {
    const int64_t W = engine().work_threads;
    if (batchSize <= 0 || thread_lo < 0 || thread_hi > W || thread_lo >= thread_hi) {
        if (first_position) *first_position = 0;
        return 0;
    }
    int64_t lo, hi, tmp;
    thread_slice(batchSize, W, thread_lo, lo, tmp);
    thread_slice(batchSize, W, thread_hi - 1, tmp, hi);
    if (first_position) *first_position = lo;
    return hi - lo;
}

int kge_sampling_attach(int32_t *d_h, int32_t *d_t, int32_t *d_r, INT batchSize, INT negRate, INT negRelRate,
                        INT thread_lo, INT thread_hi, INT out_stride, INT *n_local, void *stream) {
    int64_t nl = 0;
    int rc = attach_sampler(d_h, d_t, d_r, batchSize, negRate, negRelRate, thread_lo, thread_hi, out_stride, &nl, (hipStream_t)stream);
    if (n_local) *n_local = nl;
    return rc;
}

int kge_sampling_flush(void *stream) { return flush_attached_sampler((hipStream_t)stream); }

int kge_sampling_device(int32_t *d_h, int32_t *d_t, int32_t *d_r, INT batchSize, INT negRate, INT negRelRate,
                        INT thread_lo, INT thread_hi, INT out_stride, INT *n_local, void *stream) {
    int64_t nl = 0;
    int rc = launch_sampler(d_h, d_t, d_r, batchSize, negRate, negRelRate, thread_lo, thread_hi, out_stride, &nl,
                            (hipStream_t)stream);
    if (n_local) *n_local = nl;
    return rc;
}

int kge_table_shape(const kge_model_desc *m, int table, int64_t *rows, int64_t *cols) {
    if (!m || !rows || !cols) return KGE_ERR_BAD_ARG;
    *rows = 0; *cols = 0;
    switch (table) {
        case 0: *rows = m->ent_total; *cols = m->ent_dim; break;
        case 1: *rows = m->rel_total; *cols = m->rel_dim; break;
        case 2:
            if (m->model == KGE_TRANSH || m->model == KGE_TRANSD) { *rows = m->rel_total; *cols = m->rel_dim; }
            if (m->model == KGE_TRANSR) { *rows = m->rel_total; *cols = (int64_t)m->ent_dim * m->rel_dim; }
            break;
        case 3:
            if (m->model == KGE_TRANSD) { *rows = m->ent_total; *cols = m->ent_dim; }
            break;
        default: return KGE_ERR_BAD_ARG;
    }
    return KGE_OK;
}

int kge_forward_backward(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], const int32_t *d_h,
                         const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom,
                         float *const grads[KGE_MAX_TABLES], float *d_loss, void *stream) {
    if (!m || !tables || !grads || !d_loss) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward: null argument");
    return launch_forward_backward(*m, tables, d_h, d_t, d_r, n_pos, n_neg, stride, denom, grads, d_loss, (hipStream_t)stream, false);
}

int kge_forward_backward_sgd_rows(const kge_model_desc *m, float *const tables[KGE_MAX_TABLES], const int32_t *d_h, const int32_t *d_t,
                                  const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, float lr, float *d_loss, void *stream) {
    if (!m || !tables || !d_loss) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward_sgd_rows: null argument");
    if (!(lr > 0.f)) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward_sgd_rows: learning rate must be positive");
    return launch_forward_backward(*m, tables, d_h, d_t, d_r, n_pos, n_neg, stride, denom, tables, d_loss, (hipStream_t)stream, false, lr);
}

int kge_forward_backward_records(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], const int32_t *d_h, const int32_t *d_t,
                                 const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom, INT n_pos_total, float *d_rec,
                                 int32_t *d_dst, INT rec_offset, INT rec_slice, float *d_loss, void *stream) {
    if (!m || !tables || !d_loss) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward_records: null argument");
    return launch_forward_backward_records(*m, tables, d_h, d_t, d_r, n_pos, n_neg, stride, denom, n_pos_total, d_rec, d_dst, rec_offset,
                                           rec_slice, d_loss, (hipStream_t)stream);
}

int kge_float_records_apply(const kge_model_desc *m, float *const tables[KGE_MAX_TABLES], const float *d_rec, int32_t *d_dst, INT n_records,
                            INT n_pos_total, INT n_neg, float lr, void *stream) {
    if (!m || !tables) return fail(KGE_ERR_BAD_ARG, "kge_float_records_apply: null argument");
    return launch_float_records_apply(*m, tables, d_rec, d_dst, n_records, n_pos_total, n_neg, lr, (hipStream_t)stream);
}

int kge_loss_limbs_target(int32_t *d_limbs4) {
    engine().loss_limbs = d_limbs4;
    return KGE_OK;
}

int kge_sgd_rows_skipped(int32_t *n_negatives) {
    if (!n_negatives) return fail(KGE_ERR_BAD_ARG, "kge_sgd_rows_skipped: null argument");
    return sgd_rows_skipped(n_negatives);
}

int kge_pair_path_active(const kge_model_desc *m, INT n_pos, INT n_neg) { return m && pair_path_active(*m, n_pos, n_neg) ? 1 : 0; }

int kge_stream_wait_emit(void *stream) {
    Engine &e = engine();
    // nothing recorded since the last wait (recording off, or this step's path has no emit kernel): the caller must order the
    // side stream some other way -- returning OK here would leave it with no dependency on the step at all
    if (!e.emit_done || e.emit_seq == e.emit_waited) return KGE_NO_EVENT;
    e.emit_waited = e.emit_seq;
    return hip_check(hipStreamWaitEvent((hipStream_t)stream, e.emit_done, 0), "wait for the emit kernel");
}

int kge_forward_backward_sampled(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], const int32_t *d_h,
                                 const int32_t *d_t, const int32_t *d_r, INT n_pos, INT n_neg, INT stride, INT denom,
                                 float *const grads[KGE_MAX_TABLES], float *d_loss, void *stream) {
    if (!m || !tables || !grads || !d_loss) return fail(KGE_ERR_BAD_ARG, "kge_forward_backward_sampled: null argument");
    return launch_forward_backward(*m, tables, d_h, d_t, d_r, n_pos, n_neg, stride, denom, grads, d_loss, (hipStream_t)stream, true);
}

int kge_sgd_update(float *d_p, float *d_g, int64_t n, float lr, void *stream) {
    tables_written();
    return launch_sgd(d_p, d_g, n, lr, (hipStream_t)stream);
}

int kge_adam_update(float *d_p, float *d_m, float *d_v, float *d_g, int64_t n, float lr_t, float beta1, float beta2,
                    float eps, void *stream) {
    tables_written();
    return launch_adam(d_p, d_m, d_v, d_g, n, lr_t, beta1, beta2, eps, (hipStream_t)stream);
}

int kge_sgd_update_tables(int32_t n_tables, float *const d_p[KGE_MAX_TABLES], float *const d_g[KGE_MAX_TABLES],
                          const INT numel[KGE_MAX_TABLES], float lr, void *stream) {
    tables_written();
    return launch_sgd_tables(n_tables, d_p, d_g, (const int64_t *)numel, lr, (hipStream_t)stream);
}

int kge_adam_update_tables(int32_t n_tables, float *const d_p[KGE_MAX_TABLES], float *const d_m[KGE_MAX_TABLES],
                           float *const d_v[KGE_MAX_TABLES], float *const d_g[KGE_MAX_TABLES], const INT numel[KGE_MAX_TABLES],
                           float lr_t, float beta1, float beta2, float eps, void *stream) {
    tables_written();
    return launch_adam_tables(n_tables, d_p, d_m, d_v, d_g, (const int64_t *)numel, lr_t, beta1, beta2, eps, (hipStream_t)stream);
}

int kge_predict(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], const int32_t *d_h,
                const int32_t *d_t, const int32_t *d_r, INT n, float *d_out, void *stream) {
    if (!m || !tables || !d_out) return fail(KGE_ERR_BAD_ARG, "kge_predict: null argument");
    return launch_predict(*m, tables, d_h, d_t, d_r, n, d_out, (hipStream_t)stream);
}

}  // extern "C"
