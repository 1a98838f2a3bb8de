// Link-prediction evaluation: the inputs of base/Reader.h:186-449 and the
// ranker of base/Test.h:11-249, with the Base.so entry points kept (importTestFiles, importTypeFiles,
// importOntologyFiles, getHeadBatch, getTailBatch, testHead, testTail) and a device-native batch
// entry (kge_link_prediction) that never moves the E-long score vectors over PCIe.
//
// Ranker on the device: one workgroup per (test triple, side).  Every thread walks candidates
// j = tid, tid+256, ...; a candidate scoring strictly lower than the expected entity bumps the raw
// count, the filtered count unless (j,r,t)/(h,r,j) is a known triple (binary search in the
// (h,r,t)-sorted union of train+valid+test, Corrupt.h:104-115), and the type-constrained counts when
// j is in the relation's sorted head/tail type list.  The four arg-mins are lexicographic minima
// over (score, j) -- the sequential loop's strict `<` keeps the first (smallest j) minimum -- and
// their ontology classes are resolved by one thread with the reference's non-rewinding cursors.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine.hpp"

namespace kge {

struct EvalHost {
    bool loaded = false, types = false, onto = false;
    int64_t test_total = 0, valid_total = 0, triple_total = 0;
    std::vector<Int4> test;         // (h,t,r,0) sorted by (r,h,t)   Reader.h:256
    std::vector<Int4> all;          // (h,r,t,0) sorted by (h,r,t)   Reader.h:255
    std::vector<int32_t> head_lef, head_rig, tail_lef, tail_rig, head_type, tail_type;
    std::vector<int32_t> sup_lef, sup_rig, sub_lef, sub_rig, sup_type, sub_type;
};

struct EvalDev {
    bool uploaded = false;
    int4 *test = nullptr, *all = nullptr, *all_t = nullptr;   // all_t: the same triples as (t,r,h,0) sorted by (t,r,h)
    int32_t *head_lef = nullptr, *head_rig = nullptr, *tail_lef = nullptr, *tail_rig = nullptr, *head_type = nullptr, *tail_type = nullptr;
    int32_t *sup_lef = nullptr, *sup_rig = nullptr, *sub_lef = nullptr, *sub_rig = nullptr, *sup_type = nullptr, *sub_type = nullptr;
    float *scores = nullptr;      // staging for testHead/testTail and kge_link_prediction
    int64_t scores_cap = 0;
    int32_t *cand = nullptr;      // 3 * cap candidate ids
    long long *out = nullptr;     // 8 * requests
    int64_t out_cap = 0;
};

static EvalHost g_eh;
static EvalDev g_ed;
// triple classification: valid triples (h,t,r) sorted by (r,h,t) (Reader.h:257) and the per-relation
// ranges of the sorted valid / test lists (Reader.h:263-291)
static std::vector<Int4> g_valid;
static std::vector<int32_t> g_valid_lef, g_valid_rig, g_test_lef, g_test_rig;

template <typename T, typename V>
static int up(T *&dst, const std::vector<V> &src, const char *what) {
    static_assert(sizeof(T) == sizeof(V), "size mismatch");
    if (dst) { (void)hipFree(dst); dst = nullptr; }
    int rc = hip_check(hipMalloc(&dst, sizeof(T) * (src.size() ? src.size() : 1)), what);
    if (rc) return rc;
    if (!src.empty()) rc = hip_check(hipMemcpy(dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice), what);
    return rc;
}

static int ensure_eval_device() {
    if (!g_eh.loaded) return fail(KGE_ERR_NO_DATASET, "importTestFiles has not been called");
    if (!device_ok()) return fail(KGE_ERR_NO_DEVICE, "no usable HIP device: the ranker has no CPU fallback");
    if (g_ed.uploaded) return KGE_OK;
    int rc;
    const int64_t E = engine().index.ent_total, R = engine().index.rel_total;
    if (g_eh.head_lef.empty()) { g_eh.head_lef.assign(R, 0); g_eh.head_rig.assign(R, 0); g_eh.tail_lef.assign(R, 0); g_eh.tail_rig.assign(R, 0); }
    if (g_eh.sup_lef.empty()) { g_eh.sup_lef.assign(E, 0); g_eh.sup_rig.assign(E, 0); g_eh.sub_lef.assign(E, 0); g_eh.sub_rig.assign(E, 0); }
    if ((rc = up(g_ed.test, g_eh.test, "upload test"))) return rc;
    if ((rc = up(g_ed.all, g_eh.all, "upload triples"))) return rc;
    {   // second order for head requests: the known heads of a (t, r) pair are then contiguous
        std::vector<Int4> by_tail(g_eh.all.size());
        for (size_t i = 0; i < by_tail.size(); i++) by_tail[i] = Int4{g_eh.all[i].z, g_eh.all[i].y, g_eh.all[i].x, 0};
        std::sort(by_tail.begin(), by_tail.end(), [](const Int4 &a, const Int4 &b) {
            if (a.x != b.x) return a.x < b.x; if (a.y != b.y) return a.y < b.y; return a.z < b.z; });
        if ((rc = up(g_ed.all_t, by_tail, "upload triples by tail"))) return rc;
    }
    if ((rc = up(g_ed.head_lef, g_eh.head_lef, "upload types"))) return rc;
    if ((rc = up(g_ed.head_rig, g_eh.head_rig, "upload types"))) return rc;
    if ((rc = up(g_ed.tail_lef, g_eh.tail_lef, "upload types"))) return rc;
    if ((rc = up(g_ed.tail_rig, g_eh.tail_rig, "upload types"))) return rc;
    if ((rc = up(g_ed.head_type, g_eh.head_type, "upload types"))) return rc;
    if ((rc = up(g_ed.tail_type, g_eh.tail_type, "upload types"))) return rc;
    if ((rc = up(g_ed.sup_lef, g_eh.sup_lef, "upload ontology"))) return rc;
    if ((rc = up(g_ed.sup_rig, g_eh.sup_rig, "upload ontology"))) return rc;
    if ((rc = up(g_ed.sub_lef, g_eh.sub_lef, "upload ontology"))) return rc;
    if ((rc = up(g_ed.sub_rig, g_eh.sub_rig, "upload ontology"))) return rc;
    if ((rc = up(g_ed.sup_type, g_eh.sup_type, "upload ontology"))) return rc;
    if ((rc = up(g_ed.sub_type, g_eh.sub_type, "upload ontology"))) return rc;
    g_ed.uploaded = true;
    return KGE_OK;
}

// ---------------------------------------------------------------------------------------------
struct RankArgs {
    const float *scores;     // [n_req][E]
    const int4 *test, *all, *all_t;
    long long n_all;
    const int32_t *head_lef, *head_rig, *tail_lef, *tail_rig, *head_type, *tail_type;
    const int32_t *sup_lef, *sup_rig, *sub_lef, *sub_rig, *sup_type, *sub_type;
    const int32_t *req_index;  // [n_req] test triple index
    const int32_t *req_head;   // [n_req] 1 = replace head, 0 = replace tail
    long long *out;            // [n_req][8]
    int E;
};

// [lo, hi) of the entries whose first two fields are (a, b) in an array sorted by (x, y, z): the third fields of that
// range are the known tails of (h, r) in `all`, or the known heads of (t, r) in `all_t`, in increasing order
__device__ __forceinline__ void pair_range(const int4 *__restrict__ arr, long long n, int a, int b, long long &lo, long long &hi) {
    long long l = 0, r = n;
    while (l < r) { const long long mid = (l + r) >> 1; const int4 m = arr[mid]; if (m.x < a || (m.x == a && m.y < b)) l = mid + 1; else r = mid; }
    lo = l;
    r = n;
    while (l < r) { const long long mid = (l + r) >> 1; const int4 m = arr[mid]; if (m.x < a || (m.x == a && m.y <= b)) l = mid + 1; else r = mid; }
    hi = l;
}
__device__ __forceinline__ bool in_range(const int4 *__restrict__ arr, long long lo, long long hi, int j) {
    const long long end = hi;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (arr[mid].z < j) lo = mid + 1; else hi = mid; }
    return lo < end && arr[lo].z == j;
}

struct MinPair { float v; int j; };
__device__ __forceinline__ void take(MinPair &a, float v, int j) { if (v < a.v || (v == a.v && j < a.j)) { a.v = v; a.j = j; } }

__global__ __launch_bounds__(256) void rank_kernel(RankArgs a) {
    const int req = blockIdx.x;
    const float *con = a.scores + (long long)req * a.E;
    const int4 tt = a.test[a.req_index[req]];  // (h, t, r)
    const bool head = a.req_head[req] != 0;
    const int h = tt.x, t = tt.y, r = tt.z;
    const int target = head ? h : t;
    const float minimal = con[target];
    const int lef = head ? a.head_lef[r] : a.tail_lef[r], rig = head ? a.head_rig[r] : a.tail_rig[r];
    const int32_t *types = head ? a.head_type : a.tail_type;
    // the filter (Corrupt.h:104-115 `_find` over train+valid+test): the known tails of (h, r) / heads of (t, r) are one
    // contiguous, sorted range, located once per request; a candidate then costs a search in that short range
    __shared__ long long s_known[2];
    if (threadIdx.x == 0) {
        long long lo, hi;
        if (head) pair_range(a.all_t, a.n_all, t, r, lo, hi); else pair_range(a.all, a.n_all, h, r, lo, hi);
        s_known[0] = lo; s_known[1] = hi;
    }
    __syncthreads();
    const long long known_lo = s_known[0], known_hi = s_known[1];
    const int4 *known_arr = head ? a.all_t : a.all;
    long long c[4] = {0, 0, 0, 0};
    // arg-min candidates start at (minimal, target); a candidate must be STRICTLY lower to replace it
    MinPair m[4];
    for (int i = 0; i < 4; i++) { m[i].v = minimal; m[i].j = 0x7fffffff; }
    for (int j = threadIdx.x; j < a.E; j += 256) {
        if (j == target) continue;
        const float value = con[j];
        if (!(value < minimal)) continue;
        const bool known = in_range(known_arr, known_lo, known_hi, j);
        bool typed = false;
        {   // lower_bound in the relation's sorted type list
            int lo = lef, hi = rig;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (types[mid] < j) lo = mid + 1; else hi = mid; }
            typed = lo < rig && types[lo] == j;
        }
        c[0]++; take(m[0], value, j);
        if (!known) c[1]++;
        if (head || !known) take(m[1], value, j);   // Test.h:69-74: the head version updates this arg-min outside the filter
        if (typed) { c[2]++; take(m[2], value, j); if (!known) { c[3]++; take(m[3], value, j); } }
    }
    __shared__ long long sc[4][256];
    __shared__ float sv[4][256];
    __shared__ int sj[4][256];
    for (int i = 0; i < 4; i++) { sc[i][threadIdx.x] = c[i]; sv[i][threadIdx.x] = m[i].v; sj[i][threadIdx.x] = m[i].j; }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            for (int i = 0; i < 4; i++) {
                sc[i][threadIdx.x] += sc[i][threadIdx.x + w];
                MinPair x = {sv[i][threadIdx.x], sj[i][threadIdx.x]};
                take(x, sv[i][threadIdx.x + w], sj[i][threadIdx.x + w]);
                sv[i][threadIdx.x] = x.v; sj[i][threadIdx.x] = x.j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        long long *o = a.out + (long long)req * 8;
        int arg[4];
        for (int i = 0; i < 4; i++) { o[i] = sc[i][0]; arg[i] = sv[i][0] < minimal ? sj[i][0] : target; }
        // ontology classes with the reference's shared, never-rewinding cursors (Test.h:113-135)
        int lsup = a.sup_lef[target], rsup = a.sup_rig[target], lsub = a.sub_lef[target], rsub = a.sub_rig[target];
        for (int i = 0; i < 4; i++) {
            const int v = arg[i];
            long long cls = 3;
            if (v == target) cls = 0;
            else {
                while (lsup < rsup && a.sup_type[lsup] < v) lsup++;
                if (lsup < rsup && a.sup_type[lsup] == v) cls = 1;
                else {
                    while (lsub < rsub && a.sub_type[lsub] < v) lsub++;
                    if (lsub < rsub && a.sub_type[lsub] == v) cls = 2;
                }
            }
            o[4 + i] = cls;
        }
    }
}

// candidates of getTailBatch / getHeadBatch (Test.h:11-26) for a batch of requests, on the device
__global__ void candidates_kernel(const int4 *__restrict__ test, const int32_t *__restrict__ req_index,
                                  const int32_t *__restrict__ req_head, int n_req, int E, int32_t *__restrict__ ch,
                                  int32_t *__restrict__ ct, int32_t *__restrict__ cr) {
    const long long total = (long long)n_req * E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i / E), j = (int)(i - (long long)q * E);
        const int4 tt = test[req_index[q]];
        const bool head = req_head[q] != 0;
        ch[i] = head ? j : tt.x;
        ct[i] = head ? tt.y : j;
        cr[i] = tt.z;
    }
}

static int rank_requests(const float *d_scores, const std::vector<int32_t> &index, const std::vector<int32_t> &head,
                         long long *h_out, hipStream_t stream, const int32_t *d_req_index, const int32_t *d_req_head) {
    const int n = (int)index.size();
    if (g_ed.out_cap < n) {
        if (g_ed.out) (void)hipFree(g_ed.out);
        g_ed.out = nullptr;
        int rc = hip_check(hipMalloc(&g_ed.out, sizeof(long long) * 8 * (size_t)n), "alloc rank out");
        if (rc) return rc;
        g_ed.out_cap = n;
    }
    RankArgs a;
    a.scores = d_scores; a.test = g_ed.test; a.all = g_ed.all; a.all_t = g_ed.all_t; a.n_all = (long long)g_eh.all.size();
    a.head_lef = g_ed.head_lef; a.head_rig = g_ed.head_rig; a.tail_lef = g_ed.tail_lef; a.tail_rig = g_ed.tail_rig;
    a.head_type = g_ed.head_type; a.tail_type = g_ed.tail_type;
    a.sup_lef = g_ed.sup_lef; a.sup_rig = g_ed.sup_rig; a.sub_lef = g_ed.sub_lef; a.sub_rig = g_ed.sub_rig;
    a.sup_type = g_ed.sup_type; a.sub_type = g_ed.sub_type;
    a.req_index = d_req_index; a.req_head = d_req_head; a.out = g_ed.out; a.E = (int)engine().index.ent_total;
    hipLaunchKernelGGL(rank_kernel, dim3(n), dim3(256), 0, stream, a);
    int rc = hip_check(hipMemcpyAsync(h_out, g_ed.out, sizeof(long long) * 8 * (size_t)n, hipMemcpyDeviceToHost, stream), "copy ranks");
    if (rc) return rc;
    return hip_check(hipStreamSynchronize(stream), "rank sync");
}

}  // namespace kge

using namespace kge;

extern "C" {

void importTestFiles(void) {
    Engine &e = engine();
    g_eh = EvalHost();
    g_ed.uploaded = false;
    g_valid.clear();
    std::vector<int64_t> te, tr, va, tmp;
    const std::string &d = e.in_path;
    if (!read_all_longs(d + "relation2id.txt", tmp)) { set_error("`" + d + "relation2id.txt` does not exist"); return; }
    if (!read_all_longs(d + "entity2id.txt", tmp)) { set_error("`" + d + "entity2id.txt` does not exist"); return; }
    if (!read_all_longs(d + "test2id.txt", te)) { set_error("`" + d + "test2id.txt` does not exist"); return; }
    if (!read_all_longs(d + "train2id.txt", tr)) { set_error("`" + d + "train2id.txt` does not exist"); return; }
    if (!read_all_longs(d + "valid2id.txt", va)) { set_error("`" + d + "valid2id.txt` does not exist"); return; }
    const int64_t T = te.empty() ? 0 : te[0], Tr = tr.empty() ? 0 : tr[0], V = va.empty() ? 0 : va[0];
    if ((int64_t)te.size() < 1 + 3 * T || (int64_t)tr.size() < 1 + 3 * Tr || (int64_t)va.size() < 1 + 3 * V) { set_error("importTestFiles: truncated triple file"); return; }
    g_eh.test_total = T; g_eh.valid_total = V; g_eh.triple_total = T + Tr + V;
    auto push = [&](const std::vector<int64_t> &src, int64_t n, bool is_test) {
        for (int64_t i = 0; i < n; i++) {  // on disk: head, tail, relation
            const int32_t h = (int32_t)src[1 + 3 * i], t = (int32_t)src[2 + 3 * i], r = (int32_t)src[3 + 3 * i];
            g_eh.all.push_back(Int4{h, r, t, 0});
            if (is_test) g_eh.test.push_back(Int4{h, t, r, 0});
        }
    };
    push(te, T, true); push(tr, Tr, false); push(va, V, false);
    std::sort(g_eh.all.begin(), g_eh.all.end(), [](const Int4 &a, const Int4 &b) {        // Triple.h:18-20
        if (a.x != b.x) return a.x < b.x; if (a.y != b.y) return a.y < b.y; return a.z < b.z; });
    std::sort(g_eh.test.begin(), g_eh.test.end(), [](const Int4 &a, const Int4 &b) {      // Triple.h:30-32 (r,h,t)
        if (a.z != b.z) return a.z < b.z; if (a.x != b.x) return a.x < b.x; return a.y < b.y; });
    std::printf("The total of test triples is %ld.\n", (long)T);
    std::printf("The total of valid triples is %ld.\n", (long)V);
    g_eh.loaded = true;
}

void importTypeFiles(void) {
    Engine &e = engine();
    std::vector<int64_t> a;
    if (!read_all_longs(e.in_path + "type_constrain.txt", a)) { set_error("`" + e.in_path + "type_constrain.txt` does not exist"); return; }
    const int64_t R = e.index.rel_total;
    g_eh.head_lef.assign(R, 0); g_eh.head_rig.assign(R, 0); g_eh.tail_lef.assign(R, 0); g_eh.tail_rig.assign(R, 0);
    g_eh.head_type.clear(); g_eh.tail_type.clear();
    size_t p = 1;
    for (int64_t i = 0; i < R && p + 1 < a.size(); i++) {   // Reader.h:344-362
        int64_t rel = a[p], tot = a[p + 1]; p += 2;
        if (rel < 0 || rel >= R || p + tot > a.size()) { set_error("type_constrain.txt: malformed"); return; }
        g_eh.head_lef[rel] = (int32_t)g_eh.head_type.size();
        for (int64_t j = 0; j < tot; j++) g_eh.head_type.push_back((int32_t)a[p++]);
        g_eh.head_rig[rel] = (int32_t)g_eh.head_type.size();
        std::sort(g_eh.head_type.begin() + g_eh.head_lef[rel], g_eh.head_type.end());
        if (p + 1 >= a.size()) break;
        rel = a[p]; tot = a[p + 1]; p += 2;
        if (rel < 0 || rel >= R || p + tot > a.size()) { set_error("type_constrain.txt: malformed"); return; }
        g_eh.tail_lef[rel] = (int32_t)g_eh.tail_type.size();
        for (int64_t j = 0; j < tot; j++) g_eh.tail_type.push_back((int32_t)a[p++]);
        g_eh.tail_rig[rel] = (int32_t)g_eh.tail_type.size();
        std::sort(g_eh.tail_type.begin() + g_eh.tail_lef[rel], g_eh.tail_type.end());
    }
    g_eh.types = true;
    g_ed.uploaded = false;
}

void importOntologyFiles(void) {
    Engine &e = engine();
    std::printf("Reading %sontology_constrain.txt\n", e.in_path.c_str());
    std::vector<int64_t> a;
    if (!read_all_longs(e.in_path + "ontology_constrain.txt", a)) { set_error("`" + e.in_path + "ontology_constrain.txt` does not exist"); return; }
    const int64_t E = e.index.ent_total;
    g_eh.sup_lef.assign(E, 0); g_eh.sup_rig.assign(E, 0); g_eh.sub_lef.assign(E, 0); g_eh.sub_rig.assign(E, 0);
    g_eh.sup_type.clear(); g_eh.sub_type.clear();
    const int64_t n = a.empty() ? 0 : a[0];
    size_t p = 1;
    for (int64_t i = 0; i < n && p + 1 < a.size(); i++) {   // Reader.h:425-446
        int64_t ent = a[p], tot = a[p + 1]; p += 2;
        if (ent < 0 || ent >= E || p + tot > a.size()) { set_error("ontology_constrain.txt: malformed"); return; }
        g_eh.sup_lef[ent] = (int32_t)g_eh.sup_type.size();
        for (int64_t j = 0; j < tot; j++) g_eh.sup_type.push_back((int32_t)a[p++]);
        g_eh.sup_rig[ent] = (int32_t)g_eh.sup_type.size();
        std::sort(g_eh.sup_type.begin() + g_eh.sup_lef[ent], g_eh.sup_type.end());
        if (p + 1 >= a.size()) break;
        ent = a[p]; tot = a[p + 1]; p += 2;
        if (ent < 0 || ent >= E || p + tot > a.size()) { set_error("ontology_constrain.txt: malformed"); return; }
        g_eh.sub_lef[ent] = (int32_t)g_eh.sub_type.size();
        for (int64_t j = 0; j < tot; j++) g_eh.sub_type.push_back((int32_t)a[p++]);
        g_eh.sub_rig[ent] = (int32_t)g_eh.sub_type.size();
        std::sort(g_eh.sub_type.begin() + g_eh.sub_lef[ent], g_eh.sub_type.end());
    }
    g_eh.onto = true;
    g_ed.uploaded = false;
}

INT kge_eval_test_total(void) { return g_eh.test_total; }
INT kge_eval_valid_total(void) { return g_eh.valid_total; }
INT kge_eval_triple_total(void) { return g_eh.triple_total; }

void getHeadBatch(INT index, INT *ph, INT *pt, INT *pr) {   // Test.h:11-17
    if (!g_eh.loaded || index < 0 || index >= g_eh.test_total) { set_error("getHeadBatch: bad index / importTestFiles missing"); return; }
    const Int4 &tt = g_eh.test[(size_t)index];
    for (INT i = 0; i < engine().index.ent_total; i++) { ph[i] = i; pt[i] = tt.y; pr[i] = tt.z; }
}

void getTailBatch(INT index, INT *ph, INT *pt, INT *pr) {   // Test.h:20-26
    if (!g_eh.loaded || index < 0 || index >= g_eh.test_total) { set_error("getTailBatch: bad index / importTestFiles missing"); return; }
    const Int4 &tt = g_eh.test[(size_t)index];
    for (INT i = 0; i < engine().index.ent_total; i++) { ph[i] = tt.x; pt[i] = i; pr[i] = tt.z; }
}

static INT *rank_one_host_scores(INT index, REAL *con, int head) {
    // the reference returns `new INT[8]` that its callers never free (Test.h:37, Config.py:36-39);
    // here a small ring of result slots is reused instead
    static long long ring[64][8];
    static int slot = 0;
    long long *out = ring[slot];
    slot = (slot + 1) % 64;
    std::memset(out, 0, sizeof(long long) * 8);
    if (ensure_eval_device()) return (INT *)out;
    if (index < 0 || index >= g_eh.test_total) { set_error("testHead/testTail: index out of range"); return (INT *)out; }
    const int64_t E = engine().index.ent_total;
    if (g_ed.scores_cap < E) {
        if (g_ed.scores) (void)hipFree(g_ed.scores);
        g_ed.scores = nullptr;
        if (hip_check(hipMalloc(&g_ed.scores, sizeof(float) * (size_t)E), "alloc scores")) return (INT *)out;
        g_ed.scores_cap = E;
    }
    static int32_t *d_req = nullptr;
    if (!d_req && hip_check(hipMalloc(&d_req, sizeof(int32_t) * 2), "alloc req")) return (INT *)out;
    int32_t req[2] = {(int32_t)index, head};
    if (hip_check(hipMemcpy(g_ed.scores, con, sizeof(float) * (size_t)E, hipMemcpyHostToDevice), "upload scores")) return (INT *)out;
    if (hip_check(hipMemcpy(d_req, req, sizeof(req), hipMemcpyHostToDevice), "upload req")) return (INT *)out;
    std::vector<int32_t> idx{(int32_t)index}, hd{head};
    rank_requests(g_ed.scores, idx, hd, out, nullptr, d_req, d_req + 1);
    return (INT *)out;
}

INT *testHead(INT index, REAL *con) { return rank_one_host_scores(index, con, 1); }   // Test.h:31-136
INT *testTail(INT index, REAL *con) { return rank_one_host_scores(index, con, 0); }   // Test.h:141-249

/* ------------------------------------------------------------------------------------------------
 * Triple classification (Test.h:262-444).  These are the reference's HOST
 * routines over validTotal / testTotal-long score arrays (run once per early-stop check, not a hot
 * path): negatives drawn with libc rand() from the relation's tail type list until unknown
 * (Corrupt.h:118-137), a per-relation threshold grid search with step `interval` = 0.01f
 * (Setting.h:118), accuracy / TP-FP counts.  Restated in float exactly as written there.
 * ---------------------------------------------------------------------------------------------- */
static const float kInterval = 0.01f;   // Setting.h:118
// min_score + i * interval (Test.h:326).  The reference is built with -O3 -march=native (make.sh:1), where
// GCC contracts this expression into ONE fused multiply-add; a separate multiply and add rounds twice and
// lands on different thresholds.  The fused form is what an FMA-capable host produces, so it is pinned here.
static inline float grid_point(float mn, INT i) { return fmaf((float)i, kInterval, mn); }

static bool find_host(int h, int t, int r) {   // Corrupt.h:104-115 on the host copy
    const std::vector<Int4> &all = g_eh.all;
    long long lef = 0, rig = (long long)all.size() - 1;
    while (lef + 1 < rig) {
        const long long mid = (lef + rig) >> 1;
        const Int4 &m = all[(size_t)mid];
        if (m.x < h || (m.x == h && m.y < r) || (m.x == h && m.y == r && m.z < t)) lef = mid; else rig = mid;
    }
    const Int4 &a = all[(size_t)lef], &b = all[(size_t)rig];
    return (a.x == h && a.y == r && a.z == t) || (b.x == h && b.y == r && b.z == t);
}

// corrupt_head(0, h, r) on the host index (Corrupt.h:7-37), consuming rng stream 0; used only after 1000
// failed type-constrained draws (Corrupt.h:131-133)
static int corrupt_head_stream0(int h, int r) {
    Engine &e = engine();
    if (e.dev.streams_sync == 2 && e.dev.streams)
        (void)hipMemcpy(e.streams.data(), e.dev.streams, sizeof(uint64_t) * e.streams.size(), hipMemcpyDeviceToHost);
    if (e.streams.empty()) e.streams.assign(1, 0);
    e.streams[0] = e.streams[0] * kLcgMul + kLcgAdd;
    e.dev.streams_sync = 0;
    // known tails of (h, r): any file-order triple of the group carries its [offset, length]
    const KgIndex &ix = e.index;
    int off = 0, len = 0;
    for (size_t i = 0; i < ix.pos.size(); i++)
        if (ix.pos[i].x == h && ix.pos[i].z == r) { off = ix.grp[i].x; len = ix.grp[i].y; break; }
    const long long tmp = (long long)(e.streams[0] % (uint64_t)(ix.ent_total - len));
    int lo = 0, hi = len;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)ix.tails_hr[(size_t)off + mid] - mid <= tmp) lo = mid + 1; else hi = mid; }
    return (int)(tmp + lo);
}

static int corrupt_typed(int h, int r) {   // Corrupt.h:118-137
    Engine &e = engine();
    const int ll = g_eh.tail_lef.empty() ? 0 : g_eh.tail_lef[(size_t)r], rr = g_eh.tail_rig.empty() ? 0 : g_eh.tail_rig[(size_t)r];
    if (rr <= ll) { set_error("getValidBatch/getTestBatch: relation without tail type list (importTypeFiles)"); return 0; }
    for (int loop = 0;;) {
        const int t = g_eh.tail_type[(size_t)(ll + e.libc.next() % (rr - ll))];   // Random.h:38-40 rand(a,b)
        if (!find_host(h, t, r)) return t;
        if (++loop >= 1000) return corrupt_head_stream0(h, r);
    }
}


static void build_rel_ranges(const std::vector<Int4> &list, std::vector<int32_t> &lef, std::vector<int32_t> &rig) {
    const int64_t R = engine().index.rel_total;
    lef.assign((size_t)R, -1); rig.assign((size_t)R, -1);
    for (size_t i = 0; i < list.size(); i++) {
        const int r = list[i].z;
        if (lef[(size_t)r] < 0) lef[(size_t)r] = (int32_t)i;
        rig[(size_t)r] = (int32_t)i;
    }
}

static bool ensure_classification_lists() {
    if (!g_eh.loaded) { set_error("triple classification: importTestFiles has not been called"); return false; }
    if (g_valid.size() != (size_t)g_eh.valid_total) {
        // the valid triples are part of `all`; re-read them in file order is not needed: only the sorted list is used
        std::vector<int64_t> va;
        if (!read_all_longs(engine().in_path + "valid2id.txt", va)) { set_error("valid2id.txt does not exist"); return false; }
        g_valid.clear();
        for (int64_t i = 0; i < g_eh.valid_total; i++)
            g_valid.push_back(Int4{(int32_t)va[1 + 3 * i], (int32_t)va[2 + 3 * i], (int32_t)va[3 + 3 * i], 0});
        std::sort(g_valid.begin(), g_valid.end(), [](const Int4 &a, const Int4 &b) {
            if (a.z != b.z) return a.z < b.z; if (a.x != b.x) return a.x < b.x; return a.y < b.y; });
    }
    build_rel_ranges(g_valid, g_valid_lef, g_valid_rig);
    build_rel_ranges(g_eh.test, g_test_lef, g_test_rig);
    return true;
}

static void fill_batch(const std::vector<Int4> &list, INT *ph, INT *pt, INT *pr, INT *nh, INT *nt, INT *nr) {
    for (size_t i = 0; i < list.size(); i++) {   // Test.h:252-300: negative = positive with a new tail
        ph[i] = list[i].x; pt[i] = list[i].y; pr[i] = list[i].z;
        nh[i] = list[i].x; nr[i] = list[i].z;
        nt[i] = corrupt_typed(list[i].x, list[i].z);
    }
}

void getTestBatch(INT *ph, INT *pt, INT *pr, INT *nh, INT *nt, INT *nr) {
    if (!ensure_classification_lists()) return;
    fill_batch(g_eh.test, ph, pt, pr, nh, nt, nr);
}

void getValidBatch(INT *ph, INT *pt, INT *pr, INT *nh, INT *nt, INT *nr) {
    if (!ensure_classification_lists()) return;
    fill_batch(g_valid, ph, pt, pr, nh, nt, nr);
}

// min / max of the relation's validation scores and the grid size (Test.h:310-322, :394-410)
static bool score_range(INT r, const REAL *sp, const REAL *sn, float &mn, float &mx, INT &n_interval) {
    if (r < 0 || r >= (INT)g_valid_lef.size() || g_valid_lef[(size_t)r] == -1) return false;
    const int lo = g_valid_lef[(size_t)r], hi = g_valid_rig[(size_t)r];
    mn = sp[lo]; if (sn[lo] < mn) mn = sn[lo];
    mx = sp[lo]; if (sn[lo] > mx) mx = sn[lo];
    for (int i = lo + 1; i <= hi; i++) {
        if (sp[i] < mn) mn = sp[i];
        if (sp[i] > mx) mx = sp[i];
        if (sn[i] < mn) mn = sn[i];
        if (sn[i] > mx) mx = sn[i];
    }
    n_interval = (INT)((mx - mn) / kInterval);
    return true;
}

void getBestThreshold(REAL *relThresh, REAL *score_pos, REAL *score_neg) {   // Test.h:304-341
    if (!ensure_classification_lists()) return;
    for (INT r = 0; r < engine().index.rel_total; r++) {
        float mn, mx; INT n_interval;
        if (!score_range(r, score_pos, score_neg, mn, mx, n_interval)) continue;
        const int lo = g_valid_lef[(size_t)r], hi = g_valid_rig[(size_t)r];
        const INT total = (INT)(hi - lo + 1) * 2;
        float bestThresh = 0.f, bestAcc = 0.f;
        for (INT i = 0; i <= n_interval; i++) {
            const float tmpThresh = grid_point(mn, i);
            INT correct = 0;
            for (int j = lo; j <= hi; j++) {
                if (score_pos[j] <= tmpThresh) correct++;
                if (score_neg[j] > tmpThresh) correct++;
            }
            const float tmpAcc = (float)(1.0 * correct / total);
            if (i == 0 || tmpAcc > bestAcc) { bestAcc = tmpAcc; bestThresh = tmpThresh; }
        }
        relThresh[r] = bestThresh;
    }
}

void test_triple_classification(REAL *relThresh, REAL *score_pos, REAL *score_neg, REAL *acc_addr) {   // Test.h:347-387
    if (!ensure_classification_lists()) return;
    INT TP = 0, TN = 0, FP = 0, FN = 0;
    for (INT r = 0; r < engine().index.rel_total; r++) {
        if (g_valid_lef[(size_t)r] == -1 || g_test_lef[(size_t)r] == -1) continue;
        for (int i = g_test_lef[(size_t)r]; i <= g_test_rig[(size_t)r]; i++) {
            if (score_pos[i] <= relThresh[r]) TP++; else FN++;
            if (score_neg[i] > relThresh[r]) TN++; else FP++;
        }
    }
    const double accuracy = 1.0 * (TP + TN) / (TP + TN + FP + FN);
    const double precision = 1.0 * TP / (TP + FP);
    const double recall = 1.0 * TP / (TP + FN);
    const double fmeasure = (2 * precision * recall) / (precision + recall);
    std::printf("triple classification accuracy is %lf\n", accuracy);
    std::printf("triple classification precision is %lf\n", precision);
    std::printf("triple classification recall is %lf\n", recall);
    std::printf("triple classification f-measure is %lf\n", fmeasure);
    if (acc_addr) acc_addr[0] = (float)(1.0 * (TP + TN) / (TP + TN + FP + FN));
}

INT get_n_interval(INT r, REAL *score_pos, REAL *score_neg) {   // Test.h:390-407
    if (!ensure_classification_lists()) return 0;
    float mn, mx; INT n;
    return score_range(r, score_pos, score_neg, mn, mx, n) ? n : 0;
}

INT *get_TPFP(INT r, REAL *score_pos, REAL *score_neg, REAL *score_pos_test, REAL *score_neg_test) {   // Test.h:410-444
    static std::vector<INT> buf;   // the reference leaks a new INT[] per call
    if (!ensure_classification_lists()) return nullptr;
    float mn, mx; INT n_interval;
    if (!score_range(r, score_pos, score_neg, mn, mx, n_interval)) return nullptr;
    buf.assign((size_t)(n_interval + 1) * 2, 0);
    for (INT i = 0; i <= n_interval; i++) {
        const float tmpThresh = grid_point(mn, i);
        INT TP = 0, FP = 0;
        if (g_test_lef[(size_t)r] != -1)
            for (int j = g_test_lef[(size_t)r]; j <= g_test_rig[(size_t)r]; j++) {
                if (score_pos_test[j] <= tmpThresh) TP++;
                if (score_neg_test[j] <= tmpThresh) FP++;
            }
        buf[(size_t)i] = TP;
        buf[(size_t)(i + n_interval + 1)] = FP;
    }
    return buf.data();
}

/* Device-native evaluation of test triples [first, first+count): for each, the model's predict op over
 * ALL entities as tail (and, if test_head != 0, as head) candidates, then the ranker.  out receives
 * count x 2 x 8 int64: [i][0] = testTail result, [i][1] = testHead result (zeros when test_head == 0),
 * each as the reference's 8-vector (distribute_training.py:465-590 consumes exactly these). */
static int link_prediction_v1(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], INT first, INT count,
                              INT test_head, int64_t *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = ensure_eval_device();
    if (rc) return rc;
    if (!m || !tables || !out || first < 0 || count < 0 || first + count > g_eh.test_total) return fail(KGE_ERR_BAD_ARG, "kge_link_prediction: bad range");
    const int64_t E = m->ent_total;
    const int sides = test_head ? 2 : 1;
    const int max_req = m->model == KGE_TRANSR ? 1 : 32;   // TransR's predict uses ONE matrix per call (TransR.py:83)
    if (g_ed.scores_cap < (int64_t)max_req * E) {
        if (g_ed.scores) (void)hipFree(g_ed.scores);
        if (g_ed.cand) (void)hipFree(g_ed.cand);
        g_ed.scores = nullptr; g_ed.cand = nullptr;
        if ((rc = hip_check(hipMalloc(&g_ed.scores, sizeof(float) * (size_t)max_req * E), "alloc scores"))) return rc;
        if ((rc = hip_check(hipMalloc(&g_ed.cand, sizeof(int32_t) * 3 * (size_t)max_req * E), "alloc candidates"))) return rc;
        g_ed.scores_cap = (int64_t)max_req * E;
    } else if (!g_ed.cand) {
        if ((rc = hip_check(hipMalloc(&g_ed.cand, sizeof(int32_t) * 3 * (size_t)g_ed.scores_cap), "alloc candidates"))) return rc;
    }
    static int32_t *d_req = nullptr;
    if (!d_req && (rc = hip_check(hipMalloc(&d_req, sizeof(int32_t) * 2 * 64), "alloc req"))) return rc;
    std::memset(out, 0, sizeof(int64_t) * 16 * (size_t)count);
    std::vector<int32_t> idx, hd;
    std::vector<long long> res;
    for (INT base = 0; base < count * sides; base += max_req) {
        idx.clear(); hd.clear();
        for (INT q = base; q < base + max_req && q < count * sides; q++) { idx.push_back((int32_t)(first + q / sides)); hd.push_back((int32_t)(q % sides)); }
        const int n = (int)idx.size();
        if ((rc = hip_check(hipMemcpyAsync(d_req, idx.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, stream), "upload req"))) return rc;
        if ((rc = hip_check(hipMemcpyAsync(d_req + 64, hd.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, stream), "upload req"))) return rc;
        int32_t *ch = g_ed.cand, *ct = g_ed.cand + (size_t)max_req * E, *cr = g_ed.cand + 2 * (size_t)max_req * E;
        long long tot = (long long)n * E;
        int blocks = (int)std::min<long long>((tot + 255) / 256, 4096);
        hipLaunchKernelGGL(candidates_kernel, dim3(blocks), dim3(256), 0, stream, g_ed.test, d_req, d_req + 64, n, (int)E, ch, ct, cr);
        if ((rc = launch_predict(*m, tables, ch, ct, cr, tot, g_ed.scores, stream))) return rc;
        res.assign((size_t)n * 8, 0);
        if ((rc = rank_requests(g_ed.scores, idx, hd, res.data(), stream, d_req, d_req + 64))) return rc;
        for (int q = 0; q < n; q++) {
            const INT i = idx[q] - first;
            std::memcpy(out + (i * 2 + hd[q]) * 8, res.data() + (size_t)q * 8, sizeof(int64_t) * 8);
        }
    }
    return KGE_OK;
}


// Relation-grouped ranker (models.hip lp_table_kernel / lp_score_kernel): importTestFiles keeps the test triples sorted
// by (r, h, t) (Triple.h:30-32), so the requests of a range are already grouped by relation.  Per relation: one table
// of all candidates' projected + normalised vectors, one scoring pass per chunk of requests, the rank kernel; the 8-vectors
// of all requests come back in ONE copy at the end.
int kge_link_prediction(const kge_model_desc *m, const float *const tables[KGE_MAX_TABLES], INT first, INT count,
                        INT test_head, int64_t *out, void *stream_) {
    if (engine().lp_v1) return link_prediction_v1(m, tables, first, count, test_head, out, stream_);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = ensure_eval_device();
    if (rc) return rc;
    if (!m || !tables || !out || first < 0 || count < 0 || first + count > g_eh.test_total) return fail(KGE_ERR_BAD_ARG, "kge_link_prediction: bad range");
    std::memset(out, 0, sizeof(int64_t) * 16 * (size_t)count);
    if (count == 0) return KGE_OK;
    const int64_t E = m->ent_total;
    const int sides = test_head ? 2 : 1;
    const int64_t n_req = count * sides;
    const int D = m->model == KGE_TRANSR ? m->rel_dim : m->ent_dim;
    // requests in the order (test triple, tail then head)
    std::vector<int32_t> idx((size_t)n_req), hd((size_t)n_req), fixed((size_t)n_req), rel((size_t)n_req);
    for (int64_t q = 0; q < n_req; q++) {
        const int64_t i = first + q / sides;
        const Int4 &tt = g_eh.test[(size_t)i];   // (h, t, r)
        idx[(size_t)q] = (int32_t)i;
        hd[(size_t)q] = (int32_t)(q % sides);
        fixed[(size_t)q] = hd[(size_t)q] ? tt.y : tt.x;   // head request: the tail stays; tail request: the head stays
        rel[(size_t)q] = tt.z;
    }
    static int32_t *d_req = nullptr;     // [3][cap]: test index, head flag, fixed entity
    static int64_t req_cap = 0;
    static float *d_T = nullptr, *d_P = nullptr;
    static int64_t t_cap = 0, p_cap = 0;
    if (n_req > req_cap) {
        if (d_req) (void)hipFree(d_req);
        d_req = nullptr;
        if ((rc = hip_check(hipMalloc(&d_req, sizeof(int32_t) * 3 * (size_t)n_req), "alloc lp requests"))) return rc;
        req_cap = n_req;
    }
    if (E * D > t_cap) {
        if (d_T) (void)hipFree(d_T);
        d_T = nullptr;
        if ((rc = hip_check(hipMalloc(&d_T, sizeof(float) * (size_t)(E * D)), "alloc lp table"))) return rc;
        t_cap = E * D;
    }
    if (m->model == KGE_TRANSR && (E + 1) * D > p_cap) {
        if (d_P) (void)hipFree(d_P);
        d_P = nullptr;
        if ((rc = hip_check(hipMalloc(&d_P, sizeof(float) * (size_t)((E + 1) * D)), "alloc lp projections"))) return rc;
        p_cap = (E + 1) * D;
    }
    // scores for a chunk of requests: at most 1 GiB
    int64_t qmax = (int64_t(1) << 28) / (E > 0 ? E : 1);
    if (qmax < 1) qmax = 1;
    if (qmax > 4096) qmax = 4096;
    if (qmax > n_req) qmax = n_req;
    if (g_ed.scores_cap < qmax * E) {
        if (g_ed.scores) (void)hipFree(g_ed.scores);
        if (g_ed.cand) (void)hipFree(g_ed.cand);
        g_ed.scores = nullptr; g_ed.cand = nullptr;
        if ((rc = hip_check(hipMalloc(&g_ed.scores, sizeof(float) * (size_t)(qmax * E)), "alloc scores"))) return rc;
        g_ed.scores_cap = qmax * E;
    }
    if (g_ed.out_cap < n_req) {
        if (g_ed.out) (void)hipFree(g_ed.out);
        g_ed.out = nullptr;
        if ((rc = hip_check(hipMalloc(&g_ed.out, sizeof(long long) * 8 * (size_t)n_req), "alloc rank out"))) return rc;
        g_ed.out_cap = (int)n_req;
    }
    int32_t *d_idx = d_req, *d_hd = d_req + req_cap, *d_fixed = d_req + 2 * req_cap;
    if ((rc = hip_check(hipMemcpyAsync(d_idx, idx.data(), sizeof(int32_t) * (size_t)n_req, hipMemcpyHostToDevice, stream), "upload req"))) return rc;
    if ((rc = hip_check(hipMemcpyAsync(d_hd, hd.data(), sizeof(int32_t) * (size_t)n_req, hipMemcpyHostToDevice, stream), "upload req"))) return rc;
    if ((rc = hip_check(hipMemcpyAsync(d_fixed, fixed.data(), sizeof(int32_t) * (size_t)n_req, hipMemcpyHostToDevice, stream), "upload req"))) return rc;
    RankArgs a;
    a.test = g_ed.test; a.all = g_ed.all; a.all_t = g_ed.all_t; a.n_all = (long long)g_eh.all.size();
    a.head_lef = g_ed.head_lef; a.head_rig = g_ed.head_rig; a.tail_lef = g_ed.tail_lef; a.tail_rig = g_ed.tail_rig;
    a.head_type = g_ed.head_type; a.tail_type = g_ed.tail_type;
    a.sup_lef = g_ed.sup_lef; a.sup_rig = g_ed.sup_rig; a.sub_lef = g_ed.sub_lef; a.sub_rig = g_ed.sub_rig;
    a.sup_type = g_ed.sup_type; a.sub_type = g_ed.sub_type; a.E = (int)E;
    bool have_table = false;
    for (int64_t q0 = 0; q0 < n_req;) {
        const int32_t r = rel[(size_t)q0];
        int64_t q1 = q0;
        while (q1 < n_req && rel[(size_t)q1] == r) q1++;
        if (m->model != KGE_TRANSE || !have_table) {   // TransE's candidates do not depend on the relation
            if (m->model == KGE_TRANSR && (rc = transr_project_all(*m, tables, r, d_P, stream))) return rc;
            if ((rc = launch_lp_table(*m, tables, d_P, r, d_T, stream))) return rc;
            have_table = true;
        }
        for (int64_t c0 = q0; c0 < q1; c0 += qmax) {
            const int64_t n = std::min<int64_t>(qmax, q1 - c0);
            if ((rc = launch_lp_scores(*m, tables, d_T, r, d_fixed + c0, d_hd + c0, n, g_ed.scores, stream))) return rc;
            a.scores = g_ed.scores; a.req_index = d_idx + c0; a.req_head = d_hd + c0; a.out = g_ed.out + c0 * 8;
            hipLaunchKernelGGL(rank_kernel, dim3((unsigned)n), dim3(256), 0, stream, a);
        }
        q0 = q1;
    }
    std::vector<long long> res((size_t)n_req * 8);
    if ((rc = hip_check(hipMemcpyAsync(res.data(), g_ed.out, sizeof(long long) * 8 * (size_t)n_req, hipMemcpyDeviceToHost, stream), "copy ranks"))) return rc;
    if ((rc = hip_check(hipStreamSynchronize(stream), "rank sync"))) return rc;
    for (int64_t q = 0; q < n_req; q++) {
        const int64_t i = q / sides;
        std::memcpy(out + (i * 2 + hd[(size_t)q]) * 8, res.data() + (size_t)q * 8, sizeof(int64_t) * 8);
    }
    return KGE_OK;
}

}  // extern "C"
