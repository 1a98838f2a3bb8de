// Host-side loader / index builder (see kg_index.hpp).  Pure C++17, no device code.
#include "kg_index.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace kge {

namespace {

// Reads a whole text file; returns false if it cannot be opened.
bool slurp(const std::string &path, std::vector<char> &buf) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(static_cast<size_t>(n) + 1);
    size_t got = n > 0 ? std::fread(buf.data(), 1, static_cast<size_t>(n), f) : 0;
    buf[got] = 0;
    buf.resize(got + 1);
    std::fclose(f);
    return true;
}

// fscanf("%ld") semantics: skip whitespace, parse one signed integer; false at end / on junk.
bool next_long(const char *&p, int64_t &out) {
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') ++p;
    if (!*p) return false;
    char *end = nullptr;
    long v = std::strtol(p, &end, 10);
    if (end == p) return false;
    p = end;
    out = v;
    return true;
}

bool first_long(const std::string &path, int64_t &out, bool &exists) {
    std::vector<char> buf;
    exists = slurp(path, buf);
    if (!exists) return false;
    const char *p = buf.data();
    return next_long(p, out);
}

struct U3 { int32_t h, r, t; };

}  // namespace

bool read_all_longs(const std::string &path, std::vector<int64_t> &out) {
    std::vector<char> buf;
    out.clear();
    if (!slurp(path, buf)) return false;
    const char *p = buf.data();
    int64_t v;
    while (next_long(p, v)) out.push_back(v);
    return true;
}

std::string load_openke_dir(const std::string &dir, int64_t &ent_total, int64_t &rel_total, int64_t &new_batch,
                            std::vector<int64_t> &h, std::vector<int64_t> &t, std::vector<int64_t> &r) {
    bool exists = false;
    rel_total = 0; ent_total = 0; new_batch = 0;
    first_long(dir + "relation2id.txt", rel_total, exists);  // Reader.h:35-42
    if (!exists) return "`" + dir + "relation2id.txt` does not exist";
    first_long(dir + "entity2id.txt", ent_total, exists);  // Reader.h:46-54
    if (!exists) return "`" + dir + "entity2id.txt` does not exist";
    int64_t nb = 0;
    first_long(dir + "batch2id.txt", nb, exists);  // Reader.h:61-67 (optional file)
    if (exists) new_batch = nb;
    std::vector<char> buf;
    if (!slurp(dir + "train2id.txt", buf)) return "`" + dir + "train2id.txt` does not exist";  // Reader.h:71-75
    const char *p = buf.data();
    int64_t n = 0;
    if (!next_long(p, n) || n < 0) n = 0;
    h.assign(static_cast<size_t>(n), 0);
    t.assign(static_cast<size_t>(n), 0);
    r.assign(static_cast<size_t>(n), 0);
    for (int64_t i = 0; i < n; i++) {  // Reader.h:91-95: order on disk is head, tail, relation
        if (!next_long(p, h[i])) break;
        if (!next_long(p, t[i])) break;
        if (!next_long(p, r[i])) break;
    }
    return "";
}

std::string build_index(KgIndex &ix, int64_t E, int64_t R, int64_t new_batch, int64_t n, const int64_t *h,
                        const int64_t *t, const int64_t *r) {
    ix = KgIndex();
    if (E < 0 || R < 0 || n < 0) return "negative totals";
    if (E >= (int64_t(1) << 31) || R >= (int64_t(1) << 31) || n >= (int64_t(1) << 31))
        return "entity / relation / triple counts must fit int32 for the device index";
    for (int64_t i = 0; i < n; i++)
        if (h[i] < 0 || h[i] >= E || t[i] < 0 || t[i] >= E || r[i] < 0 || r[i] >= R)
            return "train2id.txt: id out of range at line " + std::to_string(i + 2);
    if (new_batch < 0 || new_batch > n) return "batch2id.txt: newBatchTotal out of range";
    ix.ent_total = E; ix.rel_total = R; ix.train_dup = n; ix.new_batch = new_batch;

    // argsort of the file order by (h,r,t)  (Reader.h:103), then dedup (Reader.h:106-123)
    std::vector<int32_t> order(static_cast<size_t>(n));
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (h[a] != h[b]) return h[a] < h[b];
        if (r[a] != r[b]) return r[a] < r[b];
        return t[a] < t[b];
    });
    std::vector<U3> uniq;
    uniq.reserve(static_cast<size_t>(n));
    std::vector<int32_t> file2uniq(static_cast<size_t>(n));
    for (int64_t k = 0; k < n; k++) {
        int32_t i = order[k];
        if (k == 0 || h[i] != uniq.back().h || r[i] != uniq.back().r || t[i] != uniq.back().t)
            uniq.push_back(U3{int32_t(h[i]), int32_t(r[i]), int32_t(t[i])});
        file2uniq[i] = int32_t(uniq.size() - 1);
    }
    const int64_t U = int64_t(uniq.size());
    ix.train_uniq = U;

    std::vector<int64_t> freq_rel(static_cast<size_t>(R), 0), groups_hr(static_cast<size_t>(R), 0),
        groups_tr(static_cast<size_t>(R), 0);
    std::vector<int32_t> hr_off(U), hr_len(U), tr_off(U), tr_len(U), ht_off(U), ht_len(U);

    // (h,r,t) order: tails grouped by (h,r)  == trainHead
    ix.tails_hr.resize(U);
    for (int64_t u = 0, start = 0; u < U; u++) {
        ix.tails_hr[u] = uniq[u].t;
        freq_rel[uniq[u].r]++;
        bool last = (u + 1 == U) || uniq[u + 1].h != uniq[u].h || uniq[u + 1].r != uniq[u].r;
        if (last) {
            groups_hr[uniq[u].r]++;
            for (int64_t q = start; q <= u; q++) { hr_off[q] = int32_t(start); hr_len[q] = int32_t(u - start + 1); }
            start = u + 1;
        }
    }
    // (t,r,h) order: heads grouped by (t,r)  == trainTail (Reader.h:126)
    std::vector<int32_t> perm(U);
    std::iota(perm.begin(), perm.end(), 0);
    std::sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
        if (uniq[a].t != uniq[b].t) return uniq[a].t < uniq[b].t;
        if (uniq[a].r != uniq[b].r) return uniq[a].r < uniq[b].r;
        return uniq[a].h < uniq[b].h;
    });
    ix.heads_tr.resize(U);
    for (int64_t j = 0, start = 0; j < U; j++) {
        const U3 &c = uniq[perm[j]];
        ix.heads_tr[j] = c.h;
        bool last = (j + 1 == U) || uniq[perm[j + 1]].t != c.t || uniq[perm[j + 1]].r != c.r;
        if (last) {
            groups_tr[c.r]++;
            for (int64_t q = start; q <= j; q++) { tr_off[perm[q]] = int32_t(start); tr_len[perm[q]] = int32_t(j - start + 1); }
            start = j + 1;
        }
    }
    // (h,t,r) order: relations grouped by (h,t)  == trainRel (Reader.h:127)
    std::iota(perm.begin(), perm.end(), 0);
    std::sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
        if (uniq[a].h != uniq[b].h) return uniq[a].h < uniq[b].h;
        if (uniq[a].t != uniq[b].t) return uniq[a].t < uniq[b].t;
        return uniq[a].r < uniq[b].r;
    });
    ix.rels_ht.resize(U);
    for (int64_t j = 0, start = 0; j < U; j++) {
        const U3 &c = uniq[perm[j]];
        ix.rels_ht[j] = c.r;
        bool last = (j + 1 == U) || uniq[perm[j + 1]].h != c.h || uniq[perm[j + 1]].t != c.t;
        if (last) {
            for (int64_t q = start; q <= j; q++) { ht_off[perm[q]] = int32_t(start); ht_len[perm[q]] = int32_t(j - start + 1); }
            start = j + 1;
        }
    }
    // per file-order triple records
    ix.pos.resize(n); ix.grp.resize(n); ix.ht.resize(n);
    for (int64_t i = 0; i < n; i++) {
        int32_t u = file2uniq[i];
        ix.pos[i] = Int4{int32_t(h[i]), int32_t(t[i]), int32_t(r[i]), 0};
        ix.grp[i] = Int4{hr_off[u], hr_len[u], tr_off[u], tr_len[u]};
        ix.ht[i] = Int2{ht_off[u], ht_len[u]};
    }
    relation_means(ix, freq_rel, groups_hr, groups_tr);
    ix.loaded = true;
    return "";
}

void relation_means(KgIndex &ix, const std::vector<int64_t> &freq_rel, const std::vector<int64_t> &groups_hr,
                    const std::vector<int64_t> &groups_tr) {
    const int64_t R = ix.rel_total;
    // Reader.h:160-177.  The reference counts groups by adding 1.0f to a float, which stops
    // growing at 2^24, and divides a long by that float.  Reproduce both effects.
    ix.left_mean.resize(R); ix.right_mean.resize(R); ix.bern_prob.resize(R);
    for (int64_t q = 0; q < R; q++) {
        float gl = float(std::min<int64_t>(groups_hr[q], int64_t(1) << 24));
        float gr = float(std::min<int64_t>(groups_tr[q], int64_t(1) << 24));
        volatile float lm = float(freq_rel[q]) / gl;
        volatile float rm = float(freq_rel[q]) / gr;
        ix.left_mean[q] = lm;
        ix.right_mean[q] = rm;
        // Base.cpp:117, evaluated in float exactly as written there: (1000*right)/(right+left)
        volatile float num = 1000.0f * rm;
        volatile float den = rm + lm;
        ix.bern_prob[q] = num / den;
    }
}

LibcRand::LibcRand() {
    int32_t w = 1;
    s_[0] = uint32_t(w);
    for (int i = 1; i < 31; i++) {
        int32_t hi = w / 127773, lo = w % 127773;
        w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        s_[i] = uint32_t(w);
    }
    f_ = 3; r_ = 0;
    for (int k = 0; k < 310; k++) next();
}

int32_t LibcRand::next() {
    s_[f_] += s_[r_];
    uint32_t v = s_[f_];
    f_ = (f_ + 1) % 31;
    r_ = (r_ + 1) % 31;
    return int32_t(v >> 1);
}

LcgJumpTable make_jump_table() {
    LcgJumpTable tab;
    uint64_t a = kLcgMul, c = kLcgAdd;
    for (int j = 0; j < 64; j++) {
        tab.mulA[j] = a;
        tab.addC[j] = c;
        c = a * c + c;  // composing x -> a x + c with itself
        a = a * a;
    }
    return tab;
}

uint64_t lcg_jump(const LcgJumpTable &tab, uint64_t state, uint64_t steps) {
    for (int j = 0; steps; j++, steps >>= 1)
        if (steps & 1) state = tab.mulA[j] * state + tab.addC[j];
    return state;
}

}  // namespace kge
