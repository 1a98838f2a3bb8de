// Filter-index build on the device: what kg_index.cpp's build_index does with four
// std::sort calls on the host (Reader.h:102-177: dedup, the (h,r,t) / (t,r,h) / (h,t,r) orders, the group
// ranges, tails-per-head / heads-per-tail), done with rocPRIM radix sorts and scans over packed 64-bit keys.
// At 500 M triples the host build is minutes of std::sort; here it is three 64-bit key sorts of n elements.
//
// Every array it produces is bit-identical to the host build (tests/test_gpu_index.py compares them all),
// which in turn is pinned to the compiled reference through the sampler fixtures.
//
//   triple (a,b,c) -> key = a << (bits_b + bits_c) | b << bits_c | c     (needs 2*bits(E) + bits(R) <= 64)
//   order  (h,r,t): sort file-order triples, flag first-of-equal-key, scan -> unique ids (dedup, Reader.h:106-123)
//   groups (h,r):   flag on key >> bits(E), scan, head positions -> [offset,length] of every triple's group
//   order  (t,r,h) and (h,t,r): same over the unique triples, results scattered back through the sort permutation
#include "engine.hpp"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace kge {

namespace {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t count, const char *what) {
        if (p) { (void)hipFree(p); p = nullptr; }
        return hip_check(hipMalloc(&p, sizeof(T) * (count ? count : 1)), what);
    }
    T *release() { T *q = p; p = nullptr; return q; }
};

int bits_for(int64_t count) {  // bits needed for values in [0, count)
    int b = 1;
    while ((int64_t(1) << b) < count) b++;
    return b;
}

constexpr int TPB = 256;
inline unsigned grid_for(int64_t n) {
    int64_t b = (n + TPB - 1) / TPB;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
}
#define KGE_GRID_LOOP(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// int64 ids -> int32, with the host build's range check (first offending line wins)
__global__ void narrow_kernel(const int64_t *__restrict__ h, const int64_t *__restrict__ t, const int64_t *__restrict__ r, long long n,
                              long long E, long long R, int32_t *__restrict__ h32, int32_t *__restrict__ t32, int32_t *__restrict__ r32,
                              unsigned long long *first_bad) {
    KGE_GRID_LOOP(i, n) {
        const long long a = h[i], b = t[i], c = r[i];
        if (a < 0 || a >= E || b < 0 || b >= E || c < 0 || c >= R) atomicMin(first_bad, (unsigned long long)i);
        h32[i] = (int32_t)a; t32[i] = (int32_t)b; r32[i] = (int32_t)c;
    }
}

__global__ void pack_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, const int32_t *__restrict__ c, long long n,
                            int bits_b, int bits_c, uint64_t *__restrict__ keys, int32_t *__restrict__ idx) {
    KGE_GRID_LOOP(i, n) {
        keys[i] = ((uint64_t)(uint32_t)a[i] << (bits_b + bits_c)) | ((uint64_t)(uint32_t)b[i] << bits_c) | (uint64_t)(uint32_t)c[i];
        idx[i] = (int32_t)i;
    }
}

// flag[k] = 1 where the group key (key >> shift) changes
__global__ void flag_kernel(const uint64_t *__restrict__ keys, long long n, int shift, int32_t *__restrict__ flag) {
    KGE_GRID_LOOP(k, n) flag[k] = (k == 0 || (keys[k] >> shift) != (keys[k - 1] >> shift)) ? 1 : 0;
}

// gid = inclusive scan of the flags (1-based group id).  head_pos[g] = first position of group g; head_pos[G] = n
__global__ void heads_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ gid, long long n, int shift,
                             int32_t *__restrict__ head_pos) {
    KGE_GRID_LOOP(k, n) {
        if (k == 0 || (keys[k] >> shift) != (keys[k - 1] >> shift)) head_pos[gid[k] - 1] = (int32_t)k;
        if (k == n - 1) head_pos[gid[k]] = (int32_t)n;
    }
}

// dedup of the (h,r,t)-sorted file-order triples: unique keys, and the unique id of every file-order line
__global__ void dedup_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ uid, const int32_t *__restrict__ order,
                             long long n, uint64_t *__restrict__ ukeys, int32_t *__restrict__ file2uniq) {
    KGE_GRID_LOOP(k, n) {
        const int32_t u = uid[k] - 1;
        file2uniq[order[k]] = u;
        if (k == 0 || keys[k] != keys[k - 1]) ukeys[u] = keys[k];
    }
}

__global__ void unpack_hrt_kernel(const uint64_t *__restrict__ ukeys, long long U, int bits_e, int bits_r, int32_t *__restrict__ uh,
                                  int32_t *__restrict__ ur, int32_t *__restrict__ ut) {
    const uint64_t me = (uint64_t(1) << bits_e) - 1, mr = (uint64_t(1) << bits_r) - 1;
    KGE_GRID_LOOP(u, U) {
        const uint64_t k = ukeys[u];
        ut[u] = (int32_t)(k & me);
        ur[u] = (int32_t)((k >> bits_e) & mr);
        uh[u] = (int32_t)(k >> (bits_e + bits_r));
    }
}

// per sorted position j (unique triple perm[j], or j itself when perm == nullptr): its group's [offset,length]
// written at the TRIPLE's slot; value[j] = low field of the key; optional per-relation counters
__global__ void groups_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ gid, const int32_t *__restrict__ head_pos,
                              const int32_t *__restrict__ perm, long long n, int shift, uint64_t value_mask, int rel_shift,
                              uint64_t rel_mask, int32_t *__restrict__ value, int32_t *__restrict__ off, int32_t *__restrict__ len,
                              unsigned long long *freq_rel, unsigned long long *groups_rel) {
    KGE_GRID_LOOP(j, n) {
        const uint64_t k = keys[j];
        const int32_t g = gid[j] - 1;
        const int32_t start = head_pos[g];
        const long long slot = perm ? perm[j] : j;
        off[slot] = start;
        len[slot] = head_pos[g + 1] - start;
        value[j] = (int32_t)(k & value_mask);
        const unsigned rel = (unsigned)((k >> rel_shift) & rel_mask);
        if (freq_rel) atomicAdd(freq_rel + rel, 1ULL);
        if (groups_rel && (j == 0 || (k >> shift) != (keys[j - 1] >> shift))) atomicAdd(groups_rel + rel, 1ULL);
    }
}

__global__ void assemble_kernel(const int32_t *__restrict__ h, const int32_t *__restrict__ t, const int32_t *__restrict__ r,
                                const int32_t *__restrict__ file2uniq, const int32_t *__restrict__ hr_off, const int32_t *__restrict__ hr_len,
                                const int32_t *__restrict__ tr_off, const int32_t *__restrict__ tr_len, const int32_t *__restrict__ ht_off,
                                const int32_t *__restrict__ ht_len, long long n, int4 *__restrict__ pos, int4 *__restrict__ grp,
                                int2 *__restrict__ ht) {
    KGE_GRID_LOOP(i, n) {
        const int32_t u = file2uniq[i];
        pos[i] = make_int4(h[i], t[i], r[i], 0);
        grp[i] = make_int4(hr_off[u], hr_len[u], tr_off[u], tr_len[u]);
        ht[i] = make_int2(ht_off[u], ht_len[u]);
    }
}

struct SortScratch {
    DevBuf<char> tmp;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return KGE_OK;
        int rc = tmp.alloc(need, "index build: sort scratch");
        if (!rc) bytes = need;
        return rc;
    }
};

int sort_pairs(SortScratch &sc, const uint64_t *kin, uint64_t *kout, const int32_t *vin, int32_t *vout, size_t n, int bits) {
    size_t need = 0;
    int rc = hip_check(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, n, 0, (unsigned)bits, nullptr), "index sort size");
    if (rc) return rc;
    if ((rc = sc.ensure(need))) return rc;
    return hip_check(rocprim::radix_sort_pairs(sc.tmp.p, need, kin, kout, vin, vout, n, 0, (unsigned)bits, nullptr), "index sort");
}

int scan_flags(SortScratch &sc, int32_t *flags, size_t n) {
    size_t need = 0;
    int rc = hip_check(rocprim::inclusive_scan(nullptr, need, flags, flags, n, rocprim::plus<int32_t>(), nullptr), "index scan size");
    if (rc) return rc;
    if ((rc = sc.ensure(need))) return rc;
    return hip_check(rocprim::inclusive_scan(sc.tmp.p, need, flags, flags, n, rocprim::plus<int32_t>(), nullptr), "index scan");
}

template <typename T>
int download(std::vector<T> &dst, const void *src, size_t count, const char *what) {
    dst.resize(count);
    if (!count) return KGE_OK;
    return hip_check(hipMemcpy(dst.data(), src, sizeof(T) * count, hipMemcpyDeviceToHost), what);
}

}  // namespace

bool device_index_build_supported(int64_t E, int64_t R, int64_t n) {
    return n > 0 && E > 0 && R > 0 && n < (int64_t(1) << 31) && E < (int64_t(1) << 31) && R < (int64_t(1) << 31) &&
           2 * bits_for(E) + bits_for(R) <= 64;
}

// Builds DeviceIndex arrays in place (dev.uploaded = true) and the host mirror `ix` that the other consumers read.
std::string build_index_device(KgIndex &ix, DeviceIndex &dev, int64_t E, int64_t R, int64_t new_batch, int64_t n, const int64_t *h,
                               const int64_t *t, const int64_t *r) {
    ix = KgIndex();
    if (!device_index_build_supported(E, R, n)) return "device index build: sizes not supported";
    if (new_batch < 0 || new_batch > n) return "batch2id.txt: newBatchTotal out of range";
    const int be = bits_for(E), br = bits_for(R);
    const int total_bits = 2 * be + br;
    const size_t N = (size_t)n;
    int rc;
#define KGE_TRY(expr) if ((rc = (expr))) return "device index build failed: " + engine().last_error
    DevBuf<int32_t> h32, t32, r32;
    KGE_TRY(h32.alloc(N, "index h")); KGE_TRY(t32.alloc(N, "index t")); KGE_TRY(r32.alloc(N, "index r"));
    {
        DevBuf<int64_t> h64, t64, r64;
        DevBuf<unsigned long long> bad;
        KGE_TRY(h64.alloc(N, "index h64")); KGE_TRY(t64.alloc(N, "index t64")); KGE_TRY(r64.alloc(N, "index r64"));
        KGE_TRY(bad.alloc(1, "index flag"));
        KGE_TRY(hip_check(hipMemcpy(h64.p, h, sizeof(int64_t) * N, hipMemcpyHostToDevice), "upload h"));
        KGE_TRY(hip_check(hipMemcpy(t64.p, t, sizeof(int64_t) * N, hipMemcpyHostToDevice), "upload t"));
        KGE_TRY(hip_check(hipMemcpy(r64.p, r, sizeof(int64_t) * N, hipMemcpyHostToDevice), "upload r"));
        KGE_TRY(hip_check(hipMemset(bad.p, 0xFF, sizeof(unsigned long long)), "index flag init"));
        hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, h64.p, t64.p, r64.p, (long long)n, (long long)E,
                           (long long)R, h32.p, t32.p, r32.p, bad.p);
        unsigned long long first_bad = 0;
        KGE_TRY(hip_check(hipMemcpy(&first_bad, bad.p, sizeof(first_bad), hipMemcpyDeviceToHost), "index flag read"));
        if (first_bad != ~0ULL) return "train2id.txt: id out of range at line " + std::to_string((long long)first_bad + 2);
    }
    SortScratch sc;
    DevBuf<uint64_t> keys_a, keys_b;
    DevBuf<int32_t> idx_a, idx_b, flags, head_pos;
    KGE_TRY(keys_a.alloc(N, "index keys")); KGE_TRY(keys_b.alloc(N, "index keys"));
    KGE_TRY(idx_a.alloc(N, "index idx")); KGE_TRY(idx_b.alloc(N, "index idx"));
    KGE_TRY(flags.alloc(N, "index flags")); KGE_TRY(head_pos.alloc(N + 1, "index heads"));

    // ---- (h,r,t) order of the file lines, dedup ----------------------------------------------------
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, h32.p, r32.p, t32.p, (long long)n, br, be, keys_a.p, idx_a.p);
    KGE_TRY(sort_pairs(sc, keys_a.p, keys_b.p, idx_a.p, idx_b.p, N, total_bits));   // keys_b sorted, idx_b = order
    hipLaunchKernelGGL(flag_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, keys_b.p, (long long)n, 0, flags.p);
    KGE_TRY(scan_flags(sc, flags.p, N));
    int32_t U32 = 0;
    KGE_TRY(hip_check(hipMemcpy(&U32, flags.p + (N - 1), sizeof(int32_t), hipMemcpyDeviceToHost), "index unique count"));
    const int64_t U = U32;
    const size_t UU = (size_t)U;
    DevBuf<uint64_t> ukeys;
    DevBuf<int32_t> file2uniq, uh, ur, ut;
    KGE_TRY(ukeys.alloc(UU, "index ukeys")); KGE_TRY(file2uniq.alloc(N, "index file2uniq"));
    KGE_TRY(uh.alloc(UU, "index uh")); KGE_TRY(ur.alloc(UU, "index ur")); KGE_TRY(ut.alloc(UU, "index ut"));
    hipLaunchKernelGGL(dedup_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, keys_b.p, flags.p, idx_b.p, (long long)n, ukeys.p, file2uniq.p);
    hipLaunchKernelGGL(unpack_hrt_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, ukeys.p, (long long)U, be, br, uh.p, ur.p, ut.p);

    DevBuf<unsigned long long> counters;   // freq_rel | groups_hr | groups_tr
    KGE_TRY(counters.alloc(3 * (size_t)R, "index relation counters"));
    KGE_TRY(hip_check(hipMemset(counters.p, 0, sizeof(unsigned long long) * 3 * (size_t)R), "index counters init"));
    const uint64_t mask_e = (uint64_t(1) << be) - 1, mask_r = (uint64_t(1) << br) - 1;

    DevBuf<int32_t> tails_hr, heads_tr, rels_ht, hr_off, hr_len, tr_off, tr_len, ht_off, ht_len;
    KGE_TRY(tails_hr.alloc(UU, "index tails")); KGE_TRY(heads_tr.alloc(UU, "index heads")); KGE_TRY(rels_ht.alloc(UU, "index rels"));
    KGE_TRY(hr_off.alloc(UU, "index hr_off")); KGE_TRY(hr_len.alloc(UU, "index hr_len"));
    KGE_TRY(tr_off.alloc(UU, "index tr_off")); KGE_TRY(tr_len.alloc(UU, "index tr_len"));
    KGE_TRY(ht_off.alloc(UU, "index ht_off")); KGE_TRY(ht_len.alloc(UU, "index ht_len"));

    // ---- groups (h,r) over the unique triples (already in (h,r,t) order): tails_hr ----------------
    hipLaunchKernelGGL(flag_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, ukeys.p, (long long)U, be, flags.p);
    KGE_TRY(scan_flags(sc, flags.p, UU));
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, ukeys.p, flags.p, (long long)U, be, head_pos.p);
    hipLaunchKernelGGL(groups_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, ukeys.p, flags.p, head_pos.p, (const int32_t *)nullptr,
                       (long long)U, be, mask_e, be, mask_r, tails_hr.p, hr_off.p, hr_len.p, counters.p, counters.p + R);

    // ---- (t,r,h) order: heads_tr, groups (t,r) -----------------------------------------------------
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, ut.p, ur.p, uh.p, (long long)U, br, be, keys_a.p, idx_a.p);
    KGE_TRY(sort_pairs(sc, keys_a.p, keys_b.p, idx_a.p, idx_b.p, UU, total_bits));
    hipLaunchKernelGGL(flag_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, keys_b.p, (long long)U, be, flags.p);
    KGE_TRY(scan_flags(sc, flags.p, UU));
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, keys_b.p, flags.p, (long long)U, be, head_pos.p);
    hipLaunchKernelGGL(groups_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, keys_b.p, flags.p, head_pos.p, idx_b.p, (long long)U, be,
                       mask_e, be, mask_r, heads_tr.p, tr_off.p, tr_len.p, (unsigned long long *)nullptr, counters.p + 2 * R);

    // ---- (h,t,r) order: rels_ht, groups (h,t) ------------------------------------------------------
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, uh.p, ut.p, ur.p, (long long)U, be, br, keys_a.p, idx_a.p);
    KGE_TRY(sort_pairs(sc, keys_a.p, keys_b.p, idx_a.p, idx_b.p, UU, total_bits));
    hipLaunchKernelGGL(flag_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, keys_b.p, (long long)U, br, flags.p);
    KGE_TRY(scan_flags(sc, flags.p, UU));
    hipLaunchKernelGGL(heads_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, keys_b.p, flags.p, (long long)U, br, head_pos.p);
    hipLaunchKernelGGL(groups_kernel, dim3(grid_for(U)), dim3(TPB), 0, nullptr, keys_b.p, flags.p, head_pos.p, idx_b.p, (long long)U, br,
                       mask_r, 0, mask_r, rels_ht.p, ht_off.p, ht_len.p, (unsigned long long *)nullptr, (unsigned long long *)nullptr);

    // ---- per file-order line records ----------------------------------------------------------------
    DevBuf<int4> pos, grp;
    DevBuf<int2> ht;
    KGE_TRY(pos.alloc(N, "index pos")); KGE_TRY(grp.alloc(N, "index grp")); KGE_TRY(ht.alloc(N, "index ht"));
    hipLaunchKernelGGL(assemble_kernel, dim3(grid_for(n)), dim3(TPB), 0, nullptr, h32.p, t32.p, r32.p, file2uniq.p, hr_off.p, hr_len.p,
                       tr_off.p, tr_len.p, ht_off.p, ht_len.p, (long long)n, pos.p, grp.p, ht.p);
    KGE_TRY(hip_check(hipDeviceSynchronize(), "index build"));
    KGE_TRY(hip_check(hipGetLastError(), "index build launch"));

    // ---- relation statistics: the three integer histograms come back, the float arithmetic is the host's ----
    std::vector<unsigned long long> cnt(3 * (size_t)R);
    KGE_TRY(hip_check(hipMemcpy(cnt.data(), counters.p, sizeof(unsigned long long) * cnt.size(), hipMemcpyDeviceToHost), "index counters"));
    ix.ent_total = E; ix.rel_total = R; ix.train_dup = n; ix.new_batch = new_batch; ix.train_uniq = U;
    std::vector<int64_t> freq_rel((size_t)R), groups_hr((size_t)R), groups_tr((size_t)R);
    for (int64_t q = 0; q < R; q++) {
        freq_rel[(size_t)q] = (int64_t)cnt[(size_t)q];
        groups_hr[(size_t)q] = (int64_t)cnt[(size_t)(R + q)];
        groups_tr[(size_t)q] = (int64_t)cnt[(size_t)(2 * R + q)];
    }
    relation_means(ix, freq_rel, groups_hr, groups_tr);

    // host mirror (kge_index_copy, the evaluation loader)
    KGE_TRY(download(ix.pos, pos.p, N, "mirror pos")); KGE_TRY(download(ix.grp, grp.p, N, "mirror grp"));
    KGE_TRY(download(ix.ht, ht.p, N, "mirror ht"));
    KGE_TRY(download(ix.tails_hr, tails_hr.p, UU, "mirror tails")); KGE_TRY(download(ix.heads_tr, heads_tr.p, UU, "mirror heads"));
    KGE_TRY(download(ix.rels_ht, rels_ht.p, UU, "mirror rels"));

    // adopt the device arrays
    auto drop = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    drop(dev.pos); drop(dev.grp); drop(dev.ht); drop(dev.tails_hr); drop(dev.heads_tr); drop(dev.rels_ht); drop(dev.bern_prob);
    dev.pos = pos.release(); dev.grp = grp.release(); dev.ht = ht.release();
    dev.tails_hr = tails_hr.release(); dev.heads_tr = heads_tr.release(); dev.rels_ht = rels_ht.release();
    KGE_TRY(hip_check(hipMalloc(&dev.bern_prob, sizeof(float) * (size_t)(R ? R : 1)), "index bern"));
    KGE_TRY(hip_check(hipMemcpy(dev.bern_prob, ix.bern_prob.data(), sizeof(float) * (size_t)R, hipMemcpyHostToDevice), "index bern"));
    dev.uploaded = true;
    ix.loaded = true;
#undef KGE_TRY
    return "";
}

}  // namespace kge
