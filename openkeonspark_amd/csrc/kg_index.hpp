// Host-side knowledge-graph loader and filter-index builder.
//
// Produces the arrays the device sampler reads.  The reference keeps five copies of the training
// set as 24-byte {h,r,t} structs plus six per-entity range arrays and runs three binary searches
// per corruption (base/Reader.h:82-158, base/Corrupt.h:7-37).  Here the two searches that only
// locate the (anchor, relation) group are done ONCE at load time: every file-order triple carries
// the [offset,length] of its (h,r), (t,r) and (h,t) groups inside three flat int32 value arrays, so
// the device does one 16-byte record load and one monotone search per corruption.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace kge {

struct Int4 { int32_t x, y, z, w; };
struct Int2 { int32_t x, y; };

struct KgIndex {
    int64_t ent_total = 0, rel_total = 0;
    int64_t train_dup = 0;   // trainTotal_  (Reader.h:77)
    int64_t train_uniq = 0;  // trainTotal after dedup (Reader.h:106-123)
    int64_t new_batch = 0;   // newBatchTotal (Reader.h:61-67)
    std::vector<Int4> pos;       // [train_dup]  (h,t,r,0) in file order      == trainList_no
    std::vector<Int4> grp;       // [train_dup]  (hr_off,hr_len,tr_off,tr_len)
    std::vector<Int2> ht;        // [train_dup]  (ht_off,ht_len)
    std::vector<int32_t> tails_hr;  // [train_uniq] == trainHead[].t
    std::vector<int32_t> heads_tr;  // [train_uniq] == trainTail[].h
    std::vector<int32_t> rels_ht;   // [train_uniq] == trainRel[].r
    std::vector<float> left_mean, right_mean, bern_prob;  // [rel_total]
    bool loaded = false;
};

// Reader.h:27-100 text parse (first line of relation2id/entity2id/batch2id, N + N lines "h t r").
// Returns an empty string on success, else the reference's "`path` does not exist" style message.
std::string load_openke_dir(const std::string &dir, int64_t &ent_total, int64_t &rel_total, int64_t &new_batch,
                            std::vector<int64_t> &h, std::vector<int64_t> &t, std::vector<int64_t> &r);

// every whitespace-separated integer of a text file (fscanf("%ld") semantics); false if it cannot be opened
bool read_all_longs(const std::string &path, std::vector<int64_t> &out);

// Reader.h:102-177: dedup, the three sort orders, group ranges, tails-per-head / heads-per-tail.
std::string build_index(KgIndex &ix, int64_t ent_total, int64_t rel_total, int64_t new_batch, int64_t n,
                        const int64_t *h, const int64_t *t, const int64_t *r);

// Reader.h:160-177 + Base.cpp:117 from the three per-relation integer counts (shared by the host and device builds)
void relation_means(KgIndex &ix, const std::vector<int64_t> &freq_rel, const std::vector<int64_t> &groups_hr,
                    const std::vector<int64_t> &groups_tr);

// glibc rand() with the default seed, continuing across calls (Random.h:9-13 never calls srand).
class LibcRand {
   public:
    LibcRand();
    int32_t next();

   private:
    uint32_t s_[31];
    int f_, r_;
};

// 64-bit LCG of Random.h:16-19 and its jump-ahead table: after 2^j steps x -> mulA[j]*x + addC[j].
constexpr uint64_t kLcgMul = 25214903917ULL;
constexpr uint64_t kLcgAdd = 11ULL;
struct LcgJumpTable { uint64_t mulA[64]; uint64_t addC[64]; };
LcgJumpTable make_jump_table();
uint64_t lcg_jump(const LcgJumpTable &tab, uint64_t state, uint64_t steps);

// Base.cpp:85-92: half-open slice [lef,rig) of batch positions owned by virtual thread `id`.
inline void thread_slice(int64_t batch, int64_t threads, int64_t id, int64_t &lef, int64_t &rig) {
    if (batch % threads == 0) {
        lef = id * (batch / threads);
        rig = (id + 1) * (batch / threads);
    } else {
        lef = id * (batch / threads + 1);
        rig = (id + 1) * (batch / threads + 1);
        if (rig > batch) rig = batch;
        if (lef > batch) lef = batch;
    }
}

}  // namespace kge
