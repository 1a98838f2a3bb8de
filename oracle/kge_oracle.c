/*
 * kge_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the OpenKEonSpark hot path.
 *
 * This file is the *checker* for the HIP engine in openkeonspark_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product never
 * calls into it and has no CPU fallback.
 *
 * Parity status
 *   - sampler part (rng, loader, corruption, batch fill): PINNED.  Checked bit-exactly against
 *     the reference's own C++ compiled from /root/reference/base (oracle/_ref/Base.so) and
 *     against the committed fixtures tests/golden/*.npz generated from that build.
 *   - model part (TransE/H/R/D forward, backward, SGD, Adam): PARITY UNPINNED versus TensorFlow
 *     1.x (un-vendored dependency of the reference, not installable here).  It follows the text of
 *     TransE.py / TransH.py / TransR.py / TransD.py / Model.py / distribute_training.py:95-101 and
 *     the published TF 1.x op semantics, and is cross-checked against fp64 torch.autograd in
 *     tests/test_oracle_models.py.
 *
 * Every function cites the reference file:line it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------
 * glibc rand(), TYPE_3 additive feedback generator with the default seed 1.
 * Random.h:9-13 seeds every sampler stream with rand() and never calls srand(), so the seeds are
 * the first outputs of an unseeded glibc generator: 1804289383, 846930886, 1681692777, ...
 * Restated from the published glibc algorithm (random_r.c) so the oracle does not depend on the
 * process-wide libc state.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t s[31]; /* the 31-word state table */
    int f, r;       /* front / rear cursors, 3 apart */
} OrcLibcRand;

static uint32_t libc_rand_step(OrcLibcRand *g) {
    g->s[g->f] += g->s[g->r];
    uint32_t v = g->s[g->f];
    g->f = (g->f + 1) % 31;
    g->r = (g->r + 1) % 31;
    return v >> 1;
}

void orc_libc_rand_init(OrcLibcRand *g) {
    int32_t w = 1; /* default seed */
    g->s[0] = (uint32_t)w;
    for (int i = 1; i < 31; i++) {
        /* w = 16807 * w mod 2147483647 via Schrage's method */
        int32_t hi = w / 127773, lo = w % 127773;
        w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        g->s[i] = (uint32_t)w;
    }
    g->f = 3; g->r = 0;
    for (int k = 0; k < 310; k++) libc_rand_step(g); /* glibc discards 10*31 outputs */
}

int32_t orc_libc_rand_next(OrcLibcRand *g) { return (int32_t)libc_rand_step(g); }

/* ------------------------------------------------------------------------------------------
 * Knowledge graph + sampler index.  Restates Reader.h:27-179 (importTrainFiles) and the data
 * layout of Triple.h:5-34.  Three sorted orders are kept as separate int64 column arrays.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    i64 h, r, t;
} OrcTriple;

typedef struct OrcKG {
    i64 ent_total, rel_total;
    i64 train_dup;   /* trainTotal_  : file count, duplicates kept  (Reader.h:77) */
    i64 train_uniq;  /* trainTotal   : after dedup                  (Reader.h:106-123) */
    i64 new_batch;   /* newBatchTotal: first line of batch2id.txt   (Reader.h:61-67) */
    OrcTriple *file_order;             /* trainList_no */
    OrcTriple *by_head, *by_tail, *by_rel; /* trainHead (h,r,t) / trainTail (t,r,h) / trainRel (h,t,r) */
    i64 *lef_head, *rig_head, *lef_tail, *rig_tail, *lef_rel, *rig_rel;
    float *left_mean, *right_mean;
    /* rng: one 64-bit LCG stream per virtual worker thread (Random.h:6-19) */
    i64 work_threads;
    u64 *stream;
    i64 bern;
    OrcLibcRand libc;
} OrcKG;

static int cmp_hrt(const void *pa, const void *pb) { /* Triple.h:18-20 */
    const OrcTriple *a = pa, *b = pb;
    if (a->h != b->h) return a->h < b->h ? -1 : 1;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    if (a->t != b->t) return a->t < b->t ? -1 : 1;
    return 0;
}
static int cmp_trh(const void *pa, const void *pb) { /* Triple.h:22-24 */
    const OrcTriple *a = pa, *b = pb;
    if (a->t != b->t) return a->t < b->t ? -1 : 1;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    if (a->h != b->h) return a->h < b->h ? -1 : 1;
    return 0;
}
static int cmp_htr(const void *pa, const void *pb) { /* Triple.h:26-28 */
    const OrcTriple *a = pa, *b = pb;
    if (a->h != b->h) return a->h < b->h ? -1 : 1;
    if (a->t != b->t) return a->t < b->t ? -1 : 1;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    return 0;
}

/* Build every derived structure from the file-order list.  Reader.h:82-177. */
static void orc_kg_build(OrcKG *kg) {
    i64 n = kg->train_dup, E = kg->ent_total, R = kg->rel_total;
    OrcTriple *sorted = malloc(sizeof(OrcTriple) * (size_t)(n > 0 ? n : 1));
    memcpy(sorted, kg->file_order, sizeof(OrcTriple) * (size_t)n);
    qsort(sorted, (size_t)n, sizeof(OrcTriple), cmp_hrt); /* Reader.h:103 */
    i64 *freq_rel = calloc((size_t)(R > 0 ? R : 1), sizeof(i64));
    /* dedup (Reader.h:106-123) */
    i64 u = 0;
    for (i64 i = 0; i < n; i++) {
        if (i == 0 || cmp_hrt(&sorted[i], &sorted[i - 1]) != 0) {
            sorted[u++] = sorted[i];
            freq_rel[sorted[u - 1].r]++;
        }
    }
    kg->train_uniq = u;
    kg->by_head = malloc(sizeof(OrcTriple) * (size_t)(u > 0 ? u : 1));
    kg->by_tail = malloc(sizeof(OrcTriple) * (size_t)(u > 0 ? u : 1));
    kg->by_rel = malloc(sizeof(OrcTriple) * (size_t)(u > 0 ? u : 1));
    memcpy(kg->by_head, sorted, sizeof(OrcTriple) * (size_t)u);
    memcpy(kg->by_tail, sorted, sizeof(OrcTriple) * (size_t)u);
    memcpy(kg->by_rel, sorted, sizeof(OrcTriple) * (size_t)u);
    free(sorted);
    qsort(kg->by_tail, (size_t)u, sizeof(OrcTriple), cmp_trh); /* Reader.h:126 */
    qsort(kg->by_rel, (size_t)u, sizeof(OrcTriple), cmp_htr);  /* Reader.h:127 */

    /* inclusive per-entity ranges; rig initialised to -1, lef to 0 (Reader.h:130-158) */
    size_t eb = sizeof(i64) * (size_t)(E > 0 ? E : 1);
    kg->lef_head = calloc(1, eb); kg->rig_head = malloc(eb);
    kg->lef_tail = calloc(1, eb); kg->rig_tail = malloc(eb);
    kg->lef_rel = calloc(1, eb);  kg->rig_rel = malloc(eb);
    for (i64 e = 0; e < E; e++) kg->rig_head[e] = kg->rig_tail[e] = kg->rig_rel[e] = -1;
    for (i64 i = 0; i < u; i++) {
        i64 eh = kg->by_head[i].h, et = kg->by_tail[i].t, er = kg->by_rel[i].h;
        if (i == 0 || kg->by_head[i - 1].h != eh) kg->lef_head[eh] = i;
        kg->rig_head[eh] = i;
        if (i == 0 || kg->by_tail[i - 1].t != et) kg->lef_tail[et] = i;
        kg->rig_tail[et] = i;
        if (i == 0 || kg->by_rel[i - 1].h != er) kg->lef_rel[er] = i;
        kg->rig_rel[er] = i;
    }
    /* tails-per-head / heads-per-tail (Reader.h:160-177): float accumulators, long/float division */
    kg->left_mean = calloc((size_t)(R > 0 ? R : 1), sizeof(float));
    kg->right_mean = calloc((size_t)(R > 0 ? R : 1), sizeof(float));
    for (i64 i = 0; i < u; i++) {
        if (i == 0 || kg->by_head[i - 1].h != kg->by_head[i].h || kg->by_head[i - 1].r != kg->by_head[i].r)
            kg->left_mean[kg->by_head[i].r] += 1.0f;
        if (i == 0 || kg->by_tail[i - 1].t != kg->by_tail[i].t || kg->by_tail[i - 1].r != kg->by_tail[i].r)
            kg->right_mean[kg->by_tail[i].r] += 1.0f;
    }
    for (i64 r = 0; r < R; r++) {
        kg->left_mean[r] = freq_rel[r] / kg->left_mean[r];
        kg->right_mean[r] = freq_rel[r] / kg->right_mean[r];
    }
    free(freq_rel);
}

OrcKG *orc_kg_from_arrays(i64 E, i64 R, i64 n, const i64 *h, const i64 *t, const i64 *r, i64 new_batch) {
    OrcKG *kg = calloc(1, sizeof(OrcKG));
    kg->ent_total = E; kg->rel_total = R; kg->train_dup = n; kg->new_batch = new_batch;
    kg->file_order = malloc(sizeof(OrcTriple) * (size_t)(n > 0 ? n : 1));
    for (i64 i = 0; i < n; i++) { kg->file_order[i].h = h[i]; kg->file_order[i].t = t[i]; kg->file_order[i].r = r[i]; }
    orc_kg_build(kg);
    kg->work_threads = 1;
    kg->stream = calloc(1, sizeof(u64));
    orc_libc_rand_init(&kg->libc);
    return kg;
}

/* Reader.h:27-100: relation2id.txt / entity2id.txt first line = count, optional batch2id.txt first
 * line = newBatchTotal, train2id.txt = N then N lines "head tail rel". */
OrcKG *orc_kg_load(const char *dir) {
    char path[4096];
    i64 E = 0, R = 0, n = 0, nb = 0;
    FILE *f;
    snprintf(path, sizeof path, "%srelation2id.txt", dir);
    if (!(f = fopen(path, "r"))) return NULL;
    if (fscanf(f, "%ld", &R) != 1) R = 0;
    fclose(f);
    snprintf(path, sizeof path, "%sentity2id.txt", dir);
    if (!(f = fopen(path, "r"))) return NULL;
    if (fscanf(f, "%ld", &E) != 1) E = 0;
    fclose(f);
    snprintf(path, sizeof path, "%sbatch2id.txt", dir);
    if ((f = fopen(path, "r"))) { if (fscanf(f, "%ld", &nb) != 1) nb = 0; fclose(f); }
    snprintf(path, sizeof path, "%strain2id.txt", dir);
    if (!(f = fopen(path, "r"))) return NULL;
    if (fscanf(f, "%ld", &n) != 1) n = 0;
    i64 *h = malloc(sizeof(i64) * (size_t)(n + 1)), *t = malloc(sizeof(i64) * (size_t)(n + 1)), *r = malloc(sizeof(i64) * (size_t)(n + 1));
    for (i64 i = 0; i < n; i++) {
        h[i] = t[i] = r[i] = 0;
        if (fscanf(f, "%ld", &h[i]) != 1) break;
        if (fscanf(f, "%ld", &t[i]) != 1) break;
        if (fscanf(f, "%ld", &r[i]) != 1) break;
    }
    fclose(f);
    OrcKG *kg = orc_kg_from_arrays(E, R, n, h, t, r, nb);
    free(h); free(t); free(r);
    return kg;
}

void orc_kg_free(OrcKG *kg) {
    if (!kg) return;
    free(kg->file_order); free(kg->by_head); free(kg->by_tail); free(kg->by_rel);
    free(kg->lef_head); free(kg->rig_head); free(kg->lef_tail); free(kg->rig_tail);
    free(kg->lef_rel); free(kg->rig_rel); free(kg->left_mean); free(kg->right_mean);
    free(kg->stream); free(kg);
}

i64 orc_kg_ent_total(const OrcKG *kg) { return kg->ent_total; }     /* Setting.h:63-66 */
i64 orc_kg_rel_total(const OrcKG *kg) { return kg->rel_total; }     /* Setting.h:68-71 */
i64 orc_kg_train_total(const OrcKG *kg) { return kg->train_uniq; }  /* Setting.h:78-81 */
i64 orc_kg_train_total_dup(const OrcKG *kg) { return kg->train_dup; } /* Setting.h:84-87 */
i64 orc_kg_batch_total(const OrcKG *kg) { return kg->new_batch; }   /* Setting.h:90-93 */
const float *orc_kg_left_mean(const OrcKG *kg) { return kg->left_mean; }
const float *orc_kg_right_mean(const OrcKG *kg) { return kg->right_mean; }
/* sorted copies, as (h,r,t) int64 triples, for index-parity tests */
const i64 *orc_kg_by_head(const OrcKG *kg) { return (const i64 *)kg->by_head; }
const i64 *orc_kg_by_tail(const OrcKG *kg) { return (const i64 *)kg->by_tail; }
const i64 *orc_kg_by_rel(const OrcKG *kg) { return (const i64 *)kg->by_rel; }

/* Setting.h:36-39 + Random.h:8-13: (re)allocate W streams and seed them from the *continuing*
 * libc sequence (a second randReset in the reference continues where the first stopped). */
void orc_set_work_threads(OrcKG *kg, i64 w) { kg->work_threads = w; }
void orc_set_bern(OrcKG *kg, i64 flag) { kg->bern = flag; } /* Setting.h:110-113 */
void orc_rand_reset(OrcKG *kg) {
    free(kg->stream);
    kg->stream = calloc((size_t)(kg->work_threads > 0 ? kg->work_threads : 1), sizeof(u64));
    for (i64 i = 0; i < kg->work_threads; i++) kg->stream[i] = (u64)(i64)orc_libc_rand_next(&kg->libc);
}
u64 orc_stream_state(const OrcKG *kg, i64 id) { return kg->stream[id]; }
void orc_set_stream_state(OrcKG *kg, i64 id, u64 s) { kg->stream[id] = s; }

/* Random.h:16-19 */
static inline u64 lcg_next(OrcKG *kg, i64 id) {
    kg->stream[id] = kg->stream[id] * 25214903917ULL + 11ULL;
    return kg->stream[id];
}
/* Random.h:22-27 (the negative-result loop can never run: the remainder is unsigned) */
static inline i64 draw_below(OrcKG *kg, i64 id, i64 x) { return (i64)(lcg_next(kg, id) % (u64)x); }

/* Corrupt.h:7-37 / 39-69 / 71-101, one routine for the three mirrored cases.
 * `arr` is one of the sorted copies restricted to the anchor entity's inclusive range [lo,hi];
 * `mid_of(i)` is the middle sort key (r for head/tail corruption, t for relation corruption) and
 * `val_of(i)` the last sort key (the candidates to be excluded, strictly increasing in the
 * sub-range).  Exactly one draw.  */
typedef enum { C_KEEP_HEAD, C_KEEP_TAIL, C_REL } CorruptKind;

static i64 corrupt(OrcKG *kg, i64 id, CorruptKind kind, i64 anchor, i64 key) {
    const OrcTriple *arr; i64 lo, hi, universe;
    if (kind == C_KEEP_HEAD) { arr = kg->by_head; lo = kg->lef_head[anchor]; hi = kg->rig_head[anchor]; universe = kg->ent_total; }
    else if (kind == C_KEEP_TAIL) { arr = kg->by_tail; lo = kg->lef_tail[anchor]; hi = kg->rig_tail[anchor]; universe = kg->ent_total; }
    else { arr = kg->by_rel; lo = kg->lef_rel[anchor]; hi = kg->rig_rel[anchor]; universe = kg->rel_total; }
#define MIDKEY(i) (kind == C_REL ? arr[i].t : arr[i].r)
#define VAL(i) (kind == C_KEEP_HEAD ? arr[i].t : (kind == C_KEEP_TAIL ? arr[i].h : arr[i].r))
    /* first index with midkey >= key  (Corrupt.h:9-16) */
    i64 a = lo - 1, b = hi;
    while (a + 1 < b) { i64 m = (a + b) >> 1; if (MIDKEY(m) >= key) b = m; else a = m; }
    i64 ll = b;
    /* last index with midkey <= key   (Corrupt.h:17-24) */
    a = lo; b = hi + 1;
    while (a + 1 < b) { i64 m = (a + b) >> 1; if (MIDKEY(m) <= key) a = m; else b = m; }
    i64 rr = a;
    i64 tmp = draw_below(kg, id, universe - (rr - ll + 1));      /* Corrupt.h:25 */
    if (tmp < VAL(ll)) return tmp;                                  /* Corrupt.h:26 */
    if (tmp > VAL(rr) - rr + ll - 1) return tmp + rr - ll + 1;      /* Corrupt.h:27 */
    a = ll; b = rr + 1;                                             /* Corrupt.h:28-36 */
    while (a + 1 < b) { i64 m = (a + b) >> 1; if (VAL(m) - m + ll - 1 < tmp) a = m; else b = m; }
    return tmp + a - ll + 1;
#undef MIDKEY
#undef VAL
}

/* Base.cpp:74-143 (getBatch) for one virtual thread, Base.cpp:149-172 (sampling) for all of them.
 * Threads write disjoint slices from independent streams, so running them one after another gives
 * the reference's output whatever the pthread schedule was. */
static void sampling_impl(OrcKG *kg, i64 *bh, i64 *bt, i64 *br, float *by, i64 B, i64 neg, i64 negrel, int parallel) {
    i64 W = kg->work_threads;
#pragma omp parallel for schedule(static, 1) if (parallel)
    for (i64 id = 0; id < W; id++) {
        i64 lef, rig;
        if (B % W == 0) { lef = id * (B / W); rig = (id + 1) * (B / W); }           /* Base.cpp:85-87 */
        else { lef = id * (B / W + 1); rig = (id + 1) * (B / W + 1); if (rig > B) rig = B; } /* :88-92 */
        float prob = 500;
        for (i64 b = lef; b < rig; b++) {
            i64 i;
            if (kg->new_batch > 0)                                                    /* Base.cpp:101-106 */
                i = (i64)(lcg_next(kg, id) % (u64)kg->new_batch) + (kg->train_dup - kg->new_batch); /* Random.h:32-34 */
            else
                i = draw_below(kg, id, kg->train_dup);
            OrcTriple p = kg->file_order[i];
            bh[b] = p.h; bt[b] = p.t; br[b] = p.r; by[b] = 1;                         /* Base.cpp:109-112 */
            i64 slot = b + B;
            for (i64 k = 0; k < neg; k++, slot += B) {                                /* Base.cpp:115-131 */
                if (kg->bern) prob = 1000 * kg->right_mean[p.r] / (kg->right_mean[p.r] + kg->left_mean[p.r]);
                if (lcg_next(kg, id) % 1000 < prob) {
                    bh[slot] = p.h; bt[slot] = corrupt(kg, id, C_KEEP_HEAD, p.h, p.r); br[slot] = p.r;
                } else {
                    bh[slot] = corrupt(kg, id, C_KEEP_TAIL, p.t, p.r); bt[slot] = p.t; br[slot] = p.r;
                }
                by[slot] = -1;
            }
            for (i64 k = 0; k < negrel; k++, slot += B) {                             /* Base.cpp:133-139 */
                bh[slot] = p.h; bt[slot] = p.t; br[slot] = corrupt(kg, id, C_REL, p.h, p.t); by[slot] = -1;
            }
        }
    }
}

void orc_sampling(OrcKG *kg, i64 *bh, i64 *bt, i64 *br, float *by, i64 B, i64 neg, i64 negrel) {
    sampling_impl(kg, bh, bt, br, by, B, neg, negrel, 0);
}
/* the same with one OS thread per virtual thread, as the reference's pthreads run it (Base.cpp:151-171): same output */
void orc_sampling_parallel(OrcKG *kg, i64 *bh, i64 *bt, i64 *br, float *by, i64 B, i64 neg, i64 negrel) {
    sampling_impl(kg, bh, bt, br, by, B, neg, negrel, 1);
}

/* ==========================================================================================
 * Model arithmetic.  fp32 throughout, written after the TF graph the reference builds.
 *
 * Batch layout (Model.py:55-74, matches Base.cpp:109-139): flat arrays of length B*(1+N), N =
 * negative_ent + negative_rel; positive b at [b], negative k of positive b at [B*(k+1)+b].
 *
 * Tables (variable names are the checkpoint contract):
 *   TransE: ent_embeddings[E,D] rel_embeddings[R,D]                         TransE.py:21-22
 *   TransH: + normal_vectors[R,D]                                           TransH.py:26-28
 *   TransR: ent[E,De] rel[R,Dr] transfer_matrix[R,De*Dr] (row-major De x Dr) TransR.py:29-31
 *   TransD: ent[E,D] rel[R,D] ent_transfer[E,D] rel_transfer[R,D]           TransD.py:37-40
 * ======================================================================================== */
enum { ORC_TRANSE = 0, ORC_TRANSH = 1, ORC_TRANSR = 2, ORC_TRANSD = 3 };
enum { ORC_SGD = 0, ORC_ADAM = 1 };

typedef struct {
    int model;
    i64 E, R;
    int De, Dr;      /* TransE/H/D: De == Dr == hidden_size */
    float margin;
    int negative_rel; /* TransR.py:57: pos matrix reused for negatives when 0 */
    /* parameter tables; unused ones NULL.  tab[0]=ent tab[1]=rel tab[2]=model specific tab[3]=.. */
    float *ent, *rel;
    float *aux_rel;  /* TransH normal_vectors | TransR transfer_matrix | TransD rel_transfer */
    float *aux_ent;  /* TransD ent_transfer */
} OrcModel;

static inline float sgnf(float x) { return (x > 0.f) - (x < 0.f); } /* TF Sign: sign(0)=0 */

/* tf.nn.l2_normalize(x,-1): x * rsqrt(max(sum(x^2), 1e-12))  (TransE.py:12-14).  returns inv, flag */
static inline float l2n(const float *x, int d, float *out, int *unclipped) {
    float ss = 0.f;
    for (int i = 0; i < d; i++) ss += x[i] * x[i];
    float m = ss >= 1e-12f ? ss : 1e-12f;
    float inv = 1.0f / sqrtf(m);
    for (int i = 0; i < d; i++) out[i] = x[i] * inv;
    *unclipped = ss >= 1e-12f;
    return inv;
}
/* backward of l2n: g_x = inv * (g_y - [unclipped] * y * <y, g_y>) */
static inline void l2n_bwd(const float *y, const float *gy, int d, float inv, int unclipped, float *gx) {
    float dot = 0.f;
    if (unclipped) for (int i = 0; i < d; i++) dot += y[i] * gy[i];
    for (int i = 0; i < d; i++) gx[i] = inv * (gy[i] - dot * y[i]);
}

#define MAXD 2048
typedef struct {
    /* projected (pre-normalisation) vectors and everything the backward needs */
    float hp[MAXD], tp[MAXD], hn[MAXD], tn[MAXD], rn[MAXD], wn[MAXD];
    float inv_h, inv_t, inv_r, inv_w, ah, at;
    int uc_h, uc_t, uc_r, uc_w;
} Scratch;

/* forward score of one triple (h,t,r).  mr = relation index whose projection is used (TransR). */
static float score_fwd(const OrcModel *M, i64 h, i64 t, i64 r, i64 mr, Scratch *S) {
    int De = M->De, Dr = M->Dr;
    const float *eh = M->ent + h * De, *et = M->ent + t * De, *er = M->rel + r * Dr;
    switch (M->model) {
    case ORC_TRANSE: /* TransE.py:35-45 */
        memcpy(S->hp, eh, sizeof(float) * De); memcpy(S->tp, et, sizeof(float) * De);
        break;
    case ORC_TRANSH: { /* TransH.py:12-14,54-59: e - sum(e*n)*n with n normalised */
        const float *w = M->aux_rel + r * Dr;
        S->inv_w = l2n(w, Dr, S->wn, &S->uc_w);
        float ah = 0.f, at = 0.f;
        for (int i = 0; i < Dr; i++) { ah += eh[i] * S->wn[i]; at += et[i] * S->wn[i]; }
        S->ah = ah; S->at = at;
        for (int i = 0; i < Dr; i++) { S->hp[i] = eh[i] - ah * S->wn[i]; S->tp[i] = et[i] - at * S->wn[i]; }
        break; }
    case ORC_TRANSR: { /* TransR.py:16-17,52-60: row vector [De] x matrix [De,Dr] */
        const float *Mx = M->aux_rel + mr * (i64)De * Dr;
        for (int j = 0; j < Dr; j++) { S->hp[j] = 0.f; S->tp[j] = 0.f; }
        for (int i = 0; i < De; i++) {
            float a = eh[i], b = et[i];
            const float *row = Mx + (i64)i * Dr;
            for (int j = 0; j < Dr; j++) { S->hp[j] += a * row[j]; S->tp[j] += b * row[j]; }
        }
        break; }
    case ORC_TRANSD: { /* TransD.py:23-25,62-67: e + sum(e*e_p)*r_p */
        const float *hpv = M->aux_ent + h * De, *tpv = M->aux_ent + t * De, *rp = M->aux_rel + r * Dr;
        float ah = 0.f, at = 0.f;
        for (int i = 0; i < De; i++) { ah += eh[i] * hpv[i]; at += et[i] * tpv[i]; }
        S->ah = ah; S->at = at;
        for (int i = 0; i < Dr; i++) { S->hp[i] = eh[i] + ah * rp[i]; S->tp[i] = et[i] + at * rp[i]; }
        break; }
    }
    /* _calc: abs(l2n(h) + l2n(r) - l2n(t)), reduce_sum  (TransE.py:11-15,48-49) */
    S->inv_h = l2n(S->hp, Dr, S->hn, &S->uc_h);
    S->inv_t = l2n(S->tp, Dr, S->tn, &S->uc_t);
    S->inv_r = l2n(er, Dr, S->rn, &S->uc_r);
    float s = 0.f;
    for (int i = 0; i < Dr; i++) s += fabsf(S->hn[i] + S->rn[i] - S->tn[i]);
    return s;
}

/* Sink for one gradient slice: table id, row, vector.  Used either to accumulate densely
 * (IndexedSlices dedup-sum) or to apply scatter_sub immediately. */
typedef struct {
    float *g[4];      /* dense accumulators per table (ent, rel, aux_rel, aux_ent) or NULL */
    float *p[4];      /* parameter tables for immediate scatter_sub */
    int width[4];
    float lr;
    int immediate;
    int atomic;       /* several threads share the accumulators */
} Sink;

static inline void emit(Sink *K, int tab, i64 row, const float *v) {
    int w = K->width[tab];
    if (K->immediate) { float *p = K->p[tab] + row * w; for (int i = 0; i < w; i++) p[i] -= v[i] * K->lr; }
    else if (K->atomic) {
        float *g = K->g[tab] + row * w;
        for (int i = 0; i < w; i++) {
#pragma omp atomic
            g[i] += v[i];
        }
    }
    else { float *g = K->g[tab] + row * w; for (int i = 0; i < w; i++) g[i] += v[i]; }
}

/* backward of one scored triple given dL/dscore = gs; S holds the forward state of THIS triple
 * computed from the (snapshot) parameters in M. */
static void score_bwd(const OrcModel *M, i64 h, i64 t, i64 r, i64 mr, const Scratch *S, float gs, Sink *K) {
    int De = M->De, Dr = M->Dr;
    float ge[MAXD], gneg[MAXD], ghp[MAXD], gtp[MAXD], gr[MAXD];
    for (int i = 0; i < Dr; i++) { ge[i] = gs * sgnf(S->hn[i] + S->rn[i] - S->tn[i]); gneg[i] = -ge[i]; }
    l2n_bwd(S->hn, ge, Dr, S->inv_h, S->uc_h, ghp);
    l2n_bwd(S->tn, gneg, Dr, S->inv_t, S->uc_t, gtp);
    l2n_bwd(S->rn, ge, Dr, S->inv_r, S->uc_r, gr);
    emit(K, 1, r, gr);
    const float *eh = M->ent + h * De, *et = M->ent + t * De;
    switch (M->model) {
    case ORC_TRANSE:
        emit(K, 0, h, ghp); emit(K, 0, t, gtp);
        break;
    case ORC_TRANSH: {
        float gh[MAXD], gt[MAXD], gwn[MAXD], gw[MAXD];
        float dh = 0.f, dt = 0.f;
        for (int i = 0; i < Dr; i++) { dh += ghp[i] * S->wn[i]; dt += gtp[i] * S->wn[i]; }
        for (int i = 0; i < Dr; i++) {
            gh[i] = ghp[i] - dh * S->wn[i];
            gt[i] = gtp[i] - dt * S->wn[i];
            gwn[i] = -(dh * eh[i] + S->ah * ghp[i]) - (dt * et[i] + S->at * gtp[i]);
        }
        l2n_bwd(S->wn, gwn, Dr, S->inv_w, S->uc_w, gw);
        emit(K, 0, h, gh); emit(K, 0, t, gt); emit(K, 2, r, gw);
        break; }
    case ORC_TRANSR: {
        const float *Mx = M->aux_rel + mr * (i64)De * Dr;
        float gh[MAXD], gt[MAXD];
        float *gM = malloc(sizeof(float) * (size_t)De * Dr);
        for (int i = 0; i < De; i++) {
            const float *row = Mx + (i64)i * Dr;
            float a = 0.f, b = 0.f;
            for (int j = 0; j < Dr; j++) { a += ghp[j] * row[j]; b += gtp[j] * row[j]; }
            gh[i] = a; gt[i] = b;
            for (int j = 0; j < Dr; j++) gM[(i64)i * Dr + j] = eh[i] * ghp[j] + et[i] * gtp[j];
        }
        emit(K, 0, h, gh); emit(K, 0, t, gt); emit(K, 2, mr, gM);
        free(gM);
        break; }
    case ORC_TRANSD: {
        const float *hpv = M->aux_ent + h * De, *tpv = M->aux_ent + t * De, *rp = M->aux_rel + r * Dr;
        float gh[MAXD], gt[MAXD], ghv[MAXD], gtv[MAXD], grp[MAXD];
        float dh = 0.f, dt = 0.f;
        for (int i = 0; i < Dr; i++) { dh += ghp[i] * rp[i]; dt += gtp[i] * rp[i]; }
        for (int i = 0; i < De; i++) {
            gh[i] = ghp[i] + dh * hpv[i]; gt[i] = gtp[i] + dt * tpv[i];
            ghv[i] = dh * eh[i]; gtv[i] = dt * et[i];
            grp[i] = S->ah * ghp[i] + S->at * gtp[i];
        }
        emit(K, 0, h, gh); emit(K, 0, t, gt); emit(K, 3, h, ghv); emit(K, 3, t, gtv); emit(K, 2, r, grp);
        break; }
    }
}

static void table_shapes(const OrcModel *M, i64 rows[4], int width[4]) {
    rows[0] = M->E; width[0] = M->De; rows[1] = M->R; width[1] = M->Dr;
    rows[2] = 0; width[2] = 0; rows[3] = 0; width[3] = 0;
    if (M->model == ORC_TRANSH) { rows[2] = M->R; width[2] = M->Dr; }
    if (M->model == ORC_TRANSR) { rows[2] = M->R; width[2] = M->De * M->Dr; }
    if (M->model == ORC_TRANSD) { rows[2] = M->R; width[2] = M->Dr; rows[3] = M->E; width[3] = M->De; }
}

/* loss = reduce_mean(max(p - n + margin, 0)) over B*N terms (TransE.py:51).  Also returns the
 * per-positive active counts and per-negative active flags for the backward. */
static float forward_all(const OrcModel *M, const i64 *bh, const i64 *bt, const i64 *br, i64 B, i64 N,
                         float *pos_score, float *neg_score, int nthreads) {
    (void)nthreads;
#pragma omp parallel num_threads(nthreads)
    {
        Scratch *S = malloc(sizeof(Scratch));
#pragma omp for schedule(static)
        for (i64 b = 0; b < B; b++) {
            pos_score[b] = score_fwd(M, bh[b], bt[b], br[b], br[b], S);
            for (i64 k = 0; k < N; k++) {
                i64 j = B * (k + 1) + b;
                i64 mr = M->negative_rel == 0 ? br[b] : br[j]; /* TransR.py:57-65 */
                neg_score[b * N + k] = score_fwd(M, bh[j], bt[j], br[j], mr, S);
            }
        }
        free(S);
    }
    /* the fp32 hinge values are added in double: a sequential fp32 sum of B*N ~ 10^5..10^6 terms is off by ~1e-5 relative on
     * its own (TF's reduce_mean is a blocked / pairwise reduction, not a sequential one), which is the tolerance itself */
    double total = 0.0;
    for (i64 b = 0; b < B; b++)
        for (i64 k = 0; k < N; k++) {
            float v = pos_score[b] - neg_score[b * N + k] + M->margin;
            total += v > 0.f ? (double)v : 0.0;
        }
    return (float)(total / (double)(B * N));
}

/* scores of the B positives and of the B*N negatives (ns[b*N+k]); lets a test see how close a hinge is to 0 */
void orc_scores(const OrcModel *M, const i64 *bh, const i64 *bt, const i64 *br, i64 B, i64 N, float *ps, float *ns) {
    forward_all(M, bh, bt, br, B, N, ps, ns, 1);
}

float orc_loss(const OrcModel *M, const i64 *bh, const i64 *bt, const i64 *br, i64 B, i64 N) {
    float *ps = malloc(sizeof(float) * (size_t)B), *ns = malloc(sizeof(float) * (size_t)(B * N));
    float l = forward_all(M, bh, bt, br, B, N, ps, ns, 1);
    free(ps); free(ns);
    return l;
}

/* Dense gradients of the loss w.r.t. every table (the dedup-summed IndexedSlices).  grads[i] must
 * be zero-initialised arrays of the table shapes (NULL for absent tables).  B_total/N_total allow a
 * data-parallel shard to be differentiated with the global mean's denominator.
 * With nthreads > 1 the threads split the positives and add into the shared accumulators with
 * atomic float adds (summation order then varies in the last bits); nthreads = 1 is the plain
 * sequential sum. */
float orc_grad(const OrcModel *M, const i64 *bh, const i64 *bt, const i64 *br, i64 B, i64 N,
               i64 denom, float *grads[4], int nthreads) {
    float *ps = malloc(sizeof(float) * (size_t)B), *ns = malloc(sizeof(float) * (size_t)(B * N));
    /* nthreads < 0: |nthreads| threads with THREAD-PRIVATE accumulators summed at the end (no atomic adds): the fast CPU
     * form, used as bench.py's cpu_baseline */
    const int private_acc = nthreads < -1;
    if (private_acc) nthreads = -nthreads;
    if (nthreads < 1) nthreads = 1;
    float loss = forward_all(M, bh, bt, br, B, N, ps, ns, nthreads);
    if (denom != B * N) loss = loss * (float)(B * N) / (float)denom;
    float unit = 1.0f / (float)denom;
    i64 rows[4]; int width[4];
    table_shapes(M, rows, width);
    static float *priv[64][4];      /* cached per-thread accumulators of the private form */
    static size_t priv_elems[4];
    if (private_acc) {
        if (nthreads > 64) nthreads = 64;
        for (int i = 0; i < 4; i++) {
            size_t n = grads[i] ? (size_t)rows[i] * (size_t)width[i] : 0;
            if (n != priv_elems[i]) {
                for (int t = 0; t < 64; t++) { free(priv[t][i]); priv[t][i] = NULL; }
                priv_elems[i] = n;
            }
        }
    }
#pragma omp parallel num_threads(nthreads)
    {
        Sink K; memset(&K, 0, sizeof K);
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        for (int i = 0; i < 4; i++) {
            K.g[i] = grads[i]; K.width[i] = width[i];
            if (private_acc && grads[i]) {
                if (!priv[tid][i]) priv[tid][i] = malloc(sizeof(float) * priv_elems[i]);
                memset(priv[tid][i], 0, sizeof(float) * priv_elems[i]);
                K.g[i] = priv[tid][i];
            }
        }
        K.atomic = nthreads > 1 && !private_acc;
        Scratch *S = malloc(sizeof(Scratch));
#pragma omp for schedule(static)
        for (i64 b = 0; b < B; b++) {
            /* d loss / d p_b = (#active negatives)/denom ; d loss / d n_bk = -[active]/denom.
             * TF maximum(x,0) routes the gradient to x when x >= 0. */
            float gp = 0.f;
            for (i64 k = 0; k < N; k++) if (ps[b] - ns[b * N + k] + M->margin >= 0.f) gp += unit;
            if (gp != 0.f) {
                score_fwd(M, bh[b], bt[b], br[b], br[b], S);
                score_bwd(M, bh[b], bt[b], br[b], br[b], S, gp, &K);
            }
            for (i64 k = 0; k < N; k++) {
                if (!(ps[b] - ns[b * N + k] + M->margin >= 0.f)) continue;
                i64 j = B * (k + 1) + b;
                i64 mr = M->negative_rel == 0 ? br[b] : br[j];
                score_fwd(M, bh[j], bt[j], br[j], mr, S);
                score_bwd(M, bh[j], bt[j], br[j], mr, S, -unit, &K);
            }
        }
        free(S);
        if (private_acc) {   /* every thread sums its share of the elements over all the private copies, in thread order */
#pragma omp barrier
            for (int i = 0; i < 4; i++) {
                if (!grads[i]) continue;
#pragma omp for schedule(static)
                for (i64 e = 0; e < (i64)priv_elems[i]; e++) {
                    float acc = grads[i][e];
                    for (int t = 0; t < nthreads; t++) acc += priv[t][i][e];
                    grads[i][e] = acc;
                }
            }
        }
    }
    free(ps); free(ns);
    return loss;
}

/* tf.train.GradientDescentOptimizer on IndexedSlices: var.scatter_sub(values*lr) — duplicates
 * accumulate one by one (distribute_training.py:98,101).  Forward/backward are evaluated on a
 * snapshot so every slice sees the pre-step parameters, exactly as one sess.run does. */
float orc_sgd_step_sequential(OrcModel *M, const i64 *bh, const i64 *bt, const i64 *br, i64 B, i64 N, float lr) {
    i64 rows[4]; int width[4];
    table_shapes(M, rows, width);
    float *live[4] = { M->ent, M->rel, M->aux_rel, M->aux_ent };
    OrcModel snap = *M;
    float *copy[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) if (rows[i]) {
        copy[i] = malloc(sizeof(float) * (size_t)(rows[i] * width[i]));
        memcpy(copy[i], live[i], sizeof(float) * (size_t)(rows[i] * width[i]));
    }
    snap.ent = copy[0]; snap.rel = copy[1]; snap.aux_rel = copy[2]; snap.aux_ent = copy[3];
    float *ps = malloc(sizeof(float) * (size_t)B), *ns = malloc(sizeof(float) * (size_t)(B * N));
    float loss = forward_all(&snap, bh, bt, br, B, N, ps, ns, 1);
    float unit = 1.0f / (float)(B * N);
    Sink K; memset(&K, 0, sizeof K);
    K.immediate = 1; K.lr = lr;
    for (int i = 0; i < 4; i++) { K.p[i] = live[i]; K.width[i] = width[i]; }
    Scratch *S = malloc(sizeof(Scratch));
    for (i64 b = 0; b < B; b++) {
        float gp = 0.f;
        for (i64 k = 0; k < N; k++) if (ps[b] - ns[b * N + k] + M->margin >= 0.f) gp += unit;
        if (gp != 0.f) { score_fwd(&snap, bh[b], bt[b], br[b], br[b], S); score_bwd(&snap, bh[b], bt[b], br[b], br[b], S, gp, &K); }
    }
    for (i64 k = 0; k < N; k++)
        for (i64 b = 0; b < B; b++) {
            if (!(ps[b] - ns[b * N + k] + M->margin >= 0.f)) continue;
            i64 j = B * (k + 1) + b;
            i64 mr = M->negative_rel == 0 ? br[b] : br[j];
            score_fwd(&snap, bh[j], bt[j], br[j], mr, S);
            score_bwd(&snap, bh[j], bt[j], br[j], mr, S, -unit, &K);
        }
    free(S); free(ps); free(ns);
    for (int i = 0; i < 4; i++) free(copy[i]);
    return loss;
}

/* SGD with the duplicate slices summed first: p -= lr * G.  Same mathematics as above, different
 * rounding order; this is the order the HIP engine uses. */
void orc_sgd_apply_dense(float *p, const float *g, i64 n, float lr) {
#pragma omp parallel for schedule(static) if (n > 100000)
    for (i64 i = 0; i < n; i++) p[i] -= lr * g[i];
}

/* tf.train.AdamOptimizer._apply_sparse_shared on the dedup-summed gradient (TF 1.x adam.py,
 * distribute_training.py:95-96): m and v decay for EVERY row, then the touched rows receive the
 * scaled gradient, then EVERY row moves.  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller
 * in fp32 with TF's op order (see oracle.py: adam_lr_t). */
void orc_adam_apply_dense(float *p, float *m, float *v, const float *g, i64 n,
                          float lr_t, float beta1, float beta2, float eps) {
    float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
#pragma omp parallel for schedule(static) if (n > 100000)
    for (i64 i = 0; i < n; i++) {
        float mi = m[i] * beta1;
        float vi = v[i] * beta2;
        float gi = g[i];
        if (gi != 0.f) { /* scatter_add of (1-b)*g on touched rows; adding an exact 0 is the identity */
            mi = mi + gi * omb1;
            vi = vi + (gi * gi) * omb2;
        }
        m[i] = mi; v[i] = vi;
        p[i] -= (lr_t * mi) / (sqrtf(vi) + eps);
    }
}

/* predict ops: TransE = reduce_mean over the embedding dim (TransE.py:58); the others reduce_sum
 * (TransH.py:82, TransR.py:87, TransD.py:98). */
void orc_predict(const OrcModel *M, const i64 *ph, const i64 *pt, const i64 *pr, i64 n, float *out) {
    Scratch *S = malloc(sizeof(Scratch));
    for (i64 i = 0; i < n; i++) {
        i64 mr = M->model == ORC_TRANSR ? pr[0] : pr[i]; /* TransR.py:83 uses predict_r[0] only */
        float s = score_fwd(M, ph[i], pt[i], pr[i], mr, S);
        out[i] = M->model == ORC_TRANSE ? s / (float)M->Dr : s;
    }
    free(S);
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ==========================================================================================
 * Link-prediction evaluation inputs and ranker.  PINNED against the
 * reference build (tests/golden/lp_*.npz).
 * ======================================================================================== */
typedef struct OrcEval {
    i64 ent_total, rel_total, test_total, valid_total, triple_total;
    OrcTriple *test_list;    /* sorted by (r,h,t)  (Reader.h:256, Triple.h:30-32) */
    OrcTriple *triple_list;  /* test + train + valid, sorted by (h,r,t) (Reader.h:255) */
    /* type constraints (Reader.h:302-365): per relation sorted candidate heads / tails */
    i64 *head_lef, *head_rig, *tail_lef, *tail_rig, *head_type, *tail_type;
    /* ontology (Reader.h:376-449): per entity sorted super / sub classes */
    i64 *sup_lef, *sup_rig, *sub_lef, *sub_rig, *sup_type, *sub_type;
} OrcEval;

static int cmp_rht(const void *pa, const void *pb) { /* Triple.h:30-32 cmp_rel2 */
    const OrcTriple *a = pa, *b = pb;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    if (a->h != b->h) return a->h < b->h ? -1 : 1;
    if (a->t != b->t) return a->t < b->t ? -1 : 1;
    return 0;
}
static int cmp_i64(const void *a, const void *b) { i64 x = *(const i64 *)a, y = *(const i64 *)b; return (x > y) - (x < y); }

static i64 *read_all_longs(const char *path, i64 *n_out) {
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    i64 cap = 1024, n = 0, v;
    i64 *a = malloc(sizeof(i64) * (size_t)cap);
    while (fscanf(f, "%ld", &v) == 1) { if (n == cap) { cap *= 2; a = realloc(a, sizeof(i64) * (size_t)cap); } a[n++] = v; }
    fclose(f);
    *n_out = n;
    return a;
}

/* Reader.h:186-292 */
OrcEval *orc_eval_load(const char *dir) {
    char path[4096];
    OrcEval *ev = calloc(1, sizeof(OrcEval));
    i64 n;
    i64 *a;
    snprintf(path, sizeof path, "%srelation2id.txt", dir); a = read_all_longs(path, &n); if (!a) return NULL; ev->rel_total = a[0]; free(a);
    snprintf(path, sizeof path, "%sentity2id.txt", dir); a = read_all_longs(path, &n); if (!a) return NULL; ev->ent_total = a[0]; free(a);
    i64 nt, ntr, nv;
    snprintf(path, sizeof path, "%stest2id.txt", dir); i64 *te = read_all_longs(path, &nt); if (!te) return NULL;
    snprintf(path, sizeof path, "%strain2id.txt", dir); i64 *tr = read_all_longs(path, &ntr); if (!tr) return NULL;
    snprintf(path, sizeof path, "%svalid2id.txt", dir); i64 *va = read_all_longs(path, &nv); if (!va) return NULL;
    i64 T = te[0], Tr = tr[0], V = va[0];
    ev->test_total = T; ev->valid_total = V; ev->triple_total = T + Tr + V;
    ev->test_list = calloc((size_t)(T > 0 ? T : 1), sizeof(OrcTriple));
    ev->triple_list = calloc((size_t)(ev->triple_total > 0 ? ev->triple_total : 1), sizeof(OrcTriple));
    i64 k = 0;
    for (i64 i = 0; i < T; i++) { OrcTriple x = { te[1 + 3 * i], te[3 + 3 * i], te[2 + 3 * i] }; ev->test_list[i] = x; ev->triple_list[k++] = x; }
    for (i64 i = 0; i < Tr; i++) { OrcTriple x = { tr[1 + 3 * i], tr[3 + 3 * i], tr[2 + 3 * i] }; ev->triple_list[k++] = x; }
    for (i64 i = 0; i < V; i++) { OrcTriple x = { va[1 + 3 * i], va[3 + 3 * i], va[2 + 3 * i] }; ev->triple_list[k++] = x; }
    free(te); free(tr); free(va);
    qsort(ev->triple_list, (size_t)ev->triple_total, sizeof(OrcTriple), cmp_hrt);
    qsort(ev->test_list, (size_t)T, sizeof(OrcTriple), cmp_rht);
    /* type_constrain.txt (Reader.h:302-365): count, then per relation "rel n heads..." and "rel n tails..." */
    snprintf(path, sizeof path, "%stype_constrain.txt", dir);
    a = read_all_longs(path, &n);
    if (a) {
        i64 R = ev->rel_total;
        ev->head_lef = calloc((size_t)R, sizeof(i64)); ev->head_rig = calloc((size_t)R, sizeof(i64));
        ev->tail_lef = calloc((size_t)R, sizeof(i64)); ev->tail_rig = calloc((size_t)R, sizeof(i64));
        ev->head_type = malloc(sizeof(i64) * (size_t)(n + 1)); ev->tail_type = malloc(sizeof(i64) * (size_t)(n + 1));
        i64 p = 1, nh = 0, ntl = 0;
        for (i64 i = 0; i < R && p + 1 < n; i++) {
            i64 rel = a[p], tot = a[p + 1]; p += 2;
            ev->head_lef[rel] = nh;
            for (i64 j = 0; j < tot; j++) ev->head_type[nh++] = a[p++];
            ev->head_rig[rel] = nh;
            qsort(ev->head_type + ev->head_lef[rel], (size_t)tot, sizeof(i64), cmp_i64);
            rel = a[p]; tot = a[p + 1]; p += 2;
            ev->tail_lef[rel] = ntl;
            for (i64 j = 0; j < tot; j++) ev->tail_type[ntl++] = a[p++];
            ev->tail_rig[rel] = ntl;
            qsort(ev->tail_type + ev->tail_lef[rel], (size_t)tot, sizeof(i64), cmp_i64);
        }
        free(a);
    }
    /* ontology_constrain.txt (Reader.h:376-449): count, then per entity "ent n supers..." and "ent n subs..." */
    snprintf(path, sizeof path, "%sontology_constrain.txt", dir);
    a = read_all_longs(path, &n);
    if (a) {
        i64 E = ev->ent_total;
        ev->sup_lef = calloc((size_t)E, sizeof(i64)); ev->sup_rig = calloc((size_t)E, sizeof(i64));
        ev->sub_lef = calloc((size_t)E, sizeof(i64)); ev->sub_rig = calloc((size_t)E, sizeof(i64));
        ev->sup_type = malloc(sizeof(i64) * (size_t)(n + 1)); ev->sub_type = malloc(sizeof(i64) * (size_t)(n + 1));
        i64 tot_ont = a[0], p = 1, ns = 0, nb = 0;
        for (i64 i = 0; i < tot_ont; i++) {
            i64 ent = a[p], tot = a[p + 1]; p += 2;
            ev->sup_lef[ent] = ns;
            for (i64 j = 0; j < tot; j++) ev->sup_type[ns++] = a[p++];
            ev->sup_rig[ent] = ns;
            qsort(ev->sup_type + ev->sup_lef[ent], (size_t)tot, sizeof(i64), cmp_i64);
            ent = a[p]; tot = a[p + 1]; p += 2;
            ev->sub_lef[ent] = nb;
            for (i64 j = 0; j < tot; j++) ev->sub_type[nb++] = a[p++];
            ev->sub_rig[ent] = nb;
            qsort(ev->sub_type + ev->sub_lef[ent], (size_t)tot, sizeof(i64), cmp_i64);
        }
        free(a);
    }
    return ev;
}

i64 orc_eval_test_total(const OrcEval *ev) { return ev->test_total; }
i64 orc_eval_valid_total(const OrcEval *ev) { return ev->valid_total; }
i64 orc_eval_triple_total(const OrcEval *ev) { return ev->triple_total; }
void orc_eval_test_triple(const OrcEval *ev, i64 i, i64 *htr) { htr[0] = ev->test_list[i].h; htr[1] = ev->test_list[i].t; htr[2] = ev->test_list[i].r; }

/* Corrupt.h:104-115 */
static int eval_find(const OrcEval *ev, i64 h, i64 t, i64 r) {
    i64 lef = 0, rig = ev->triple_total - 1;
    while (lef + 1 < rig) {
        i64 mid = (lef + rig) >> 1;
        const OrcTriple *m = &ev->triple_list[mid];
        if (m->h < h || (m->h == h && m->r < r) || (m->h == h && m->r == r && m->t < t)) lef = mid; else rig = mid;
    }
    const OrcTriple *a = &ev->triple_list[lef], *b = &ev->triple_list[rig];
    if (a->h == h && a->r == r && a->t == t) return 1;
    if (b->h == h && b->r == r && b->t == t) return 1;
    return 0;
}

/* ontology classes of the four arg-mins w.r.t. the expected entity (Test.h:113-135 / :226-248):
 * 0 correct, 1 a superclass (generalisation), 2 a subclass (specialisation), 3 anything else.
 * The reference walks ONE pair of cursors over the sorted super/sub lists for all four values in
 * turn and never rewinds them, so a later arg-min smaller than an earlier one can be missed --
 * reproduced by keeping the cursors across the four lookups. */
static void onto_classes(const OrcEval *ev, i64 expected, i64 *arr /* [4] in, [4] out */) {
    i64 lsup = ev->sup_lef ? ev->sup_lef[expected] : 0, rsup = ev->sup_lef ? ev->sup_rig[expected] : 0;
    i64 lsub = ev->sub_lef ? ev->sub_lef[expected] : 0, rsub = ev->sub_lef ? ev->sub_rig[expected] : 0;
    for (int i = 0; i < 4; i++) {
        i64 v = arr[i];
        if (v == expected) { arr[i] = 0; continue; }
        while (lsup < rsup && ev->sup_type[lsup] < v) lsup++;
        if (lsup < rsup && ev->sup_type[lsup] == v) { arr[i] = 1; continue; }
        while (lsub < rsub && ev->sub_type[lsub] < v) lsub++;
        if (lsub < rsub && ev->sub_type[lsub] == v) { arr[i] = 2; continue; }
        arr[i] = 3;
    }
}

/* Test.h:31-136 (head) and :141-249 (tail).  out[0..3] = number of candidates scoring strictly lower
 * (raw, filtered, type-constrained, both); out[4..7] = ontology class of the four arg-mins.  The
 * head version's filtered arg-min is updated OUTSIDE the `if (not _find)` (missing braces,
 * Test.h:69-74) -- reproduced. */
void orc_test_rank(const OrcEval *ev, i64 index, const float *con, int head, i64 *out) {
    i64 h = ev->test_list[index].h, t = ev->test_list[index].t, r = ev->test_list[index].r;
    i64 target = head ? h : t;
    float minimal = con[target];
    i64 s = 0, fs = 0, cs = 0, fcs = 0;
    i64 mn = target, fmn = target, cmn = target, fcmn = target;
    float mv = minimal, fmv = minimal, cmv = minimal, fcmv = minimal;
    i64 lef = 0, rig = 0;
    const i64 *types = NULL;
    if (ev->head_lef) { lef = head ? ev->head_lef[r] : ev->tail_lef[r]; rig = head ? ev->head_rig[r] : ev->tail_rig[r]; types = head ? ev->head_type : ev->tail_type; }
    for (i64 j = 0; j < ev->ent_total; j++) {
        if (j == target) continue;
        float value = con[j];
        int known = head ? eval_find(ev, j, t, r) : eval_find(ev, h, j, r);
        if (value < minimal) {
            s++;
            if (value < mv) { mv = value; mn = j; }
            if (!known) fs++;
            if (head ? 1 : !known) { if (value < fmv) { fmv = value; fmn = j; } }
        }
        while (lef < rig && types[lef] < j) lef++;
        if (lef < rig && types[lef] == j && value < minimal) {
            cs++;
            if (value < cmv) { cmv = value; cmn = j; }
            if (!known) { fcs++; if (value < fcmv) { fcmv = value; fcmn = j; } }
        }
    }
    out[0] = s; out[1] = fs; out[2] = cs; out[3] = fcs;
    out[4] = mn; out[5] = fmn; out[6] = cmn; out[7] = fcmn;
    onto_classes(ev, target, out + 4);
}
