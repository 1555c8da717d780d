/*
 * mmidx_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See mmidx_oracle.h.
 *
 * PARITY STATUS: "parity unpinned" (the Java reference has no tests / golden vectors and cannot
 * be built here). Every function cites the reference lines it restates;
 * J/ = /root/reference/src/main/java/gr/iti/mklab/visual/
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile). Arithmetic order is the
 * reference's: `acc += (a - b) * (a - b)` evaluates (a-b) twice, multiplies, then adds -- each
 * step rounded to binary64, no fused multiply-add.
 */
#include "mmidx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * BoundedPriorityQueue<Result>  (com.aliasi.util.BoundedPriorityQueue, LingPipe 4.0.1; NOT in
 * the reference tree -- assumption A1).  Call sites: IVFPQ.java:409,445,576,590-598;
 * PQ.java:291,318,335,368-370; Linear.java:140,156-158.
 *
 * Semantics restated:
 *   - entries ordered best-first by the user comparator (Result.compare: smaller distance is
 *     "greater", J/utilities/Result.java:38-45); ties ordered by insertion counter with the
 *     LATER-inserted entry first;
 *   - offer(o): size < max -> insert. Otherwise reject when compare(o, last) <= 0, i.e. when
 *     o.distance >= worst.distance (a candidate EQUAL to the current worst is rejected); else
 *     insert and remove last() (the worst; among equal-worst the EARLIEST inserted);
 *   - last() = worst kept, poll() = remove best, iteration / toArray = best -> worst.
 * Kept as a sorted array: key (distance ascending, insertion counter descending).
 * ---------------------------------------------------------------------------------------- */
/* The two behaviours above that cannot be read off the reference tree (LingPipe is a jar): what happens to a candidate EQUAL to
 * the current worst, and how equal entries are ordered.  Rule 0 is assumption A1.  Rules 1 and 2 are the two plausible
 * alternatives; they exist so that tests can show which answers do NOT depend on the assumption (tests/test_queue_rules_cpu.py):
 *   1  accept-equal-to-worst: offer() rejects only when o.distance > worst.distance (the evicted entry is still last());
 *   2  earlier-inserted-first among equals (a new entry goes BEHIND its equals; last() is then the LATEST inserted equal-worst).
 * Process-wide, set before any search; the product never sees it. */
static int g_queue_rule = 0;
void mmo_set_queue_rule(int rule) { g_queue_rule = (rule == 1 || rule == 2) ? rule : 0; }
int mmo_get_queue_rule(void) { return g_queue_rule; }

struct mmo_bpq {
    int cap, size;
    long long next_ins;
    double *dist;
    int *id;
    long long *ins;
};

mmo_bpq *mmo_bpq_new(int max_size) {
    if (max_size < 1) return NULL; /* LingPipe ctor rejects max size < 1 */
    mmo_bpq *q = (mmo_bpq *)calloc(1, sizeof(*q));
    q->cap = max_size;
    q->dist = (double *)malloc(sizeof(double) * (size_t)max_size);
    q->id = (int *)malloc(sizeof(int) * (size_t)max_size);
    q->ins = (long long *)malloc(sizeof(long long) * (size_t)max_size);
    return q;
}
void mmo_bpq_free(mmo_bpq *q) {
    if (!q) return;
    free(q->dist);
    free(q->id);
    free(q->ins);
    free(q);
}
void mmo_bpq_clear(mmo_bpq *q) {
    q->size = 0;
    q->next_ins = 0;
}
int mmo_bpq_size(const mmo_bpq *q) { return q->size; }
double mmo_bpq_last_dist(const mmo_bpq *q) { return q->dist[q->size - 1]; }

/* insert keeping (dist asc, ins desc); a new entry has the largest ins so it goes BEFORE all
 * existing entries of equal distance. */
static void bpq_insert(mmo_bpq *q, int id, double dist) {
    int lo = 0, hi = q->size;
    while (lo < hi) { /* first position whose dist >= new dist (rule 2: > new dist, behind its equals) */
        int mid = (lo + hi) >> 1;
        if (q->dist[mid] < dist || (g_queue_rule == 2 && q->dist[mid] == dist)) lo = mid + 1;
        else hi = mid;
    }
    int n = q->size - lo;
    memmove(q->dist + lo + 1, q->dist + lo, sizeof(double) * (size_t)n);
    memmove(q->id + lo + 1, q->id + lo, sizeof(int) * (size_t)n);
    memmove(q->ins + lo + 1, q->ins + lo, sizeof(long long) * (size_t)n);
    q->dist[lo] = dist;
    q->id[lo] = id;
    q->ins[lo] = q->next_ins++;
    q->size++;
}

int mmo_bpq_offer(mmo_bpq *q, int id, double dist) {
    if (q->size < q->cap) {
        bpq_insert(q, id, dist);
        return 1;
    }
    /* Result.compare(o, last) <= 0  <=>  !(o.dist < last.dist)   (rule 1: < 0, an equal candidate gets in) */
    if (g_queue_rule == 1 ? (dist > q->dist[q->size - 1]) : !(dist < q->dist[q->size - 1])) return 0;
    q->size--; /* remove last(): worst distance, earliest inserted among equals */
    bpq_insert(q, id, dist);
    return 1;
}

int mmo_bpq_poll(mmo_bpq *q, int *id, double *dist) {
    if (q->size == 0) return 0;
    if (id) *id = q->id[0];
    if (dist) *dist = q->dist[0];
    q->size--;
    memmove(q->dist, q->dist + 1, sizeof(double) * (size_t)q->size);
    memmove(q->id, q->id + 1, sizeof(int) * (size_t)q->size);
    memmove(q->ins, q->ins + 1, sizeof(long long) * (size_t)q->size);
    return 1;
}

int mmo_bpq_to_arrays(const mmo_bpq *q, int *ids, double *dists) {
    for (int i = 0; i < q->size; i++) {
        if (ids) ids[i] = q->id[i];
        if (dists) dists[i] = q->dist[i];
    }
    return q->size;
}

/* ------------------------------------------------------------------------------------------
 * java.util.Random (48-bit LCG, JDK javadoc) and Collections.shuffle(list, rnd):
 *   for (int i = size; i > 1; i--) swap(list, i - 1, rnd.nextInt(i));
 * J/utilities/RandomPermutation.java:29-40.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t s; } jrand;
static const uint64_t JMASK = (1ULL << 48) - 1;
static void jrand_seed(jrand *r, int64_t seed) { r->s = ((uint64_t)seed ^ 0x5DEECE66DULL) & JMASK; }
static int32_t jrand_next(jrand *r, int bits) {
    r->s = (r->s * 0x5DEECE66DULL + 0xBULL) & JMASK;
    return (int32_t)((int64_t)r->s >> (48 - bits)); /* (int)(seed >>> (48 - bits)) */
}
static int32_t jrand_next_int_bound(jrand *r, int32_t bound) {
    int32_t bits, val;
    if ((bound & (-bound)) == bound) /* power of two */
        return (int32_t)(((int64_t)bound * (int64_t)jrand_next(r, 31)) >> 31);
    do {
        bits = jrand_next(r, 31);
        val = bits % bound;
        /* while (bits - val + (bound - 1) < 0) with Java int wrap-around */
    } while ((int32_t)((uint32_t)bits - (uint32_t)val + (uint32_t)(bound - 1)) < 0);
    return val;
}
int32_t mmo_jdk_first_next_int(int64_t seed) {
    jrand r;
    jrand_seed(&r, seed);
    return jrand_next(&r, 32);
}
void mmo_random_permutation(int64_t seed, int dim, int32_t *perm) {
    jrand r;
    jrand_seed(&r, seed);
    for (int i = 0; i < dim; i++) perm[i] = i;
    for (int i = dim; i > 1; i--) {
        int j = jrand_next_int_bound(&r, i);
        int32_t t = perm[i - 1];
        perm[i - 1] = perm[j];
        perm[j] = t;
    }
}
void mmo_permute(const int32_t *perm, int dim, const double *v, double *out) {
    for (int i = 0; i < dim; i++) out[i] = v[perm[i]]; /* RandomPermutation.java:52-54 */
}
void mmo_rotate(const double *R, int dim, const double *v, double *out) {
    /* CommonOps.mult(original 1xD, randomMatrix DxD, transformed) RandomRotation.java:47.
     * A2: each output element accumulated sequentially over the inner index from 0. */
    for (int j = 0; j < dim; j++) {
        double total = 0;
        for (int i = 0; i < dim; i++) total += v[i] * R[(size_t)i * dim + j];
        out[j] = total;
    }
}

/* ------------------------------------------------------------------------------------------
 * Normalization.java
 * ---------------------------------------------------------------------------------------- */
void mmo_normalize_l2(double *v, int n) { /* :21-37 */
    double norm2 = 0;
    for (int i = 0; i < n; i++) norm2 += v[i] * v[i];
    norm2 = sqrt(norm2);
    if (norm2 == 0) {
        for (int i = 0; i < n; i++) v[i] = 1; /* Arrays.fill(vector, 1) */
    } else {
        for (int i = 0; i < n; i++) v[i] = v[i] / norm2;
    }
}
void mmo_normalize_l1(double *v, int n) { /* :47-62 */
    double norm1 = 0;
    for (int i = 0; i < n; i++) norm1 += fabs(v[i]);
    if (norm1 == 0) {
        for (int i = 0; i < n; i++) v[i] = 1.0 / n;
    } else {
        for (int i = 0; i < n; i++) v[i] = v[i] / norm1;
    }
}
static double jsignum(double x) { return (x == 0.0 || x != x) ? x : (x > 0 ? 1.0 : -1.0); }
void mmo_normalize_power(double *v, int n, double a) { /* :74-79 */
    for (int i = 0; i < n; i++) {
        /* Math.pow(x, 0.5) is specified within 1 ulp; for a == 0.5 sqrt() is the correctly
         * rounded value StrictMath/fdlibm pow returns for exact squares and what the GPU
         * path computes; other exponents use libm pow. */
        double p = (a == 0.5) ? sqrt(fabs(v[i])) : pow(fabs(v[i]), a);
        v[i] = jsignum(v[i]) * p;
    }
}
void mmo_normalize_ssr(double *v, int n) { /* :89-93 */
    mmo_normalize_power(v, n, 0.5);
    mmo_normalize_l2(v, n);
}

/* ------------------------------------------------------------------------------------------
 * Linear.computeNearestNeighborsInternal  Linear.java:138-163
 * ---------------------------------------------------------------------------------------- */
static int linear_search_q(mmo_bpq *nn, const double *X, int n, int D, const double *q, int k,
                           int *ids, double *dists) {
    (void)k;
    mmo_bpq_clear(nn);
    double lowest = 1.7976931348623157e308; /* Double.MAX_VALUE */
    for (int i = 0; i < n; i++) {
        int skip = 0;
        const double *x = X + (size_t)i * D;
        double l2 = 0;
        for (int j = 0; j < D; j++) {
            l2 += (q[j] - x[j]) * (q[j] - x[j]);
            if (l2 > lowest) {
                skip = 1;
                break;
            }
        }
        if (!skip) {
            mmo_bpq_offer(nn, i, l2);
            if (i >= nn->cap) lowest = mmo_bpq_last_dist(nn);
        }
    }
    return mmo_bpq_to_arrays(nn, ids, dists);
}
int mmo_linear_search(const double *X, int n, int D, const double *q, int k, int *ids,
                      double *dists) {
    mmo_bpq *nn = mmo_bpq_new(k);
    if (!nn) return -1;
    int c = linear_search_q(nn, X, n, D, q, k, ids, dists);
    mmo_bpq_free(nn);
    return c;
}

/* ------------------------------------------------------------------------------------------
 * PQ / IVFPQ index
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int len, cap;
    int *iids;      /* TIntArrayList invertedLists[c]      IVFPQ.java:83 */
    int8_t *bcodes; /* TByteArrayList pqByteCodes[c]       IVFPQ.java:72  (biased idx-128) */
    int16_t *scodes;/* TShortArrayList pqShortCodes[c]     IVFPQ.java:78 */
} mmo_list;

struct mmo_index {
    int kind, D, m, ks, dsub, C, transform, w;
    int load_counter;
    double *coarse; /* [C][D] */
    double *pq;     /* [m][ks][dsub] */
    int32_t *perm;
    double *rot;
    int nlists;
    mmo_list *lists; /* C lists for IVFPQ; 1 list for PQ (iid == position, PQ.java:303,318) */
};

mmo_index *mmo_index_new(int kind, int D, int m, int ks, int C, int transform,
                         const int32_t *perm, const double *rot) {
    if (m <= 0 || D % m > 0) return NULL; /* IVFPQ.java:181-183, PQ.java:148-150 */
    mmo_index *ix = (mmo_index *)calloc(1, sizeof(*ix));
    ix->kind = kind;
    ix->D = D;
    ix->m = m;
    ix->ks = ks;
    ix->dsub = D / m;
    ix->C = (kind == MMO_KIND_IVFPQ) ? C : 0;
    ix->transform = transform;
    ix->w = (int)(C * 0.1); /* IVFPQ.java:188 */
    if (transform == MMO_TR_PERMUTATION) {
        ix->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)D);
        if (perm) memcpy(ix->perm, perm, sizeof(int32_t) * (size_t)D);
        else mmo_random_permutation(1, D, ix->perm); /* seed = 1: IVFPQ.java:136, PQ.java:108 */
    } else if (transform == MMO_TR_ROTATION) {
        /* EJML RandomMatrices.createOrthogonal cannot be restated (source absent): the matrix
         * is an input. */
        if (!rot) {
            free(ix);
            return NULL;
        }
        ix->rot = (double *)malloc(sizeof(double) * (size_t)D * D);
        memcpy(ix->rot, rot, sizeof(double) * (size_t)D * D);
    }
    ix->nlists = (kind == MMO_KIND_IVFPQ) ? C : 1;
    ix->lists = (mmo_list *)calloc((size_t)ix->nlists, sizeof(mmo_list));
    return ix;
}
void mmo_index_free(mmo_index *ix) {
    if (!ix) return;
    for (int i = 0; i < ix->nlists; i++) {
        free(ix->lists[i].iids);
        free(ix->lists[i].bcodes);
        free(ix->lists[i].scodes);
    }
    free(ix->lists);
    free(ix->coarse);
    free(ix->pq);
    free(ix->perm);
    free(ix->rot);
    free(ix);
}
void mmo_index_set_coarse(mmo_index *ix, const double *coarse) {
    size_t n = (size_t)ix->C * ix->D;
    free(ix->coarse);
    ix->coarse = (double *)malloc(sizeof(double) * n);
    memcpy(ix->coarse, coarse, sizeof(double) * n);
}
void mmo_index_set_pq(mmo_index *ix, const double *pq) {
    size_t n = (size_t)ix->m * ix->ks * ix->dsub;
    free(ix->pq);
    ix->pq = (double *)malloc(sizeof(double) * n);
    memcpy(ix->pq, pq, sizeof(double) * n);
}
void mmo_index_set_w(mmo_index *ix, int w) { ix->w = w; }
int mmo_index_get_w(const mmo_index *ix) { return ix->w; }
int mmo_index_size(const mmo_index *ix) { return ix->load_counter; }
int8_t mmo_transform_to_byte(int idx) { return (int8_t)(idx - 128); } /* PQ.java:555 */

/* computeNearestCoarseIndex IVFPQ.java:547-564 (first index wins ties: update only on <) */
static int nearest_coarse_index(const mmo_index *ix, const double *v) {
    int centroid = -1;
    double min_d = 1.7976931348623157e308;
    for (int i = 0; i < ix->C; i++) {
        const double *c = ix->coarse + (size_t)i * ix->D;
        double d = 0;
        for (int j = 0; j < ix->D; j++) {
            d += (c[j] - v[j]) * (c[j] - v[j]);
            if (d >= min_d) break;
        }
        if (d < min_d) {
            min_d = d;
            centroid = i;
        }
    }
    return centroid;
}
/* computeNearestProductIndex IVFPQ.java:613-631 / PQ.java:411-429 */
static int nearest_product_index(const mmo_index *ix, const double *sub, int s) {
    int centroid = -1;
    double min_d = 1.7976931348623157e308;
    const double *base = ix->pq + (size_t)s * ix->ks * ix->dsub;
    for (int i = 0; i < ix->ks; i++) {
        const double *c = base + (size_t)i * ix->dsub;
        double d = 0;
        for (int j = 0; j < ix->dsub; j++) {
            d += (c[j] - sub[j]) * (c[j] - sub[j]);
            if (d >= min_d) break;
        }
        if (d < min_d) {
            min_d = d;
            centroid = i;
        }
    }
    return centroid;
}
/* computeResidualVector IVFPQ.java:642-648 : centroid - vector (note the sign) */
static void residual_vector(const mmo_index *ix, const double *v, int cell, double *out) {
    const double *c = ix->coarse + (size_t)cell * ix->D;
    for (int i = 0; i < ix->D; i++) out[i] = c[i] - v[i];
}
static void apply_transform(const mmo_index *ix, const double *in, double *out) {
    if (ix->transform == MMO_TR_ROTATION) mmo_rotate(ix->rot, ix->D, in, out);
    else if (ix->transform == MMO_TR_PERMUTATION) mmo_permute(ix->perm, ix->D, in, out);
    else memcpy(out, in, sizeof(double) * (size_t)ix->D);
}

void mmo_index_encode(const mmo_index *ix, const double *v, int *cell_out, int *code_out) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)ix->D * 2);
    double *res = tmp, *tr = tmp + ix->D;
    int cell = -1;
    if (ix->kind == MMO_KIND_IVFPQ) {
        cell = nearest_coarse_index(ix, v);  /* IVFPQ.java:315 */
        residual_vector(ix, v, cell, res);   /* :316 */
        apply_transform(ix, res, tr);        /* :319-323 */
    } else {
        apply_transform(ix, v, tr);          /* PQ.java:237-241 */
    }
    for (int s = 0; s < ix->m; s++)          /* IVFPQ.java:328-335 / PQ.java:245-252 */
        code_out[s] = nearest_product_index(ix, tr + (size_t)s * ix->dsub, s);
    if (cell_out) *cell_out = cell;
    free(tmp);
}

static void list_append(mmo_index *ix, mmo_list *L, int iid, const int *code) {
    if (L->len == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 16;
        L->iids = (int *)realloc(L->iids, sizeof(int) * (size_t)L->cap);
        if (ix->ks <= 256)
            L->bcodes = (int8_t *)realloc(L->bcodes, (size_t)L->cap * ix->m);
        else
            L->scodes = (int16_t *)realloc(L->scodes, sizeof(int16_t) * (size_t)L->cap * ix->m);
    }
    L->iids[L->len] = iid;
    for (int s = 0; s < ix->m; s++) {
        if (ix->ks <= 256) L->bcodes[(size_t)L->len * ix->m + s] = (int8_t)(code[s] - 128); /* PQ.java:555 */
        else L->scodes[(size_t)L->len * ix->m + s] = (int16_t)code[s];                      /* PQ.java:547 */
    }
    L->len++;
}

int mmo_index_add_code(mmo_index *ix, int iid, int cell, const int *code) {
    mmo_list *L = (ix->kind == MMO_KIND_IVFPQ) ? &ix->lists[cell] : &ix->lists[0];
    list_append(ix, L, iid, code);
    ix->load_counter++;
    return iid;
}

int mmo_index_load_lists(mmo_index *ix, int nlists, const int64_t *off, const int32_t *iids,
                         const void *codes) {
    if (nlists != ix->nlists) return -1;
    for (int c = 0; c < nlists; c++) {
        mmo_list *L = &ix->lists[c];
        const int64_t b = off[c], n = off[c + 1] - off[c];
        if (n == 0) continue;
        const int64_t need = L->len + n;
        L->iids = (int *)realloc(L->iids, sizeof(int) * (size_t)need);
        memcpy(L->iids + L->len, iids + b, sizeof(int) * (size_t)n);
        if (ix->ks <= 256) {
            L->bcodes = (int8_t *)realloc(L->bcodes, (size_t)need * ix->m);
            memcpy(L->bcodes + (size_t)L->len * ix->m, (const int8_t *)codes + (size_t)b * ix->m,
                   (size_t)n * ix->m);
        } else {
            L->scodes = (int16_t *)realloc(L->scodes, sizeof(int16_t) * (size_t)need * ix->m);
            memcpy(L->scodes + (size_t)L->len * ix->m, (const int16_t *)codes + (size_t)b * ix->m,
                   sizeof(int16_t) * (size_t)n * ix->m);
        }
        L->len = (int)need;
        L->cap = (int)need;
        ix->load_counter += (int)n;
    }
    return 0;
}

int mmo_index_add_vector(mmo_index *ix, const double *v) {
    int cell;
    int *code = (int *)malloc(sizeof(int) * (size_t)ix->m);
    mmo_index_encode(ix, v, &cell, code);
    int iid = ix->load_counter; /* invertedLists[cell].add(loadCounter) IVFPQ.java:339 */
    mmo_index_add_code(ix, iid, cell, code);
    free(code);
    return iid;
}

/* computeLookupADC IVFPQ.java:525-538 / PQ.java:387-399 */
void mmo_index_lookup_adc(const mmo_index *ix, const double *qv, double *lut) {
    for (int i = 0; i < ix->m; i++) {
        int start = i * ix->dsub;
        for (int j = 0; j < ix->ks; j++) {
            const double *c = ix->pq + ((size_t)i * ix->ks + j) * ix->dsub;
            double acc = 0; /* new double[][] is zero-initialised */
            for (int k = 0; k < ix->dsub; k++)
                acc += (qv[start + k] - c[k]) * (qv[start + k] - c[k]);
            lut[(size_t)i * ix->ks + j] = acc;
        }
    }
}

/* computeNearestCoarseIndices IVFPQ.java:575-601 */
static void nearest_coarse_indices(const mmo_index *ix, mmo_bpq *bpq, const double *v, int k,
                                   int *out) {
    mmo_bpq_clear(bpq);
    double lowest = 1.7976931348623157e308;
    for (int i = 0; i < ix->C; i++) {
        int skip = 0;
        const double *c = ix->coarse + (size_t)i * ix->D;
        double l2 = 0;
        for (int j = 0; j < ix->D; j++) {
            l2 += (c[j] - v[j]) * (c[j] - v[j]);
            if (l2 > lowest) {
                skip = 1;
                break;
            }
        }
        if (!skip) {
            mmo_bpq_offer(bpq, i, l2);
            if (i >= k) lowest = mmo_bpq_last_dist(bpq);
        }
    }
    /* for (i < k) nn[i] = bpq.poll().getId();  -- NPE in Java when k > C; here: -1 padding */
    for (int i = 0; i < k; i++) {
        int id = -1;
        if (!mmo_bpq_poll(bpq, &id, NULL)) id = -1;
        out[i] = id;
    }
}
void mmo_index_nearest_coarse(const mmo_index *ix, const double *q, int w, int *cells_out) {
    mmo_bpq *b = mmo_bpq_new(w);
    nearest_coarse_indices(ix, b, q, w, cells_out);
    mmo_bpq_free(b);
}

typedef struct {
    mmo_bpq *nn, *coarse_q;
    double *res, *tr, *lut;
    int *cells;
    int k, w;
} search_ws;

static search_ws *ws_new(const mmo_index *ix, int k) {
    search_ws *ws = (search_ws *)calloc(1, sizeof(*ws));
    ws->k = k;
    ws->w = ix->w;
    ws->nn = mmo_bpq_new(k);
    if (ix->kind == MMO_KIND_IVFPQ && ix->w >= 1) {
        ws->coarse_q = mmo_bpq_new(ix->w);
        ws->cells = (int *)malloc(sizeof(int) * (size_t)ix->w);
    }
    ws->res = (double *)malloc(sizeof(double) * (size_t)ix->D);
    ws->tr = (double *)malloc(sizeof(double) * (size_t)ix->D);
    ws->lut = (double *)malloc(sizeof(double) * (size_t)ix->m * ix->ks);
    return ws;
}
static void ws_free(search_ws *ws) {
    mmo_bpq_free(ws->nn);
    mmo_bpq_free(ws->coarse_q);
    free(ws->cells);
    free(ws->res);
    free(ws->tr);
    free(ws->lut);
    free(ws);
}

/* inner scan shared by IVFPQ.java:429-446 and PQ.java:303-319 */
static void scan_list(const mmo_index *ix, const mmo_list *L, const double *lut, mmo_bpq *nn) {
    const int m = ix->m, ks = ix->ks;
    for (int j = 0; j < L->len; j++) {
        int iid = L->iids[j];
        double l2 = 0;
        if (ks <= 256) {
            const int8_t *code = L->bcodes + (size_t)j * m;
            for (int s = 0; s < m; s++) l2 += lut[(size_t)s * ks + (code[s] + 128)]; /* :437 */
        } else {
            const int16_t *code = L->scodes + (size_t)j * m;
            for (int s = 0; s < m; s++) l2 += lut[(size_t)s * ks + code[s]];         /* :442 */
        }
        mmo_bpq_offer(nn, iid, l2); /* :445 */
    }
}

static int search_ws_run(const mmo_index *ix, search_ws *ws, const double *q, int *ids,
                         double *dists) {
    mmo_bpq_clear(ws->nn);
    if (ix->kind == MMO_KIND_IVFPQ) {
        /* computeKnnIVFADC IVFPQ.java:408-450 */
        nearest_coarse_indices(ix, ws->coarse_q, q, ix->w, ws->cells); /* :412 */
        for (int i = 0; i < ix->w; i++) {
            int cell = ws->cells[i];
            if (cell < 0) break;
            residual_vector(ix, q, cell, ws->res);       /* :417 */
            apply_transform(ix, ws->res, ws->tr);        /* :420-424 */
            mmo_index_lookup_adc(ix, ws->tr, ws->lut);   /* :427 */
            scan_list(ix, &ix->lists[cell], ws->lut, ws->nn);
        }
    } else {
        /* computeKnnADC PQ.java:290-322 ; iid == position i (PQ.java:318) */
        apply_transform(ix, q, ws->tr);
        mmo_index_lookup_adc(ix, ws->tr, ws->lut);
        scan_list(ix, &ix->lists[0], ws->lut, ws->nn);
    }
    return mmo_bpq_to_arrays(ws->nn, ids, dists); /* ASS.lookUp: best -> worst */
}

int mmo_index_search(const mmo_index *ix, int k, const double *q, int *ids, double *dists) {
    if (k < 1) return -1;
    if (ix->kind == MMO_KIND_IVFPQ && (ix->w < 1 || ix->w > ix->C)) return -1;
    search_ws *ws = ws_new(ix, k);
    int c = search_ws_run(ix, ws, q, ids, dists);
    ws_free(ws);
    return c;
}

long long mmo_index_probed_codes(const mmo_index *ix, const double *q) {
    if (ix->kind != MMO_KIND_IVFPQ) return ix->lists[0].len;
    int *cells = (int *)malloc(sizeof(int) * (size_t)ix->w);
    mmo_index_nearest_coarse(ix, q, ix->w, cells);
    long long t = 0;
    for (int i = 0; i < ix->w; i++)
        if (cells[i] >= 0) t += ix->lists[cells[i]].len;
    free(cells);
    return t;
}
void mmo_index_list_sizes(const mmo_index *ix, int *out) {
    for (int i = 0; i < ix->nlists; i++) out[i] = ix->lists[i].len;
}

/* ---- per-id utilities: the reference reads the BDB record {listId, code} of the id (IVFPQ.java:760-772 layout);
 * here the record is found by scanning the in-memory lists (test sizes only). ---- */
static int find_record(const mmo_index *ix, int iid, int *cell_out, int *pos_out) {
    for (int c = 0; c < ix->nlists; c++) {
        const mmo_list *L = &ix->lists[c];
        for (int j = 0; j < L->len; j++)
            if (L->iids[j] == iid) {
                *cell_out = c;
                *pos_out = j;
                return 1;
            }
    }
    return 0;
}
/* getInvertedListId IVFPQ.java:865-880 + getPQCodeByte :801-824 / getPQCodeShort :833-856: cell (-1 for PQ) and the STORED
 * code values (byte: idx-128, short: idx) widened to int; returns 0 when the id does not exist ("Id does not exist!") */
int mmo_index_get_record(const mmo_index *ix, int iid, int *cell_out, int *code_out) {
    int c, j;
    if (!find_record(ix, iid, &c, &j)) return 0;
    const mmo_list *L = &ix->lists[c];
    *cell_out = ix->kind == MMO_KIND_IVFPQ ? c : -1;
    for (int s = 0; s < ix->m; s++)
        code_out[s] = ix->ks <= 256 ? (int)L->bcodes[(size_t)j * ix->m + s] : (int)L->scodes[(size_t)j * ix->m + s];
    return 1;
}
/* computeDistanceIVFADC IVFPQ.java:464-497: residual w.r.t. the id's cell (:470), transformation (:473-477), lookup table
 * (:480), sum over the sub-quantizers of the entries the stored code selects (:482-495).  PQ: the query itself is
 * transformed (no residual).  Returns 0 when the id does not exist. */
int mmo_index_distance(const mmo_index *ix, const double *q, int iid, double *dist_out) {
    int c, j;
    if (!find_record(ix, iid, &c, &j)) return 0;
    const mmo_list *L = &ix->lists[c];
    double *res = (double *)malloc(sizeof(double) * (size_t)ix->D);
    double *tr = (double *)malloc(sizeof(double) * (size_t)ix->D);
    double *lut = (double *)malloc(sizeof(double) * (size_t)ix->m * (size_t)ix->ks);
    if (ix->kind == MMO_KIND_IVFPQ) {
        residual_vector(ix, q, c, res);
        apply_transform(ix, res, tr);
    } else {
        apply_transform(ix, q, tr);
    }
    mmo_index_lookup_adc(ix, tr, lut);
    double distance = 0; /* :465 */
    for (int s = 0; s < ix->m; s++) {
        if (ix->ks <= 256) distance += lut[(size_t)s * ix->ks + (L->bcodes[(size_t)j * ix->m + s] + 128)]; /* :486 */
        else distance += lut[(size_t)s * ix->ks + L->scodes[(size_t)j * ix->m + s]];                         /* :491 */
    }
    free(res);
    free(tr);
    free(lut);
    *dist_out = distance;
    return 1;
}

/* computeKnnSDC PQ.java:334-374 (byte codes; the reference dereferences pqByteCodes
 * unconditionally at :350, so short codes NPE there -- not restated). */
int mmo_pq_search_sdc(const mmo_index *ix, int k, int iid, int *ids, double *dists) {
    if (ix->kind != MMO_KIND_PQ || ix->ks > 256 || k < 1) return -1;
    const mmo_list *L = &ix->lists[0];
    if (iid < 0 || iid >= L->len) return -1;
    mmo_bpq *nn = mmo_bpq_new(k);
    const int m = ix->m, ks = ix->ks, dsub = ix->dsub;
    const int8_t *cq = L->bcodes + (size_t)iid * m;
    double lowest = 1.7976931348623157e308;
    for (int i = 0; i < ix->load_counter; i++) {
        double l2 = 0;
        for (int j = 0; j < m; j++) {
            int a = L->bcodes[(size_t)i * m + j] + 128;
            int b = cq[j] + 128;
            const double *pa = ix->pq + ((size_t)j * ks + a) * dsub;
            const double *pb = ix->pq + ((size_t)j * ks + b) * dsub;
            for (int t = 0; t < dsub; t++) {
                l2 += (pa[t] - pb[t]) * (pa[t] - pb[t]);
                if (l2 > lowest) break;
            }
            if (l2 > lowest) break;
        }
        mmo_bpq_offer(nn, i, l2); /* offered even when abandoned (:368): then l2 > lowest */
        if (i >= k) lowest = mmo_bpq_last_dist(nn);
    }
    int c = mmo_bpq_to_arrays(nn, ids, dists);
    mmo_bpq_free(nn);
    return c;
}

/* ---- multi-threaded batch drivers (timed CPU baseline) ---- */
typedef struct {
    const mmo_index *ix;
    const double *X;
    int n, D, k, nq, t, nt;
    const double *Q;
    int *ids;
    double *dists;
    int *counts;
} batch_arg;

static void *batch_worker(void *p) {
    batch_arg *a = (batch_arg *)p;
    int D = a->ix ? a->ix->D : a->D;
    if (a->ix) {
        search_ws *ws = ws_new(a->ix, a->k);
        for (int i = a->t; i < a->nq; i += a->nt)
            a->counts[i] = search_ws_run(a->ix, ws, a->Q + (size_t)i * D,
                                         a->ids + (size_t)i * a->k, a->dists + (size_t)i * a->k);
        ws_free(ws);
    } else {
        mmo_bpq *nn = mmo_bpq_new(a->k);
        for (int i = a->t; i < a->nq; i += a->nt)
            a->counts[i] = linear_search_q(nn, a->X, a->n, D, a->Q + (size_t)i * D, a->k,
                                           a->ids + (size_t)i * a->k, a->dists + (size_t)i * a->k);
        mmo_bpq_free(nn);
    }
    return NULL;
}
static void run_batch(batch_arg proto, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    batch_arg *args = (batch_arg *)malloc(sizeof(batch_arg) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        args[t] = proto;
        args[t].t = t;
        args[t].nt = nthreads;
        if (nthreads == 1) batch_worker(&args[t]);
        else pthread_create(&th[t], NULL, batch_worker, &args[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(args);
}
void mmo_index_search_batch(const mmo_index *ix, int k, int nq, const double *Q, int *ids,
                            double *dists, int *counts, int nthreads) {
    batch_arg a;
    memset(&a, 0, sizeof(a));
    a.ix = ix;
    a.k = k;
    a.nq = nq;
    a.Q = Q;
    a.ids = ids;
    a.dists = dists;
    a.counts = counts;
    run_batch(a, nthreads);
}
void mmo_linear_search_batch(const double *X, int n, int D, int k, int nq, const double *Q,
                             int *ids, double *dists, int *counts, int nthreads) {
    batch_arg a;
    memset(&a, 0, sizeof(a));
    a.X = X;
    a.n = n;
    a.D = D;
    a.k = k;
    a.nq = nq;
    a.Q = Q;
    a.ids = ids;
    a.dists = dists;
    a.counts = counts;
    run_batch(a, nthreads);
}

/* ------------------------------------------------------------------------------------------
 * PCA  (J/dimreduction/PCA.java)
 * ---------------------------------------------------------------------------------------- */
void mmo_pca_whiten(double *Vt, const double *eig, int nc, int ss) {
    /* W(i,i) = pow(eig_i, -0.5) :283-285 ; V_t <- W * V_t :311. W is diagonal, so the EJML
     * product row i is w_ii * V_t[i][j] plus exact zeros. */
    for (int i = 0; i < nc; i++) {
        double w = pow(eig[i], -0.5);
        for (int j = 0; j < ss; j++) Vt[(size_t)i * ss + j] = w * Vt[(size_t)i * ss + j];
    }
}
void mmo_pca_project(const double *Vt, const double *means, int nc, int ss, int whitening,
                     const double *x, double *y) {
    /* sample - means :199 ; V_t * sample :201 (matrix-vector: y_i = sum_j V_t[i][j]*xc[j],
     * sequential j, assumption A2) ; normalizeL2 iff whitening :203-204 */
    double *xc = (double *)malloc(sizeof(double) * (size_t)ss);
    for (int j = 0; j < ss; j++) xc[j] = x[j] - means[j];
    for (int i = 0; i < nc; i++) {
        double total = 0;
        const double *row = Vt + (size_t)i * ss;
        for (int j = 0; j < ss; j++) total += row[j] * xc[j];
        y[i] = total;
    }
    free(xc);
    if (whitening) mmo_normalize_l2(y, nc);
}

/* ------------------------------------------------------------------------------------------
 * VLAD  (J/aggregation)
 * ---------------------------------------------------------------------------------------- */
int mmo_nearest_centroid(const double *cb, int nc, int dl, const double *desc) {
    /* AbstractFeatureAggregator.java:136-155 */
    int centroid = -1;
    double min_d = 1.7976931348623157e308;
    for (int i = 0; i < nc; i++) {
        const double *c = cb + (size_t)i * dl;
        double d = 0;
        for (int j = 0; j < dl; j++) {
            d += (c[j] - desc[j]) * (c[j] - desc[j]);
            if (d >= min_d) break;
        }
        if (d < min_d) {
            min_d = d;
            centroid = i;
        }
    }
    return centroid;
}
void mmo_vlad_aggregate(const double *cb, int nc, int dl, const double *descs, int ndesc,
                        double *vlad) {
    /* VladAggregator.java:56-70 */
    memset(vlad, 0, sizeof(double) * (size_t)nc * dl);
    for (int d = 0; d < ndesc; d++) {
        const double *desc = descs + (size_t)d * dl;
        int nn = mmo_nearest_centroid(cb, nc, dl, desc);
        for (int i = 0; i < dl; i++) vlad[(size_t)nn * dl + i] += desc[i] - cb[(size_t)nn * dl + i];
    }
}
void mmo_vlad_aggregate_multi(const double *cbs, const int *nc, int nvocab, int dl,
                              const double *descs, int ndesc, int norms_on, double *out) {
    /* VladAggregatorMultipleVocabularies.java:84-101 */
    size_t shift = 0, cb_off = 0;
    for (int v = 0; v < nvocab; v++) {
        double *sub = out + shift;
        mmo_vlad_aggregate(cbs + cb_off, nc[v], dl, descs, ndesc, sub);
        if (norms_on) {
            mmo_normalize_power(sub, nc[v] * dl, 0.5);
            mmo_normalize_l2(sub, nc[v] * dl);
        }
        shift += (size_t)nc[v] * dl;
        cb_off += (size_t)nc[v] * dl;
    }
    if (nvocab > 1 && norms_on) mmo_normalize_l2(out, (int)shift);
}
