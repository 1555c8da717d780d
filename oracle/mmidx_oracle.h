/*
 * mmidx_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Loop-faithful plain-C restatement of the search path of MKLab-ITI/multimedia-indexing
 * (gr.iti.mklab.visual.datastructures Linear / PQ / IVFPQ, plus the PCA projection and VLAD
 * aggregation that feed it).  Citations are relative to /root/reference, with
 *   J/ = src/main/java/gr/iti/mklab/visual/
 *
 * PARITY STATUS: "parity unpinned".  The reference is pure Java, ships no tests, golden
 * vectors or fixtures, and cannot be compiled or run in this environment (no JDK, none of the
 * LingPipe / Trove / BDB-JE / EJML jars).  This oracle is pinned only by (a) hand-derived KATs
 * (tests/test_oracle_kats.py), (b) an independent numpy twin written in a different code shape
 * (tests/np_twin.py) and (c) the JDK-specified java.util.Random / Collections.shuffle algorithm.
 * Two behaviours come from third-party jars whose source is not in the reference tree:
 *   A1  com.aliasi:lingpipe:4.0.1  com.aliasi.util.BoundedPriorityQueue (tie / eviction order)
 *   A2  ejml:0.23 CommonOps.mult (sequential accumulation over the inner index)
 * They are restated from their published behaviour and flagged wherever they matter.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (libmmidx_hip.so) never links, loads or calls it.
 *
 * All arithmetic is IEEE-754 binary64, no FMA contraction (build with -ffp-contract=off),
 * strictly in the reference's left-to-right order.
 */
#ifndef MMIDX_ORACLE_H
#define MMIDX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bounded priority queue (LingPipe BoundedPriorityQueue<Result> + Result comparator) ---- */
/* J/utilities/Result.java:38-45 : smaller distance = higher priority. Assumption A1 for ties. */
typedef struct mmo_bpq mmo_bpq;
/* queue rule: 0 = assumption A1 (default), 1 = accept-equal-to-worst, 2 = earlier-inserted-first among equals (tests only) */
void mmo_set_queue_rule(int rule);
int mmo_get_queue_rule(void);
mmo_bpq *mmo_bpq_new(int max_size);
void mmo_bpq_free(mmo_bpq *q);
void mmo_bpq_clear(mmo_bpq *q);
int mmo_bpq_offer(mmo_bpq *q, int id, double dist); /* 1 = accepted, 0 = rejected */
int mmo_bpq_size(const mmo_bpq *q);
double mmo_bpq_last_dist(const mmo_bpq *q); /* worst kept (queue must be non-empty) */
int mmo_bpq_poll(mmo_bpq *q, int *id, double *dist); /* remove best; 0 if empty */
int mmo_bpq_to_arrays(const mmo_bpq *q, int *ids, double *dists); /* best -> worst, returns size */

/* ---- java.util.Random + Collections.shuffle (J/utilities/RandomPermutation.java:29-40) ---- */
int32_t mmo_jdk_first_next_int(int64_t seed); /* new Random(seed).nextInt() */
void mmo_random_permutation(int64_t seed, int dim, int32_t *perm_out);
/* J/utilities/RandomPermutation.java:50-56 : out[i] = v[perm[i]] */
void mmo_permute(const int32_t *perm, int dim, const double *v, double *out);
/* J/utilities/RandomRotation.java:44-49 : out(1xD) = v(1xD) * R(DxD), R row-major (A2) */
void mmo_rotate(const double *R, int dim, const double *v, double *out);

/* ---- J/utilities/Normalization.java ---- */
void mmo_normalize_l2(double *v, int n);               /* :21-37 */
void mmo_normalize_l1(double *v, int n);               /* :47-62 */
void mmo_normalize_power(double *v, int n, double a);  /* :74-79 */
void mmo_normalize_ssr(double *v, int n);              /* :89-93 */

/* ---- Linear (J/datastructures/Linear.java:138-163) ---- */
int mmo_linear_search(const double *X, int n, int D, const double *q, int k, int *ids,
                      double *dists);

/* ---- PQ / IVFPQ index ---- */
enum { MMO_KIND_PQ = 1, MMO_KIND_IVFPQ = 2 };
/* PQ.TransformationType ordinal order, J/datastructures/PQ.java:78-80 */
enum { MMO_TR_NONE = 0, MMO_TR_ROTATION = 1, MMO_TR_PERMUTATION = 2 };

typedef struct mmo_index mmo_index;

/* ctor checks follow IVFPQ.java:174-194 / PQ.java:142-159 (D % m != 0 -> NULL).
 * perm (D ints) is used when transform == PERMUTATION, rot (DxD row-major) when ROTATION;
 * pass NULL perm to get RandomPermutation(seed = 1, D) exactly as IVFPQ.java:136,193. */
mmo_index *mmo_index_new(int kind, int D, int m, int ks, int C, int transform,
                         const int32_t *perm, const double *rot);
void mmo_index_free(mmo_index *ix);
void mmo_index_set_coarse(mmo_index *ix, const double *coarse /* [C][D] */);
void mmo_index_set_pq(mmo_index *ix, const double *pq /* [m][ks][dsub], file order */);
void mmo_index_set_w(mmo_index *ix, int w); /* IVFPQ.java:95-97 ; default (int)(0.1*C) :188 */
int mmo_index_get_w(const mmo_index *ix);
int mmo_index_size(const mmo_index *ix);

/* encode: IVFPQ.java:309-335 (+:547-564, :613-631, :642-648) / PQ.java:232-252 (+:411-429).
 * code_out receives m centroid indices (0..ks-1, before the -128 byte bias). cell_out = -1 for PQ. */
void mmo_index_encode(const mmo_index *ix, const double *v, int *cell_out, int *code_out);
/* indexVectorInternal: encode + append; returns the iid (= loadCounter before increment) */
int mmo_index_add_vector(mmo_index *ix, const double *v);
/* indexPQCode (IVFPQ.java:357-386) / loadIndexInMemory (:680-728): append a precomputed code.
 * code = centroid indices 0..ks-1. */
int mmo_index_add_code(mmo_index *ix, int iid, int cell, const int *code);
/* bulk form of loadIndexInMemory (IVFPQ.java:680-728 / PQ.java:436-483): nlists lists given
 * list-major: off[nlists+1], iids[n], codes[n][m] in the STORED form (int8 idx-128 / int16). */
int mmo_index_load_lists(mmo_index *ix, int nlists, const int64_t *off, const int32_t *iids,
                         const void *codes);
/* Java-side stored byte of centroid index idx: PQ.transformToByte, PQ.java:552-558 */
int8_t mmo_transform_to_byte(int idx);

/* computeKnnIVFADC IVFPQ.java:408-450 / computeKnnADC PQ.java:290-322; returns result count,
 * ids/dists best -> worst (ASS.lookUp, AbstractSearchStructure.java:345-358). */
int mmo_index_search(const mmo_index *ix, int k, const double *q, int *ids, double *dists);
/* computeNearestCoarseIndices IVFPQ.java:575-601 (nearest first) */
void mmo_index_nearest_coarse(const mmo_index *ix, const double *q, int w, int *cells_out);
/* computeLookupADC IVFPQ.java:525-538 / PQ.java:387-399 : lut[m][ks] */
void mmo_index_lookup_adc(const mmo_index *ix, const double *qvec, double *lut);
/* computeKnnSDC PQ.java:334-374 (byte codes only; the reference NPEs on short codes) */
/* per-id utilities: IVFPQ.java:865-880 / :801-856 (record of an id) and :464-497 (computeDistanceIVFADC); 0 = unknown id */
int mmo_index_get_record(const mmo_index *ix, int iid, int *cell_out, int *code_out);
int mmo_index_distance(const mmo_index *ix, const double *q, int iid, double *dist_out);
int mmo_pq_search_sdc(const mmo_index *ix, int k, int iid, int *ids, double *dists);

/* batch driver used for the timed CPU baseline: nthreads concurrent readers, each issuing whole
 * single-query calls (legal: computeNearestNeighbors is unsynchronised, ASS:281). */
void mmo_index_search_batch(const mmo_index *ix, int k, int nq, const double *Q, int *ids,
                            double *dists, int *counts, int nthreads);
void mmo_linear_search_batch(const double *X, int n, int D, int k, int nq, const double *Q,
                             int *ids, double *dists, int *counts, int nthreads);
/* sum over the probed lists of their lengths for one query (algorithmic bytes = m * this) */
long long mmo_index_probed_codes(const mmo_index *ix, const double *q);
void mmo_index_list_sizes(const mmo_index *ix, int *sizes_out /* [C] or [1] */);

/* ---- PCA projection (J/dimreduction/PCA.java:188-208, load-time whitening :275-313) ---- */
/* Vt [nc][ss] row-major, ALREADY whitened if whitening (use mmo_pca_whiten). y[nc]. */
void mmo_pca_project(const double *Vt, const double *means, int nc, int ss, int whitening,
                     const double *x, double *y);
/* V_t <- W * V_t with W = diag(eig^-0.5), PCA.java:283-313 */
void mmo_pca_whiten(double *Vt, const double *eig, int nc, int ss);

/* ---- VLAD (J/aggregation/VladAggregator.java:56-70, AbstractFeatureAggregator.java:136-155,
 *            VladAggregatorMultipleVocabularies.java:84-101) ---- */
int mmo_nearest_centroid(const double *codebook, int nc, int dl, const double *desc);
void mmo_vlad_aggregate(const double *codebook, int nc, int dl, const double *descs, int ndesc,
                        double *vlad_out /* nc*dl */);
/* multi-vocabulary: codebooks concatenated, nc[i] centroids each, same dl */
void mmo_vlad_aggregate_multi(const double *codebooks, const int *nc, int nvocab, int dl,
                              const double *descs, int ndesc, int normalizations_on,
                              double *out);

#ifdef __cplusplus
}
#endif
#endif
