/*
 * mmidx.h -- C ABI of the MI355X-native PQ / IVFPQ search engine (libmmidx_hip.so).
 *
 * This is the drop-in boundary for the search path of MKLab-ITI/multimedia-indexing
 * (gr.iti.mklab.visual.datastructures.{PQ,IVFPQ}).  The reference is pure Java and has no FFI
 * layer; its seam is the Template-Method contract of AbstractSearchStructure (5 protected hooks,
 * AbstractSearchStructure.java:267,305,342,729,755).  Each entry point below states the
 * reference method(s) whose body it replaces.  Citations are relative to the reference root,
 * J/ = src/main/java/gr/iti/mklab/visual/.  INTEGRATION.md shows the JNI stub and the Java
 * subclasses (GpuIVFPQ / GpuPQ) that bind these symbols.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types.  Every call returns an int status
 *     (MMIDX_OK == 0); mmidx_last_error() returns the message for the calling thread.  Status
 *     codes map 1:1 onto the reference's exceptions (see the enum).  Native code never exits.
 *   - vectors are row-major IEEE binary64, exactly the Java double[] contents.
 *   - PQ codes cross the boundary in the reference's stored form: for numProductCentroids <= 256
 *     one int8 per sub-quantizer holding (index - 128) (PQ.transformToByte, J/datastructures/
 *     PQ.java:552-558); otherwise one int16 per sub-quantizer holding the index
 *     (PQ.transformToShort, PQ.java:544-550).
 *   - internal ids (iid) are int32, as in the reference (loadCounter, ASS:65).
 *   - "_device" variants take pointers into the HBM of the handle's GPU and run asynchronously on
 *     the given hipStream_t (passed as void*; NULL = the HIP default stream, which is also what
 *     PyTorch's default stream is).  Host variants stage through HBM on a private stream and
 *     are synchronous.
 *   - threading: adds are serialised per handle (indexVector / indexPQCode are `synchronized`,
 *     ASS:229, IVFPQ.java:357); searches may run concurrently with each other but not with adds
 *     (the reference offers no reader/writer exclusion either).
 *   - arithmetic: all distances are fp64, accumulated in the reference's left-to-right order
 *     without FMA contraction; returned ids are bit-exact w.r.t. the reference restatement and
 *     distances are bit-equal (tolerance promised to callers: 1e-5).
 */
#ifndef MMIDX_H
#define MMIDX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMIDX_ABI_VERSION 8

typedef struct mmidx_index mmidx_index; /* opaque handle: one index on one GPU */

enum mmidx_status {
    MMIDX_OK = 0,
    MMIDX_ERR_INVALID_SUBVECTORS = 1, /* "The given number of subvectors is not valid!"  IVFPQ.java:181-183, PQ.java:148-150 */
    MMIDX_ERR_WRONG_DIM = 2,          /* "The dimensionality of the vector is wrong!"    IVFPQ.java:310-312, PQ.java:233-235 */
    MMIDX_ERR_NOT_IN_MEMORY = 3,      /* "Cannot execute query because the index is not loaded in memory!" ASS:282-284 */
    MMIDX_ERR_BYTE_OVERFLOW = 4,      /* "Byte is not sufficient to enumerate the centroids of the product quantizer!" IVFPQ.java:358-361 */
    MMIDX_ERR_CAPACITY = 5,           /* "Maximum index capacity reached..." (return false) ASS:232-235 */
    MMIDX_ERR_INVALID_ARG = 6,        /* null pointer, k < 1 (LingPipe queue ctor), w outside 1..C (NPE at IVFPQ.java:597-599) */
    MMIDX_ERR_NOT_READY = 7,          /* quantizers not loaded (NullPointerException in the reference) */
    MMIDX_ERR_NO_DEVICE = 8,          /* no usable MI355X / HIP runtime: the product path has NO CPU fallback */
    MMIDX_ERR_HIP = 9,                /* HIP runtime error; message carries hipGetErrorString */
    MMIDX_ERR_UNSUPPORTED = 10        /* configuration outside the kernel envelope (see DESIGN.md limits) */
};

enum mmidx_kind { MMIDX_KIND_PQ = 1, MMIDX_KIND_IVFPQ = 2 };
/* PQ.TransformationType ordinals, J/datastructures/PQ.java:78-80 */
enum mmidx_transform { MMIDX_TR_NONE = 0, MMIDX_TR_ROTATION = 1, MMIDX_TR_PERMUTATION = 2 };

const char *mmidx_last_error(void);
int mmidx_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU / runtime) */
int mmidx_device_count(void);

/* ---- lifecycle ------------------------------------------------------------------------------
 * Replaces the in-memory half of the constructors IVFPQ.java:174-237 / PQ.java:142-174 (the
 * BDB half stays in Java).  D = vectorLength, m = numSubVectors, ks = numProductCentroids,
 * C = numCoarseCentroids (ignored for PQ).  transform PERMUTATION: perm = D int32 indices
 * (RandomPermutation, J/utilities/RandomPermutation.java:29-56; pass NULL to have the library
 * derive RandomPermutation(seed = 1, D) with the JDK LCG, IVFPQ.java:136,193).  transform
 * ROTATION: rot = D*D row-major doubles computed by the Java side with EJML
 * (J/utilities/RandomRotation.java:30-35; the EJML stream cannot be re-derived natively).
 * Default w = (int)(0.1 * C), IVFPQ.java:188. */
int mmidx_create(int kind, int D, int m, int ks, int C, int transform, const int32_t *perm,
                 const double *rot, int device, mmidx_index **out);
int mmidx_destroy(mmidx_index *h); /* closeInternal, IVFPQ.java:888-890 */

/* loadCoarseQuantizer IVFPQ.java:297-300 : coarse[C][D] */
int mmidx_set_coarse(mmidx_index *h, const double *coarse);
/* loadProductQuantizer IVFPQ.java:275-288 / PQ.java:210-223 : pq[m][ks][D/m], file order */
int mmidx_set_pq(mmidx_index *h, const double *pq);
/* setW IVFPQ.java:95-97 */
int mmidx_set_w(mmidx_index *h, int w);
int mmidx_get_w(const mmidx_index *h, int *w_out);
/* loadCounter (ASS:65): number of indexed codes */
int mmidx_size(const mmidx_index *h, int64_t *n_out);
/* outputItemsPerList IVFPQ.java:654-673 : sizes[C] (sizes[1] for PQ) */
int mmidx_list_sizes(mmidx_index *h, int32_t *sizes_out);

/* ---- indexing -------------------------------------------------------------------------------
 * mmidx_encode: the arithmetic of indexVectorInternal, IVFPQ.java:309-335 (+computeNearest-
 * CoarseIndex :547-564, computeResidualVector :642-648, computeNearestProductIndex :613-631) /
 * PQ.java:232-252, for n vectors X[n][D], WITHOUT appending.  cell_out[n] (-1 for PQ),
 * code_out[n][m] in stored form (int8 or int16, see above).  Lets the Java side persist the
 * record to BDB exactly as appendPersistentIndex does (IVFPQ.java:760-792). */
int mmidx_encode(mmidx_index *h, int64_t n, const double *X, int32_t *cell_out, void *code_out);
/* mmidx_add_vectors: indexVectorInternal for a batch: encode + append with the given iids
 * (iids == NULL: iid = current size + i, i.e. loadCounter semantics ASS:244-251).  cell_out /
 * code_out may be NULL. */
int mmidx_add_vectors(mmidx_index *h, int64_t n, const double *X, const int32_t *iids,
                      int32_t *cell_out, void *code_out);
/* mmidx_add_codes: append precomputed records -- serves indexPQCode (IVFPQ.java:357-386) and
 * loadIndexInMemory (IVFPQ.java:680-728, PQ.java:436-483).  cells may be NULL for PQ. */
int mmidx_add_codes(mmidx_index *h, int64_t n, const int32_t *iids, const int32_t *cells,
                    const void *codes);
/* device-resident bulk variants (X, iids, cells, codes in HBM). */
int mmidx_add_vectors_device(mmidx_index *h, int64_t n, const double *dX, const int32_t *d_iids,
                             int32_t iid0, void *stream);
int mmidx_add_codes_device(mmidx_index *h, int64_t n, const int32_t *d_iids,
                           const int32_t *d_cells, const void *d_codes, void *stream);
int mmidx_encode_device(mmidx_index *h, int64_t n, const double *dX, int32_t *d_cell_out,
                        void *d_code_out, void *stream);
/* build the device-side inverted-list layout now (otherwise done lazily by the next search) */
int mmidx_sync_index(mmidx_index *h);
/* snapshot of the in-memory index in list-major order -- the arrays loadIndexInMemory builds
 * (invertedLists / pqByteCodes, IVFPQ.java:680-728; the per-id getters getPQCodeByte :801,
 * getInvertedListId :865 read the same records from BDB): list_off_out[nlists+1],
 * iids_out[n], codes_out[n][m] in stored form.  iids_out / codes_out may be NULL. */
int mmidx_export(mmidx_index *h, int64_t *list_off_out, int32_t *iids_out, void *codes_out);
/* native flat snapshot for fast restart alongside BDB (ABI 8).  mmidx_save writes what mmidx_export returns -- header
 * (magic "MMIDXSN1", version, kind, D, m, ks, C, code width, transform, n), list_off[nlists + 1], iids[n], codes[n][m] in stored
 * form, little-endian -- to `path` (through path.tmp + rename).  mmidx_load fills an EMPTY handle of the same shape (quantizers set
 * as after the constructor) from such a file: the work of loadIndexInMemory (IVFPQ.java:680-728, PQ.java:436-483) without the
 * BDB cursor; every list keeps its arrival order.  A shape mismatch, a damaged file or a non-empty index is MMIDX_ERR_INVALID_ARG.
 * Ids (String <-> iid) stay where they are: in the Java side's BDB. */
int mmidx_save(mmidx_index *h, const char *path);
int mmidx_load(mmidx_index *h, const char *path);

/* ---- per-id utilities of IVFPQ / PQ ----------------------------------------------------------------
 * mmidx_get_dims: the constructor arguments of the handle (vectorLength, numSubVectors, numProductCentroids,
 *   numCoarseCentroids) and the bytes per stored code entry (1: byte codes, 2: short codes); any pointer may be NULL.
 *   (Lets a binding check array lengths before it hands them over.)
 * mmidx_get_codes: getInvertedListId (IVFPQ.java:865-880) and getPQCodeByte / getPQCodeShort (:801-855) for n internal
 *   ids: cell_out[n] (-1 for PQ) and code_out[n][m] in stored form (int8 or int16).  An id that was never indexed fails
 *   with MMIDX_ERR_INVALID_ARG, message "Id does not exist!" (IVFPQ.java:803-805).  Either output may be NULL.
 * mmidx_distance: computeDistanceIVFADC(double[] qVector, String existingVecId), IVFPQ.java:464-497, for n (query, id) pairs:
 *   residual of Q[i] w.r.t. the cell of iids[i] (:470), the handle's transformation (:473-477), then the sum over the
 *   sub-quantizers of the lookup-table entries the stored code selects (:482-495), in the reference's order -- bit-equal to
 *   the distance a search reports for that candidate.  For a PQ handle the query itself takes the place of the residual. */
int mmidx_get_dims(const mmidx_index *h, int *D, int *m, int *ks, int *C, int *code_bytes);
int mmidx_get_codes(mmidx_index *h, int64_t n, const int32_t *iids, int32_t *cell_out, void *code_out);
int mmidx_distance(mmidx_index *h, int64_t n, const double *Q, const int32_t *iids, double *dist_out);

/* ---- search ---------------------------------------------------------------------------------
 * computeNearestNeighborsInternal(k, double[]) : computeKnnIVFADC IVFPQ.java:408-450 /
 * computeKnnADC PQ.java:290-322, for nq queries Q[nq][D].  Row i of iid_out / dist_out holds
 * count_out[i] = min(k, #candidates) results, best first: ascending squared distance, equal
 * distances in the bounded queue's order (ASS.lookUp, ASS:345-358).  Unused tail entries are
 * iid -1 / +inf.  k must be in 1..4095 (the reference's queue has no bound, IVFPQ.java:409; here the per-block
 * candidate buffers share the 160 KiB LDS with the lookup table: larger k -> MMIDX_ERR_INVALID_ARG).
 * mmidx_search may be called from any number of threads (computeNearestNeighbors is not synchronized,
 * ASS:281-291): callers that arrive while a batch is running are served together as one device batch
 * (same k), each with the answer of its own call. */
int mmidx_search(mmidx_index *h, int k, int64_t nq, const double *Q, int32_t *iid_out,
                 double *dist_out, int32_t *count_out);
int mmidx_search_device(mmidx_index *h, int k, int64_t nq, const double *dQ, int32_t *d_iid_out,
                        double *d_dist_out, int32_t *d_count_out, void *stream);
/* computeNearestNeighborsInternal(k, int iid) of PQ: computeKnnSDC, PQ.java:334-374 -- the query is
 * the stored code of the vector with internal id iids[i]; distances are code-to-code (one sequential
 * chain over all D dimensions).  PQ with byte codes only: IVFPQ.computeKnnIVFSDC returns null in the
 * reference (IVFPQ.java:509-511) -> MMIDX_ERR_UNSUPPORTED. */
int mmidx_search_sdc(mmidx_index *h, int k, int64_t nq, const int32_t *iids, int32_t *iid_out,
                     double *dist_out, int32_t *count_out);

/* computeNearestCoarseIndex (IVFPQ.java:547-564) for n device-resident vectors: d_cell_out[i] = the exact
 * fp64 argmin cell, first index wins ties.  Needs only the coarse quantizer (used by the codebook learner). */
int mmidx_assign_device(mmidx_index *h, int64_t n, const double *dX, int32_t *d_cell_out, void *stream);

/* ---- sharded search, building blocks (one process per GPU over torch.distributed / MPI; the single-process form is
 * mmidx_create_sharded below, which drives these same phases itself) ---------------------
 * mmidx_coarse_device: computeNearestCoarseIndices IVFPQ.java:575-601 for nq queries ->
 *   d_cells_out[nq][w] (nearest first) and, when d_cdist_out is not NULL, the exact squared distance
 *   of every selected cell, d_cdist_out[nq][w] (what pass B's coarse bound needs: ranks that did not
 *   run the coarse stage for a query receive it with the all-gather of the cells).
 * mmidx_search_partial_device: scan of this shard's lists for the given probe cells; writes the
 *   shard's best k+1 candidates per query, sorted: d_pdist[nq][k+1] (fp64), d_pkey[nq][k+1]
 *   (int64 = probe_rank << 32 | iid -- the reference's offer order), d_pcount[nq].
 * mmidx_merge_partials_device: merges nshards partial lists into final results; no index handle
 *   needed.  Layout: dense [nshards][nq][k+1] (d_poff NULL), or ragged -- only the d_pcount[s][q] valid
 *   entries of every list, concatenated, list (s, q) starting at element d_poff[s * nq + q] (what a
 *   variable-size all-to-all delivers). */
int mmidx_coarse_device(mmidx_index *h, int64_t nq, const double *dQ, int32_t *d_cells_out,
                        double *d_cdist_out, void *stream);
int mmidx_search_partial_device(mmidx_index *h, int k, int64_t nq, const double *dQ,
                                const int32_t *d_cells, double *d_pdist, int64_t *d_pkey,
                                int32_t *d_pcount, void *stream);
/* Two-phase form of the partial search, so that shards can share their thresholds:
 *   pass A scans probe rank 0 of every query on this shard and exports the shard's threshold per
 *   query (the (k+1)-th best distance so far as a double, +inf when there are fewer candidates);
 *   the host MIN-all-reduces that array over ranks; pass B imports it, drops every probe whose
 *   coarse bound exceeds it, scans the rest and writes the sorted partial lists.  pass B must follow
 *   pass A on the same handle with the same (k, nq, dQ, d_cells).  d_cdist (may be NULL: the bound
 *   is then evaluated from the centroids) = the distances mmidx_coarse_device delivered. */
int mmidx_shard_pass_a_device(mmidx_index *h, int k, int64_t nq, const double *dQ,
                              const int32_t *d_cells, double *d_T_out, void *stream);
int mmidx_shard_pass_b_device(mmidx_index *h, int k, int64_t nq, const double *dQ,
                              const int32_t *d_cells, const double *d_cdist, const double *d_T_in,
                              double *d_pdist, int64_t *d_pkey, int32_t *d_pcount, void *stream);
/* dense partial lists [nq][k+1] -> ragged (list q = its d_pcount[q] valid entries at element d_poff[q]):
 * what a rank sends through the variable-size all-to-all */
int mmidx_compact_partials_device(int device, int k, int64_t nq, const double *d_pdist,
                                  const int64_t *d_pkey, const int32_t *d_pcount, const int64_t *d_poff,
                                  double *d_out_dist, int64_t *d_out_key, void *stream);
int mmidx_merge_partials_device(int device, int k, int64_t nq, int nshards, const double *d_pdist,
                                const int64_t *d_pkey, const int32_t *d_pcount, const int64_t *d_poff,
                                int32_t *d_iid_out, double *d_dist_out, int32_t *d_count_out,
                                int32_t *d_flag_out, void *stream);
/* Straddling ties across shards (ABI version 4).  mmidx_merge_partials_device sets d_flag_out[q] = 1 (d_flag_out may be NULL)
 * when the k-th and (k+1)-th merged distances are equal: which of the equal candidates the reference's single bounded queue
 * (IVFPQ.java:409, :445) keeps depends on the offer order over ALL probed lists, which no rank sees alone.  The owner collects
 * the flagged queries (d_fq[nf] = query index into dQ / d_cells, -1 = unused slot; d_tau[nf] = the k-th distance) and every
 * rank runs three passes over its own lists with a reduction over ranks in between:
 *   phase 0  d_counts[nf][w][2] (zeroed by the caller) <- per local list: offers with d <= tau, offers with d == tau;  SUM all-reduce
 *   phase 1  d_pB[nf] (zeroed) <- ties among the first offers of the list that holds the k-th "d <= tau" offer;          SUM all-reduce
 *   phase 2  d_tie_iids[nf][k] (filled with -1) <- iids of this rank's kept ties at their answer slots;                 MAX all-reduce
 * after which the owner overwrites answer slot s of a flagged query with d_tie_iids[f][s] wherever that is >= 0.  The result is
 * the single queue's (the same closed form k_tie_resolve applies on one GPU).  Within a list, entries are replayed in list
 * position = arrival order, as the reference appends them. */
int mmidx_shard_tie_phase_device(mmidx_index *h, int phase, int k, int64_t nf, const double *dQ, const int32_t *d_cells,
                                 const int32_t *d_fq, const double *d_tau, int32_t *d_counts, int32_t *d_pB,
                                 int32_t *d_tie_iids, void *stream);

/* ---- one index over several GPUs, ONE process (ABI version 5) --------------------------------------------------------------
 * The reference's caller is a single JVM that holds the whole index and queries it through one object
 * (YFCC100MExample.java:93-99, :155; Example.java:96-110).  mmidx_create_sharded makes that object span n_dev GPUs of one node:
 * whole inverted lists are partitioned (list c lives on shard c mod n_dev -- invertedLists[c] / pqByteCodes[c], IVFPQ.java:72-83,
 * stay intact, so the offer order inside a list is the reference's), the codebooks are replicated, and every entry point that
 * takes host pointers works on the returned handle exactly as on a plain one:
 *   mmidx_set_coarse / mmidx_set_pq / mmidx_set_w / mmidx_set_option / mmidx_set_profiling   applied to every shard
 *   mmidx_encode / mmidx_add_vectors     the batch is encoded 1/n_dev per GPU, each record goes to the shard that owns its list
 *   mmidx_add_codes                      indexPQCode / loadIndexInMemory: records routed by list id
 *   mmidx_search                         computeKnnIVFADC over all shards (below); concurrent callers are combined as usual
 *   mmidx_size / mmidx_list_sizes / mmidx_export / mmidx_get_codes / mmidx_distance / mmidx_get_stats / mmidx_sync_index
 * Search, per round of queries: shard r owns 1/n_dev of the queries; RCCL all-gather of the query vectors and of the probe cells
 * (each shard runs the coarse stage for its own queries), pass A on every shard's local lists, RCCL all-reduce (MIN) of the
 * thresholds, pass B under the global thresholds, every sorted partial top-(k+1) list stored straight into its owner's buffers
 * over xGMI (peer access; option "shard_exchange" = 1: ncclSend / ncclRecv of the dense lists instead), merge at the owner,
 * cross-shard replay of the bounded queue for queries whose k-th and (k+1)-th distances tie (RCCL all-reduces).  Results are
 * those of a plain handle holding the same records: ids bit-exact, distances bit-equal, ties included.
 * devs[i] = HIP device of shard i.  Pairwise distinct devices (the production form, n_dev = 1 included) use RCCL
 * (ncclCommInitAll, one communicator per device, one host thread per device); a device listed more than once makes virtual
 * shards on it with in-process collectives (RCCL refuses duplicate devices): the functional test form on a one-GPU box.
 * The _device entry points of a plain handle take ONE device's pointers and are refused on a sharded handle
 * (MMIDX_ERR_UNSUPPORTED); the _sliced_device forms below are their counterparts: slice r lives in the HBM of shard r's device.
 *   mmidx_search_sliced_device   shard r hands in nq_per_shard queries dQ[r][nq_per_shard][D] and receives their answers in
 *                                d_iid_out[r] / d_dist_out[r] / d_count_out[r] (synchronous: the answers are complete on return)
 *   mmidx_add_vectors_sliced_device   the batch is the concatenation of the slices dX[r][n_per_shard[r]][D]; row i of it gets
 *                                iid0 + i (arrival order = batch order, as indexVector calls in that order would give)
 *   mmidx_shard_count / mmidx_shard_info   number of shards (1 for a plain handle); a shard's device, its number of records,
 *                                and whether the group's collectives run on RCCL */
int mmidx_create_sharded(int kind, int D, int m, int ks, int C, int transform, const int32_t *perm, const double *rot, int n_dev,
                         const int *devs, mmidx_index **out);
int mmidx_shard_count(const mmidx_index *h, int *n_out);
int mmidx_shard_info(const mmidx_index *h, int shard, int *device_out, int64_t *size_out, int *uses_rccl_out);
int mmidx_search_sliced_device(mmidx_index *h, int k, int64_t nq_per_shard, const double *const *dQ, int32_t *const *d_iid_out,
                               double *const *d_dist_out, int32_t *const *d_count_out);
int mmidx_add_vectors_sliced_device(mmidx_index *h, const int64_t *n_per_shard, const double *const *dX, int32_t iid0);

/* ---- Linear: exhaustive exact search (J/datastructures/Linear.java; BASELINE config 1) ------------------------
 * add = indexVectorInternal (:111-122), search = computeNearestNeighborsInternal(k, double[]) (:138-163): exact
 * sequential fp64 squared distances, bounded-queue order and ties, ids = insertion order; get_vector = getVector
 * (:253-263).  The vectors go through the coarse-stage kernels as if they were centroids (same arithmetic, same
 * queue rule). */
typedef struct mmidx_linear mmidx_linear;
int mmidx_linear_create(int D, int64_t capacity, int device, mmidx_linear **out);
int mmidx_linear_destroy(mmidx_linear *l);
int mmidx_linear_add(mmidx_linear *l, int64_t n, const double *X);
int mmidx_linear_size(const mmidx_linear *l, int64_t *n_out);
int mmidx_linear_get_dim(const mmidx_linear *l, int *D_out); /* vectorLength of the handle (bindings check array lengths against it) */
int mmidx_linear_get_vector(const mmidx_linear *l, int64_t iid, double *out);
int mmidx_linear_search(mmidx_linear *l, int k, int64_t nq, const double *Q, int32_t *iid_out, double *dist_out,
                        int32_t *count_out);

/* ---- codebook learning (J/visual/quantization/AbstractQuantizerLearning.java:39-81) ------
 * k-means on the GPU in place of Weka's SimpleKMeans, which the reference calls with setSeed(seed),
 * setNumClusters(k), setMaxIterations(max_iter) and, optionally, k-means++ seeding.  flags:
 *   MMIDX_KMEANS_PLUS_PLUS  k-means++ seeding (else SimpleKMeans' default random seeding),
 *   MMIDX_KMEANS_NORMALIZE  min-max attribute normalisation inside the distance (Weka's default).
 * init_centroids (host, [k][d], may be NULL) overrides the seeding.  Empty clusters are dropped as Weka does:
 * *k_out <= k centroids are written to centroids_out (host, room for [k][d]); assign_out[i] indexes them.
 * sse = squared error in the space the clustering ran in.  Weka is an absent third-party dependency: the
 * algorithm is restated, its random stream and summation order are not reproduced (parity unpinned). */
enum mmidx_kmeans_flags { MMIDX_KMEANS_PLUS_PLUS = 1, MMIDX_KMEANS_NORMALIZE = 2 };
int mmidx_kmeans_device(int device, int64_t n, int d, int k, int max_iter, int64_t seed, int flags,
                        const double *dX, const double *init_centroids, double *centroids_out,
                        int32_t *d_assign_out, double *sse_out, int32_t *iters_out, int32_t *k_out,
                        void *stream);
int mmidx_kmeans(int device, int64_t n, int d, int k, int max_iter, int64_t seed, int flags,
                 const double *X, const double *init_centroids, double *centroids_out,
                 int32_t *assign_out, double *sse_out, int32_t *iters_out, int32_t *k_out);

/* ---- instrumentation -------------------------------------------------------------------------
 * Per-handle statistics accumulated over the search calls since the last mmidx_get_stats, measured
 * with HIP events recorded on the stream the kernels are launched on (no host synchronisation
 * until mmidx_get_stats).  scan_ms = time inside the list-scan kernel (the HBM-bound
 * kernel of the path), scan_codes = sum over (query, probed list) of list lengths (algorithmic
 * bytes = m * scan_codes), launches = number of scan launches. */
typedef struct mmidx_stats {
    double total_ms, coarse_ms, scan_ms, merge_ms;
    int64_t scan_codes;
    int32_t scan_launches;
    int32_t tie_fallbacks;
    /* pass A alone (the dominant kernel: the exact scan of every query's nearest list): time between the events around
     * its launches, the codes of the probe-rank-0 lists, the number of calls that ran it (ABI version 3) */
    double passa_ms;
    int64_t passa_codes;
    int32_t passa_launches;
    /* (query, probe) pairs of the most recent search call that survived the coarse bound and went through pass B
     * (-1: unknown); read from the pinned word the device writes for the launch-size hint */
    int32_t passb_items_last;
    /* codes of pass B that survived the lower-bound filter and had their exact distance computed (ABI version 4) */
    int64_t verified_codes;
    /* K3m, the matrix-core lower bound of pass B (csrc/mmidx_scan_mfma.h; ABI version 6): (query, code) pairs its bound could not
     * drop (each verified exactly: they are part of verified_codes), and queries it handed back to K3f (no finite threshold yet, a
     * full survivor list or pool) */
    int64_t mfma_survivors;
    int64_t mfma_redo_queries;
    /* ... and, with full profiling, the launch durations of its scan kernel (k_scan_mfma) and of the exact verification
     * (k_mfma_verify) from HIP events around them, summed over mfma_launches calls */
    double mfma_scan_ms, mfma_verify_ms;
    int32_t mfma_launches;
    int32_t reserved0;
    /* K3ma, pass A on the matrix cores (csrc/mmidx_scan_mfma_a.h; ABI version 7): calls whose pass A went through it, and with full
     * profiling the durations of its four stages (HIP events): sweep 1 (k_scan_mfma<.., 1>), threshold selection (k_a1_select),
     * sweep 2 (k_scan_mfma<.., 2>) and the exact verification (k_a1_verify), summed over those calls.  Its verified codes count in
     * verified_codes / mfma_survivors, the queries it handed to the exact kernels in mfma_redo_queries. */
    int64_t passa_mfma_launches;
    double passa_mfma_sweep1_ms, passa_mfma_select_ms, passa_mfma_sweep2_ms, passa_mfma_verify_ms;
} mmidx_stats;
/* enabled: 0 off; 1 full (six events per search call and the code counters: every field below); 2 light (only the two
 * events around pass A: passa_ms / passa_launches -- an event record is a ~5 us bubble in the stream, so a throughput
 * run carries as few as its roofline figure needs) */
int mmidx_set_profiling(mmidx_index *h, int enabled);
/* measurement switches; results are identical in every setting.  "exhaustive" = 1: every probed
 * code is read and summed in fp64 (no lower-bound filter, no coarse-bound probe pruning) -- the
 * configuration the HBM roofline of the scan kernel is quoted on; "no_filter", "no_bound",
 * "exact_coarse" switch the individual devices (DESIGN.md sections 5.2, 5.4, 5.5); "combine" = 0:
 * concurrent mmidx_search callers are served one at a time instead of together (section 5.11).
 * A/B and test switches: "coarse_v1" (K1c/K1d instead of K1e/K1f), "coarse_fused" (K1f as one kernel instead of
 * front end + selection), "coarse_nodma" (K1e with register staging
 * instead of the LDS-DMA kernel), "passa_hist" (1 / 0 / -1: K3h always / never / for long lists), "passa_wide" (K3h with
 * 512-thread blocks), "passa_prefix", "no_grp" (pass B through K3f only), "grp_blocks",
 * "passb_main_grid", "no_item_compaction", "passa_item_min", "passa_item_margin" (a shard's pass-A item list,
 * section 6), "smin_pre" (K3s, the certified Smin of every far pair in front of pass B's sort, section 5.15: 1 always, 0 never,
 * -1 = default: by the figures the device reported for the call before), "smin_valu" (1: K3s with packed VALU FMAs instead of
 * the matrix cores), "smin_bf16" (0: without K3s's bf16 first stage), "coarse_dma_kc" (0: long vectors through K1e's register staging instead of the
 * LDS-DMA kernel), "no_union" (K3g's instances that rank the union of the verified candidates: -1 always, 1 never, 0 hint).  On a sharded handle: "shard_exchange" (0: partial lists stored into the owners' buffers over xGMI, 1: ncclSend /
 * ncclRecv), "tie_slots" (flagged queries per owner and replay round, 0 = no replay), "shard_max_round"; every other option goes
 * to every shard.  Round 5: "no_split_table" (1: an exact table of twice the LDS -- m = 128 byte codes -- stays in global scratch instead
 * of being taken in two sweeps with half of it in LDS, k_scan_split).  Round 4: "no_mfma" (1: pass B through K3g / K3f instead of the matrix-core bound K3m / K3mk), "mfma_sub" (codes per
 * work item of K3m / K3mk, 0 = sized from the call), "mfma_qcap" (survivor records per launch; a small value sends queries through the
 * redo path), "mfma_blocks", "mfma_kc_v1" (1: K3mk without LDS-DMA, k_scan_mfma_kc, also where k_scan_mfma_kc2 applies),
 * "mfma_kc_tpw" (8 / 16 code tiles per wave of k_scan_mfma_kc), "lut_pre" (pass A's tables built by their own kernel),
 * "coarse_wave_sel" (0: the coarse stage's exact selection always by a block per query instead of k_coarse_front_sel),
 * "passb_small" (0: pass B through K3m / K3g also when the call before kept at most 64 pairs; default 1: K3f's looping kernel alone,
 * one launch), "passa_mfma" (round 5, K3ma: pass A through the matrix-core bound in two sweeps -- 1 wherever the shape allows, 0 never,
 * -1 = default: from 8 queries per list of a long-list index),
 * "shard_route_host" (round 5; 1: mmidx_add_vectors_sliced_device sends its records through the host as before round 5 instead of routing
 * them between the devices),
 * "shard_pipeline" (1: the query exchange of a sharded handle on a second stream / communicator; the default for in-process shards, off
 * by default on two or more physical devices until a multi-device run has passed).  Round 6: "passa_q" (K3q, pass A decided on packed
 * integer table sums with four queries of a nearest list per block, mmidx_scan_q.h: 1 wherever the shape allows -- IVFPQ, byte codes,
 * ks = 256, m = 16, dsub in {4, 8, 16}, k <= 151 --, 0 never, -1 = default: from 1.25 queries per non-empty list of a long-list index),
 * "host_slots" (default 1: host-pointer searches of more than 4096 queries from several threads take one of three slots -- own copy
 * stream and buffers -- and hold the handle's lock only while their kernels are enqueued; 0: one request at a time).
 * None of them changes a result. */
int mmidx_set_option(mmidx_index *h, const char *name, int value);
int mmidx_get_stats(mmidx_index *h, mmidx_stats *out);
/* Which kernel family served each stage of the most recent search sub-batch of this handle (ABI version 7), as text:
 * "coarse=<K1..>;pass_a=<K3 | K3h | K3ma | ..>;pre=<K3s | ->;pass_b=<K3m | K3mk | K3g | K3f.. | ->" (DESIGN.md section 5.0 lists the gates;
 * tests/test_gpu_dispatch.py asserts the table).  A sharded handle reports its first shard.  Diagnostic: not a stable format. */
int mmidx_get_dispatch(mmidx_index *h, char *out, int cap);

/* Measured ceilings for the rooflines bench.py reports (csrc/mmidx_probe.hip; SURVEY 8d "secondary: LDS gather rate", A5):
 * mmidx_probe_lds_gather: the gather of the exact scan alone -- per code m random 8-byte LDS reads over 2 KiB rows, summed
 *   in fp64, `chains` (1 or 3) codes in flight per lane, 256-thread blocks at the scan kernel's occupancy; m in {8, 16, 32}.
 *   out[0] = wave-level ds_read_b64 per second, out[1] = GB/s of "algorithmic bytes" (codes x m), out[2] = blocks per CU,
 *   out[3] = seconds.
 * mmidx_probe_split_gather: the m = 16 loop with the last kg (1, 2 or 4) rows gathered through the vector-memory path (a
 *   per-block copy in global memory, L1-resident) and the others from the LDS -- the experiment behind DESIGN section 9's
 *   "share the gather between the two pipes"; out as mmidx_probe_lds_gather.
 * mmidx_probe_f64_mfma: back-to-back v_mfma_f64_16x16x4_f64 on every SIMD; out[0] = TFLOP/s, out[1] = seconds. */
int mmidx_probe_lds_gather(int device, int m, int chains, double *out);
int mmidx_probe_split_gather(int device, int kg, double *out);
int mmidx_probe_f64_mfma(int device, double *out);

/* ---- front end of BASELINE config 5 ----------------------------------------------------------
 * PCA projection: PCA.loadPCAFromFile (J/dimreduction/PCA.java:257-318) + sampleToEigenSpace
 * (:188-208).  means[ss] = line 1 of the PCA file, eig[nc] = line 2 (used iff whitening: the
 * library folds W = diag(eig^-0.5) into V_t exactly as :283-313 does), Vt[nc][ss] = the component
 * lines.  project: Y[n][nc] = V_t (X[n] - means), L2-normalised iff whitening (:203-204).  This is
 * the one dense contraction of the path and runs on the f64 matrix cores; results agree with the
 * reference to 1e-12 (relative to the row norm), not bit-for-bit (EJML summation order, A2). */
typedef struct mmidx_pca mmidx_pca;
int mmidx_pca_create(int nc, int ss, int whitening, const double *means, const double *eig,
                     const double *Vt, int device, mmidx_pca **out);
int mmidx_pca_destroy(mmidx_pca *p);
int mmidx_pca_get_dims(const mmidx_pca *p, int *nc_out, int *ss_out); /* numComponents, sampleSize of the handle */
int mmidx_pca_project(mmidx_pca *p, int64_t n, const double *X, double *Y);
int mmidx_pca_project_device(mmidx_pca *p, int64_t n, const double *dX, double *dY, void *stream);

/* VLAD aggregation: VladAggregator.aggregateInternal (J/aggregation/VladAggregator.java:56-70) with
 * computeNearestCentroid (AbstractFeatureAggregator.java:136-155), and the multi-vocabulary wrapper
 * with power + L2 normalisation (VladAggregatorMultipleVocabularies.java:84-101).  codebooks =
 * the nvocab codebooks concatenated, ncent[i] centroids of dl doubles each.  desc_off[nimg+1]
 * delimits each image's descriptors in descs[total][dl].  out[nimg][sum ncent*dl].  Without
 * normalisation the output is bit-exact (accumulation in descriptor order); with it, 1e-12. */
typedef struct mmidx_vlad mmidx_vlad;
int mmidx_vlad_create(int nvocab, const int32_t *ncent, int dl, const double *codebooks,
                      int normalizations_on, int device, mmidx_vlad **out);
/* "exact" = 1: the one-kernel form (fp64 brute-force nearest centroid inside the image's block) instead of the default -- the
 * nearest centroid of every descriptor of the call by the encoder's certified bf16-MFMA argmin (identical result: first index wins,
 * AFA:136-155; flagged descriptors redone in fp64), then the ordered accumulation.  A/B and test switch (ABI version 6).
 * "two_pass" = 1 (ABI version 7): assignment and accumulation as two kernels (K8') also where the default one-kernel form K8''
 * (k_vlad_fused: 64-dimensional descriptors, vocabularies of <= 128 centroids -- one pass over the descriptors, no host
 * synchronisation inside mmidx_vlad_aggregate_device) applies. */
int mmidx_vlad_set_option(mmidx_vlad *v, const char *name, int value);
int mmidx_vlad_destroy(mmidx_vlad *v);
int mmidx_vlad_vector_length(const mmidx_vlad *v, int *len_out);
int mmidx_vlad_descriptor_length(const mmidx_vlad *v, int *dl_out);
int mmidx_vlad_aggregate(mmidx_vlad *v, int64_t nimg, const int64_t *desc_off, const double *descs,
                         double *out);
int mmidx_vlad_aggregate_device(mmidx_vlad *v, int64_t nimg, const int64_t *d_desc_off,
                                const double *d_descs, int max_desc, double *d_out, void *stream);

/* ImageVectorization.transformToVector (J/vectorization/ImageVectorization.java:169-208) for a batch of images:
 * aggregate (VLAD) then PCA.sampleToEigenSpace in one call; the VLAD vectors stay on the device.  out[nimg][nc]. */
int mmidx_vectorize(mmidx_vlad *v, mmidx_pca *p, int64_t nimg, const int64_t *desc_off, const double *descs,
                    double *out);
int mmidx_vectorize_device(mmidx_vlad *v, mmidx_pca *p, int64_t nimg, const int64_t *d_desc_off,
                           const double *d_descs, int max_desc, double *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
