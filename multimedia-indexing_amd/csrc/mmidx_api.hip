// mmidx_api.hip -- host side of libmmidx_hip.so: the C ABI of include/mmidx.h.
//
// One handle = one PQ / IVFPQ index resident in the HBM of one MI355X.  There is no CPU
// fallback: without a HIP device every entry point fails with MMIDX_ERR_NO_DEVICE.
//
// HBM layout (per handle)
//   coarse   [C][D]  f64   + coarseT [D][C]        (transposed copy: coalesced per-centroid reads)
//   pq       [m][ks][dsub] + pqT [m][dsub][ks]     (transposed copy: coalesced per-entry reads)
//   list_off [nlists+1] i64, codes [n][m] u8|u16 (list-major, arrival order inside a list),
//   ids      [n] i32                               (CSR form of invertedLists / pqByteCodes,
//                                                   IVFPQ.java:72-83)
//   pending  (cells, ids, codes) of records added since the last CSR build
#include "../../include/mmidx.h"
#include "mmidx_kernels.h"
#include "mmidx_scan_grp.h"
#include "mmidx_scan_mfma.h"
#include "mmidx_scan_mfma_kc.h"
#include "mmidx_scan_mfma_a.h"
#include "mmidx_scan_q.h"
#include "mmidx_frontend.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <limits>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCK(expr)                                                                          \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(MMIDX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// ---- java.util.Random + Collections.shuffle (RandomPermutation.java:29-40) --------------------
struct JRandom {
    uint64_t s;
    explicit JRandom(int64_t seed) : s(((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next(int bits) {
        s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)((int64_t)s >> (48 - bits));
    }
    int32_t nextInt(int32_t bound) {
        if ((bound & (-bound)) == bound) return (int32_t)(((int64_t)bound * (int64_t)next(31)) >> 31);
        int32_t bits, val;
        do {
            bits = next(31);
            val = bits % bound;
        } while ((int32_t)((uint32_t)bits - (uint32_t)val + (uint32_t)(bound - 1)) < 0);
        return val;
    }
};
void jdk_random_permutation(int64_t seed, int dim, int32_t *perm) {
    JRandom r(seed);
    for (int i = 0; i < dim; i++) perm[i] = i;
    for (int i = dim; i > 1; i--) std::swap(perm[i - 1], perm[r.nextInt(i)]);
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// one host-pointer search call waiting to be served (see mmidx_search)
struct SearchReq {
    int k;
    int64_t nq;
    const double *Q;
    int32_t *iid;
    double *dist;
    int32_t *cnt;
    int rc = MMIDX_OK;
    bool done = false;
    std::string err;
    std::condition_variable cv;  // its caller sleeps here: woken when served, or when it is the oldest and nobody leads
};

// Concurrent callers of a host-pointer search are combined into one device batch (the reference's API is one query
// per call, from many reader threads): requests queue here, one caller at a time leads and serves the queue.
struct Combiner {
    std::mutex mu;
    std::deque<SearchReq *> q;
    bool busy = false;
    int enabled = 1;  // 0 = every call runs on its own
};

// Queues `me`, leads batches (serve(requests, count) -> status) until `me` has been served, returns its status.
template <class Serve>
int combiner_submit(Combiner &c, SearchReq &me, int64_t max_q, Serve serve) {
    std::unique_lock<std::mutex> lk(c.mu);
    c.q.push_back(&me);
    while (!me.done) {
        if (c.busy) {
            me.cv.wait(lk);
            continue;
        }
        // lead: the oldest request and everything behind it with the same k, up to the batch limit
        c.busy = true;
        std::vector<SearchReq *> batch;
        {
            SearchReq *first = c.q.front();
            c.q.pop_front();
            batch.push_back(first);
            int64_t tot = first->nq;
            if (c.enabled && tot <= max_q) {
                for (auto it = c.q.begin(); it != c.q.end();) {
                    if ((*it)->k == first->k && tot + (*it)->nq <= max_q) {
                        tot += (*it)->nq;
                        batch.push_back(*it);
                        it = c.q.erase(it);
                    } else {
                        ++it;
                    }
                }
            }
        }
        lk.unlock();
        const int brc = serve(batch.data(), batch.size());
        lk.lock();
        for (SearchReq *r : batch) {
            r->rc = brc;
            if (brc && r != &me) r->err = g_err;  // the message lives in the leader's thread
            r->done = true;
            if (r != &me) r->cv.notify_one();
        }
        c.busy = false;
        // hand the lead to the oldest waiting caller (only that thread is woken); if this call is still unserved
        // -- the batch was another k's -- it leads again itself
        if (me.done && !c.q.empty()) c.q.front()->cv.notify_one();
    }
    lk.unlock();
    if (me.rc && !me.err.empty()) g_err = me.err;
    return me.rc;
}

}  // namespace

struct ShardGroup;  // mmidx_sharded.h

struct mmidx_index {
    // a handle made by mmidx_create_sharded is a parent: it owns no device memory itself, `grp` holds one sub-index per shard
    ShardGroup *grp = nullptr;
    // a sub-index of such a group: where pass B's partial lists go (MergeParams::dest), set per search round by the group
    const ShardDest *shard_dest = nullptr;
    int shard_dest_per = 0, shard_dest_me = 0;
    int kind = 0, D = 0, m = 0, ks = 0, dsub = 0, C = 0, transform = 0, w = 0, device = 0;
    int nlists = 1;
    size_t code_bytes = 1;  // per sub-quantizer
    bool coarse_set = false, pq_set = false;
    bool no_filter = false;  // MMIDX_NO_FILTER=1: exact scan only (A/B switch for measurements)
    bool no_bound = false;   // MMIDX_NO_BOUND=1: no coarse-bound pruning of probes
    bool debug_sync = false; // MMIDX_DEBUG_SYNC=1
    bool passa_512 = false;  // MMIDX_PASSA_512=1: pass A with 512-thread blocks
    int32_t *pin_hint = nullptr;  // pinned host word: pass B's item count of the previous call (launch sizing hint)
    int passb_main_grid = 0;      // > 0: fixed size of pass B's main launch (tests: force the looping tail kernel)
    int passa_hist = -1;     // MMIDX_PASSA_HIST: 1 = always use K3h in pass A, 0 = never, -1 = lists of >= 4096 codes on average
    int passa_wide = 0;      // option "passa_wide" / MMIDX_PASSA_WIDE=1: K3h with 512-thread blocks
    int passa_prefix = 0;    // MMIDX_PASSA_PREFIX=n: pass A scans n codes exactly, the rest of the list filtered
    bool passa_su2 = false;  // MMIDX_PASSA_SU2=1: pass A with 2 codes per thread per segment (A/B switch)
    bool passa_filter = false;  // MMIDX_PASSA_FILTER=1
    bool no_seed = true;        // MMIDX_SEED=1: pass A with the seeded scan K3s (measured slower than K3: 1.45 vs 1.22 ms
                                // per 8192 queries -- one block per query is latency-bound, not LDS-bound; kept for study)
    double rmax = 0.0;       // sqrt(sum_s max_j ||pq[s][j]||^2) * (1 + 1e-12)
    double rot_shrink = 0.0; // RandomRotation: |v R| >= rot_shrink |v| for every v (0: the matrix is too far from orthogonal to say)
    // K3g (grouped pass B, mmidx_scan_grp.h): fp32 copy of the codebook (entry index innermost), ||p||^2 and their maxima
    float *d_pq32T = nullptr, *d_pn32 = nullptr;
    double *d_pnmax = nullptr;
    bool grp_valid = false;  // tables match the current quantizers
    int no_grp = 0;          // option "no_grp" = 1: pass B through K3f only (A/B switch)
    int grp_blocks = 0;      // option "grp_blocks": persistent blocks of K3g (0 = occupancy x CUs)
    int num_cus = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // every search / encode entry point (host or device pointers) shares the handle's workspaces: calls are serialised on the
    // host by this mutex, and on the device by stream order -- a call arriving on another stream than the previous one first
    // waits for that stream (DeviceCall below).  Recursive: the host entry points call the device ones.
    std::recursive_mutex search_mu;
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    std::recursive_mutex add_mu;  // adds are `synchronized` in the reference (ASS:229, IVFPQ.java:357): one at a time
    Combiner comb;  // concurrent mmidx_search callers are served together (see mmidx_search)
    unsigned char *pin_stage = nullptr;  // pinned staging: queries in, (distances | ids | counts) out
    // large host-pointer requests (> MMIDX_COMB_MAX_Q queries): a few callers in flight at once, each with its own copy stream and
    // device buffers -- one caller's queries and answers are on the bus while another's kernels run (see search_host_big)
    struct HostSlot {
        hipStream_t st = nullptr;
        hipEvent_t e_in = nullptr, e_done = nullptr;
        DevBuf<double> dQ;
        DevBuf<unsigned char> out;
        bool busy = false;
    };
    static constexpr int N_HOST_SLOTS = 3;
    int host_slots_on = 1;  // option "host_slots": 0 = large requests through the combiner's queue, one at a time (rounds 1-5)
    HostSlot host_slot[N_HOST_SLOTS];
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    size_t pin_stage_cap = 0;

    double *d_coarse = nullptr, *d_coarseT = nullptr, *d_pq = nullptr, *d_pqT = nullptr,
           *d_rot = nullptr, *d_cn = nullptr, *d_cnorm = nullptr;
    float *d_coarseT32 = nullptr;
    // K1e/K1f (bf16-split coarse dot products + group minima): padded bf16 head/tail copies of the centroids
    unsigned short *d_Ch = nullptr, *d_Cl = nullptr;
    double *d_cn_pad = nullptr;  // [Cp] |c|^2, +inf on the padding rows
    int Cp = 0, Dp = 0;          // C rounded up to 128, D rounded up to 32
    bool coarse_v1 = false;      // MMIDX_COARSE_V1=1: K1c/K1d (fp32 MFMA, full d~ matrix) instead
    bool coarse_fused = false;   // option "coarse_fused": K1f as one kernel (front end + selection), as in round 1 (A/B switch)
    int coarse_dma_kc = 1;       // option "coarse_dma_kc": K1e' with LDS-DMA also for vectors of several k chunks (Dp a multiple of 128)
    bool coarse_nodma = false;   // option "coarse_nodma": K1e with register staging also when Dp == 128 (A/B switch)
    bool no_item_compaction = false;  // option "no_item_compaction": a shard's pass A over every query (A/B switch)
    int passa_item_min = 4096;        // option "passa_item_min": fewest queries per call for that compaction
    int passa_item_margin = 1024;     // option "passa_item_margin": blocks launched beyond 1.15 x the expected count (tests: 0)
    double cn_max = 0.0, cnorm_max = 0.0;
    bool exact_coarse = false;  // MMIDX_EXACT_COARSE=1: fp64 distances to every centroid (K1a/K1b)
    bool cdsel_valid = false;   // ws_cdsel holds the selected cells' exact distances for the current batch
    int32_t *d_perm = nullptr;

    // CSR
    int64_t n_csr = 0;
    std::vector<int64_t> h_off;  // [nlists+1]
    int64_t max_list_len = 0;
    int64_t nonempty_lists = 0;  // (a shard holds only its share of the lists: averages are taken over these)
    int64_t *d_off = nullptr;
    void *d_codes = nullptr;
    int32_t *d_ids = nullptr;
    // pending (device, arrival order)
    int64_t n_pend = 0, cap_pend = 0;
    int32_t *d_pcell = nullptr, *d_pid = nullptr;
    void *d_pcodes = nullptr;

    // workspaces
    DevBuf<int32_t> ws_fb;
    DevBuf<unsigned short> ws_Qh, ws_Ql;
    DevBuf<float> ws_gmin;
    DevBuf<double> ws_glut;   // lookup tables in global memory when m * ks * 8 bytes do not fit the LDS: one per block
    size_t glut_slots = 0;
    DevBuf<int32_t> ws_clist;  // K1f in two kernels: candidate lists [nq][MMIDX_CLIST], entry 0 = length
    DevBuf<double> ws_Q, ws_cdist, ws_odist, ws_X, ws_Xa, ws_qn, ws_cdsel, ws_sdc;
    DevBuf<float> ws_Q32, ws_S;
    DevBuf<int32_t> ws_cells, ws_oiid, ws_ocnt, ws_flag, ws_ecell, ws_pcount, ws_pstart, ws_pcursor, ws_order;
    DevBuf<u64> ws_T, ws_pkey, ws_pval;
    DevBuf<u32> ws_pcnt;
    DevBuf<unsigned char> ws_ecode, ws_tmp, ws_keep, ws_amb, ws_out;
    DevBuf<int32_t> ws_aidx, ws_acell;
    int64_t last_ambiguous = 0;  // vectors of the last encode call that needed the exact redo
    DevBuf<long long> ws_dest;
    DevBuf<int4> ws_gdesc;
    DevBuf<int32_t> ws_cand;  // K3s: the pairs behind the coarse bound (+ their count in the last word)
    DevBuf<double> ws_smin;   // K3s: certified lower bounds of their Smin
    DevBuf<double> ws_Qp;     // K3s: the call's queries in transformed (permuted) order
    DevBuf<int32_t> ws_cand2; // K3s: what the bf16 stage could not drop (+ count in the last word)
    DevBuf<double> ws_smin1;  // K3s: the bf16 stage's bounds
    double *d_coarseP = nullptr;  // K3s: the centroids in transformed order (RandomPermutation only; built with the K3g tables)
    DevBuf<int32_t> ws_inv;   // iid -> position in the list-major arrays (-1: absent), built on demand
    bool inv_valid = false;
    int64_t inv_size = 0;
    DevBuf<int32_t> ws_gfb;
    DevBuf<int64_t> ws_flatoff;  // flat PQ through K3g: chunk offsets standing in for list offsets
    DevBuf<double> ws_flatlut;   // ... and the queries' exact lookup tables [nq][m][256] (k_flat_lut)
    DevBuf<u32> ws_ghist;        // K3g: per-query histogram of accepted candidates [nq][256] (thresholds from the union over lists)
    DevBuf<u64> ws_T0;           // ... and pass A's thresholds as the launch found them
    int no_union = 0;            // option "no_union": K3g without that histogram (A/B switch)
    int smin_pre = -1;           // option "smin_pre": K3s (k_pair_smin) in front of pass B's counting sort: 1 always, 0 never, -1 when the
                                 // device-reported figures of the call before say that at least half of the pairs end at Smin >= T
    int smin_bf16 = 1;           // option "smin_bf16": K3s's bf16 first stage for 16-dimensional sub-quantizers (0: fp32 only)
    int smin_valu = 0;           // option "smin_valu": K3s with packed VALU FMAs instead of the matrix cores (A/B)
    bool pre_on = false;         // the call before ran K3s (which pair of hint words describes it)
    int pre_skip = 0;            // calls K3s still sits out in front of K3mk (it removed next to nothing the last time)
    int flat_chunk = 0;          // option "flat_chunk": codes per chunk of a flat PQ list (0 = sized from the batch)
    // K3m (mmidx_scan_mfma.h): pass B as a certified lower bound on the matrix cores
    unsigned short *d_pq16 = nullptr;  // fp16 codebook [D / 8][256][8], scaled by 2^pq_ep
    double *d_pn64 = nullptr;          // [m][256] ||p_sj||^2
    DevBuf<float> xn;                  // ||x||^2 of every stored code (list-major)
    bool xn_valid = false, mfma_valid = false, mfma_ok = false;
    int pq_ep = 0;
    double pq_maxabs = 0.0;            // largest |codebook element| (set_pq)
    int no_mfma = 0;                   // option "no_mfma" = 1: pass B through K3g / K3f (A/B switch)
    int no_split_table = 0;            // option "no_split_table" = 1: a table of twice the LDS (m = 128) stays in global scratch (A/B switch)
    int mfma_sub = 0;                  // option "mfma_sub": codes per K3m item (0 = sized from the call)
    int mfma_qcap = 0;                 // option "mfma_qcap": survivor records per launch (0 = sized from the call; tests force the redo path)
    int mfma_blocks = 0;               // option "mfma_blocks": persistent blocks (0 = occupancy x CUs)
    int lut_pre = -1;                  // option "lut_pre": pass A's tables built ahead of K3h by k_lut_pre: 1 always, 0 never, -1 = where it pays (m >= 32)
    DevBuf<double> ws_lutpre;
    DevBuf<uint4> ws_surv;
    DevBuf<double> ws_R;               // RandomRotation: the kept pairs' exact rotated residuals [pairs][D]
    DevBuf<u32> ws_defer;              // k_coarse_front_sel: count + list of the queries left to k_coarse_select_defer
    int passb_small = 1;               // option "passb_small": 0 = pass B through K3m / K3g also when the call before kept at most 64 pairs
    int hint_calls = 0;                // IVF pass-B stages launched so far (pin_hint[0] describes the last one)
    int64_t hint_last_nq = -1, hint_prev_nq = -1;  // batch size and probes of the stage launched last / the one before it (whose count pin_hint[0] holds
    int hint_last_w = -1, hint_prev_w = -1;        //  while the current call is being enqueued): "passb_small" trusts the hint for a like call only
    DevBuf<unsigned short> ws_R16;     // K3mk: the kept pairs' fp16 residuals [pairs][D]
    DevBuf<double> ws_nrow;            // ... and ||r||^2
    double coarse_maxabs = 0.0;        // largest |centroid element| (set_coarse)
    int coarse_wave_sel = 1;           // option "coarse_wave_sel": 0 = the coarse stage's exact selection always by k_coarse_select_list (a block per query)
    int mfma_kc_v1 = 0;                // option "mfma_kc_v1": 1 = K3mk without LDS-DMA (k_scan_mfma_kc) also where k_scan_mfma_kc2 applies
    int mfma_kc_tpw = 8;               // option "mfma_kc_tpw": code tiles per wave of K3mk (8 or 16)
    DevBuf<u32> ws_mfctl, ws_psnap;
    DevBuf<unsigned char> ws_redo;
    void *d_grpx = nullptr, *pin_grpx = nullptr;  // K3g's GrpExtra on the device and its pinned mirror
    double *d_zero = nullptr;    // ... and the zero "centroid"

    // profiling: HIP events recorded on the launch stream, resolved lazily by mmidx_get_stats
    int profiling = 0;  // 0 off, 1 full (six events per call + code counters), 2 light (the pass-A pair only)
    mmidx_stats stats{};
    std::vector<hipEvent_t> evpool;  // groups of 6: start, coarse end, scan start, scan end, end, pass A end
    size_t ev_used = 0;
    std::vector<hipEvent_t> mf_ev;   // K3m, full profiling: groups of 3 (scan start, scan end, verification end)
    size_t mf_ev_used = 0;
    // which kernel family served each stage of the most recent search sub-batch (mmidx_get_dispatch; host-side words, a few stores per call)
    const char *disp_coarse = "-", *disp_passa = "-", *disp_passb = "-", *disp_pre = "-";
    // K3ma (pass A on the matrix cores, mmidx_scan_mfma_a.h)
    hipEvent_t host_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // mmidx_search (host buffers): the answers' slices on their way back
    int passa_q = -1;                  // option "passa_q": K3q (mmidx_scan_q.h) 1 always (where the shape allows), 0 never, -1 = from 1.25 queries per non-empty list of a long-list index
    double *d_pqstat = nullptr;        // K3q: [m * dsub] mean_j p_sj[t], then [m] mean_j ||p_sj||^2
    float *d_pqT32 = nullptr;          // K3q: the transposed codebook in fp32, dimension pairs side by side [m][dsub / 2][256][2] (ks = 256, even dsub)
    int passa_mfma = -1;               // option "passa_mfma": 1 always (where the shape allows), 0 never, -1 = from 8 queries per list of a long-list index
    int a_wide = 1;                    // option "passa_mfma_wide": 0 = K3ma's sweeps by the four-wave instance alone (A/B switch)
    DevBuf<float> ws_acand;            // [pairs][pieces][256] sweep 1's kept accumulator values
    DevBuf<double2> ws_arowc;          // [pairs] upper-bound maps
    DevBuf<unsigned char> ws_abm;      // [items][stride] sweep 2's compare masks
    DevBuf<u32> ws_aicnt;              // [items + 1] their set bits per item, then the exclusive prefix
    DevBuf<uint2> ws_arec;             // the flat record list {pair slot, position}
    DevBuf<double> ws_arows;           // [pairs][D] exact residuals
    DevBuf<int4> ws_ameta;             // [pairs] {query, rank, list start}
    DevBuf<u64> ws_ametaT;             // [pairs] thresholds
    DevBuf<uint2> ws_arnd;             // [rounds] slot ranges
    std::vector<hipEvent_t> a_ev;      // full profiling: groups of 5 (sweep 1 start / end, selection end, sweep 2 end, verification end)
    size_t a_ev_used = 0;
    long long a_launches = 0;
    u64 *d_counters = nullptr;       // [0] scan codes, [1] tie fallbacks, [2] codes of the probe-rank-0 lists (pass A), [3] verified codes (K3g); [4] add-validation flag
    int64_t host_codes = 0;          // PQ: nq * n, known on the host
    int32_t launches = 0;
    int32_t passa_launches = 0;
    int64_t host_passa_codes = 0;
};

namespace {

// a handle made by mmidx_create_sharded (h->grp != null): mmidx_sharded.h, included at the end of this file
int sharded_search(mmidx_index *h, int k, int64_t nq, const double *Q, int32_t *iid_out, double *dist_out, int32_t *count_out);
int sharded_add_vectors(mmidx_index *h, int64_t n, const double *X, const double *const *dXs, const int64_t *ns, const int32_t *iids,
                        int32_t iid0, int32_t *cell_out, void *code_out);
int sharded_add_codes(mmidx_index *h, int64_t n, const int32_t *iids, const int32_t *cells, const void *codes);
int sharded_encode(mmidx_index *h, int64_t n, const double *X, int32_t *cell_out, void *code_out);
int sharded_sync(mmidx_index *h);
int sharded_list_sizes(mmidx_index *h, int32_t *sizes_out);
int sharded_export(mmidx_index *h, int64_t *list_off_out, int32_t *iids_out, void *codes_out);
int sharded_get_codes(mmidx_index *h, int64_t n, const int32_t *iids, int32_t *cell_out, void *code_out);
int sharded_distance(mmidx_index *h, int64_t n, const double *Q, const int32_t *iids, double *dist_out);
int sharded_get_stats(mmidx_index *h, mmidx_stats *out);
int sharded_set_option(mmidx_index *h, const char *name, int value);
int sharded_for_each(mmidx_index *h, const std::function<int(mmidx_index *)> &f);
const mmidx_index *sharded_first(const mmidx_index *h);
int64_t sharded_total(const mmidx_index *h);
constexpr int32_t MMIDX_IID_AUTO = INT32_MIN;  // sharded_add_vectors: number the vectors from the handle's total, read under its lock
void sharded_destroy(mmidx_index *h);
#define NOT_ON_SHARDED(h, name)                                                                                              \
    do {                                                                                                                     \
        if ((h) && (h)->grp)                                                                                                 \
            return fail(MMIDX_ERR_UNSUPPORTED, name " takes a plain handle: a sharded handle has one device per shard (use the " \
                                                    "host-pointer entry points or the _sliced_device forms)");              \
    } while (0)

// Serialises the calls that use a handle's workspaces (see mmidx_index::search_mu).  A _device entry point is asynchronous on
// the caller's stream; when the previous call ran on a DIFFERENT stream its kernels may still be reading the workspaces, so
// the new call waits for that stream first (no cost while a handle is driven from one stream, the usual case).
struct DeviceCall {
    mmidx_index *h;
    hipStream_t st;
    std::unique_lock<std::recursive_mutex> lk;
    DeviceCall(mmidx_index *h_, hipStream_t st_) : h(h_), st(st_), lk(h_->search_mu) {
        if (h->last_stream_valid && h->last_stream != st) {
            // (a caller may have destroyed the stream of its previous call -- stream pools, short-lived streams: HIP then
            //  rejects the handle, and whatever that call left running is waited for device-wide instead)
            if (hipStreamSynchronize(h->last_stream) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
        }
    }
    ~DeviceCall() {
        h->last_stream = st;
        h->last_stream_valid = true;
    }
};

int set_device(const mmidx_index *h) {
    HIPCK(hipSetDevice(h->device));
    return MMIDX_OK;
}

int64_t total_size(const mmidx_index *h) { return h->n_csr + h->n_pend; }

int ensure_pending(mmidx_index *h, int64_t extra) {
    int64_t need = h->n_pend + extra;
    if (need <= h->cap_pend) return MMIDX_OK;
    int64_t cap = std::max<int64_t>(need, h->cap_pend * 2);
    cap = std::max<int64_t>(cap, 1024);
    int32_t *ncell = nullptr, *nid = nullptr;
    void *ncodes = nullptr;
    HIPCK(hipMalloc((void **)&ncell, cap * sizeof(int32_t)));
    HIPCK(hipMalloc((void **)&nid, cap * sizeof(int32_t)));
    HIPCK(hipMalloc(&ncodes, (size_t)cap * h->m * h->code_bytes));
    if (h->n_pend) {
        HIPCK(hipMemcpyAsync(ncell, h->d_pcell, h->n_pend * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
        HIPCK(hipMemcpyAsync(nid, h->d_pid, h->n_pend * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
        HIPCK(hipMemcpyAsync(ncodes, h->d_pcodes, (size_t)h->n_pend * h->m * h->code_bytes,
                             hipMemcpyDeviceToDevice, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    if (h->d_pcell) (void)hipFree(h->d_pcell);
    if (h->d_pid) (void)hipFree(h->d_pid);
    if (h->d_pcodes) (void)hipFree(h->d_pcodes);
    h->d_pcell = ncell;
    h->d_pid = nid;
    h->d_pcodes = ncodes;
    h->cap_pend = cap;
    return MMIDX_OK;
}

// Fold the pending records into the CSR layout (stable: arrival order inside every list, as
// invertedLists[c].add / pqByteCodes[c].add do, IVFPQ.java:337-346, :376-377, :700-715).
int build_csr(mmidx_index *h) {
    if (h->n_pend == 0 && h->d_off) return MMIDX_OK;
    const int nl = h->nlists;
    const int64_t np = h->n_pend, nold = h->n_csr, ntot = np + nold;
    std::vector<int32_t> pcell((size_t)np);
    if (np) {
        HIPCK(hipMemcpyAsync(pcell.data(), h->d_pcell, np * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    std::vector<int64_t> cnt((size_t)nl, 0), off_new((size_t)nl + 1, 0), cursor((size_t)nl);
    for (int64_t i = 0; i < np; i++) {
        int c = pcell[(size_t)i];
        if (c < 0 || c >= nl) {
            h->n_pend = 0;  // (drop the pending batch rather than fail every later call on this handle)
            return fail(MMIDX_ERR_INVALID_ARG, "list id %d outside 0..%d: %lld pending records dropped", c, nl - 1, (long long)np);
        }
        cnt[(size_t)c]++;
    }
    if (h->h_off.empty()) h->h_off.assign((size_t)nl + 1, 0);
    for (int c = 0; c < nl; c++) {
        const int64_t len_old = h->h_off[(size_t)c + 1] - h->h_off[(size_t)c];
        off_new[(size_t)c + 1] = off_new[(size_t)c] + len_old + cnt[(size_t)c];
        cursor[(size_t)c] = off_new[(size_t)c] + len_old;
    }
    std::vector<long long> dest((size_t)np);
    for (int64_t i = 0; i < np; i++) dest[(size_t)i] = cursor[(size_t)pcell[(size_t)i]]++;

    int64_t *d_off_new = nullptr;
    void *d_codes_new = nullptr;
    int32_t *d_ids_new = nullptr;
    HIPCK(hipMalloc((void **)&d_off_new, ((size_t)nl + 1) * sizeof(int64_t)));
    HIPCK(hipMalloc(&d_codes_new, std::max<size_t>((size_t)ntot * h->m * h->code_bytes, 16)));
    HIPCK(hipMalloc((void **)&d_ids_new, std::max<size_t>((size_t)ntot * sizeof(int32_t), 16)));
    HIPCK(hipMemcpyAsync(d_off_new, off_new.data(), ((size_t)nl + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    if (nold) {
        const unsigned grid = (unsigned)((nold + 255) / 256);
        if (h->code_bytes == 1)
            hipLaunchKernelGGL(k_move_old<unsigned char>, dim3(grid), dim3(256), 0, h->stream, h->d_off, d_off_new, nl,
                               (const unsigned char *)h->d_codes, h->d_ids, (unsigned char *)d_codes_new, d_ids_new, h->m, (long long)nold);
        else
            hipLaunchKernelGGL(k_move_old<unsigned short>, dim3(grid), dim3(256), 0, h->stream, h->d_off, d_off_new, nl,
                               (const unsigned short *)h->d_codes, h->d_ids, (unsigned short *)d_codes_new, d_ids_new, h->m, (long long)nold);
    }
    if (np) {
        HIPCK(h->ws_dest.reserve((size_t)np));
        HIPCK(hipMemcpyAsync(h->ws_dest.p, dest.data(), (size_t)np * sizeof(long long), hipMemcpyHostToDevice, h->stream));
        const unsigned grid = (unsigned)((np + 255) / 256);
        if (h->code_bytes == 1)
            hipLaunchKernelGGL(k_place_new<unsigned char>, dim3(grid), dim3(256), 0, h->stream, h->ws_dest.p,
                               (const unsigned char *)h->d_pcodes, h->d_pid, (unsigned char *)d_codes_new, d_ids_new, h->m, (long long)np);
        else
            hipLaunchKernelGGL(k_place_new<unsigned short>, dim3(grid), dim3(256), 0, h->stream, h->ws_dest.p,
                               (const unsigned short *)h->d_pcodes, h->d_pid, (unsigned short *)d_codes_new, d_ids_new, h->m, (long long)np);
    }
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    if (h->d_off) (void)hipFree(h->d_off);
    if (h->d_codes) (void)hipFree(h->d_codes);
    if (h->d_ids) (void)hipFree(h->d_ids);
    h->d_off = d_off_new;
    h->d_codes = d_codes_new;
    h->d_ids = d_ids_new;
    h->h_off = off_new;
    h->n_csr = ntot;
    h->n_pend = 0;
    // release the pending buffers of a bulk load (they can be as large as the index)
    if (h->cap_pend > (1 << 20)) {
        (void)hipFree(h->d_pcell);
        (void)hipFree(h->d_pid);
        (void)hipFree(h->d_pcodes);
        h->d_pcell = h->d_pid = nullptr;
        h->d_pcodes = nullptr;
        h->cap_pend = 0;
        h->ws_dest.release();
    }
    h->inv_valid = false;  // (iid -> position map of mmidx_get_codes / mmidx_distance)
    h->xn_valid = false;   // (K3m's per-code norms follow the codes)
    h->max_list_len = 0;
    h->nonempty_lists = 0;
    for (int c = 0; c < nl; c++) {
        const int64_t len = off_new[(size_t)c + 1] - off_new[(size_t)c];
        h->max_list_len = std::max(h->max_list_len, len);
        h->nonempty_lists += len > 0;
    }
    return MMIDX_OK;
}

int check_ready(const mmidx_index *h) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (!h->pq_set) return fail(MMIDX_ERR_NOT_READY, "product quantizer not loaded (loadProductQuantizer)");
    if (h->kind == MMIDX_KIND_IVFPQ && !h->coarse_set)
        return fail(MMIDX_ERR_NOT_READY, "coarse quantizer not loaded (loadCoarseQuantizer)");
    return MMIDX_OK;
}

size_t scan_lds_bytes(const mmidx_index *h, int cap) {
    return (size_t)h->m * h->ks * 8 + (h->transform ? 2 : 1) * (size_t)h->D * 8 + (size_t)cap * 12 + 16;
}

// encode n device-resident vectors into (cell, code); code_out holds centroid indices (CodeT)
// nearest coarse centroid of n device-resident vectors (computeNearestCoarseIndex, IVFPQ.java:547-564): exact fp64 argmin,
// first index wins ties; -1 for a PQ index
int assign_device(mmidx_index *h, int64_t n, const double *dX, int32_t *d_cell, hipStream_t st) {
    if (n == 0) return MMIDX_OK;
    const int ivf = h->kind == MMIDX_KIND_IVFPQ;
    const size_t asg_lds = (size_t)((ASG_BM * (h->D + 2) + 3) & ~3) * 4 + (size_t)ASG_BK * ASG_BN * 4;
    const bool split16 = h->d_Ch && !h->coarse_v1;  // (K6a' stages centroid tiles only: any vector length; K6a holds whole vectors in LDS)
    if (ivf && !h->exact_coarse && (split16 || asg_lds <= 160 * 1024) && h->C >= 2) {
        // certified approximate assignment (bf16-split or fp32 MFMA) + exact redo of the flagged vectors
        constexpr int QT = 16;
        if (!split16) HIPCK(h->ws_Q32.reserve((size_t)n * h->D));
        HIPCK(h->ws_qn.reserve((size_t)n));
        HIPCK(h->ws_amb.reserve((size_t)n));
        HIPCK(h->ws_aidx.reserve((size_t)n + 1));
        HIPCK(hipMemsetAsync(h->ws_aidx.p + n, 0, sizeof(int32_t), st));
        if (split16) {
            // bf16-split dot products on the matrix cores (K6a'), 5x the rate of the fp32 MFMA
            const size_t l16 = 2 * (size_t)G16_BC * G16_STRIDE;
            const bool fromx = h->Dp <= G16_KC && (h->D % 8) == 0 && ((uintptr_t)dX & 15) == 0 && !getenv("MMIDX_ASSIGN_SPLIT");
            if (fromx) {  // one k chunk: the kernel splits the fp64 rows itself (no k_split_bf16 round trip through HBM)
                HIPCK(hipFuncSetAttribute((const void *)k_assign_gmin16_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16));
                hipLaunchKernelGGL(k_assign_gmin16_t<true>, dim3((unsigned)((n + G16_BQ - 1) / G16_BQ)), dim3(MMIDX_BLOCK), l16, st, (const __bf16 *)nullptr,
                                   (const __bf16 *)nullptr, (const __bf16 *)h->d_Ch, (const __bf16 *)h->d_Cl, h->d_cn_pad, (const double *)nullptr, d_cell,
                                   h->ws_amb.p, h->cnorm_max, h->cn_max, h->Cp, h->Dp, (long long)n, dX, h->D, h->ws_aidx.p, h->ws_aidx.p + n);
            } else {
                HIPCK(h->ws_Qh.reserve((size_t)n * h->Dp));
                HIPCK(h->ws_Ql.reserve((size_t)n * h->Dp));
                hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)((n + 3) / 4)), dim3(MMIDX_BLOCK), 0, st, dX, (__bf16 *)h->ws_Qh.p, (__bf16 *)h->ws_Ql.p,
                                   (float *)nullptr, h->ws_qn.p, h->D, h->Dp, (long long)n);
                HIPCK(hipFuncSetAttribute((const void *)k_assign_gmin16_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16));
                hipLaunchKernelGGL(k_assign_gmin16_t<false>, dim3((unsigned)((n + G16_BQ - 1) / G16_BQ)), dim3(MMIDX_BLOCK), l16, st, (const __bf16 *)h->ws_Qh.p,
                                   (const __bf16 *)h->ws_Ql.p, (const __bf16 *)h->d_Ch, (const __bf16 *)h->d_Cl, h->d_cn_pad, h->ws_qn.p, d_cell,
                                   h->ws_amb.p, h->cnorm_max, h->cn_max, h->Cp, h->Dp, (long long)n, (const double *)nullptr, h->D, h->ws_aidx.p,
                                   h->ws_aidx.p + n);
            }
        } else {
            hipLaunchKernelGGL(k_query_prep, dim3((unsigned)((n + 3) / 4)), dim3(MMIDX_BLOCK), 0, st, dX, h->ws_Q32.p, h->ws_qn.p, h->D, (long long)n);
            HIPCK(hipFuncSetAttribute((const void *)k_assign_approx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asg_lds));
            hipLaunchKernelGGL(k_assign_approx, dim3((unsigned)((n + ASG_BM - 1) / ASG_BM)), dim3(MMIDX_BLOCK), asg_lds, st, h->d_coarseT32,
                               h->ws_Q32.p, h->d_cn, h->ws_qn.p, d_cell, h->ws_amb.p, h->cnorm_max, h->cn_max, h->C, h->D, (long long)n);
        }
        HIPCK(hipGetLastError());
        // the flagged vectors (second - best within the certified error: ~1e-4 of them) are listed ON THE DEVICE and only their
        // number travels to the host -- the flags themselves (a byte per vector: 10 MB per 10 M SURF descriptors of a VLAD call,
        // 100 MB per 100 M indexed vectors) used to be copied out and scanned by one host thread
        int32_t *d_na = h->ws_aidx.p + n;
        if (!split16) {  // (the fp32-MFMA kernel only writes flags; the bf16-split kernels list the flagged vectors themselves)
            hipLaunchKernelGGL(k_compact_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->ws_amb.p, (long long)n, h->ws_aidx.p, d_na);
            HIPCK(hipGetLastError());
        }
        int32_t na32 = 0;
        HIPCK(hipMemcpyAsync(&na32, d_na, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        h->last_ambiguous = (int64_t)na32;
        if (na32 > 0) {
            const int64_t na = (int64_t)na32;
            HIPCK(h->ws_acell.reserve((size_t)na));
            HIPCK(h->ws_Xa.reserve((size_t)na * h->D));
            const long long tot = (long long)na * h->D;
            hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dX, h->ws_aidx.p, h->ws_Xa.p, h->D, (long long)na);
            hipLaunchKernelGGL(k_assign_coarse<QT>, dim3((unsigned)((na + QT - 1) / QT)), dim3(MMIDX_BLOCK), 0, st, h->d_coarseT, h->ws_Xa.p,
                               h->ws_acell.p, h->C, h->D, (long long)na);
            hipLaunchKernelGGL(k_scatter_cells, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, h->ws_aidx.p, h->ws_acell.p, d_cell, (long long)na);
            HIPCK(hipGetLastError());
        }
    } else if (ivf) {
        constexpr int QT = 16;
        const unsigned grid = (unsigned)((n + QT - 1) / QT);
        hipLaunchKernelGGL(k_assign_coarse<QT>, dim3(grid), dim3(MMIDX_BLOCK), 0, st, h->d_coarseT, dX, d_cell, h->C, h->D, (long long)n);
    } else {
        HIPCK(hipMemsetAsync(d_cell, 0xFF, (size_t)n * sizeof(int32_t), st));  // -1
    }
    return MMIDX_OK;
}

int encode_device(mmidx_index *h, int64_t n, const double *dX, int32_t *d_cell, void *d_code, hipStream_t st) {
    if (n == 0) return MMIDX_OK;
    const int ivf = h->kind == MMIDX_KIND_IVFPQ;
    int rc0 = assign_device(h, n, dX, d_cell, st);
    if (rc0) return rc0;
    if ((h->transform == MMIDX_TR_NONE || h->transform == MMIDX_TR_PERMUTATION) && (h->dsub == 4 || h->dsub == 8 || h->dsub == 16)) {
        // K6b': one thread per vector, one sub-quantizer table in LDS at a time
        const size_t rl = (size_t)h->ks * h->dsub * 8 + (size_t)MMIDX_BLOCK * h->m * h->code_bytes;
        if (rl <= 64 * 1024) {
            const unsigned g = (unsigned)((n + MMIDX_BLOCK - 1) / MMIDX_BLOCK);
#define LAUNCH_ROWS(DS, CT)                                                                                                     \
    hipLaunchKernelGGL((k_encode_pq_rows<DS, CT>), dim3(g), dim3(MMIDX_BLOCK), rl, st, dX, d_cell, h->d_coarse, h->d_pq, h->d_perm, \
                       (CT *)d_code, h->D, h->m, h->ks, h->transform, ivf, (long long)n)
            if (h->code_bytes == 1) {
                if (h->dsub == 4) LAUNCH_ROWS(4, unsigned char);
                else if (h->dsub == 8) LAUNCH_ROWS(8, unsigned char);
                else LAUNCH_ROWS(16, unsigned char);
            } else {
                if (h->dsub == 4) LAUNCH_ROWS(4, unsigned short);
                else if (h->dsub == 8) LAUNCH_ROWS(8, unsigned short);
                else LAUNCH_ROWS(16, unsigned short);
            }
#undef LAUNCH_ROWS
            HIPCK(hipGetLastError());
            return MMIDX_OK;
        }
    }
    constexpr int VT = 8;
    const size_t lds = 2 * (size_t)VT * h->D * 8 + (size_t)h->m * VT * 4 * 12;
    if (lds > 160 * 1024) return fail(MMIDX_ERR_UNSUPPORTED, "vector length %d too large for the encode kernel", h->D);
    const unsigned grid = (unsigned)((n + VT - 1) / VT);
    if (h->code_bytes == 1) {
        HIPCK(hipFuncSetAttribute((const void *)k_encode_pq<VT, unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_encode_pq<VT, unsigned char>), dim3(grid), dim3(MMIDX_BLOCK), lds, st, dX, d_cell, h->d_coarse, h->d_pqT,
                           h->d_perm, h->d_rot, (unsigned char *)d_code, h->D, h->m, h->ks, h->dsub, h->transform, ivf, (long long)n);
    } else {
        HIPCK(hipFuncSetAttribute((const void *)k_encode_pq<VT, unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_encode_pq<VT, unsigned short>), dim3(grid), dim3(MMIDX_BLOCK), lds, st, dX, d_cell, h->d_coarse, h->d_pqT,
                           h->d_perm, h->d_rot, (unsigned short *)d_code, h->D, h->m, h->ks, h->dsub, h->transform, ivf, (long long)n);
    }
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

template <int M, typename CodeT, int SU, int NT = MMIDX_BLOCK, bool SDC = false, bool GLUT = false>
int launch_scan_t(const ScanParams &P, dim3 grid, size_t lds, hipStream_t st) {
    HIPCK(hipFuncSetAttribute((const void *)k_scan<M, CodeT, SU, NT, SDC, GLUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan<M, CodeT, SU, NT, SDC, GLUT>), grid, dim3(NT), lds, st, P);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

// su = codes per thread per segment (1 or 2; 11 = 1 with 512-thread blocks); P.cap and lds sized for it
int launch_scan(const mmidx_index *h, const ScanParams &P, dim3 grid, size_t lds, hipStream_t st, int su = 2) {
    if (P.glut) {  // the table lives in global scratch (make_plan: it does not fit the LDS): generic kernels, LDS = vectors + candidates
        if ((size_t)grid.x * grid.y > h->glut_slots) return fail(MMIDX_ERR_UNSUPPORTED, "lookup-table scratch too small for %u x %u blocks", grid.x, grid.y);
        if (P.cap < P.K1 + MMIDX_SEG) return fail(MMIDX_ERR_UNSUPPORTED, "candidate buffer of %d entries too small for k + 1 = %d and a %d-code segment", P.cap, P.K1, MMIDX_SEG);
        const size_t l = lds - (size_t)h->m * h->ks * 8;
        // twice the LDS (m = 128 byte codes): two sweeps with half the table in LDS each (k_scan_split); cap >= K1 + 512 covers its
        // 512-code segments, a chunk's partial sums fit the block's table slot
        if (!P.sdc_tt && !h->no_split_table && h->code_bytes == 1 && h->ks == 256 && h->m == 128 && P.chunk <= h->m * h->ks &&
            l + (size_t)64 * 256 * 8 <= 160 * 1024) {
            const size_t sl = l + (size_t)64 * 256 * 8;
            HIPCK(hipFuncSetAttribute((const void *)k_scan_split<128, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl));
            hipLaunchKernelGGL((k_scan_split<128, 512>), grid, dim3(512), sl, st, P);
            HIPCK(hipGetLastError());
            return MMIDX_OK;
        }
        if (P.sdc_tt) return launch_scan_t<0, unsigned char, 2, MMIDX_BLOCK, true, true>(P, grid, l, st);
        if (h->code_bytes == 1) return launch_scan_t<0, unsigned char, 2, MMIDX_BLOCK, false, true>(P, grid, l, st);
        return launch_scan_t<0, unsigned short, 2, MMIDX_BLOCK, false, true>(P, grid, l, st);
    }
    if (P.sdc_tt) {  // symmetric distances: byte codes only (the reference NPEs on short codes, PQ.java:350)
        switch (h->m) {
            case 8: return launch_scan_t<8, unsigned char, 2, MMIDX_BLOCK, true>(P, grid, lds, st);
            case 16: return launch_scan_t<16, unsigned char, 2, MMIDX_BLOCK, true>(P, grid, lds, st);
            default: return launch_scan_t<0, unsigned char, 2, MMIDX_BLOCK, true>(P, grid, lds, st);
        }
    }
    if (h->code_bytes == 1) {
        if (su == 11) {  // one code per thread, 512-thread blocks
            switch (h->m) {
                case 8: return launch_scan_t<8, unsigned char, 1, 512>(P, grid, lds, st);
                case 16: return launch_scan_t<16, unsigned char, 1, 512>(P, grid, lds, st);
                case 32: return launch_scan_t<32, unsigned char, 1, 512>(P, grid, lds, st);
                default: break;
            }
        }
        if (su == 1) {
            switch (h->m) {
                case 8: return launch_scan_t<8, unsigned char, 1>(P, grid, lds, st);
                case 16: return launch_scan_t<16, unsigned char, 1>(P, grid, lds, st);
                case 32: return launch_scan_t<32, unsigned char, 1>(P, grid, lds, st);
                default: break;
            }
        }
        switch (h->m) {
            case 4: return launch_scan_t<4, unsigned char, 2>(P, grid, lds, st);
            case 8: return launch_scan_t<8, unsigned char, 2>(P, grid, lds, st);
            case 16: return launch_scan_t<16, unsigned char, 2>(P, grid, lds, st);
            case 32: return launch_scan_t<32, unsigned char, 2>(P, grid, lds, st);
            case 64: return launch_scan_t<64, unsigned char, 2>(P, grid, lds, st);
            default: return launch_scan_t<0, unsigned char, 2>(P, grid, lds, st);
        }
    }
    switch (h->m) {
        case 8: return launch_scan_t<8, unsigned short, 2>(P, grid, lds, st);
        case 16: return launch_scan_t<16, unsigned short, 2>(P, grid, lds, st);
        default: return launch_scan_t<0, unsigned short, 2>(P, grid, lds, st);
    }
}

// the certified coarse stage (fp32 / bf16-split dot products + exact distances of the nominees) applies: run_coarse's own test
bool coarse_certified(const mmidx_index *h, size_t *alds_out = nullptr, int *cch_out = nullptr) {
    const size_t sel_fixed = (size_t)MMIDX_CSEL_CAP * 12 + (size_t)(h->w + 1) * 8 + (size_t)((h->w + 2) & ~1) * 4 + 16;
    const size_t row_bytes = (size_t)(h->D + MMIDX_TERM_PAD) * 8;
    int cch = MMIDX_CAND_CHUNK;
    if (sel_fixed + (size_t)cch * row_bytes > 64 * 1024) cch = sel_fixed < 64 * 1024 ? (int)((64 * 1024 - sel_fixed) / row_bytes) : 0;
    const size_t alds = sel_fixed + (size_t)std::max(cch, 1) * row_bytes;
    if (alds_out) *alds_out = alds;
    if (cch_out) *cch_out = cch;
    return !h->exact_coarse && h->C >= MMIDX_BLOCK && h->w + 1 <= MMIDX_BLOCK && h->C <= 64 * MMIDX_BLOCK && h->w >= 1 && cch >= 1 && alds <= 64 * 1024;
}

struct SearchPlan;
int launch_scan_seeded(const mmidx_index *h, ScanParams P, const SearchPlan &pl, dim3 grid, hipStream_t st);
int launch_scan_filtered(const mmidx_index *h, ScanParams P, const SearchPlan &pl, dim3 grid, hipStream_t st, int main_grid = 0);

struct SearchPlan {
    int K1, cap, chunk, nchunks, nitems, poolq;
    bool glut;   // the lookup table goes to global scratch (pl.lds still counts it: launch_scan subtracts)
    size_t lds;
    int64_t qb;  // queries per sub-batch
};

int make_plan(mmidx_index *h, int k, int64_t nq, SearchPlan &pl, bool need_coarse = true) {
    if (k < 1 || k > MMIDX_K_MAX) return fail(MMIDX_ERR_INVALID_ARG, "k must be in 1..%d (got %d)", MMIDX_K_MAX, k);
    pl.K1 = k + 1;
    int cap = 1;
    while (cap < pl.K1 + MMIDX_SEG) cap <<= 1;
    pl.cap = cap;
    pl.lds = scan_lds_bytes(h, cap);
    pl.glut = pl.lds > 160 * 1024;  // table in global scratch, the generic exact-scan kernels only (IVFPQ.java has no such limit)
    if (pl.glut && pl.lds - (size_t)h->m * h->ks * 8 > 160 * 1024)
        return fail(MMIDX_ERR_UNSUPPORTED, "vectors and candidate buffer of k = %d do not fit the 160 KiB LDS", k);
    const int ivf = h->kind == MMIDX_KIND_IVFPQ;
    const int nprobe = ivf ? h->w : 1;
    int64_t chunk = 16384;
    if (!ivf) {
        // flat PQ: chunk 0 is pass A (exact, LDS-gather bound: its cost grows with the chunk), the other chunks go through the
        // grouped filtered scan under pass A's threshold (looser with a short chunk 0, but lowered from the union of the verified
        // candidates as pass B goes).  Measured on cfg2 (1 M x 128, m = 8, k = 100; tools/dbg/cfg2_chunk.py): 65536 codes is best or
        // within 1 % of it from 1024 to 16384 queries per call (1.16-1.17 M q/s; the batch-proportional rule of rounds 1-2 chose
        // 131072 at 4096 queries: 1.09 M, and 524288 at 16384: 0.89 M); small batches keep 32768 so that one query still spreads
        // over the chip
        chunk = nq >= 512 ? 65536 : 32768;
        // (with K3m behind it -- thresholds tighten from the survivors' upper bounds as the scan goes -- the exact, LDS-bound pass over
        //  chunk 0 can be half as long: cfg2 2.06 -> 1.93 ms per 4096 queries; 16384 and 8192 measure the same)
        if (h->mfma_ok && !h->no_mfma) chunk = 32768;
        if (h->flat_chunk >= 4096) chunk = std::min<int64_t>(h->flat_chunk, 1 << 23);  // option "flat_chunk" (A/B)
    }
    const int64_t maxlen = std::max<int64_t>(h->max_list_len, 1);
    pl.chunk = (int)chunk;
    pl.nchunks = (int)((maxlen + chunk - 1) / chunk);
    pl.nitems = nprobe * pl.nchunks;
    // a work item emits at most K1 survivors, but no query can emit more than its candidates
    // (+ MMIDX_HKEEP: a K3h pass-A item may emit up to MMIDX_HKEEP entries instead of K1)
    int64_t poolq = (int64_t)pl.nitems * pl.K1 + MMIDX_HKEEP;
    pl.poolq = (int)std::min<int64_t>(poolq, std::max<int64_t>(h->n_csr, pl.K1));
    // sub-batch so that the pool stays <= 16 GiB (sized for 288 GB of HBM: a sharded search hands a rank the
    // whole node's batch at once) and, when the coarse stage runs inside the call, its scratch matrix <= 2 GiB
    int64_t qb = std::min<int64_t>(nq, (int64_t)(1 << 30) / std::max(nprobe, 1));
    const int64_t pool_bytes_q = (int64_t)pl.poolq * 16;
    qb = std::min<int64_t>(qb, std::max<int64_t>(1, (16ll << 30) / std::max<int64_t>(pool_bytes_q, 1)));
    // (the [nq][C] matrix of exact coarse distances: the exact path writes and reads all of it, the certified paths -- K1c ... K1f --
    //  touch only the rows of queries that overflow their candidate lists, so theirs may be large: a batch of 131072 queries over
    //  8192 cells reserves 8.6 GB of the 288)
    if (ivf && need_coarse) {
        int64_t cap_bytes = (coarse_certified(h) ? 16ll : 2ll) << 30;
        // (never more than a quarter of what the device has free right now: a smaller GPU, or one shared with a framework's
        //  allocator, takes sub-batches instead of hipErrorOutOfMemory -- ADVICE r5)
        // (asked only when the scratch of this call would pass 2 GiB: hipMemGetInfo is a driver round trip)
        if (cap_bytes > (2ll << 30) && (int64_t)nq * h->C * 8 > (2ll << 30)) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_bytes = std::max<int64_t>(2ll << 30, std::min<int64_t>(cap_bytes, (int64_t)(free_b / 4)));
        }
        qb = std::min<int64_t>(qb, std::max<int64_t>(1, cap_bytes / ((int64_t)h->C * 8)));
    }
    if (pl.glut) {  // one table per block in global scratch: at most 2 GiB of it
        const int64_t lut_bytes = (int64_t)h->m * h->ks * 8;
        qb = std::min<int64_t>(qb, std::max<int64_t>(1, (2ll << 30) / (lut_bytes * std::max(pl.nitems, 1))));
        // (pass B's grid is the pair count rounded up to a multiple of 8: the slots cover the padded grid)
        const size_t slots = (((size_t)qb * (size_t)std::max(nprobe, 1) + 7) / 8 * 8 + 8) * (size_t)std::max(pl.nchunks, 1);
        HIPCK(h->ws_glut.reserve(slots * (size_t)h->m * h->ks));
        h->glut_slots = slots;
    }
    pl.qb = qb;
    return MMIDX_OK;
}

template <int M>
int launch_filt_t(const mmidx_index *h, ScanParams P, dim3 grid, size_t lds, hipStream_t st, int main_grid = 0) {
    HIPCK(hipFuncSetAttribute((const void *)k_scan_filt<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned worst = grid.x;
    P.vgrid = worst;
    P.vb_base = 0;
    unsigned g1 = worst;
    if (P.order) {
        // device-side item count: size the main launch from the count seen last time (read back asynchronously into
        // pinned memory: a hint, possibly stale), let a small grid of looping blocks cover the rest
        const int32_t seen = h->pin_hint ? *(volatile int32_t *)h->pin_hint : -1;
        const unsigned hint = seen >= 0 ? (unsigned)seen : worst;
        const unsigned want = main_grid > 0 ? (unsigned)main_grid
                                            : (h->passb_main_grid > 0 ? (unsigned)h->passb_main_grid : 2u * hint + 2048u);
        g1 = std::min(worst, (want + 7u) & ~7u);
    }
    if (P.order && main_grid < 0) g1 = 0;  // (the looping kernel alone: what K3g hands back is normally nothing)
    grid.x = g1;
    if (g1 > 0) hipLaunchKernelGGL((k_scan_filt<M>), grid, dim3(MMIDX_BLOCK), lds, st, P);
    if (g1 < worst) {
        HIPCK(hipFuncSetAttribute((const void *)k_scan_filt_tail<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        P.vb_base = g1;
        grid.x = (std::min(worst - g1, 2048u) + 7u) & ~7u;
        hipLaunchKernelGGL((k_scan_filt_tail<M>), grid, dim3(MMIDX_BLOCK), lds, st, P);
    }
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

// pass B: lower-bound filtered scan where it applies (byte codes, templated m), else the exact scan
int launch_scan_filtered(const mmidx_index *h, ScanParams P, const SearchPlan &pl, dim3 grid, hipStream_t st, int main_grid) {
    const bool ok = h->code_bytes == 1 && h->ks <= 256 && (h->m == 8 || h->m == 16 || h->m == 32) && !h->no_filter && !P.sdc_tt;
    if (!ok) return launch_scan(h, P, grid, pl.lds, st);
    int cap = 1;
    while (cap < pl.K1 + MMIDX_VROUND) cap <<= 1;
    P.cap = cap;
    const size_t lds = (size_t)h->m * h->ks * 8 + 2 * (size_t)h->D * 8 + (size_t)cap * 12 + 4 * (size_t)h->m * 8 + 16 +
                       (size_t)MMIDX_SURV_CAP * 4 + 16 + (size_t)h->m * 256;
    if (lds > 160 * 1024) return launch_scan(h, P, grid, pl.lds, st);
    switch (h->m) {
        case 8: return launch_filt_t<8>(h, P, grid, lds, st, main_grid);
        case 16: return launch_filt_t<16>(h, P, grid, lds, st, main_grid);
        default: return launch_filt_t<32>(h, P, grid, lds, st, main_grid);
    }
}

template <int M>
int launch_seed_t(const ScanParams &P, dim3 grid, size_t lds, hipStream_t st) {
    HIPCK(hipFuncSetAttribute((const void *)k_scan_seed<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan_seed<M>), grid, dim3(MMIDX_BLOCK), lds, st, P);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

template <int M, int KS, int NT>
int launch_hist_nt(const ScanParams &P, dim3 grid, size_t lds, hipStream_t st) {
    if (KS == 256) {
        // the KS = 256 kernel addresses its table from LDS address 0 (byte_x8): it must not own static LDS
        static int static_lds = -1;
        if (static_lds < 0) {
            hipFuncAttributes fa{};
            HIPCK(hipFuncGetAttributes(&fa, (const void *)k_scan_hist<M, KS, NT>));
            static_lds = (int)fa.sharedSizeBytes;
        }
        if (static_lds != 0)
            return fail(MMIDX_ERR_UNSUPPORTED, "k_scan_hist owns %d bytes of static LDS: its table is not at LDS address 0", static_lds);
    }
    HIPCK(hipFuncSetAttribute((const void *)k_scan_hist<M, KS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan_hist<M, KS, NT>), grid, dim3(NT), lds, st, P);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

template <int M>
int launch_hist_t(const ScanParams &P, dim3 grid, size_t lds, hipStream_t st, bool wide) {
    if constexpr (M == 64) {  // 128 KiB of table: one block per CU, sixteen waves
        if (P.ks == 256) return launch_hist_nt<M, 256, 1024>(P, grid, lds, st);
        return launch_hist_nt<M, 0, 1024>(P, grid, lds, st);
    } else {
        if (P.ks == 256) return wide ? launch_hist_nt<M, 256, 512>(P, grid, lds, st) : launch_hist_nt<M, 256, 256>(P, grid, lds, st);
        return wide ? launch_hist_nt<M, 0, 512>(P, grid, lds, st) : launch_hist_nt<M, 0, 256>(P, grid, lds, st);
    }
}

// pass A over long lists: histogram-thresholded exact scan (K3h) + a K3 launch over the items it hands back.
// Returns 1 when K3h does not apply (the caller uses K3).
int launch_scan_hist(mmidx_index *h, ScanParams P, const SearchPlan &pl, dim3 grid, hipStream_t st) {
    const bool ok = h->code_bytes == 1 && (h->m == 8 || h->m == 16 || h->m == 32 || h->m == 64) && !P.sdc_tt && pl.K1 <= MMIDX_HKEEP &&
                    !P.order && !P.xcd_remap && pl.chunk <= (1 << 24);
    if (!ok) return 1;
    // a shard (at most half of the lists live here): launch over the queries whose nearest list is non-empty only
    const bool compact_items = P.ivf && P.nrank == 1 && P.rank_lo == 0 && grid.y == 1 && !h->no_item_compaction &&
                               h->nonempty_lists * 2 <= (int64_t)h->C && (int64_t)grid.x >= (int64_t)h->passa_item_min;
    const size_t fixed = (size_t)h->m * h->ks * 8 + (h->transform ? 2 : 1) * (size_t)h->D * 8 + 2 * MMIDX_HWV * 8 + 16 + MMIDX_HB * 4 + MMIDX_HCNT * 4;
    // 256 threads: four blocks per CU; 512 threads (option "passa_wide"): three, six waves per SIMD; m = 64 (the reference's
    // flagship shape, YFCC100MExample.java:85-90: 64 x 256 doubles = 128 KiB of table): ONE block of 1024 threads per CU --
    // the same sixteen waves per CU over one table instead of four
    const bool wide = h->passa_wide != 0;
    const bool whole_cu = h->m == 64;
    // position buffer: what is left of the block's share of the CU's LDS, within [768, 4096] entries
    int64_t room = (int64_t)(160 * 1024 / (whole_cu ? 1 : (wide ? 3 : 4))) - 256 - (int64_t)fixed;
    int cap = (int)std::min<int64_t>(4096, std::max<int64_t>(768, room / 4)) & ~15;  // an equal share per wave
    const size_t lds = fixed + (size_t)cap * 4;
    if (lds > (whole_cu ? 160 * 1024 : 64 * 1024)) return 1;
    const size_t nfb = (size_t)grid.x * grid.y;
    if (h->ws_fb.cap < 2 * nfb + 4) {  // (search_batch_device reserves and zeroes the header; a direct caller would land here)
        HIPCK(h->ws_fb.reserve(2 * nfb + 4));
        HIPCK(hipMemsetAsync(h->ws_fb.p, 0, 4 * sizeof(int32_t), st));
    }
    P.cap = cap;
    // the queries' exact tables ahead of the scan (k_lut_pre: a codebook row read once per 64 queries instead of once per query)
    {
        const bool want = h->lut_pre > 0 || (h->lut_pre < 0 && h->m >= 32);
        const long long nqa = (long long)grid.x;  // (pass A: one item per query)
        const size_t tab = (size_t)h->m * h->ks;
        if (want && P.nrank == 1 && P.rank_lo == 0 && grid.y == 1 && h->transform != MMIDX_TR_ROTATION && (tab & 1) == 0 && (h->dsub == 8 || h->dsub == 16 || h->dsub == 4) &&
            h->ks <= 256 && (size_t)nqa * tab * 8 <= ((size_t)4 << 30)) {
            HIPCK(h->ws_lutpre.reserve((size_t)nqa * tab));
            const dim3 g((unsigned)h->m, (unsigned)((nqa + LUTPRE_QB - 1) / LUTPRE_QB));
            if (g.y <= 65535) {
                if (h->dsub == 16) hipLaunchKernelGGL(k_lut_pre<16>, g, dim3(256), 0, st, P.Q, P.coarse, P.cells, P.perm, P.pqT, h->ws_lutpre.p, h->D, h->m, h->ks, P.w, P.ivf, nqa);
                else if (h->dsub == 8) hipLaunchKernelGGL(k_lut_pre<8>, g, dim3(256), 0, st, P.Q, P.coarse, P.cells, P.perm, P.pqT, h->ws_lutpre.p, h->D, h->m, h->ks, P.w, P.ivf, nqa);
                else hipLaunchKernelGGL(k_lut_pre<4>, g, dim3(256), 0, st, P.Q, P.coarse, P.cells, P.perm, P.pqT, h->ws_lutpre.p, h->D, h->m, h->ks, P.w, P.ivf, nqa);
                HIPCK(hipGetLastError());
                P.lut_pre = h->ws_lutpre.p;
            }
        }
    }
    P.fb_count = (u32 *)h->ws_fb.p;
    P.fb_items = h->ws_fb.p + 4;
    P.fb_ch = h->ws_fb.p + 4 + nfb;
    if (compact_items) {
        const long long nq_items = (long long)grid.x;
        const double share = (double)h->nonempty_lists / (double)h->C;  // expected fraction of queries served here
        long long gm = (long long)(1.15 * share * (double)nq_items) + h->passa_item_margin;
        if (gm < 1) gm = 1;
        if (gm > nq_items) gm = nq_items;
        HIPCK(h->ws_order.reserve((size_t)nq_items * (size_t)P.w));  // (pass B's size: its reserve later must not reallocate under this launch)
        int32_t *cnt = h->ws_fb.p + 1;  // (zeroed with the hand-back header)
        hipLaunchKernelGGL(k_passa_items, dim3((unsigned)((nq_items + MMIDX_BLOCK - 1) / MMIDX_BLOCK)), dim3(MMIDX_BLOCK), 0, st, P.cells, P.w,
                           P.list_off, nq_items, (int)gm, h->ws_order.p, cnt, P.fb_count, P.fb_items, P.fb_ch);
        HIPCK(hipGetLastError());
        P.order = h->ws_order.p;
        P.n_order = cnt;
        grid.x = (unsigned)gm;
    }
    int rc;
    switch (h->m) {
        case 8: rc = launch_hist_t<8>(P, grid, lds, st, wide); break;
        case 16: rc = launch_hist_t<16>(P, grid, lds, st, wide); break;
        case 64: rc = launch_hist_t<64>(P, grid, lds, st, wide); break;
        default: rc = launch_hist_t<32>(P, grid, lds, st, wide); break;
    }
    if (rc) return rc;
    if (h->debug_sync) {
        int32_t c4[4];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(c4, h->ws_fb.p, sizeof(c4), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mmidx] K3h: %d of %zu items handed back to K3 (> %d entries under the final bucket; cap %d, lds %zu); "
                "debug build only: %d second passes, %.1f appended and %.1f kept per item\n", c4[0], nfb, MMIDX_HKEEP, cap, lds,
                c4[1], (double)(unsigned)c4[2] / (double)nfb, (double)(unsigned)c4[3] / (double)nfb);
    }
    // the handed-back items (device-side count; normally none: the blocks exit at once)
    ScanParams F = P;
    F.cap = pl.cap;
    F.order = P.fb_items;
    F.n_order = (const int32_t *)P.fb_count;
    F.order_ch = P.ivf ? P.fb_ch : nullptr;
    F.n_items = (int)nfb;
    return launch_scan(h, F, dim3((unsigned)nfb, 1), pl.lds, st);
}

// pass A: seeded scan (exact sample -> histogram of u8 lower bounds -> exact verify) where it applies
int launch_scan_seeded(const mmidx_index *h, ScanParams P, const SearchPlan &pl, dim3 grid, hipStream_t st) {
    const bool ok = h->code_bytes == 1 && h->ks <= 256 && (h->m == 8 || h->m == 16 || h->m == 32) && !h->no_filter &&
                    !h->no_seed && !P.sdc_tt;
    if (!ok) return launch_scan(h, P, grid, pl.lds, st);
    int cap = 1;
    while (cap < pl.K1 + MMIDX_SEED_N0) cap <<= 1;
    P.cap = cap;
    const size_t lds = (size_t)h->m * h->ks * 8 + 2 * (size_t)h->D * 8 + (size_t)cap * 12 + 4 * (size_t)h->m * 8 + 16 +
                       (size_t)MMIDX_SURV_CAP * 4 + 256 * 4 + 16 + (size_t)h->m * 256 + (size_t)((pl.chunk + 15) & ~15);
    if (lds > 160 * 1024) return launch_scan(h, P, grid, pl.lds, st);
    switch (h->m) {
        case 8: return launch_seed_t<8>(P, grid, lds, st);
        case 16: return launch_seed_t<16>(P, grid, lds, st);
        default: return launch_seed_t<32>(P, grid, lds, st);
    }
}

// MMIDX_DEBUG_SYNC=1: synchronise after every stage and report the first failing one
#define DBG_SYNC(name)                                                                        \
    do {                                                                                      \
        if (h->debug_sync) {                                                                  \
            hipError_t e__ = hipStreamSynchronize(st);                                        \
            fprintf(stderr, "[mmidx] %s: %s\n", name, hipGetErrorString(e__));               \
            if (e__ != hipSuccess) return fail(MMIDX_ERR_HIP, "%s failed: %s", name, hipGetErrorString(e__)); \
        }                                                                                     \
    } while (0)

// ---- K3g (mmidx_scan_grp.h): grouped, list-major pass B -------------------------------------------------------
// index-side tables: built once per (coarse, product) quantizer pair, on the handle's stream, synchronously
int build_grp_tables(mmidx_index *h) {
    if (h->grp_valid) return MMIDX_OK;
    const bool shape_ok = (h->kind == MMIDX_KIND_IVFPQ || h->kind == MMIDX_KIND_PQ) && h->code_bytes == 1 && h->ks <= 256 &&
                          (h->m == 8 || h->m == 16 || h->m == 32 || h->m == 64) && h->transform != MMIDX_TR_ROTATION;
    if (!shape_ok || !h->pq_set) return MMIDX_OK;
    if (!h->d_pq32T) HIPCK(hipMalloc((void **)&h->d_pq32T, (size_t)h->m * h->dsub * 256 * sizeof(float)));
    if (!h->d_pn32) HIPCK(hipMalloc((void **)&h->d_pn32, (size_t)h->m * 256 * sizeof(float)));
    if (!h->d_pnmax) HIPCK(hipMalloc((void **)&h->d_pnmax, 2 * (size_t)h->m * sizeof(double)));
    hipLaunchKernelGGL(k_pq32_table, dim3((unsigned)h->m), dim3(256), 0, h->stream, h->d_pqT, h->d_pq32T, h->d_pn32, h->m, h->ks, h->dsub);
    hipLaunchKernelGGL(k_pn_max, dim3((unsigned)h->m), dim3(256), 0, h->stream, h->d_pqT, h->d_pnmax, h->m, h->ks, h->dsub);
    if (h->kind == MMIDX_KIND_IVFPQ && h->d_perm && h->coarse_set) {  // K3s reads contiguous dimensions
        if (!h->d_coarseP) HIPCK(hipMalloc((void **)&h->d_coarseP, (size_t)h->C * h->D * sizeof(double)));
        const long long tot = (long long)h->C * h->D;
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->d_coarse, h->d_perm, h->d_coarseP, h->D,
                           (long long)h->C);
    }
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    h->grp_valid = true;
    return MMIDX_OK;
}

template <int M, int G, int DSUB, bool FLAT = false, bool UNION = false>
int launch_grp_t(mmidx_index *h, const GrpParams &GP, size_t lds, hipStream_t st) {
    {   // the scan addresses the u8 rows from LDS address 0 (GrpLds::lut8 == 0, byte_x8): the kernel must not own static LDS
        static int static_lds = -1;
        if (static_lds < 0) {
            hipFuncAttributes fa{};
            HIPCK(hipFuncGetAttributes(&fa, (const void *)k_scan_grp<M, G, DSUB, FLAT, UNION>));
            static_lds = (int)fa.sharedSizeBytes;
        }
        if (static_lds != 0) return 1;  // (K3f takes the pairs)
    }
    HIPCK(hipFuncSetAttribute((const void *)k_scan_grp<M, G, DSUB, FLAT, UNION>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = h->grp_blocks;
    if (blocks <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_scan_grp<M, G, DSUB, FLAT, UNION>, GRP_NT, lds) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks = occ * std::max(h->num_cus, 8);
    }
    blocks = std::max(8, (blocks + 7) & ~7);
    hipLaunchKernelGGL((k_scan_grp<M, G, DSUB, FLAT, UNION>), dim3((unsigned)blocks), dim3(GRP_NT), lds, st, GP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

// ---- K3ma (mmidx_scan_mfma_a.h): pass A through the matrix-core bound ---------------------------------------------------------
template <int NJ, int DSUB, int MODE, int NWV>
int launch_mfma_a_scan_w(mmidx_index *h, const MfmaParams &MP, size_t lds, hipStream_t st) {
    HIPCK(hipFuncSetAttribute((const void *)k_scan_mfma<NJ, DSUB, MODE, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = h->mfma_blocks;
    if (blocks <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_scan_mfma<NJ, DSUB, MODE, NWV>, NWV * 64, lds) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks = occ * std::max(h->num_cus, 8);
    }
    blocks = std::max(8, (blocks + 7) & ~7);
    hipLaunchKernelGGL((k_scan_mfma<NJ, DSUB, MODE, NWV>), dim3((unsigned)blocks), dim3(NWV * 64), lds, st, MP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}
// a sweep: the eight-wave instance over the items of <= 32 rows (where it is on), then the four-wave instance over the others.
// Sweep 2 only: its loop fits 128 registers (1.27 -> 1.13 ms per 131072 queries); sweep 1 keeps 16 / 32 slot values next to the
// fragments and reloads spilled registers inside the tile loop with eight waves (1.14 -> 2.66 ms), so it stays with four.
template <int NJ, int DSUB, int MODE>
int launch_mfma_a_scan_t(mmidx_index *h, const MfmaParams &MP, size_t lds, hipStream_t st) {
    if constexpr (MODE == 2) {
        if (MP.a_wide) {
            const int rc = launch_mfma_a_scan_w<NJ, DSUB, MODE, 8>(h, MP, lds, st);
            if (rc) return rc;
        }
    }
    return launch_mfma_a_scan_w<NJ, DSUB, MODE, 4>(h, MP, lds, st);
}
template <int MODE>
int launch_mfma_a_scan(mmidx_index *h, const MfmaParams &MP, size_t lds, hipStream_t st) {
    const int nj = h->D / 32;
    if (h->dsub == 4) return nj == 4 ? launch_mfma_a_scan_t<4, 4, MODE>(h, MP, lds, st) : nj == 2 ? launch_mfma_a_scan_t<2, 4, MODE>(h, MP, lds, st) : launch_mfma_a_scan_t<1, 4, MODE>(h, MP, lds, st);
    if (h->dsub == 8) return nj == 4 ? launch_mfma_a_scan_t<4, 8, MODE>(h, MP, lds, st) : nj == 2 ? launch_mfma_a_scan_t<2, 8, MODE>(h, MP, lds, st) : launch_mfma_a_scan_t<1, 8, MODE>(h, MP, lds, st);
    return nj == 4 ? launch_mfma_a_scan_t<4, 16, MODE>(h, MP, lds, st) : nj == 2 ? launch_mfma_a_scan_t<2, 16, MODE>(h, MP, lds, st) : launch_mfma_a_scan_t<1, 16, MODE>(h, MP, lds, st);
}
template <int M, int DSUB>
int launch_a1_verify_t(mmidx_index *h, const MfmaParams &MP, hipStream_t st) {
    const A1VLds L(M, DSUB);
    HIPCK(hipFuncSetAttribute((const void *)k_a1_verify<M, DSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
    const unsigned grid = (unsigned)std::max(h->num_cus, 8);  // (one block of sixteen waves per CU, walking the list)
    hipLaunchKernelGGL((k_a1_verify<M, DSUB>), dim3(grid), dim3(A1V_NT(DSUB)), L.total, st, MP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}


// ---- K3q (mmidx_scan_q.h): pass A on integers, up to four queries of a nearest list per block -----------------------------------
// does pass A of this call go through K3q?
bool passa_q_applies(const mmidx_index *h, const ScanParams &P, const SearchPlan &pl, long long nq) {
    if (h->passa_q == 0 || !P.ivf || P.sdc_tt || nq <= 0 || !h->d_pqstat || !h->d_pqT32) return false;
    if (h->code_bytes != 1 || h->ks != 256 || h->m != 16 || (h->dsub != 4 && h->dsub != 8 && h->dsub != 16)) return false;
    if (pl.K1 + 40 > MMIDX_Q_HKQ) return false;  // (~K1 + 10 candidates per query, and room for ties)
    if (h->max_list_len >= (1ll << 24) || nq * (long long)P.w >= 0x7fffff00ll) return false;
    if (h->transform == MMIDX_TR_ROTATION && !h->d_rot) return false;
    if (h->transform == MMIDX_TR_PERMUTATION && !h->d_perm) return false;
    if (h->passa_q > 0) return true;
    // a block costs the same for one query as for four: from ~1.25 queries per non-empty list the shared scan beats K3h's one block per
    // query; long lists only (K3h's gate)
    return 4 * nq >= 5 * std::max<int64_t>(1, h->nonempty_lists) && h->n_csr / std::max<int64_t>(1, h->nonempty_lists) >= 4096;
}

template <int M, int DSUB>
int launch_q_t(const QParams &QP, unsigned grid, size_t lds, hipStream_t st) {
    static int static_lds = -1;  // (the table is addressed from LDS address 0, byte_x8: the kernel must not own static LDS)
    if (static_lds < 0) {
        hipFuncAttributes fa{};
        HIPCK(hipFuncGetAttributes(&fa, (const void *)k_scan_q<M, DSUB>));
        static_lds = (int)fa.sharedSizeBytes;
    }
    if (static_lds != 0) return fail(MMIDX_ERR_UNSUPPORTED, "k_scan_q owns %d bytes of static LDS: its table is not at LDS address 0", static_lds);
    HIPCK(hipFuncSetAttribute((const void *)k_scan_q<M, DSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_scan_q<M, DSUB>), dim3(grid), dim3(256), lds, st, QP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

// Pass A of the whole sub-batch through K3q; thresholds in P.T, exact candidates in the pools, as K3h leaves them; the queries it hands
// back (ties, an unusable scale) go through the exact kernel K3.  Uses ws_pcount (the caller zeroes it again for pass B), ws_pstart /
// ws_pcursor / ws_order / ws_gdesc / ws_gfb / ws_fb.
int launch_passa_q(mmidx_index *h, const ScanParams &P, const SearchPlan &pl, long long nq, hipStream_t st) {
    constexpr int G = MMIDX_Q_G;
    const int C = h->C;
    const size_t nfb = (size_t)nq * (size_t)std::max(pl.nchunks, 1);
    HIPCK(h->ws_pcount.reserve((size_t)C + 1));
    HIPCK(h->ws_pstart.reserve((size_t)C + 1));
    HIPCK(h->ws_pcursor.reserve((size_t)C));
    HIPCK(h->ws_order.reserve((size_t)nq * (size_t)P.w));  // (pass B's size: its reserve later must not reallocate under these launches)
    const size_t max_groups = (size_t)nq / G + (size_t)std::min<long long>(nq, C) + 8;
    HIPCK(h->ws_gdesc.reserve(max_groups));
    HIPCK(h->ws_gfb.reserve(8));
    if (h->ws_fb.cap < 2 * nfb + 4) {
        HIPCK(h->ws_fb.reserve(2 * nfb + 4));
        HIPCK(hipMemsetAsync(h->ws_fb.p, 0, 4 * sizeof(int32_t), st));
    }
    // the (query, probe 0) pairs by cell, groups of <= 4
    const unsigned gq = (unsigned)((nq + 255) / 256);
    hipLaunchKernelGGL(k_a1_pair_count, dim3(gq), dim3(256), 0, st, P.cells, P.w, (long long)nq, P.list_off, h->ws_pcount.p, C);
    hipLaunchKernelGGL(k_q_scan_groups, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, C, G, h->ws_pstart.p, h->ws_pcursor.p, h->ws_gdesc.p, h->ws_gfb.p);
    hipLaunchKernelGGL(k_a1_pair_scatter, dim3(gq), dim3(256), 0, st, P.cells, P.w, (long long)nq, P.list_off, h->ws_pstart.p, h->ws_pcursor.p, h->ws_order.p);
    HIPCK(hipGetLastError());
    QParams QP{};
    QP.S = P;
    QP.S.order = h->ws_order.p;
    QP.S.n_order = h->ws_pstart.p + C;
    QP.S.rot = h->d_rot;
    QP.S.perm = h->d_perm;
    QP.S.fb_count = (u32 *)h->ws_fb.p;
    QP.S.fb_items = h->ws_fb.p + 4;
    QP.S.fb_ch = h->ws_fb.p + 4 + nfb;
    QP.gdesc = h->ws_gdesc.p;
    QP.n_groups = h->ws_gfb.p;
    QP.pq = h->d_pq;
    QP.pqstat = h->d_pqstat;
    QP.pqT32 = h->d_pqT32;
    QP.pmax = h->rmax;
    QP.timing = nullptr;
    const QLds L(h->m, h->D);
    // grid: the host's upper bound of the group count (the blocks beyond the device-side count leave at once)
    const unsigned grid = (unsigned)((std::min<size_t>(max_groups, (size_t)nq) + 7 + 7) & ~(size_t)7);  // (eight ranges of ceil(groups / 8): k_scan_q's XCD map)
    int rc;
    switch (h->dsub) {
        case 4: rc = launch_q_t<16, 4>(QP, grid, L.total, st); break;
        case 8: rc = launch_q_t<16, 8>(QP, grid, L.total, st); break;
        default: rc = launch_q_t<16, 16>(QP, grid, L.total, st); break;
    }
    if (rc) return rc;
    if (h->debug_sync) {
        int32_t c4[4], ng = 0;
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(c4, h->ws_fb.p, sizeof(c4), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&ng, h->ws_gfb.p, sizeof(ng), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mmidx] K3q: %lld queries in %d groups (grid %u, lds %zu); %d (query, chunk) items handed back to K3\n", nq, ng, grid, L.total, c4[0]);
    }
    // the handed-back queries (device-side count; normally none: the blocks exit at once)
    ScanParams F = P;
    F.cap = pl.cap;
    F.fb_count = QP.S.fb_count;
    F.fb_items = QP.S.fb_items;
    F.fb_ch = QP.S.fb_ch;
    F.order = QP.S.fb_items;
    F.n_order = (const int32_t *)QP.S.fb_count;
    F.order_ch = P.ivf ? QP.S.fb_ch : nullptr;
    F.n_items = (int)nfb;
    return launch_scan(h, F, dim3((unsigned)nfb, 1), pl.lds, st);
}

// K3ma's items: (group, piece of <= sub codes); at most eight pieces per list (k_a1_select holds a pair's values in registers).  The
// sweep-2 bitmap is max_groups x nsub x bm_stride bytes; when that exceeds 32 GiB (one very long list on an index of many cells) K3ma
// does not apply -- decided HERE, by passa_mfma_applies, so that the caller takes the K3q / K3h branch instead of failing the search
// (ADVICE r5).
struct PassaMfmaShape {
    int sub, nsub;
    size_t max_groups, bm_stride;
    bool fits;
};
PassaMfmaShape passa_mfma_shape(const mmidx_index *h, long long npairs) {
    constexpr int G = MF_QG;
    PassaMfmaShape S{};
    const long long maxlen = std::max<long long>(h->max_list_len, 1);
    int sub = h->mfma_sub;
    if (sub <= 0) {
        const long long est_groups = npairs / G + std::min<long long>(npairs, std::max<int64_t>(1, h->nonempty_lists));
        const long long want = 4ll * 2 * std::max(h->num_cus, 8);
        long long pieces = std::max<long long>(1, std::min<long long>(8, (want + est_groups - 1) / est_groups));
        long long sb = (maxlen + pieces - 1) / pieces;
        sb = std::max<long long>(1024, sb);
        sub = (int)((sb + 63) & ~63ll);
    }
    sub = (std::max(sub, 64) + 31) & ~31;
    if ((maxlen + sub - 1) / sub > 8) sub = (int)((((maxlen + 7) / 8) + 31) & ~31ll);
    S.sub = sub;
    S.nsub = (int)((maxlen + sub - 1) / sub);
    S.max_groups = (size_t)npairs / G + (size_t)std::min<long long>(npairs, h->C) + 1;
    S.bm_stride = (size_t)((sub + 31) / 32) * 256;
    S.fits = S.max_groups * (size_t)S.nsub * S.bm_stride <= ((size_t)32 << 30);
    return S;
}

// does pass A of this call go through K3ma?  (all the applicability checks: launch_passa_mfma itself must not fall back after its first launch)
bool passa_mfma_applies(const mmidx_index *h, const ScanParams &P, const SearchPlan &pl, long long nq) {
    if (h->passa_mfma == 0 || !P.ivf || h->no_mfma || !h->mfma_ok || !h->xn_valid || h->no_filter || P.sdc_tt || nq <= 0 || h->D > 128) return false;
    if (pl.K1 > 128) return false;                                             // (a pair keeps 256 values per piece: K1 of them must be there)
    if (h->max_list_len >= (1ll << 24) || nq * (long long)P.w >= 0x7fffff00ll || ((uintptr_t)P.Q & 15) != 0) return false;
    if (h->d_perm && !h->d_coarseP) return false;
    if (h->transform == MMIDX_TR_ROTATION && (!h->d_rot || (size_t)nq * h->D * 8 > ((size_t)8 << 30))) return false;
    if (!passa_mfma_shape(h, nq).fits) return false;  // (the sweep-2 bitmap would exceed 32 GiB: one very long list)
    if (h->passa_mfma > 0) return true;
    // from ~8 queries per nearest list, on an index of long lists (where K3h would run): a shard holding a part of the
    // lists sees the same number of queries per LOCAL list
    return nq >= 8ll * h->C && h->n_csr / std::max<int64_t>(1, h->nonempty_lists) >= 4096;
}

// Pass A of the whole sub-batch through K3ma; the thresholds land in P.T and the exact candidates in the pools as K3h leaves them.
// Uses ws_pcount (zeroed by k_step_init; the caller zeroes it again for pass B), ws_pstart / ws_pcursor / ws_order / ws_gdesc / ws_gfb.
int launch_passa_mfma(mmidx_index *h, const ScanParams &P, const SearchPlan &pl, long long nq, hipStream_t st) {
    constexpr int G = MF_QG;
    const int C = h->C;
    const long long npairs = nq;  // (at most: the queries whose nearest list is non-empty here)
    const size_t nfb = (size_t)npairs * (size_t)std::max(pl.nchunks, 1);
    HIPCK(h->ws_pcount.reserve((size_t)C + 1));
    HIPCK(h->ws_pstart.reserve((size_t)C + 1));
    HIPCK(h->ws_pcursor.reserve((size_t)C));
    HIPCK(h->ws_order.reserve((size_t)nq * (size_t)P.w));  // (pass B's size: its reserve later must not reallocate under these launches)
    HIPCK(h->ws_gdesc.reserve((size_t)npairs / G + (size_t)C + 8));
    HIPCK(h->ws_gfb.reserve(4 + 2 * nfb + 16));
    const PassaMfmaShape SH = passa_mfma_shape(h, npairs);
    if (!SH.fits) return fail(MMIDX_ERR_UNSUPPORTED, "K3ma's bitmap does not fit (passa_mfma_applies must have said so)");
    const int sub = SH.sub, nsub = SH.nsub;
    const size_t max_groups = SH.max_groups, bm_stride = SH.bm_stride;
    HIPCK(h->ws_T0.reserve((size_t)nq));
    HIPCK(h->ws_redo.reserve((size_t)nq));
    HIPCK(h->ws_psnap.reserve((size_t)nq));
    HIPCK(h->ws_mfctl.reserve(64));
    const int cstride = 256;  // (512: sweep 1 by the eight-wave instance -- not used, see launch_mfma_a_scan_t)
    HIPCK(h->ws_acand.reserve((size_t)npairs * nsub * cstride));
    HIPCK(h->ws_arowc.reserve((size_t)npairs));
    HIPCK(h->ws_abm.reserve(max_groups * (size_t)nsub * bm_stride));
    HIPCK(h->ws_aicnt.reserve(max_groups * (size_t)nsub + 1));
    // (about K1 + 10 % records per pair; the list holds 512 per pair -- what does not fit sends its queries to the exact kernels)
    const size_t rec_cap = h->mfma_qcap > 0 ? (size_t)h->mfma_qcap : std::min<size_t>((size_t)0xFFFFF000u, std::max<size_t>((size_t)1 << 20, (size_t)npairs * 512));
    HIPCK(h->ws_arec.reserve(rec_cap));
    const int rnd_size = A1V_RND(h->dsub);
    HIPCK(h->ws_arnd.reserve(rec_cap / (size_t)rnd_size + 2));
    if (h->transform != MMIDX_TR_ROTATION) HIPCK(h->ws_arows.reserve((size_t)npairs * h->D));
    HIPCK(h->ws_ameta.reserve((size_t)npairs));
    HIPCK(h->ws_ametaT.reserve((size_t)npairs));
    if (h->d_perm) HIPCK(h->ws_Qp.reserve((size_t)nq * h->D));
    if (h->transform == MMIDX_TR_ROTATION) HIPCK(h->ws_R.reserve((size_t)npairs * h->D));
    // ---- the (query, probe 0) pairs by cell, groups of <= 64 ----
    const unsigned gq = (unsigned)((nq + 255) / 256);
    hipLaunchKernelGGL(k_a1_pair_count, dim3(gq), dim3(256), 0, st, P.cells, P.w, (long long)nq, P.list_off, h->ws_pcount.p, C);
    hipLaunchKernelGGL(k_pair_scan, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, C, h->ws_pstart.p, h->ws_pcursor.p, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_a1_pair_scatter, dim3(gq), dim3(256), 0, st, P.cells, P.w, (long long)nq, P.list_off, h->ws_pstart.p, h->ws_pcursor.p, h->ws_order.p);
    hipLaunchKernelGGL(k_group_build, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, h->ws_pstart.p, C, G, h->ws_gdesc.p, h->ws_gfb.p, (u32 *)(h->ws_gfb.p + 1),
                       (unsigned long long *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_mfma_prep, dim3((unsigned)std::min<long long>(4096, (nq + 255) / 256)), dim3(256), 0, st, (const int32_t *)h->ws_gfb.p, (u32 *)nullptr,
                       h->ws_redo.p, h->ws_mfctl.p, P.T, h->ws_T0.p, P.pool_cnt, h->ws_psnap.p, (long long)nq);
    HIPCK(hipGetLastError());
    DBG_SYNC("K3ma pair sort");
    MfmaParams MP{};
    MP.S = P;
    MP.S.order = h->ws_order.p;
    MP.S.n_order = h->ws_pstart.p + C;
    if (h->d_perm) {
        const long long tot = (long long)nq * h->D;
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, P.Q, h->d_perm, h->ws_Qp.p, h->D, (long long)nq);
        MP.S.Q = h->ws_Qp.p;
        MP.S.coarse = h->d_coarseP;
    }
    MP.S.perm = nullptr;
    MP.R = nullptr;
    if (h->transform == MMIDX_TR_ROTATION) {
        hipLaunchKernelGGL(k_pair_rotate, dim3((unsigned)((npairs + 7) / 8)), dim3(128), 8 * (size_t)h->D * sizeof(double), st, P.Q, P.coarse, h->d_rot, P.cells,
                           MP.S.order, MP.S.n_order, (long long)npairs, P.w, h->D, 1, h->ws_R.p);
        HIPCK(hipGetLastError());
        MP.R = h->ws_R.p;
    }
    MP.pq16 = h->d_pq16;
    MP.xn = h->xn.p;
    MP.pq = h->d_pq;
    MP.flat_lut = nullptr;
    MP.gdesc = h->ws_gdesc.p;
    MP.n_groups = h->ws_gfb.p;
    MP.sub = sub;
    MP.nsub = nsub;
    MP.ep = h->pq_ep;
    MP.xmax = h->rmax;
    MP.ghist = nullptr;
    MP.T0 = h->ws_T0.p;
    MP.surv = nullptr;
    MP.surv_cnt = h->ws_mfctl.p;
    MP.surv_cap = 0;
    MP.redo = h->ws_redo.p;
    MP.pool_snap = h->ws_psnap.p;
    MP.work = h->ws_mfctl.p + 8;
    MP.a_work = h->ws_mfctl.p + 24;  // (sweep 1: words 24 .. 39; sweep 2: 8 .. 23)
    MP.a_wide = 0;  // (sweep 1: every item through the four-wave instance; set for sweep 2 below)
    MP.a_cstride = cstride;
    MP.fb_count = (u32 *)(h->ws_gfb.p + 1);
    MP.fb_items = h->ws_gfb.p + 4;
    MP.fb_ch = h->ws_gfb.p + 4 + nfb;
    MP.fb_chunk = pl.chunk;
    MP.fb_nchunks = std::max(pl.nchunks, 1);
    MP.npairs_flat = npairs;
    MP.stat = (h->profiling == 1 || h->debug_sync) ? (unsigned long long *)(h->d_counters + 3) : nullptr;
    MP.nver = (unsigned long long *)(h->d_counters + 7);
    MP.a_cand = h->ws_acand.p;
    MP.a_rowc = h->ws_arowc.p;
    MP.a_bm = h->ws_abm.p;
    MP.a_bm_stride = bm_stride;
    MP.a_icnt = h->ws_aicnt.p;
    MP.a_rec = h->ws_arec.p;
    MP.a_rec_cap = (u32)rec_cap;
    MP.a_rows = h->transform == MMIDX_TR_ROTATION ? h->ws_R.p : h->ws_arows.p;
    MP.a_meta = h->ws_ameta.p;
    MP.a_metaT = h->ws_ametaT.p;
    MP.a_rnd = h->ws_arnd.p;
    MP.a_rnd_size = rnd_size;
    const MfmaLds L(h->D);
    hipEvent_t *aev = nullptr;
    if (h->profiling == 1 && h->a_ev_used + 5 <= 5 * 4096) {
        while (h->a_ev.size() < h->a_ev_used + 5) {
            hipEvent_t e;
            HIPCK(hipEventCreate(&e));
            h->a_ev.push_back(e);
        }
        aev = h->a_ev.data() + h->a_ev_used;
        h->a_ev_used += 5;
        HIPCK(hipEventRecord(aev[0], st));
    }
    int rc = launch_mfma_a_scan<1>(h, MP, L.total, st);  // sweep 1: the slots' best four per pair
    if (rc) return rc;
    if (aev) HIPCK(hipEventRecord(aev[1], st));
    DBG_SYNC("K3ma sweep 1");
    {
        const unsigned gs = (unsigned)std::min<long long>((npairs + 3) / 4, 32ll * std::max(h->num_cus, 8));  // (the kernel loops over the device-side count)
        const int chunks = nsub * (cstride >> 8);
        if (chunks <= 1) hipLaunchKernelGGL(k_a1_select<1>, dim3(gs), dim3(256), 0, st, MP);
        else if (chunks <= 2) hipLaunchKernelGGL(k_a1_select<2>, dim3(gs), dim3(256), 0, st, MP);
        else if (chunks <= 4) hipLaunchKernelGGL(k_a1_select<4>, dim3(gs), dim3(256), 0, st, MP);
        else if (chunks <= 8) hipLaunchKernelGGL(k_a1_select<8>, dim3(gs), dim3(256), 0, st, MP);
        else hipLaunchKernelGGL(k_a1_select<16>, dim3(gs), dim3(256), 0, st, MP);
        HIPCK(hipGetLastError());
    }
    if (aev) HIPCK(hipEventRecord(aev[2], st));
    DBG_SYNC("K3ma select");
    if (h->debug_sync) HIPCK(hipMemsetAsync(h->ws_abm.p, 0, max_groups * (size_t)nsub * bm_stride, st));  // (the debug report below counts bits)
    MP.a_wide = h->a_wide ? 1 : 0;
    rc = launch_mfma_a_scan<2>(h, MP, L.total, st);  // sweep 2: compare masks
    if (rc) return rc;
    if (aev) HIPCK(hipEventRecord(aev[3], st));
    DBG_SYNC("K3ma sweep 2");
    {
        const int tp = h->D / 2, per = std::max(1, 256 / std::min(tp, 256));
        hipLaunchKernelGGL(k_a1_rows, dim3((unsigned)std::min<long long>((npairs + per - 1) / per, 32ll * std::max(h->num_cus, 8))), dim3(256), 0, st, MP, h->D);
    }
    hipLaunchKernelGGL(k_a1_item_scan, dim3(1), dim3(1024), 0, st, MP);
    hipLaunchKernelGGL(k_a1_records, dim3((unsigned)(8 * std::max(h->num_cus, 8))), dim3(A1R_NT), 0, st, MP);
    HIPCK(hipGetLastError());
    DBG_SYNC("K3ma records");
    if (h->dsub == 4) rc = h->m == 32 ? launch_a1_verify_t<32, 4>(h, MP, st) : h->m == 16 ? launch_a1_verify_t<16, 4>(h, MP, st) : launch_a1_verify_t<8, 4>(h, MP, st);
    else if (h->dsub == 8) rc = h->m == 16 ? launch_a1_verify_t<16, 8>(h, MP, st) : h->m == 8 ? launch_a1_verify_t<8, 8>(h, MP, st) : launch_a1_verify_t<4, 8>(h, MP, st);
    else rc = h->m == 8 ? launch_a1_verify_t<8, 16>(h, MP, st) : h->m == 4 ? launch_a1_verify_t<4, 16>(h, MP, st) : launch_a1_verify_t<2, 16>(h, MP, st);
    if (rc) return rc;
    if (aev) HIPCK(hipEventRecord(aev[4], st));
    DBG_SYNC("K3ma verify");
    h->a_launches++;
    const long long span = std::max<long long>(npairs, nq);
    hipLaunchKernelGGL(k_mfma_redo, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, MP, (long long)nq);
    HIPCK(hipGetLastError());
    DBG_SYNC("K3ma redo");
    if (h->debug_sync) {
        int32_t g2[2], np1 = 0;
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(g2, h->ws_gfb.p, sizeof(g2), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&np1, h->ws_pstart.p + C, 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[mmidx] K3ma: %d pairs in %d groups of <= %d x %d pieces of %d codes, %d (pair, chunk) items handed to K3f (bitmap stride %zu)\n", np1, g2[0], G, nsub, sub,
                g2[1], bm_stride);
        {
            std::vector<unsigned char> hb(max_groups * (size_t)nsub * bm_stride);
            std::vector<u32> pc((size_t)nq);
            (void)hipMemcpy(hb.data(), h->ws_abm.p, hb.size(), hipMemcpyDeviceToHost);
            (void)hipMemcpy(pc.data(), P.pool_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost);
            unsigned long long bits = 0, pool = 0;
            for (unsigned char c : hb) bits += (unsigned)__builtin_popcount(c);
            for (u32 c : pc) pool += c;
            u64 cnt[1] = {0};
            (void)hipMemcpy(cnt, h->d_counters + 3, 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[mmidx] K3ma: %llu bits set by sweep 2, %llu records verified so far (counter), %llu pool entries\n", bits, (unsigned long long)cnt[0], pool);
            if (const char *dump = getenv("MMIDX_K3MA_DUMP")) {  // (debugging aid: groups, order, thresholds and the bitmap to a file)
                FILE *f = fopen(dump, "wb");
                if (f) {
                    std::vector<int32_t> ord((size_t)std::max(np1, 1));
                    std::vector<int4> gd((size_t)std::max(g2[0], 1));
                    std::vector<u64> tt((size_t)nq);
                    (void)hipMemcpy(ord.data(), h->ws_order.p, (size_t)np1 * 4, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(gd.data(), h->ws_gdesc.p, (size_t)g2[0] * 16, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(tt.data(), P.T, (size_t)nq * 8, hipMemcpyDeviceToHost);
                    const long long hdr[8] = {np1, g2[0], nsub, sub, (long long)bm_stride, nq, P.w, 0};
                    fwrite(hdr, 8, 8, f);
                    fwrite(ord.data(), 4, (size_t)np1, f);
                    fwrite(gd.data(), 16, (size_t)g2[0], f);
                    fwrite(tt.data(), 8, (size_t)nq, f);
                    fwrite(hb.data(), 1, (size_t)g2[0] * nsub * bm_stride, f);
                    fclose(f);
                }
            }
        }
    }
    // the queries handed back (device-side count; normally the few whose list is shorter than K1): K3f's looping kernel, from T = +inf
    ScanParams F = P;
    F.order = MP.fb_items;
    F.n_order = (const int32_t *)MP.fb_count;
    F.order_ch = MP.fb_ch;
    F.n_items = (int)std::min<size_t>(nfb, (size_t)0x7fffff00);
    F.xcd_remap = 0;
    return launch_scan_filtered(h, F, pl, dim3((unsigned)F.n_items, 1), st, -1);
}

// pass B over the sorted pairs (P.order / P.n_order as for K3f; per-cell counts and starts in ws_pcount / ws_pstart).
// Returns 1 when K3g does not apply (the caller uses K3f).
// S: the scan parameters K3g runs with; F: those of the K3f launch that serves the handed-back (pair, chunk) items
int launch_grouped_common(mmidx_index *h, const ScanParams &S, ScanParams F, const SearchPlan &pl, int nlists, int nchunks, long long npairs,
                          hipStream_t st, long long nq, const double *flat_lut = nullptr) {
    const int G = h->m >= 32 ? 4 : 8;  // (the u8 rows of a group: G x m x 256 bytes <= 64 KiB, the reach of a ds_read's immediate offset)
    int cb = 1;
    while (cb < pl.K1 + GRP_VR) cb <<= 1;
    const GrpLds L(h->m, G, h->D, cb);
    if (L.total > 160 * 1024 || pl.K1 + GRP_VR > GRP_NT) return 1;
    const size_t nfb = (size_t)npairs * (size_t)nchunks;
    HIPCK(h->ws_gdesc.reserve((size_t)npairs / G + (size_t)nlists + 8));
    HIPCK(h->ws_gfb.reserve(4 + 2 * nfb + 16));
    if (h->debug_sync) HIPCK(hipMemsetAsync(h->ws_gfb.p, 0, (4 + 2 * nfb + 16) * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_group_build, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, h->ws_pstart.p, nlists, G, h->ws_gdesc.p, h->ws_gfb.p,
                       (u32 *)(h->ws_gfb.p + 1), (unsigned long long *)(h->d_counters + 7), h->pin_hint ? h->pin_hint + 1 : nullptr);
    HIPCK(hipGetLastError());
    DBG_SYNC("K3g group build");
    GrpParams GP{};
    GP.S = S;
    GP.pq = h->d_pq;
    GP.pq32T = h->d_pq32T;
    GP.pn32 = h->d_pn32;
    GP.pnmax = h->d_pnmax;
    GP.gdesc = h->ws_gdesc.p;
    GP.n_groups = h->ws_gfb.p;
    GP.nchunks = nchunks;
    GP.fb_count = (u32 *)(h->ws_gfb.p + 1);
    GP.fb_items = h->ws_gfb.p + 4;
    GP.fb_ch = h->ws_gfb.p + 4 + nfb;
    GP.cb = cb;
    GrpExtra GX{};
    GX.flat_lut = flat_lut;
    GX.nver = (unsigned long long *)(h->d_counters + 7);
    // (IVF: only when the previous call left pass B something to do -- the histogram costs a 1 KiB-per-query memset, which the
    //  separable benchmark, whose far probes all fall to the coarse bound, would pay for nothing; a stale hint costs speed, never results)
    // ... and the UNION instance only when the call before verified more than a few codes per query (k_group_build reports it)
    const bool union_worth = !S.ivf || (h->pin_hint && *(volatile int32_t *)h->pin_hint > 0 && (long long)*(volatile int32_t *)(h->pin_hint + 1) > 8 * nq);
    if ((h->no_union < 0 || (!h->no_union && union_worth)) && nq > 0 && nq * 256 * 4 <= (1ll << 31)) {  // (no_union = -1: always, A/B)
        HIPCK(h->ws_ghist.reserve((size_t)nq * 256));
        HIPCK(h->ws_T0.reserve((size_t)nq));
        HIPCK(hipMemsetAsync(h->ws_ghist.p, 0, (size_t)nq * 256 * sizeof(u32), st));
        HIPCK(hipMemcpyAsync(h->ws_T0.p, S.T, (size_t)nq * sizeof(u64), hipMemcpyDeviceToDevice, st));
        GX.ghist = h->ws_ghist.p;
        GX.T0 = h->ws_T0.p;
    }
    GP.stat = (h->profiling == 1 || h->debug_sync) ? (unsigned long long *)(h->d_counters + 3) : nullptr;  // [0] verified, [1] flag (adds), [2..3] item statistics
    int rc;
    const int ds = h->dsub;
    // flat PQ with the queries' exact tables: its own instances (m = 8, 16 -- the others verify from the codebook as IVF does)
    const bool flat_inst = GX.flat_lut && (h->m == 8 || h->m == 16);
    if (!flat_inst) GX.flat_lut = nullptr;
    // the extra pointers live in a small device struct, re-sent only when they change (pinned mirror; one search at a time per handle)
    if (!h->d_grpx) {
        HIPCK(hipMalloc((void **)&h->d_grpx, sizeof(GrpExtra)));
        HIPCK(hipHostMalloc((void **)&h->pin_grpx, sizeof(GrpExtra)));
        memset(h->pin_grpx, 0xFF, sizeof(GrpExtra));
    }
    if (memcmp(h->pin_grpx, &GX, sizeof(GrpExtra)) != 0) {
        HIPCK(hipStreamSynchronize(st));  // (an earlier copy from the mirror may still be in flight)
        memcpy(h->pin_grpx, &GX, sizeof(GrpExtra));
        HIPCK(hipMemcpyAsync(h->d_grpx, h->pin_grpx, sizeof(GrpExtra), hipMemcpyHostToDevice, st));
    }
    GP.extra = (const GrpExtra *)h->d_grpx;
    const bool un = GX.ghist != nullptr;  // (the instance that ranks the union of the verified candidates)
#define GRP_GO(M_, G_, DS_) (un ? launch_grp_t<M_, G_, DS_, false, true>(h, GP, L.total, st) : launch_grp_t<M_, G_, DS_, false, false>(h, GP, L.total, st))
    if (flat_inst) {
        if (h->m == 8)
            rc = ds == 16  ? launch_grp_t<8, 8, 16, true, true>(h, GP, L.total, st)
                 : ds == 8 ? launch_grp_t<8, 8, 8, true, true>(h, GP, L.total, st)
                 : ds == 4 ? launch_grp_t<8, 8, 4, true, true>(h, GP, L.total, st)
                           : launch_grp_t<8, 8, 0, true, true>(h, GP, L.total, st);
        else
            rc = ds == 8   ? launch_grp_t<16, 8, 8, true, true>(h, GP, L.total, st)
                 : ds == 4 ? launch_grp_t<16, 8, 4, true, true>(h, GP, L.total, st)
                           : launch_grp_t<16, 8, 0, true, true>(h, GP, L.total, st);
    } else
    switch (h->m) {
        case 8:
            rc = ds == 16  ? GRP_GO(8, 8, 16)
                 : ds == 8 ? GRP_GO(8, 8, 8)
                 : ds == 4 ? GRP_GO(8, 8, 4)
                           : launch_grp_t<8, 8, 0>(h, GP, L.total, st);
            break;
        case 16:
            rc = ds == 8   ? GRP_GO(16, 8, 8)
                 : ds == 4 ? GRP_GO(16, 8, 4)
                           : launch_grp_t<16, 8, 0>(h, GP, L.total, st);
            break;
        case 64:  // (YFCC100MExample.java:85-90: 1024 dimensions in 64 x 16)
            rc = ds == 16  ? GRP_GO(64, 4, 16)
                 : ds == 8 ? GRP_GO(64, 4, 8)
                           : launch_grp_t<64, 4, 0>(h, GP, L.total, st);
            break;
        default:
            rc = ds == 4 ? GRP_GO(32, 4, 4) : launch_grp_t<32, 4, 0>(h, GP, L.total, st);
            break;
    }
#undef GRP_GO
    if (rc) return rc;
    DBG_SYNC("K3g scan");
    if (h->debug_sync) {
        int32_t c4[2];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(c4, h->ws_gfb.p, sizeof(c4), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mmidx] K3g: %d groups of <= %d pairs, %d (pair, chunk) items handed back to K3f (lds %zu, cb %d)\n", c4[0], G, c4[1],
                L.total, cb);
    }
    // the handed-back items (device-side count; normally few): a small main grid, the looping tail covers the rest
    F.order = GP.fb_items;
    F.n_order = (const int32_t *)GP.fb_count;
    F.order_ch = GP.fb_ch;
    F.n_items = (int)std::min<size_t>(nfb, (size_t)0x7fffff00);
    F.xcd_remap = 0;
    return launch_scan_filtered(h, F, pl, dim3((unsigned)F.n_items, 1), st, -1);
}

// ---- K3m (mmidx_scan_mfma.h): pass B as a certified lower bound on the matrix cores ------------------------------
// index-side tables: the fp16 codebook (per quantizer) and ||x||^2 of every stored code (per CSR build)
int build_mfma_tables(mmidx_index *h) {
    const bool shape_ok = (h->kind == MMIDX_KIND_IVFPQ || h->kind == MMIDX_KIND_PQ) && h->code_bytes == 1 && h->ks <= 256 && h->pq_set &&
                          (((h->dsub == 4 || h->dsub == 8 || h->dsub == 16) && (h->D == 32 || h->D == 64 || h->D == 128)) ||
                           ((h->dsub == 8 || h->dsub == 16) && h->D > 128 && h->D % 128 == 0 && h->D <= 2048 && h->m <= 128));  // (long vectors: K3mk, mmidx_scan_mfma_kc.h)
    if (!shape_ok) {
        h->mfma_ok = false;
        return MMIDX_OK;
    }
    if (!h->mfma_valid) {
        h->mfma_ok = false;
        h->mfma_valid = true;
        // power-of-two scale: the largest |element| lands in [2^12, 2^13)
        int ep = 0;
        if (!(h->pq_maxabs < 1e30) || !(h->rmax < 1e30)) return MMIDX_OK;  // (also NaN: K3g / K3f serve such a codebook)
        if (h->pq_maxabs > 0.0) {
            int ex;
            (void)std::frexp(h->pq_maxabs, &ex);
            ep = 13 - ex;
        }
        if (ep < -110 || ep > 110) return MMIDX_OK;
        h->pq_ep = ep;
        if (!h->d_pq16) HIPCK(hipMalloc((void **)&h->d_pq16, (size_t)h->D * 512));
        if (!h->d_pn64) HIPCK(hipMalloc((void **)&h->d_pn64, (size_t)h->m * 256 * sizeof(double)));
        const int rows = h->D / 8 * 256;
        hipLaunchKernelGGL(k_pq16_table, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, h->d_pqT, h->d_pq16, h->m, h->ks, h->dsub,
                           std::ldexp(1.0, ep));
        hipLaunchKernelGGL(k_pn64_table, dim3((unsigned)h->m), dim3(256), 0, h->stream, h->d_pqT, h->d_pn64, h->m, h->ks, h->dsub);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(h->stream));
        h->mfma_ok = true;
        h->xn_valid = false;
    }
    if (h->mfma_ok && !h->xn_valid && h->n_csr > 0) {
        HIPCK(h->xn.reserve((size_t)h->n_csr));
        hipLaunchKernelGGL(k_code_norms, dim3((unsigned)((h->n_csr + 255) / 256)), dim3(256), 0, h->stream, (const unsigned char *)h->d_codes, h->d_pn64,
                           h->xn.p, h->m, (long long)h->n_csr);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(h->stream));
        h->xn_valid = true;
    }
    return MMIDX_OK;
}

template <int NJ, int DSUB>
int launch_mfma_scan_t(mmidx_index *h, const MfmaParams &MP, size_t lds, hipStream_t st) {
    HIPCK(hipFuncSetAttribute((const void *)k_scan_mfma<NJ, DSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = h->mfma_blocks;
    if (blocks <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_scan_mfma<NJ, DSUB>, MF_NT, lds) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks = occ * std::max(h->num_cus, 8);
    }
    blocks = std::max(8, (blocks + 7) & ~7);
    hipLaunchKernelGGL((k_scan_mfma<NJ, DSUB>), dim3((unsigned)blocks), dim3(MF_NT), lds, st, MP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}
template <int M, int DSUB>
int launch_mfma_verify_t(mmidx_index *h, const MfmaParams &MP, hipStream_t st) {
    const unsigned grid = (unsigned)(8 * std::max(h->num_cus, 8));
    if (MP.flat_lut) hipLaunchKernelGGL((k_mfma_verify<M, DSUB, true>), dim3(grid), dim3(256), 0, st, MP);
    else hipLaunchKernelGGL((k_mfma_verify<M, DSUB, false>), dim3(grid), dim3(256), 0, st, MP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

template <int DSUB, int TPW>
int launch_mfma_kc_scan_t(mmidx_index *h, const MfmaKcParams &KP, size_t lds, hipStream_t st) {
    HIPCK(hipFuncSetAttribute((const void *)k_scan_mfma_kc<DSUB, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks = h->mfma_blocks;
    if (blocks <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_scan_mfma_kc<DSUB, TPW>, MF_NT, lds) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = 1;
        }
        blocks = occ * std::max(h->num_cus, 8);
    }
    blocks = std::max(8, (blocks + 7) & ~7);
    hipLaunchKernelGGL((k_scan_mfma_kc<DSUB, TPW>), dim3((unsigned)blocks), dim3(MF_NT), lds, st, KP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

template <int DSUB, int CG>
int launch_mfma_kc2_scan_t(mmidx_index *h, const MfmaKcParams &KP, hipStream_t st) {
    int blocks = h->mfma_blocks > 0 ? h->mfma_blocks : std::max(h->num_cus, 8);  // one block of 512 threads per CU (156 KiB of LDS)
    blocks = std::max(8, (blocks + 7) & ~7);
    hipLaunchKernelGGL((k_scan_mfma_kc2<DSUB, CG>), dim3((unsigned)blocks), dim3(MFK2_NT), 0, st, KP);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

// K3mk (mmidx_scan_mfma_kc.h): pass B through the matrix-core bound for vectors of several 128-dimension chunks.  Same contract as
// launch_mfma_common (which dispatches here); returns 1 when it does not apply.
int launch_mfma_kc(mmidx_index *h, const ScanParams &S, ScanParams F, const SearchPlan &pl, int nlists, int nchunks_f, long long npairs, long long maxlen,
                   hipStream_t st, long long nq, const double *flat_lut) {
    if (h->transform == MMIDX_TR_ROTATION || (size_t)npairs * h->D * 2 > ((size_t)16 << 30) || h->m % 16 != 0 || h->m > 128) return 1;
    if (S.ivf && h->d_perm && !h->d_coarseP) return 1;  // (every applicability check ahead of the first launch)
    const int G = MFK_G;
    const size_t nfb = (size_t)npairs * (size_t)std::max(nchunks_f, 1);
    HIPCK(h->ws_gdesc.reserve((size_t)npairs / G + (size_t)nlists + 8));
    HIPCK(h->ws_gfb.reserve(4 + 2 * nfb + 16));
    // the DMA form (k_scan_mfma_kc2) where the lanes' code bytes come in aligned words: D a multiple of 256
    const int nb = 32 / h->dsub, quarter = h->m / 4;
    int cg = 0;
    if (!h->mfma_kc_v1 && h->D % 256 == 0 && h->m % 4 == 0)
        for (int c : {8, 4})  // (16 bytes per load would hold 32 registers per wave for the code words: the chunk loop spills)
            if (!cg && quarter % c == 0 && c >= 2 * nb) cg = c;
    const int tpw = cg ? MFK2_TPW * (MFK2_NT / 64) / 4 : (h->mfma_kc_tpw == 16 ? 16 : 8);  // tiles of a piece / 4
    int sub = h->mfma_sub > 0 ? ((h->mfma_sub + 63) & ~63) : 64 * tpw;
    if (cg) sub = h->mfma_sub > 0 ? std::min(sub, 1 << 20) : 8 * MFK2_TPW * (MFK2_NT / 64) * 16;  // k_scan_mfma_kc2 walks an item in passes of 1024 codes
    else sub = std::min(sub, 64 * tpw);  // (a wave holds at most TPW tiles' accumulators)
    const int nsub = (int)((maxlen + sub - 1) / sub);
    if ((long long)(npairs / G + nlists) * nsub > 0x7fffff00ll) return 1;
    hipLaunchKernelGGL(k_group_build, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, h->ws_pstart.p, nlists, G, h->ws_gdesc.p, h->ws_gfb.p,
                       (u32 *)(h->ws_gfb.p + 1), (unsigned long long *)(h->d_counters + 7), h->pin_hint ? h->pin_hint + 1 : nullptr);
    HIPCK(hipGetLastError());
    HIPCK(h->ws_ghist.reserve((size_t)nq * 256));
    HIPCK(h->ws_T0.reserve((size_t)nq));
    HIPCK(h->ws_redo.reserve((size_t)nq));
    HIPCK(h->ws_psnap.reserve((size_t)nq));
    HIPCK(h->ws_mfctl.reserve(64));
    size_t qcap = h->mfma_qcap > 0 ? (size_t)h->mfma_qcap : std::min<size_t>((size_t)1 << 28, std::max<size_t>((size_t)1 << 20, (size_t)nq * 4096));
    HIPCK(h->ws_surv.reserve(qcap));
    HIPCK(h->ws_R16.reserve((size_t)npairs * h->D));
    HIPCK(h->ws_nrow.reserve((size_t)npairs));
    hipLaunchKernelGGL(k_mfma_prep, dim3((unsigned)std::min<long long>(4096, (nq * 64 + 255) / 256)), dim3(256), 0, st, (const int32_t *)h->ws_gfb.p, h->ws_ghist.p,
                       h->ws_redo.p, h->ws_mfctl.p, S.T, h->ws_T0.p, S.pool_cnt, h->ws_psnap.p, (long long)nq);
    HIPCK(hipGetLastError());
    MfmaKcParams KP{};
    MfmaParams &MP = KP.M;
    MP.S = S;
    if (h->d_perm) {
        HIPCK(h->ws_Qp.reserve((size_t)nq * h->D));
        const long long tot = (long long)nq * h->D;
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, S.Q, h->d_perm, h->ws_Qp.p, h->D, (long long)nq);
        MP.S.Q = h->ws_Qp.p;
        if (S.ivf) MP.S.coarse = h->d_coarseP;
    }
    MP.S.perm = nullptr;
    // the launch's residual scale: |r_i| <= max |centroid element| + max |query element| (ctl words 40: query maximum, 44..45: scale)
    u32 *d_qmax = h->ws_mfctl.p + 40;
    int32_t *d_scale = (int32_t *)(h->ws_mfctl.p + 44);
    HIPCK(hipMemsetAsync(d_qmax, 0, sizeof(u32), st));
    hipLaunchKernelGGL(k_maxabs_f64, dim3(256), dim3(256), 0, st, S.Q, (long long)nq * h->D, d_qmax);
    float cmaxf = S.ivf ? (float)h->coarse_maxabs : 0.f;
    if (S.ivf && (double)cmaxf < h->coarse_maxabs) cmaxf = std::nextafter(cmaxf, INFINITY);
    hipLaunchKernelGGL(k_resid_scale, dim3(1), dim3(1), 0, st, (const u32 *)d_qmax, cmaxf, h->pq_ep, d_scale);
    const long long nrows = npairs;
    hipLaunchKernelGGL(k_pair_resid16, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st, MP.S.Q, MP.S.coarse, S.cells, S.order, S.n_order, (long long)nrows, S.w,
                       h->D, S.ivf, (const int32_t *)d_scale, h->ws_R16.p, h->ws_nrow.p);
    HIPCK(hipGetLastError());
    MP.pq16 = h->d_pq16;
    MP.xn = h->xn.p;
    MP.pq = h->d_pq;
    MP.flat_lut = flat_lut;
    MP.R = nullptr;
    MP.gdesc = h->ws_gdesc.p;
    MP.n_groups = h->ws_gfb.p;
    MP.sub = sub;
    MP.nsub = nsub;
    MP.ep = h->pq_ep;
    MP.xmax = h->rmax;
    MP.ghist = h->ws_ghist.p;
    MP.T0 = h->ws_T0.p;
    MP.surv = h->ws_surv.p;
    MP.surv_cnt = h->ws_mfctl.p;
    MP.surv_cap = (u32)std::min<size_t>(qcap, 0xFFFFFFF0u);
    MP.redo = h->ws_redo.p;
    MP.pool_snap = h->ws_psnap.p;
    MP.work = h->ws_mfctl.p + 8;
    MP.fb_count = (u32 *)(h->ws_gfb.p + 1);
    MP.fb_items = h->ws_gfb.p + 4;
    MP.fb_ch = h->ws_gfb.p + 4 + nfb;
    MP.fb_chunk = pl.chunk;
    MP.fb_nchunks = std::max(nchunks_f, 1);
    MP.npairs_flat = npairs;
    MP.stat = (h->profiling == 1 || h->debug_sync) ? (unsigned long long *)(h->d_counters + 3) : nullptr;
    MP.nver = (unsigned long long *)(h->d_counters + 7);
    KP.R16 = h->ws_R16.p;
    KP.nrow = h->ws_nrow.p;
    KP.scale = d_scale;
    KP.D = h->D;
    const MfmaKcLds L;
    hipEvent_t *mev = nullptr;
    if (h->profiling == 1 && h->mf_ev_used + 3 <= 3 * 4096) {
        while (h->mf_ev.size() < h->mf_ev_used + 3) {
            hipEvent_t e;
            HIPCK(hipEventCreate(&e));
            h->mf_ev.push_back(e);
        }
        mev = h->mf_ev.data() + h->mf_ev_used;
        h->mf_ev_used += 3;
        HIPCK(hipEventRecord(mev[0], st));
    }
    int rc;
    if (cg && h->dsub == 16) rc = cg == 8 ? launch_mfma_kc2_scan_t<16, 8>(h, KP, st) : launch_mfma_kc2_scan_t<16, 4>(h, KP, st);
    else if (cg) rc = launch_mfma_kc2_scan_t<8, 8>(h, KP, st);
    else if (h->dsub == 16) rc = tpw == 8 ? launch_mfma_kc_scan_t<16, 8>(h, KP, L.total, st) : launch_mfma_kc_scan_t<16, 16>(h, KP, L.total, st);
    else rc = tpw == 8 ? launch_mfma_kc_scan_t<8, 8>(h, KP, L.total, st) : launch_mfma_kc_scan_t<8, 16>(h, KP, L.total, st);
    if (rc) return rc;
    if (mev) HIPCK(hipEventRecord(mev[1], st));
    DBG_SYNC("K3mk scan");
    rc = 1;
#define KC_VER(MM)                                                                                                     \
    case MM: rc = h->dsub == 16 ? launch_mfma_verify_t<MM, 16>(h, MP, st) : launch_mfma_verify_t<MM, 8>(h, MP, st); break;
    switch (h->m) {  // (16 lanes per survivor, m / 16 sub-quantizers each)
        KC_VER(8) KC_VER(16) KC_VER(32) KC_VER(48) KC_VER(64) KC_VER(80) KC_VER(96) KC_VER(112) KC_VER(128)
        default: break;
    }
#undef KC_VER
    if (rc) return rc < 0 ? rc : fail(MMIDX_ERR_UNSUPPORTED, "K3mk: no verification instance for m = %d", h->m);
    if (mev) HIPCK(hipEventRecord(mev[2], st));
    if (MP.stat) hipLaunchKernelGGL(k_mfma_count, dim3(1024), dim3(256), 0, st, MP);
    DBG_SYNC("K3mk verify");
    const long long span = std::max<long long>(npairs, nq);
    hipLaunchKernelGGL(k_mfma_redo, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, MP, (long long)nq);
    HIPCK(hipGetLastError());
    if (h->debug_sync) {
        u32 c16[16];
        int32_t g2[2], sc[2];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(c16, h->ws_mfctl.p, sizeof(c16), hipMemcpyDeviceToHost);
        (void)hipMemcpy(g2, h->ws_gfb.p, sizeof(g2), hipMemcpyDeviceToHost);
        (void)hipMemcpy(sc, d_scale, sizeof(sc), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mmidx] K3mk: %d groups of <= %d pairs x %d pieces of %d codes, %u survivor slots (cap %u), %d items handed to the redo, scale 2^%d ok %d\n", g2[0], G, nsub, sub,
                c16[0], MP.surv_cap, g2[1], sc[0], sc[1]);
    }
    F.order = MP.fb_items;
    F.n_order = (const int32_t *)MP.fb_count;
    F.order_ch = MP.fb_ch;
    F.n_items = (int)std::min<size_t>(nfb, (size_t)0x7fffff00);
    F.xcd_remap = 0;
    return launch_scan_filtered(h, F, pl, dim3((unsigned)F.n_items, 1), st, -1);
}

// pass B over the sorted pairs through K3m.  S: the scan parameters as K3g would get them (list offsets / order / centroids of the
// IVF or the flat-PQ form); F: those of the K3f launch that redoes the queries K3m hands back.  Returns 1 when K3m does not apply.
int launch_mfma_common(mmidx_index *h, const ScanParams &S, ScanParams F, const SearchPlan &pl, int nlists, int nchunks_f, long long npairs,
                       long long maxlen, hipStream_t st, long long nq, const double *flat_lut) {
    if (h->no_mfma || !h->mfma_ok || !h->xn_valid || h->no_filter || S.sdc_tt || nq <= 0 || npairs <= 0 || maxlen >= (1ll << 31) ||
        nq * 256 * 4 > (1ll << 31) || npairs >= 0x7fffff00ll || ((uintptr_t)S.Q & 15) != 0)
        return 1;
    if (h->D > 128) return launch_mfma_kc(h, S, F, pl, nlists, nchunks_f, npairs, maxlen, st, nq, flat_lut);
    // (every applicability check ahead of the first launch: a fallback after k_group_build / k_mfma_prep would pay them twice and write
    //  the hint words twice)
    if (h->d_perm && S.ivf && !h->d_coarseP) return 1;
    if (h->transform == MMIDX_TR_ROTATION && (!h->d_rot || (size_t)npairs * h->D * 8 > ((size_t)8 << 30))) return 1;
    constexpr int G = MF_QG;
    const size_t nfb = (size_t)npairs * (size_t)std::max(nchunks_f, 1);
    HIPCK(h->ws_gdesc.reserve((size_t)npairs / G + (size_t)nlists + 8));
    HIPCK(h->ws_gfb.reserve(4 + 2 * nfb + 16));
    hipLaunchKernelGGL(k_group_build, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, h->ws_pstart.p, nlists, G, h->ws_gdesc.p, h->ws_gfb.p,
                       (u32 *)(h->ws_gfb.p + 1), (unsigned long long *)(h->d_counters + 7), h->pin_hint ? h->pin_hint + 1 : nullptr);
    HIPCK(hipGetLastError());
    DBG_SYNC("K3m group build");
    // items: (group, piece of `sub` codes).  One piece per list where the batch fills the chip anyway; shorter pieces when there
    // are few groups (small calls, flat PQ's long chunks), so that the persistent blocks all find work
    int sub = h->mfma_sub;
    if (sub <= 0) {
        const long long est_groups = npairs / G + std::min<long long>(npairs, nlists);
        const long long want = 4ll * 2 * std::max(h->num_cus, 8);  // (four items per resident block; eight cost cfg2 a third piece per chunk: 1.31 -> 1.21 ms)
        long long pieces = std::max<long long>(1, (want + est_groups - 1) / est_groups);
        long long sb = (maxlen + pieces - 1) / pieces;
        sb = std::max<long long>(1024, std::min<long long>(sb, 65536));
        sub = (int)((sb + 63) & ~63ll);
    }
    sub = (std::max(sub, 64) + 15) & ~15;
    const int nsub = (int)((maxlen + sub - 1) / sub);
    HIPCK(h->ws_ghist.reserve((size_t)nq * 256));
    HIPCK(h->ws_T0.reserve((size_t)nq));
    HIPCK(h->ws_redo.reserve((size_t)nq));
    HIPCK(h->ws_psnap.reserve((size_t)nq));
    HIPCK(h->ws_mfctl.reserve(64));
    size_t qcap = h->mfma_qcap > 0 ? (size_t)h->mfma_qcap : std::min<size_t>((size_t)1 << 28, std::max<size_t>((size_t)1 << 20, (size_t)nq * 2048));
    HIPCK(h->ws_surv.reserve(qcap));
    hipLaunchKernelGGL(k_mfma_prep, dim3((unsigned)std::min<long long>(4096, (nq * 64 + 255) / 256)), dim3(256), 0, st, (const int32_t *)h->ws_gfb.p, h->ws_ghist.p,
                       h->ws_redo.p, h->ws_mfctl.p, S.T, h->ws_T0.p, S.pool_cnt, h->ws_psnap.p, (long long)nq);
    HIPCK(hipGetLastError());
    MfmaParams MP{};
    MP.S = S;
    if (h->d_perm) {  // rows in transformed order: contiguous loads (the centroids once per index, the queries once per call)
        HIPCK(h->ws_Qp.reserve((size_t)nq * h->D));
        const long long tot = (long long)nq * h->D;
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, S.Q, h->d_perm, h->ws_Qp.p, h->D, (long long)nq);
        MP.S.Q = h->ws_Qp.p;
        if (S.ivf) MP.S.coarse = h->d_coarseP;
    }
    MP.S.perm = nullptr;
    MP.R = nullptr;
    if (h->transform == MMIDX_TR_ROTATION) {
        // the pairs' exact rotated residuals, once per call (16 k multiply-adds per pair at D = 128 instead of one per survivor)
        HIPCK(h->ws_R.reserve((size_t)npairs * h->D));
        hipLaunchKernelGGL(k_pair_rotate, dim3((unsigned)((npairs + 7) / 8)), dim3(128), 8 * (size_t)h->D * sizeof(double), st, S.Q, S.coarse, h->d_rot, S.cells,
                           S.order, S.n_order, (long long)npairs, S.w, h->D, S.ivf, h->ws_R.p);
        HIPCK(hipGetLastError());
        MP.R = h->ws_R.p;
        MP.flat_lut = nullptr;  // (k_flat_lut knows permutations only: the survivors' entries come from the rotated rows)
    }
    MP.pq16 = h->d_pq16;
    MP.xn = h->xn.p;
    MP.pq = h->d_pq;
    if (h->transform != MMIDX_TR_ROTATION) MP.flat_lut = flat_lut;
    MP.gdesc = h->ws_gdesc.p;
    MP.n_groups = h->ws_gfb.p;
    MP.sub = sub;
    MP.nsub = nsub;
    MP.ep = h->pq_ep;
    MP.xmax = h->rmax;
    MP.ghist = h->ws_ghist.p;
    MP.T0 = h->ws_T0.p;
    MP.surv = h->ws_surv.p;
    MP.surv_cnt = h->ws_mfctl.p;
    MP.surv_cap = (u32)std::min<size_t>(qcap, 0xFFFFFFF0u);
    MP.redo = h->ws_redo.p;
    MP.pool_snap = h->ws_psnap.p;
    MP.work = h->ws_mfctl.p + 8;
    MP.fb_count = (u32 *)(h->ws_gfb.p + 1);
    MP.fb_items = h->ws_gfb.p + 4;
    MP.fb_ch = h->ws_gfb.p + 4 + nfb;
    MP.fb_chunk = pl.chunk;
    MP.fb_nchunks = std::max(nchunks_f, 1);
    MP.npairs_flat = npairs;
    MP.stat = (h->profiling == 1 || h->debug_sync) ? (unsigned long long *)(h->d_counters + 3) : nullptr;
    MP.nver = (unsigned long long *)(h->d_counters + 7);
    const MfmaLds L(h->D);
    int rc;
    hipEvent_t *mev = nullptr;
    if (h->profiling == 1 && h->mf_ev_used + 3 <= 3 * 4096) {
        while (h->mf_ev.size() < h->mf_ev_used + 3) {
            hipEvent_t e;
            HIPCK(hipEventCreate(&e));
            h->mf_ev.push_back(e);
        }
        mev = h->mf_ev.data() + h->mf_ev_used;
        h->mf_ev_used += 3;
        HIPCK(hipEventRecord(mev[0], st));
    }
    const int nj = h->D / 32;
    if (h->dsub == 4) rc = nj == 4 ? launch_mfma_scan_t<4, 4>(h, MP, L.total, st) : nj == 2 ? launch_mfma_scan_t<2, 4>(h, MP, L.total, st) : launch_mfma_scan_t<1, 4>(h, MP, L.total, st);
    else if (h->dsub == 8) rc = nj == 4 ? launch_mfma_scan_t<4, 8>(h, MP, L.total, st) : nj == 2 ? launch_mfma_scan_t<2, 8>(h, MP, L.total, st) : launch_mfma_scan_t<1, 8>(h, MP, L.total, st);
    else rc = nj == 4 ? launch_mfma_scan_t<4, 16>(h, MP, L.total, st) : nj == 2 ? launch_mfma_scan_t<2, 16>(h, MP, L.total, st) : launch_mfma_scan_t<1, 16>(h, MP, L.total, st);
    if (rc) return rc;
    if (mev) HIPCK(hipEventRecord(mev[1], st));
    DBG_SYNC("K3m scan");
    if (h->dsub == 4) rc = h->m == 32 ? launch_mfma_verify_t<32, 4>(h, MP, st) : h->m == 16 ? launch_mfma_verify_t<16, 4>(h, MP, st) : launch_mfma_verify_t<8, 4>(h, MP, st);
    else if (h->dsub == 8) rc = h->m == 16 ? launch_mfma_verify_t<16, 8>(h, MP, st) : h->m == 8 ? launch_mfma_verify_t<8, 8>(h, MP, st) : launch_mfma_verify_t<4, 8>(h, MP, st);
    else rc = h->m == 8 ? launch_mfma_verify_t<8, 16>(h, MP, st) : h->m == 4 ? launch_mfma_verify_t<4, 16>(h, MP, st) : launch_mfma_verify_t<2, 16>(h, MP, st);
    if (rc) return rc;
    if (mev) HIPCK(hipEventRecord(mev[2], st));
    if (MP.stat) hipLaunchKernelGGL(k_mfma_count, dim3(1024), dim3(256), 0, st, MP);  // (profiling runs only)
    DBG_SYNC("K3m verify");
    const long long span = std::max<long long>(npairs, nq);
    hipLaunchKernelGGL(k_mfma_redo, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, MP, (long long)nq);
    HIPCK(hipGetLastError());
    DBG_SYNC("K3m redo");
    if (h->debug_sync) {
        u32 c16[16];
        int32_t g2[2];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(c16, h->ws_mfctl.p, sizeof(c16), hipMemcpyDeviceToHost);
        (void)hipMemcpy(g2, h->ws_gfb.p, sizeof(g2), hipMemcpyDeviceToHost);
        fprintf(stderr, "[mmidx] K3m: %d groups of <= %d pairs x %d pieces of %d codes, %u survivors (cap %u), %d (pair, chunk) items handed to K3f (lds %zu)\n", g2[0], G,
                nsub, sub, c16[0], MP.surv_cap, g2[1], L.total);
    }
    // the queries K3m handed back (device-side count; normally none): the looping tail kernel alone
    F.order = MP.fb_items;
    F.n_order = (const int32_t *)MP.fb_count;
    F.order_ch = MP.fb_ch;
    F.n_items = (int)std::min<size_t>(nfb, (size_t)0x7fffff00);
    F.xcd_remap = 0;
    return launch_scan_filtered(h, F, pl, dim3((unsigned)F.n_items, 1), st, -1);
}

// pass B over the sorted pairs (P.order / P.n_order as for K3f; per-cell counts and starts in ws_pcount / ws_pstart).
// Returns 1 when K3g does not apply (the caller uses K3f).
int launch_scan_grouped(mmidx_index *h, const ScanParams &P, const SearchPlan &pl, long long nq, long long npairs, hipStream_t st) {
    // The call before kept (next to) no pair behind the coarse bound -- the separable benchmark: every call.  K3m would walk an empty
    // pass B in six launches of ~5 us (group build, prep, scan, verification, redo, K3f tail: the host cannot know the count); K3f's
    // looping kernel alone serves whatever the count turns out to be in ONE launch -- exactly, if slowly should the guess be wrong
    // (the next call sees the real count).  Option "passb_small" = 0: off.
    if (P.ivf && h->passb_small && h->pin_hint && h->hint_calls > 1 && h->code_bytes == 1 && h->ks <= 256 && (h->m == 8 || h->m == 16 || h->m == 32) &&
        !h->no_filter && !P.sdc_tt && !h->debug_sync && h->hint_prev_nq == (int64_t)nq && h->hint_prev_w == P.w) {
        const int32_t seen = *(volatile int32_t *)h->pin_hint;
        if (seen >= 0 && seen <= 64) {
            const unsigned gx = (unsigned)(((P.n_items + 7) / 8) * 8);
            h->disp_passb = "K3f(one looping launch: the call before kept <= 64 pairs)";
            return launch_scan_filtered(h, P, pl, dim3(gx, (unsigned)pl.nchunks), st, -1);
        }
    }
    if (P.ivf) {
        h->disp_passb = h->D > 128 ? "K3mk" : "K3m";
        const int rcm = launch_mfma_common(h, P, P, pl, h->C, pl.nchunks, npairs, h->max_list_len, st, nq, nullptr);
        if (rcm != 1) return rcm;
    }
    h->disp_passb = "K3f";
    if (h->no_grp || !h->grp_valid || !h->d_pq32T || h->no_filter || P.sdc_tt || !P.ivf || h->max_list_len >= (1 << 24)) return 1;
    h->disp_passb = "K3g";
    return launch_grouped_common(h, P, P, pl, h->C, pl.nchunks, npairs, st, nq);
}

// flat PQ pass B (chunks 1 .. of every query) through K3g: the chunks stand in for inverted lists (k_flat_pairs), the
// residual is the query itself.  P: the flat scan parameters (P.w = number of chunks).  Returns 1 when K3g does not apply.
int launch_scan_grouped_flat(mmidx_index *h, const ScanParams &P, const SearchPlan &pl, long long nq, hipStream_t st) {
    const int nch = P.w;
    const long long npairs = nq * (long long)(nch - 1);
    if (h->no_filter || P.sdc_tt || P.ivf || nch < 2 || pl.chunk >= (1 << 24) || npairs >= 0x7fffff00ll || nq * (long long)nch >= 0x7fffff00ll) return 1;
    const bool grp_ok = !h->no_grp && h->grp_valid && h->d_pq32T;
    const bool mf_ok = !h->no_mfma && h->mfma_ok && h->xn_valid;
    if (!grp_ok && !mf_ok) return 1;
    HIPCK(h->ws_pcount.reserve((size_t)nch + 1));
    HIPCK(h->ws_pstart.reserve((size_t)nch + 1));
    HIPCK(h->ws_order.reserve((size_t)npairs));
    HIPCK(h->ws_flatoff.reserve((size_t)nch + 1));
    if (!h->d_zero) {
        HIPCK(hipMalloc((void **)&h->d_zero, (size_t)h->D * sizeof(double)));
        HIPCK(hipMemsetAsync(h->d_zero, 0, (size_t)h->D * sizeof(double), st));
    }
    const long long work = std::max<long long>(npairs, nch + 1);
    hipLaunchKernelGGL(k_flat_pairs, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, (long long)nq, nch, (long long)pl.chunk,
                       (long long)h->n_csr, h->ws_order.p, h->ws_pcount.p, h->ws_pstart.p, h->ws_flatoff.p);
    HIPCK(hipGetLastError());
    ScanParams S = P;
    S.coarse = h->d_zero;
    S.list_off = h->ws_flatoff.p;
    S.order = h->ws_order.p;
    S.n_order = nullptr;
    S.chunk = pl.chunk;
    // one exact table per query for the whole list (PQ.java:300): the survivors of every chunk are verified by table reads
    const double *flat_lut = nullptr;
    if (h->dsub <= 64 && (size_t)nq * h->m * 256 * 8 <= ((size_t)2 << 30)) {
        HIPCK(h->ws_flatlut.reserve((size_t)nq * h->m * 256));
        for (long long q0 = 0; q0 < nq; q0 += 65535) {  // (grid.y <= 65535)
            const dim3 g((unsigned)h->m, (unsigned)std::min<long long>(65535, nq - q0));
            const double *Qc = P.Q + (size_t)q0 * h->D;
            double *Lc = h->ws_flatlut.p + (size_t)q0 * h->m * 256;
            switch (h->dsub) {
                case 4: hipLaunchKernelGGL(k_flat_lut<4>, g, dim3(256), 0, st, Qc, h->d_perm, h->d_pqT, Lc, h->D, h->m, h->ks, h->dsub); break;
                case 8: hipLaunchKernelGGL(k_flat_lut<8>, g, dim3(256), 0, st, Qc, h->d_perm, h->d_pqT, Lc, h->D, h->m, h->ks, h->dsub); break;
                case 16: hipLaunchKernelGGL(k_flat_lut<16>, g, dim3(256), 0, st, Qc, h->d_perm, h->d_pqT, Lc, h->D, h->m, h->ks, h->dsub); break;
                default: hipLaunchKernelGGL(k_flat_lut<0>, g, dim3(256), 0, st, Qc, h->d_perm, h->d_pqT, Lc, h->D, h->m, h->ks, h->dsub); break;
            }
        }
        HIPCK(hipGetLastError());
        flat_lut = h->ws_flatlut.p;
    }
    if (mf_ok) {
        h->disp_passb = h->D > 128 ? "K3mk" : "K3m";
        const int rcm = launch_mfma_common(h, S, P, pl, nch, 1, npairs, pl.chunk, st, nq, flat_lut);
        if (rcm != 1) return rcm;
    }
    h->disp_passb = "K3f";
    if (!grp_ok) return 1;
    h->disp_passb = "K3g";
    return launch_grouped_common(h, S, P, pl, nch, 1, npairs, st, nq, flat_lut);
}

int run_coarse(mmidx_index *h, int64_t nq, const double *dQ, int32_t *d_cells, hipStream_t st) {
    constexpr int QT = 8;
    HIPCK(h->ws_cdist.reserve((size_t)nq * h->C));
    h->cdsel_valid = false;
    // the exact stage of the certified selection stages the terms of `cch` candidates at a time in LDS: MMIDX_CAND_CHUNK of them
    // where they fit a 64 KiB block, fewer for long vectors (D = 1024: 6 -- YFCC100MExample.java:85)
    const size_t sel_fixed = (size_t)MMIDX_CSEL_CAP * 12 + (size_t)(h->w + 1) * 8 + (size_t)((h->w + 2) & ~1) * 4 + 16;
    const size_t row_bytes = (size_t)(h->D + MMIDX_TERM_PAD) * 8;
    int cch = 0;
    size_t alds = 0;
    const bool approx = coarse_certified(h, &alds, &cch);
    const size_t glds = sel_fixed + std::max<size_t>((size_t)std::max(cch, 1) * row_bytes, (size_t)MMIDX_BLOCK * 4);
    const int G = h->Cp / 8;
    if (approx && !h->coarse_v1 && h->d_Ch && G >= 4 * (h->w + 1) && glds <= 64 * 1024) {
        // K1e + K1f: bf16-split dot products on the matrix cores, group minima only, certified candidates in fp64
        HIPCK(h->ws_qn.reserve((size_t)nq));
        HIPCK(h->ws_Qh.reserve((size_t)nq * h->Dp));
        HIPCK(h->ws_Ql.reserve((size_t)nq * h->Dp));
        HIPCK(h->ws_gmin.reserve((size_t)nq * G * 2));
        HIPCK(h->ws_cdsel.reserve((size_t)nq * h->w));
        HIPCK(h->ws_defer.reserve((size_t)nq + 1));
        hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)((nq + 3) / 4)), dim3(MMIDX_BLOCK), 0, st, dQ, (__bf16 *)h->ws_Qh.p, (__bf16 *)h->ws_Ql.p,
                           (float *)nullptr, h->ws_qn.p, h->D, h->Dp, (long long)nq, h->ws_defer.p);
        const int ntiles = h->Cp / G16_BC;
        const int qblocks = (int)((nq + G16_BQ - 1) / G16_BQ);
        const int csplit = std::max(1, std::min(ntiles, (512 + qblocks - 1) / qblocks));
        if (h->Dp == G16_KC && !h->coarse_nodma) {
            // one k chunk: centroid tiles by LDS-DMA into two half-tile buffers (static LDS)
            hipLaunchKernelGGL(k_coarse_gmin16_dma, dim3((unsigned)qblocks, (unsigned)csplit), dim3(MMIDX_BLOCK), 0, st, (const __bf16 *)h->ws_Qh.p,
                               (const __bf16 *)h->ws_Ql.p, (const __bf16 *)h->d_Ch, (const __bf16 *)h->d_Cl, h->d_cn_pad, h->ws_qn.p,
                               (float2 *)h->ws_gmin.p, h->Cp, (int)nq, G);
        } else if (h->Dp > G16_KC && h->Dp % G16_KC == 0 && h->coarse_dma_kc) {
            // several k chunks (long vectors): the same double buffering, (chunk, half tile) after (chunk, half tile)
            hipLaunchKernelGGL(k_coarse_gmin16_dma_kc, dim3((unsigned)qblocks, (unsigned)csplit), dim3(MMIDX_BLOCK), 0, st, (const __bf16 *)h->ws_Qh.p,
                               (const __bf16 *)h->ws_Ql.p, (const __bf16 *)h->d_Ch, (const __bf16 *)h->d_Cl, h->d_cn_pad, h->ws_qn.p,
                               (float2 *)h->ws_gmin.p, h->Cp, (int)nq, G, h->Dp);
        } else {
            const size_t l16 = 2 * (size_t)G16_BC * G16_STRIDE;
            HIPCK(hipFuncSetAttribute((const void *)k_coarse_gmin16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16));
            hipLaunchKernelGGL(k_coarse_gmin16, dim3((unsigned)qblocks, (unsigned)csplit), dim3(MMIDX_BLOCK), l16, st, (const __bf16 *)h->ws_Qh.p,
                               (const __bf16 *)h->ws_Ql.p, (const __bf16 *)h->d_Ch, (const __bf16 *)h->d_Cl, h->d_cn_pad, h->ws_qn.p,
                               (float2 *)h->ws_gmin.p, h->Cp, h->Dp, (int)nq, G);
        }
        HIPCK(hipGetLastError());
        ApproxSel A{};
        A.cand_chunk = cch;
        A.qn = h->ws_qn.p;
        A.cnorm_max = h->cnorm_max;
        A.cn_max = h->cn_max;
        A.Q = dQ;
        A.coarse = h->d_coarse;
        A.coarseT = h->d_coarseT;
        A.row_scratch = h->ws_cdist.p;
        A.cells = d_cells;
        A.cdsel = h->ws_cdsel.p;
        A.C = h->C;
        A.D = h->D;
        A.w = h->w;
        A.gpair = (const float2 *)h->ws_gmin.p;
        A.G = G;
        A.Dp = h->Dp;
        A.nq = (int)nq;
        // front end and selection as two kernels where a query's candidates fit the list between them (about 1.3 (w + 1))
        const bool split = !h->coarse_fused && G <= 1024 && 2 * (h->w + 1) + 32 < MMIDX_CLIST;
        if (split) {
            HIPCK(h->ws_clist.reserve((size_t)nq * MMIDX_CLIST));
            A.clist = (u32 *)h->ws_clist.p;
            // front end + exact stage by one wave per query where its LDS tile applies (option "coarse_wave_sel", default on); what it
            // leaves (more than 64 candidates) goes through the block-per-query form
            A.defer = h->ws_defer.p;
            const bool wave_sel = h->coarse_wave_sel && h->w < 64 && (h->D == 64 || h->D == 128 || h->D == 256);
            const unsigned fgrid = (unsigned)((nq + 3) / 4);
            h->disp_coarse = wave_sel ? "K1e'+K1f(front_sel)" : "K1e'+K1f(front+select_list)";
            if (wave_sel && h->D == 128) hipLaunchKernelGGL((k_coarse_front_sel<8>), dim3((unsigned)nq), dim3(64), 0, st, A);  // (a wave a block)
            else if (wave_sel && h->D == 64) hipLaunchKernelGGL((k_coarse_front_sel<4>), dim3((unsigned)nq), dim3(64), 0, st, A);
            else if (wave_sel) hipLaunchKernelGGL((k_coarse_front_sel<16>), dim3((unsigned)nq), dim3(64), 0, st, A);
            else hipLaunchKernelGGL(k_coarse_front, dim3(fgrid), dim3(MMIDX_BLOCK), 0, st, A);
            if (wave_sel) {
                const unsigned dgrid = (unsigned)std::min<long long>(nq, 2ll * std::max(h->num_cus, 8));
                if (h->C <= 8 * MMIDX_BLOCK) hipLaunchKernelGGL(k_coarse_select_defer<8>, dim3(dgrid), dim3(MMIDX_BLOCK), glds, st, A);
                else hipLaunchKernelGGL(k_coarse_select_defer<32>, dim3(dgrid), dim3(MMIDX_BLOCK), glds, st, A);
            } else if (h->C <= 8 * MMIDX_BLOCK)
                hipLaunchKernelGGL(k_coarse_select_list<8>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), glds, st, A);
            else
                hipLaunchKernelGGL(k_coarse_select_list<32>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), glds, st, A);
        } else if (h->C <= 8 * MMIDX_BLOCK) {
            h->disp_coarse = "K1e'+K1f(select_grp)";
            hipLaunchKernelGGL(k_coarse_select_grp<8>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), glds, st, A);
        }
        else if (h->C <= 32 * MMIDX_BLOCK) {
            h->disp_coarse = "K1e'+K1f(select_grp)";
            hipLaunchKernelGGL(k_coarse_select_grp<32>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), glds, st, A);
        } else {
            h->disp_coarse = "K1e'+K1f(select_grp)";
            hipLaunchKernelGGL(k_coarse_select_grp<64>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), glds, st, A);
        }
        HIPCK(hipGetLastError());
        h->cdsel_valid = true;
        return MMIDX_OK;
    }
    if (approx) {
        h->disp_coarse = "K1c+K1d";
        // K1c + K1d: fp32 dot products for all centroids, fp64 only for the certified candidates
        HIPCK(h->ws_Q32.reserve((size_t)nq * h->D));
        HIPCK(h->ws_qn.reserve((size_t)nq));
        HIPCK(h->ws_S.reserve((size_t)nq * h->C));
        HIPCK(h->ws_cdsel.reserve((size_t)nq * h->w));
        hipLaunchKernelGGL(k_query_prep, dim3((unsigned)((nq + 3) / 4)), dim3(MMIDX_BLOCK), 0, st, dQ, h->ws_Q32.p, h->ws_qn.p, h->D, (long long)nq);
        dim3 g1((unsigned)((h->C + DOT_BN - 1) / DOT_BN), (unsigned)((nq + DOT_BM - 1) / DOT_BM));
        hipLaunchKernelGGL(k_coarse_dot32, g1, dim3(MMIDX_BLOCK), 0, st, h->d_coarseT32, h->ws_Q32.p, h->d_cn, h->ws_qn.p, h->ws_S.p, h->C, h->D, (int)nq);
        ApproxSel A{};
        A.cand_chunk = cch;
        A.S = h->ws_S.p;
        A.qn = h->ws_qn.p;
        A.cnorm_max = h->cnorm_max;
        A.cn_max = h->cn_max;
        A.Q = dQ;
        A.coarse = h->d_coarse;
        A.coarseT = h->d_coarseT;
        A.row_scratch = h->ws_cdist.p;
        A.cells = d_cells;
        A.cdsel = h->ws_cdsel.p;
        A.C = h->C;
        A.D = h->D;
        A.w = h->w;
        if (h->C <= 8 * MMIDX_BLOCK)
            hipLaunchKernelGGL(k_coarse_select_approx<8>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), alds, st, A);
        else if (h->C <= 32 * MMIDX_BLOCK)
            hipLaunchKernelGGL(k_coarse_select_approx<32>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), alds, st, A);
        else
            hipLaunchKernelGGL(k_coarse_select_approx<64>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), alds, st, A);
        HIPCK(hipGetLastError());
        h->cdsel_valid = true;
        return MMIDX_OK;
    }
    h->disp_coarse = "K1a+K1b(exact)";
    dim3 g1((unsigned)((h->C + MMIDX_BLOCK - 1) / MMIDX_BLOCK), (unsigned)((nq + QT - 1) / QT));
    if ((int64_t)g1.x * g1.y < 2048) {  // few centroids and queries: a query a block, so that the chip has blocks to run
        g1.y = (unsigned)nq;
        hipLaunchKernelGGL(k_coarse_dist<1>, g1, dim3(MMIDX_BLOCK), 0, st, h->d_coarseT, dQ, h->ws_cdist.p, h->C, h->D, (int)nq);
    } else
        hipLaunchKernelGGL(k_coarse_dist<QT>, g1, dim3(MMIDX_BLOCK), 0, st, h->d_coarseT, dQ, h->ws_cdist.p, h->C, h->D, (int)nq);
    const size_t lds = (size_t)(h->w + 1) * 12 + 16;
    if (lds > 64 * 1024) return fail(MMIDX_ERR_UNSUPPORTED, "w = %d too large", h->w);
    const bool fast = h->C >= MMIDX_BLOCK && h->w + 1 <= MMIDX_BLOCK && h->C <= 64 * MMIDX_BLOCK;
    const size_t flds = (size_t)MMIDX_CSEL_CAP * 12 + (size_t)(h->w + 1) * 12 + 16;
    if (fast && h->C <= 8 * MMIDX_BLOCK)
        hipLaunchKernelGGL(k_coarse_select_fast<8>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), flds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    else if (fast && h->C <= 32 * MMIDX_BLOCK)
        hipLaunchKernelGGL(k_coarse_select_fast<32>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), flds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    else if (fast)
        hipLaunchKernelGGL(k_coarse_select_fast<64>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), flds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    else if (h->C <= 8 * MMIDX_BLOCK)
        hipLaunchKernelGGL(k_coarse_select_reg<8>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), lds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    else if (h->C <= 32 * MMIDX_BLOCK)
        hipLaunchKernelGGL(k_coarse_select_reg<32>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), lds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    else if (h->C <= 64 * MMIDX_BLOCK)
        hipLaunchKernelGGL(k_coarse_select_reg<64>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), lds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    else
        hipLaunchKernelGGL(k_coarse_select, dim3((unsigned)nq), dim3(MMIDX_BLOCK), lds, st, h->ws_cdist.p, h->C, h->w, d_cells);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}


// one sub-batch (nq <= plan.qb) entirely on device
int search_batch_device(mmidx_index *h, const SearchPlan &pl, int k, int64_t nq, const double *dQ, const int32_t *d_cells_in,
                        int mode, int32_t *d_iid, double *d_dist, int32_t *d_cnt, double *d_pdist, long long *d_pkey,
                        int phase, double *d_T_io, hipStream_t st, const double *sdc_tt = nullptr, const double *d_cdsel_in = nullptr) {
    // phase 0: whole search.  Sharded search splits it so that the thresholds can be MIN-reduced
    // across ranks in between: phase 1 = setup + pass A + export T, phase 2 = import T + pass B + merge.
    const int ivf = h->kind == MMIDX_KIND_IVFPQ;
    // profiling 1: all six events + the code counters; 2 ("light"): only the two events around pass A -- every event
    // record is a ~5 us bubble in the stream, and a throughput run should carry as few as the roofline figure needs
    bool prof = h->profiling == 1, plight = h->profiling == 2;
    hipEvent_t *ev = nullptr;
    if (prof || plight) {
        if (h->ev_used + 6 > 6 * 4096) {
            prof = plight = false;  // event pool exhausted: call mmidx_get_stats to drain it
        } else {
            while (h->evpool.size() < h->ev_used + 6) {
                hipEvent_t e;
                HIPCK(hipEventCreate(&e));
                h->evpool.push_back(e);
            }
            ev = h->evpool.data() + h->ev_used;
            h->ev_used += 6;
            if (prof) HIPCK(hipEventRecord(ev[0], st));
        }
    }
    const int32_t *d_cells = d_cells_in;
    if (ivf && !d_cells) {
        HIPCK(h->ws_cells.reserve((size_t)nq * h->w));
        int rc = run_coarse(h, nq, dQ, h->ws_cells.p, st);
        if (rc) return rc;
        d_cells = h->ws_cells.p;
    }
    if (prof) HIPCK(hipEventRecord(ev[1], st));
    HIPCK(h->ws_T.reserve((size_t)nq));
    HIPCK(h->ws_pcnt.reserve((size_t)nq));
    HIPCK(h->ws_pkey.reserve((size_t)nq * pl.poolq));
    HIPCK(h->ws_pval.reserve((size_t)nq * pl.poolq));
    HIPCK(h->ws_flag.reserve((size_t)nq));
    bool pcount_zeroed = false;
    if (phase != 2) {
        // thresholds, pool counters, per-cell pair counters and the K3h fallback header in one launch
        int32_t *pc = nullptr;
        if (ivf) {
            HIPCK(h->ws_pcount.reserve((size_t)h->C + 1));
            pc = h->ws_pcount.p;
            pcount_zeroed = phase == 0;  // (phase 1 ends before the pair sort; phase 2 zeroes them itself)
        }
        HIPCK(h->ws_fb.reserve(2 * (size_t)nq * (size_t)std::max(1, pl.nchunks) + 4));
        const long long span = std::max<long long>((long long)nq, ivf ? (long long)h->C + 1 : 0);
        hipLaunchKernelGGL(k_step_init, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, h->ws_T.p, h->ws_pcnt.p, (long long)nq, pc,
                           h->C, h->ws_fb.p);
        HIPCK(hipGetLastError());
    }

    ScanParams P{};
    P.Q = dQ;
    P.coarse = h->d_coarse;
    P.pqT = h->d_pqT;
    P.sdc_tt = sdc_tt;
    P.perm = h->d_perm;
    P.rot = h->d_rot;
    P.cells = ivf ? d_cells : nullptr;
    P.list_off = h->d_off;
    P.codes = h->d_codes;
    P.T = h->ws_T.p;
    P.pool_cnt = h->ws_pcnt.p;
    P.pool_key = h->ws_pkey.p;
    P.pool_val = h->ws_pval.p;
    P.D = h->D;
    P.m = h->m;
    P.ks = h->ks;
    P.dsub = h->dsub;
    P.w = ivf ? h->w : 1;
    P.transform = h->transform;
    P.ivf = ivf;
    P.glut = pl.glut ? h->ws_glut.p : nullptr;
    P.chunk = pl.chunk;
    P.code_lo = 0;
    P.code_hi = 0x7fffffff;
    P.K1 = pl.K1;
    P.cap = pl.cap;
    P.poolq = pl.poolq;
    if (h->n_csr > 0) {
        // flat PQ: the chunks of the single list play the role of probes (P.w = number of chunks,
        // item rank = chunk index, cell = 0), so the same two passes apply: chunk 0 fixes the
        // threshold, the other chunks run through the filtered scan
        const int grid_chunks = ivf ? pl.nchunks : 1;
        if (!ivf) P.w = pl.nchunks;
        const long long npairs = (long long)nq * P.w;
        const bool two_pass = P.w > 1;
        if (prof || plight) {
            HIPCK(hipEventRecord(ev[2], st));
            if (prof || phase == 2) HIPCK(hipEventRecord(ev[5], st));  // (recorded again behind pass A when it runs)
        }
        // pass A: probe rank 0 of every query (all of them for PQ) -- fixes a tight threshold
        P.order = nullptr;
        P.rank_lo = 0;
        P.nrank = two_pass ? 1 : P.w;
        P.n_items = (int)(nq * P.nrank);
        P.xcd_remap = 0;
        // (the adaptive filter also works from T = +inf, but for one list per query the exact scan
        //  measured faster: 1.23 vs ~1.4 ms per 8192 queries; MMIDX_PASSA_FILTER=1 switches)
        int rc = MMIDX_OK;
        if (phase != 2) {
            h->disp_passb = "-";
            h->disp_pre = "-";
            if (!ivf) h->disp_coarse = "-";
            if (two_pass && passa_q_applies(h, P, pl, (long long)nq) && !(h->passa_mfma > 0 && passa_mfma_applies(h, P, pl, (long long)nq))) {
                // K3q: the queries of a nearest list four to a block, decided on integers (mmidx_scan_q.h)
                h->disp_passa = "K3q";
                rc = launch_passa_q(h, P, pl, (long long)nq, st);
                // (its pair sort counted in ws_pcount -- and k_q_scan_groups zeroed the counters again: pcount_zeroed stands)
            } else if (two_pass && passa_mfma_applies(h, P, pl, (long long)nq)) {
                // K3ma: >= 8 queries per nearest list -- the list-major matrix-core form (mmidx_scan_mfma_a.h)
                h->disp_passa = "K3ma";
                rc = launch_passa_mfma(h, P, pl, (long long)nq, st);
                pcount_zeroed = false;  // (its pair sort counted in ws_pcount)
            } else if (h->passa_filter) {
                h->disp_passa = "K3f";
                rc = launch_scan_filtered(h, P, pl, dim3((unsigned)P.n_items, (unsigned)grid_chunks), st);
            } else if (!h->no_seed) {
                h->disp_passa = "K3seed";
                rc = launch_scan_seeded(h, P, pl, dim3((unsigned)P.n_items, (unsigned)grid_chunks), st);
            } else if (two_pass && !sdc_tt && h->passa_hist != 0 &&
                       (h->passa_hist > 0 || h->n_csr / std::max<int64_t>(1, ivf ? h->nonempty_lists : 1) >= 4096) &&
                       (rc = launch_scan_hist(h, P, pl, dim3((unsigned)P.n_items, (unsigned)grid_chunks), st)) != 1) {
                // (K3h ran -- its empty fallback launch is not counted as a scan launch -- or failed with rc > 1)
                h->disp_passa = "K3h";
            } else {
                h->disp_passa = pl.glut ? ((!h->no_split_table && h->code_bytes == 1 && h->ks == 256 && h->m == 128) ? "K3(table in two halves)" : "K3(table in global scratch)")
                                        : (two_pass ? "K3" : "K3(single pass)");
                rc = MMIDX_OK;
                // one code per thread per segment: smaller candidate buffer -> a fourth block per CU
                ScanParams PA = P;
                int su = 2;
                size_t lds_a = pl.lds;
                if (two_pass && ivf && !h->passa_su2 && !pl.glut && h->code_bytes == 1 && (h->m == 8 || h->m == 16 || h->m == 32)) {  // (the GLUT kernels are SU = 2 only: their buffer keeps pl.cap)
                    const int nt = h->passa_512 ? 512 : MMIDX_BLOCK;
                    int cap1 = 1;
                    while (cap1 < pl.K1 + nt) cap1 <<= 1;
                    PA.cap = cap1;
                    lds_a = scan_lds_bytes(h, cap1);
                    su = h->passa_512 ? 11 : 1;
                }
                // prefix mode: the exact scan (LDS-bound fp64 gather) covers only the first passa_prefix
                // codes of the nearest list -- enough for a useful threshold -- and the rest of that
                // list goes through the filtered scan under it
                const bool prefix = two_pass && ivf && h->passa_prefix > 0 && !sdc_tt;
                if (prefix) PA.code_hi = h->passa_prefix;
                rc = launch_scan(h, PA, dim3((unsigned)P.n_items, (unsigned)grid_chunks), lds_a, st, su);
                if (rc) return rc;
                if (prefix) {
                    ScanParams PR = P;
                    PR.code_lo = h->passa_prefix;
                    rc = launch_scan_filtered(h, PR, pl, dim3((unsigned)P.n_items, (unsigned)grid_chunks), st);
                    if (prof) h->launches += 1;
                }
            }
            if (rc) return rc;
            DBG_SYNC("pass A scan");
            if (prof || plight) {
                HIPCK(hipEventRecord(ev[5], st));
                h->passa_launches += 1;
                if (!ivf) h->host_passa_codes += nq * std::min<int64_t>(h->n_csr, two_pass ? (int64_t)pl.chunk : h->n_csr);
            }
        }
        if (phase == 1) {
            hipLaunchKernelGGL(k_T_export, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, h->ws_T.p, d_T_io, (long long)nq);
            HIPCK(hipGetLastError());
            if (prof) {
                HIPCK(hipEventRecord(ev[3], st));
                HIPCK(hipEventRecord(ev[4], st));
                h->launches += 1;
            }
            return MMIDX_OK;
        }
        if (phase == 2) {
            hipLaunchKernelGGL(k_T_import, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, d_T_io, h->ws_T.p, (long long)nq);
            HIPCK(hipGetLastError());
        }
        if (two_pass && !ivf) {
            // flat PQ pass B: chunks 1.. of every query in natural order
            P.order = nullptr;
            P.rank_lo = 1;
            P.nrank = P.w - 1;
            P.n_items = (int)(nq * P.nrank);
            P.xcd_remap = 0;
            h->disp_passb = "K3f";
            rc = launch_scan_grouped_flat(h, P, pl, (long long)nq, st);
            if (rc == 1) rc = launch_scan_filtered(h, P, pl, dim3((unsigned)P.n_items, 1), st);
            if (rc) return rc;
            DBG_SYNC("pass B scan (flat)");
        }
        if (two_pass && ivf) {
            // pass B order: pairs with probe rank >= 1 that survive the coarse bound, sorted by cell
            // (device counting sort; needs the thresholds pass A just produced)
            HIPCK(h->ws_pcount.reserve((size_t)h->C + 1));
            HIPCK(h->ws_pstart.reserve((size_t)h->C + 1));
            HIPCK(h->ws_pcursor.reserve((size_t)h->C));
            HIPCK(h->ws_order.reserve((size_t)npairs));
            HIPCK(h->ws_keep.reserve((size_t)npairs));
            if (!pcount_zeroed) HIPCK(hipMemsetAsync(h->ws_pcount.p, 0, ((size_t)h->C + 1) * sizeof(int32_t), st));
            PairBound PB{};
            PB.Q = dQ;
            PB.coarse = h->d_coarse;
            PB.T = h->ws_T.p;
            PB.cdsel = (ivf && !d_cells_in && h->cdsel_valid) ? h->ws_cdsel.p : ((ivf && d_cells_in) ? d_cdsel_in : nullptr);
            PB.cdist = (ivf && !d_cells_in && !h->cdsel_valid) ? h->ws_cdist.p : nullptr;
            PB.list_off = h->d_off;
            PB.rmax = h->rmax;
            PB.D = h->D;
            PB.C = h->C;
            PB.enabled = ((h->transform != MMIDX_TR_ROTATION || h->rot_shrink > 0.0) && !h->no_bound) ? 1 : 0;
            PB.shrink = h->transform == MMIDX_TR_ROTATION ? h->rot_shrink : 1.0;
            const unsigned g = (unsigned)((npairs + 255) / 256);
            DBG_SYNC("pair memset");
            if (h->debug_sync) {
                std::vector<u64> ht((size_t)nq);
                std::vector<int32_t> hc((size_t)npairs);
                (void)hipMemcpy(ht.data(), h->ws_T.p, (size_t)nq * 8, hipMemcpyDeviceToHost);
                (void)hipMemcpy(hc.data(), d_cells, (size_t)npairs * 4, hipMemcpyDeviceToHost);
                int ninf = 0, cmin = 1 << 30, cmax = -(1 << 30);
                for (u64 t : ht) ninf += t >= 0x7FF0000000000000ull;
                for (int32_t c : hc) { cmin = std::min(cmin, c); cmax = std::max(cmax, c); }
                fprintf(stderr, "[mmidx] nq=%lld npairs=%lld T inf=%d cells min=%d max=%d rmax=%g D=%d w=%d keep=%p cnt=%p Q=%p coarse=%p T=%p\n", (long long)nq, npairs, ninf, cmin, cmax,
                        h->rmax, h->D, P.w, (void*)h->ws_keep.p, (void*)h->ws_pcount.p, (void*)dQ, (void*)h->d_coarse, (void*)h->ws_T.p);
            }
            // K3s: pairs whose certified Smin reaches the threshold leave before the sort (only where K3g would run, and only when
            // the figures the device reported for the call before say it pays; a stale figure costs speed, never results)
            const bool pre_ok = !h->no_grp && h->grp_valid && h->d_pq32T && !h->no_filter && !P.sdc_tt && h->max_list_len < (1 << 24) &&
                                (h->dsub == 4 || h->dsub == 8 || h->dsub == 16) && npairs < 0x7fffff00ll && ((uintptr_t)dQ & 15) == 0;  // (16-byte loads of the query rows)
            bool use_pre = false;
            if (pre_ok && h->smin_pre > 0) use_pre = true;
            else if (pre_ok && h->smin_pre < 0 && h->D > 128 && h->mfma_ok && !h->no_mfma && h->xn_valid) {
                // in front of K3mk: the bound costs a tenth of the scan it can save, and there are no K3g figures to go by -- on, unless
                // the last run removed less than an eighth of the pairs it looked at ([2] in, [0] left): then seven calls go without
                use_pre = true;
                if (h->pin_hint) {
                    volatile int32_t *ph = (volatile int32_t *)h->pin_hint;
                    if (h->pre_skip > 0) {
                        h->pre_skip--;
                        use_pre = false;
                    } else if (h->pre_on) {
                        const long long in = ph[2], left = ph[0];
                        if (in > 0 && left * 8 > in * 7) {
                            h->pre_skip = 7;
                            use_pre = false;
                        }
                    }
                }
            } else if (pre_ok && h->smin_pre < 0 && h->pin_hint) {
                volatile int32_t *ph = (volatile int32_t *)h->pin_hint;
                if (h->pre_on) {  // [2] pairs K3s looked at, [0] pairs it left: it stays while it removes a quarter
                    const long long in = ph[2], left = ph[0];
                    use_pre = !(in > 0 && left * 4 > in * 3);
                } else {  // [4] pairs K3g grouped, [5] pairs alive after its table build
                    const long long in = ph[4], left = ph[5];
                    use_pre = in > 0 && left * 2 < in;
                }
            }
            h->pre_on = use_pre;
            h->disp_pre = use_pre ? "K3s" : "-";
            if (use_pre) {
                HIPCK(h->ws_cand.reserve((size_t)npairs + 4));
                HIPCK(h->ws_smin.reserve((size_t)npairs));
                int32_t *ncand = h->ws_cand.p + npairs;
                const int32_t *cand_list = h->ws_cand.p;
                HIPCK(hipMemsetAsync(ncand, 0, sizeof(int32_t), st));
                const int sgroups = h->m / SMIN_NW;
                if (sgroups > 1) HIPCK(hipMemsetAsync(h->ws_smin.p, 0, (size_t)npairs * sizeof(double), st));
                hipLaunchKernelGGL(k_pair_hist, dim3(g), dim3(256), 0, st, d_cells, P.w, 1, npairs, h->ws_pcount.p, h->ws_keep.p, PB, h->ws_cand.p, ncand);
                DBG_SYNC("pair hist (candidates)");
                SminParams SP{};
                SP.Q = dQ;
                SP.coarse = h->d_coarse;
                SP.perm = h->d_perm;
                if (h->d_perm && h->d_coarseP && !h->smin_valu) {  // transformed copies: contiguous loads instead of 8-byte gathers
                    HIPCK(h->ws_Qp.reserve((size_t)nq * h->D));
                    const long long tot = (long long)nq * h->D;
                    hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dQ, h->d_perm, h->ws_Qp.p, h->D, (long long)nq);
                    SP.Q = h->ws_Qp.p;
                    SP.coarse = h->d_coarseP;
                    SP.perm = nullptr;
                }
                SP.cells = d_cells;
                SP.cand = h->ws_cand.p;
                SP.ncand = ncand;
                SP.pq32T = h->d_pq32T;
                SP.pn32 = h->d_pn32;
                SP.pnmax = h->d_pnmax;
                SP.smin = h->ws_smin.p;
                SP.D = h->D;
                SP.w = P.w;
                SP.M = h->m;
                // (two blocks per CU, but no more blocks than batches: a one-query call would otherwise have 500 blocks load their codebook slices for nothing)
                const long long nbatch = (npairs + 31) / 32;
                const dim3 sg((unsigned)sgroups, (unsigned)std::max<long long>(1, std::min<long long>(2 * std::max(h->num_cus, 8) / sgroups, nbatch)));
                int32_t *hint_cand = h->pin_hint ? h->pin_hint + 2 : nullptr;
                if (h->dsub == 16 && h->smin_bf16 && !h->smin_valu) {
                    // first stage: the bf16 bound over every candidate; what it cannot drop becomes the fp32 stage's candidate list
                    HIPCK(h->ws_cand2.reserve((size_t)npairs + 4));
                    HIPCK(h->ws_smin1.reserve((size_t)npairs));
                    int32_t *ncand2 = h->ws_cand2.p + npairs;
                    HIPCK(hipMemsetAsync(ncand2, 0, sizeof(int32_t), st));
                    if (sgroups > 1) HIPCK(hipMemsetAsync(h->ws_smin1.p, 0, (size_t)npairs * sizeof(double), st));
                    SminParams S1 = SP;
                    S1.smin = h->ws_smin1.p;
                    hipLaunchKernelGGL(k_pair_smin_bf16, sg, dim3(SMIN_NW * 64), 0, st, S1);
                    hipLaunchKernelGGL(k_pair_filter1, dim3(g), dim3(256), 0, st, h->ws_cand.p, ncand, P.w, h->ws_T.p, h->ws_smin1.p, h->ws_keep.p,
                                       h->ws_cand2.p, ncand2, hint_cand);
                    HIPCK(hipGetLastError());
                    DBG_SYNC("K3s bf16 stage");
                    SP.cand = h->ws_cand2.p;
                    SP.ncand = ncand2;
                    ncand = ncand2;
                    cand_list = h->ws_cand2.p;
                    hint_cand = nullptr;  // (the figure the host steers by is the first stage's input)
                }
                if (h->smin_valu) {  // (A/B: the packed-FMA form)
                    if (h->dsub == 16) hipLaunchKernelGGL(k_pair_smin<16>, sg, dim3(SMIN_NW * 64), 0, st, SP);
                    else if (h->dsub == 8) hipLaunchKernelGGL(k_pair_smin<8>, sg, dim3(SMIN_NW * 64), 0, st, SP);
                    else hipLaunchKernelGGL(k_pair_smin<4>, sg, dim3(SMIN_NW * 64), 0, st, SP);
                } else {
                    if (h->dsub == 16) hipLaunchKernelGGL(k_pair_smin_mfma<16>, sg, dim3(SMIN_NW * 64), 0, st, SP);
                    else if (h->dsub == 8) hipLaunchKernelGGL(k_pair_smin_mfma<8>, sg, dim3(SMIN_NW * 64), 0, st, SP);
                    else hipLaunchKernelGGL(k_pair_smin_mfma<4>, sg, dim3(SMIN_NW * 64), 0, st, SP);
                }
                HIPCK(hipGetLastError());
                DBG_SYNC("K3s pair smin");
                if (getenv("MMIDX_SMIN_DUMP")) {  // (debugging aid: how far above the thresholds the bounds lie)
                    (void)hipStreamSynchronize(st);
                    int32_t nc = 0;
                    (void)hipMemcpy(&nc, ncand, 4, hipMemcpyDeviceToHost);
                    std::vector<double> sm((size_t)std::max(nc, 1));
                    std::vector<int32_t> cd((size_t)std::max(nc, 1));
                    std::vector<u64> tt((size_t)nq);
                    (void)hipMemcpy(sm.data(), h->ws_smin.p, (size_t)nc * 8, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(cd.data(), cand_list, (size_t)nc * 4, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(tt.data(), h->ws_T.p, (size_t)nq * 8, hipMemcpyDeviceToHost);
                    long long hist[12] = {0};
                    for (int i = 0; i < nc; i++) {
                        double td;
                        memcpy(&td, &tt[(size_t)(cd[(size_t)i] / P.w)], 8);
                        const double r = sm[(size_t)i] / td;
                        int b = r < 1.0 ? 0 : (r < 1.25 ? 1 : (r < 1.5 ? 2 : (r < 2 ? 3 : (r < 3 ? 4 : (r < 4 ? 5 : (r < 6 ? 6 : (r < 8 ? 7 : 8)))))));
                        hist[b]++;
                    }
                    fprintf(stderr, "[mmidx] K3s: %d candidates; Smin_lo / T: <1: %lld, <1.25: %lld, <1.5: %lld, <2: %lld, <3: %lld, <4: %lld, <6: %lld, <8: %lld, more: %lld\n", nc,
                            hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
                }
                hipLaunchKernelGGL(k_pair_recount, dim3(g), dim3(256), 0, st, cand_list, ncand, d_cells, P.w, h->ws_T.p, h->ws_smin.p, h->ws_keep.p,
                                   h->ws_pcount.p, h->C, hint_cand);
                DBG_SYNC("pair recount");
            } else {
                hipLaunchKernelGGL(k_pair_hist, dim3(g), dim3(256), 0, st, d_cells, P.w, 1, npairs, h->ws_pcount.p, h->ws_keep.p, PB, (int32_t *)nullptr,
                                   (int32_t *)nullptr);
            }
            DBG_SYNC("pair hist");
            hipLaunchKernelGGL(k_pair_scan, dim3(1), dim3(1024), 0, st, h->ws_pcount.p, h->C, h->ws_pstart.p, h->ws_pcursor.p, h->pin_hint);
            if (h->hint_calls < (1 << 30)) h->hint_calls++;  // (this call's count reaches pin_hint[0] when the kernel has run)
            h->hint_prev_nq = h->hint_last_nq;  // (what the count the next stage reads was measured on: the call before this one)
            h->hint_prev_w = h->hint_last_w;
            h->hint_last_nq = nq;
            h->hint_last_w = P.w;
            DBG_SYNC("pair scan");
            hipLaunchKernelGGL(k_pair_scatter, dim3(g), dim3(256), 0, st, d_cells, P.w, 1, npairs, h->ws_pstart.p, h->ws_pcursor.p,
                               h->ws_order.p, h->ws_keep.p);
            HIPCK(hipGetLastError());
            DBG_SYNC("pair sort");
            if (h->debug_sync) {
                int32_t nsurv = 0;
                (void)hipMemcpy(&nsurv, h->ws_pstart.p + h->C, 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "[mmidx] pass B: %d of %lld (query, probe) pairs survive the coarse bound (rmax %.4f)\n", nsurv,
                        (long long)nq * (P.w - 1), h->rmax);
            }
            // pass B: the surviving pairs, list-major; the grid covers the worst case, blocks past
            // the device-side count exit at once
            P.order = h->ws_order.p;
            P.n_order = h->ws_pstart.p + h->C;
            P.n_items = (int)(nq * (P.w - 1));
            P.xcd_remap = 1;
            h->disp_passb = "K3f";
            rc = launch_scan_grouped(h, P, pl, (long long)nq, npairs, st);
            if (rc == 1) {
                const unsigned gx = (unsigned)(((P.n_items + 7) / 8) * 8);
                rc = launch_scan_filtered(h, P, pl, dim3(gx, (unsigned)pl.nchunks), st);
            }
            if (rc) return rc;
            DBG_SYNC("pass B scan");
        }
        if (prof) HIPCK(hipEventRecord(ev[3], st));
        if (prof) h->launches += (two_pass ? 1 : 0) + (phase != 2 ? 1 : 0);
    } else {
        if (plight) {
            HIPCK(hipEventRecord(ev[2], st));
            HIPCK(hipEventRecord(ev[5], st));
        }
        if (prof) {
            HIPCK(hipEventRecord(ev[2], st));
            HIPCK(hipEventRecord(ev[5], st));
            HIPCK(hipEventRecord(ev[3], st));
        }
        if (phase == 1) {
            hipLaunchKernelGGL(k_T_export, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, h->ws_T.p, d_T_io, (long long)nq);
            HIPCK(hipGetLastError());
            if (prof) HIPCK(hipEventRecord(ev[4], st));
            return MMIDX_OK;
        }
    }
    MergeParams M{};
    M.T = h->ws_T.p;
    M.pool_cnt = h->ws_pcnt.p;
    M.pool_key = h->ws_pkey.p;
    M.pool_val = h->ws_pval.p;
    M.poolq = pl.poolq;
    M.cells = ivf ? d_cells : nullptr;
    M.list_off = h->d_off;
    M.ids = h->d_ids;
    M.w = P.w;
    M.k = k;
    M.mode = mode;
    M.iid_out = d_iid;
    M.dist_out = d_dist;
    M.count_out = d_cnt;
    M.flag_out = h->ws_flag.p;
    M.pdist = d_pdist;
    M.pkey = d_pkey;
    if (mode == 1 && phase == 2 && h->shard_dest) {  // a shard of a sharded handle: the lists go to the queries' owners
        M.dest = h->shard_dest;
        M.dest_per = h->shard_dest_per;
        M.dest_me = h->shard_dest_me;
    }
    int mcap = 512;
    while (mcap < 2 * pl.K1) mcap <<= 1;  // <= 8192 (k <= MMIDX_K_MAX)
    M.cap = mcap;
    const size_t mlds = (size_t)mcap * 16 + ((ivf && P.w <= 1024) ? (size_t)P.w * 8 : 0);
    if (pl.K1 <= 128) {
        HIPCK(hipFuncSetAttribute((const void *)k_merge<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(k_merge<128>, dim3((unsigned)nq), dim3(128), mlds, st, M);
    } else {
        HIPCK(hipFuncSetAttribute((const void *)k_merge<MMIDX_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(k_merge<MMIDX_BLOCK>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), mlds, st, M);
    }
    HIPCK(hipGetLastError());
    DBG_SYNC("merge");
    if (mode == 0 && h->n_csr > 0) {
        // exact replay for queries whose k-th / (k+1)-th distances tie (rare)
        TieParams TP{};
        TP.S = P;
        TP.flag = h->ws_flag.p;
        TP.ids = h->d_ids;
        TP.iid_out = d_iid;
        TP.dist_out = d_dist;
        TP.k = k;
        TP.nq = (int)nq;
        const unsigned tgrid = (unsigned)std::min<int64_t>((nq + MMIDX_BLOCK - 1) / MMIDX_BLOCK, 4096);  // (a block reads MMIDX_BLOCK flags at once; flagged queries are rare)
        const size_t tlds = (P.glut ? 0 : (size_t)h->m * h->ks * 8) + 2 * (size_t)h->D * 8;
        if (P.glut) {  // (the table in global scratch: one slot per query of the sub-batch)
            if ((size_t)nq > h->glut_slots) return fail(MMIDX_ERR_UNSUPPORTED, "lookup-table scratch too small for the tie replay");
            if (sdc_tt) {
                hipLaunchKernelGGL((k_tie_resolve<unsigned char, true, true>), dim3(tgrid), dim3(MMIDX_BLOCK), tlds, st, TP);
            } else if (h->code_bytes == 1) {
                hipLaunchKernelGGL((k_tie_resolve<unsigned char, false, true>), dim3(tgrid), dim3(MMIDX_BLOCK), tlds, st, TP);
            } else {
                hipLaunchKernelGGL((k_tie_resolve<unsigned short, false, true>), dim3(tgrid), dim3(MMIDX_BLOCK), tlds, st, TP);
            }
        } else if (sdc_tt) {
            HIPCK(hipFuncSetAttribute((const void *)k_tie_resolve<unsigned char, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
            hipLaunchKernelGGL((k_tie_resolve<unsigned char, true>), dim3(tgrid), dim3(MMIDX_BLOCK), tlds, st, TP);
        } else if (h->code_bytes == 1) {
            HIPCK(hipFuncSetAttribute((const void *)k_tie_resolve<unsigned char, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
            hipLaunchKernelGGL((k_tie_resolve<unsigned char, false>), dim3(tgrid), dim3(MMIDX_BLOCK), tlds, st, TP);
        } else {
            HIPCK(hipFuncSetAttribute((const void *)k_tie_resolve<unsigned short, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
            hipLaunchKernelGGL((k_tie_resolve<unsigned short, false>), dim3(tgrid), dim3(MMIDX_BLOCK), tlds, st, TP);
        }
        HIPCK(hipGetLastError());
    }
    if (prof) {
        HIPCK(hipEventRecord(ev[4], st));
        if (!ivf) h->host_codes += nq * h->n_csr;
        if (ivf || mode == 0) {
            const long long tot = (long long)nq * h->w;
            hipLaunchKernelGGL(k_count_stats, dim3(256), dim3(256), 0, st, ivf ? d_cells : nullptr, h->d_off, tot, h->w,
                               mode == 0 ? h->ws_flag.p : nullptr, (long long)nq, h->d_counters);
        }
        HIPCK(hipGetLastError());
    }
    return MMIDX_OK;
}

int search_common(mmidx_index *h, int k, int64_t nq, const double *dQ, const int32_t *d_cells, int mode, int32_t *d_iid,
                  double *d_dist, int32_t *d_cnt, double *d_pdist, long long *d_pkey, hipStream_t st,
                  const double *sdc_tt = nullptr) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (nq < 0) return fail(MMIDX_ERR_INVALID_ARG, "nq < 0");
    const int ivf = h->kind == MMIDX_KIND_IVFPQ;
    if (ivf && (h->w < 1 || h->w > h->C))
        return fail(MMIDX_ERR_INVALID_ARG, "w = %d outside 1..%d (setW)", h->w, h->C);
    rc = set_device(h);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        rc = build_csr(h);
        if (rc) return rc;
        rc = build_grp_tables(h);
        if (rc) return rc;
        rc = build_mfma_tables(h);
        if (rc) return rc;
    }
    SearchPlan pl;
    rc = make_plan(h, k, nq, pl, d_cells == nullptr);
    if (rc) return rc;
    for (int64_t q0 = 0; q0 < nq; q0 += pl.qb) {
        const int64_t nb = std::min<int64_t>(pl.qb, nq - q0);
        rc = search_batch_device(h, pl, k, nb, dQ + (size_t)q0 * h->D, d_cells ? d_cells + (size_t)q0 * h->w : nullptr, mode,
                                 d_iid ? d_iid + (size_t)q0 * k : nullptr, d_dist ? d_dist + (size_t)q0 * k : nullptr,
                                 d_cnt + q0, d_pdist ? d_pdist + (size_t)q0 * pl.K1 : nullptr,
                                 d_pkey ? d_pkey + (size_t)q0 * pl.K1 : nullptr, 0, nullptr, st,
                                 sdc_tt ? sdc_tt + (size_t)q0 * h->m * h->ks * h->dsub : nullptr);
        if (rc) return rc;
    }
    return MMIDX_OK;
}

// K5 on `st` (the device is current): dense [nshards][nq][k+1] lists, or ragged with d_poff
int launch_merge_partials(int k, int64_t nq, int nshards, const double *d_pdist, const int64_t *d_pkey, const int32_t *d_pcount,
                          const int64_t *d_poff, int32_t *d_iid_out, double *d_dist_out, int32_t *d_count_out, int32_t *d_flag_out,
                          int32_t *d_nflag_out, hipStream_t st) {
    // buffer: room for the kept prefix plus at least one more shard's list.  With at most 128 entries over all shards in the
    // usual case (pass B drops everything above the global threshold) small blocks keep more queries in flight per CU.
    const bool small = k + 1 <= 128;
    int mcap = small ? 512 : MMIDX_MCAP;
    while (mcap < 2 * (k + 1)) mcap <<= 1;
    const size_t mlds = (size_t)mcap * 16;
    if (small) {
        HIPCK(hipFuncSetAttribute((const void *)k_merge_partials<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(k_merge_partials<128>, dim3((unsigned)nq), dim3(128), mlds, st, k, (int)nq, nshards, mcap, d_pdist,
                           (const long long *)d_pkey, d_pcount, (const long long *)d_poff, d_iid_out, d_dist_out, d_count_out, d_flag_out, d_nflag_out);
    } else {
        HIPCK(hipFuncSetAttribute((const void *)k_merge_partials<MMIDX_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        hipLaunchKernelGGL(k_merge_partials<MMIDX_BLOCK>, dim3((unsigned)nq), dim3(MMIDX_BLOCK), mlds, st, k, (int)nq, nshards, mcap, d_pdist,
                           (const long long *)d_pkey, d_pcount, (const long long *)d_poff, d_iid_out, d_dist_out, d_count_out, d_flag_out, d_nflag_out);
    }
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char *mmidx_last_error(void) { return g_err.c_str(); }
int mmidx_abi_version(void) { return MMIDX_ABI_VERSION; }

int mmidx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mmidx_create(int kind, int D, int m, int ks, int C, int transform, const int32_t *perm, const double *rot, int device,
                 mmidx_index **out) {
    if (!out) return fail(MMIDX_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (kind != MMIDX_KIND_PQ && kind != MMIDX_KIND_IVFPQ) return fail(MMIDX_ERR_INVALID_ARG, "unknown index kind %d", kind);
    if (D < 1 || m < 1 || D % m > 0) return fail(MMIDX_ERR_INVALID_SUBVECTORS, "The given number of subvectors is not valid!");
    if (ks < 1 || ks > 65536) return fail(MMIDX_ERR_UNSUPPORTED, "numProductCentroids %d outside 1..65536", ks);
    if (kind == MMIDX_KIND_IVFPQ && C < 1) return fail(MMIDX_ERR_INVALID_ARG, "numCoarseCentroids must be >= 1");
    if (transform < 0 || transform > 2) return fail(MMIDX_ERR_INVALID_ARG, "unknown transformation %d", transform);
    if (transform == MMIDX_TR_ROTATION && !rot)
        return fail(MMIDX_ERR_INVALID_ARG, "RandomRotation needs the D x D matrix computed by the Java side");
    const int ndev = mmidx_device_count();
    if (ndev < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MMIDX_ERR_NO_DEVICE, "device %d outside 0..%d", device, ndev - 1);
    HIPCK(hipSetDevice(device));
    mmidx_index *h = new mmidx_index();
    h->kind = kind;
    h->D = D;
    h->m = m;
    h->ks = ks;
    h->dsub = D / m;
    h->C = kind == MMIDX_KIND_IVFPQ ? C : 0;
    h->transform = transform;
    h->w = (int)(C * 0.1);  // IVFPQ.java:188
    h->device = device;
    h->nlists = kind == MMIDX_KIND_IVFPQ ? C : 1;
    h->code_bytes = ks <= 256 ? 1 : 2;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) (void)hipGetLastError();
        h->num_cus = ncu > 0 ? ncu : 256;
    }
    if (hipHostMalloc((void **)&h->pin_hint, 64) == hipSuccess) {
        memset(h->pin_hint, 0, 64);  // [0] pass B's pair count of the previous call, [1] codes its K3g verified
    } else {
        h->pin_hint = nullptr;  // (launches are then sized for the worst case)
        (void)hipGetLastError();
    }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(MMIDX_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    if (hipMalloc((void **)&h->d_counters, 16 * sizeof(u64)) != hipSuccess || hipMemset(h->d_counters, 0, 16 * sizeof(u64)) != hipSuccess) {
        delete h;
        return fail(MMIDX_ERR_HIP, "hipMalloc failed");
    }
    if (transform == MMIDX_TR_PERMUTATION) {
        std::vector<int32_t> p((size_t)D);
        if (perm) {
            for (int i = 0; i < D; i++) {
                if (perm[i] < 0 || perm[i] >= D) {
                    mmidx_destroy(h);
                    return fail(MMIDX_ERR_INVALID_ARG, "permutation index out of range");
                }
                p[(size_t)i] = perm[i];
            }
        } else {
            jdk_random_permutation(1, D, p.data());  // seed = 1: IVFPQ.java:136, PQ.java:108
        }
        HIPCK(hipMalloc((void **)&h->d_perm, (size_t)D * sizeof(int32_t)));
        HIPCK(hipMemcpy(h->d_perm, p.data(), (size_t)D * sizeof(int32_t), hipMemcpyHostToDevice));
    } else if (transform == MMIDX_TR_ROTATION) {
        HIPCK(hipMalloc((void **)&h->d_rot, (size_t)D * D * sizeof(double)));
        HIPCK(hipMemcpy(h->d_rot, rot, (size_t)D * D * sizeof(double), hipMemcpyHostToDevice));
        // The matrix is an input (EJML's createOrthogonal stream cannot be reproduced: SURVEY 8c): how far is it from orthogonal?
        // v -> v R keeps lengths up to |v| sqrt(||R R^T - I||_2) <= |v| sqrt(D max|R R^T - I|): with that measured, the coarse bound
        // (which compares ||c - q|| with the codes' norms in the ROTATED space) applies with a margin instead of being switched off.
        if (D <= 512) {
            double emax = 0.0;
            for (int i = 0; i < D; i++)
                for (int j = i; j < D; j++) {
                    double a = 0.0;
                    for (int t = 0; t < D; t++) a += rot[(size_t)i * D + t] * rot[(size_t)j * D + t];
                    emax = std::max(emax, std::fabs(a - (i == j ? 1.0 : 0.0)));
                }
            const double dev = emax * D;
            if (dev < 1e-6) h->rot_shrink = 1.0 - 2.0 * dev - 1e-12;  // |v R| >= |v| sqrt(1 - dev) >= |v| (1 - dev)
        }
    }
    h->h_off.assign((size_t)h->nlists + 1, 0);
    {
        const char *nf = getenv("MMIDX_NO_FILTER");
        h->no_filter = nf && nf[0] == '1';
        const char *nb = getenv("MMIDX_NO_BOUND");
        h->no_bound = (nb && nb[0] == '1') || h->no_filter;
        const char *ds = getenv("MMIDX_DEBUG_SYNC");
        h->debug_sync = ds && ds[0] == '1';
        const char *ec = getenv("MMIDX_EXACT_COARSE");
        h->exact_coarse = ec && ec[0] == '1';
        const char *p5 = getenv("MMIDX_PASSA_512");
        h->passa_512 = p5 && p5[0] == '1';
        const char *p2 = getenv("MMIDX_PASSA_SU2");
        h->passa_su2 = p2 && p2[0] == '1';
        const char *cv1 = getenv("MMIDX_COARSE_V1");
        h->coarse_v1 = cv1 && cv1[0] == '1';
        const char *ph = getenv("MMIDX_PASSA_HIST");
        if (ph) h->passa_hist = atoi(ph);
        const char *pw = getenv("MMIDX_PASSA_WIDE");
        if (pw) h->passa_wide = atoi(pw);
        const char *pp = getenv("MMIDX_PASSA_PREFIX");
        if (pp) h->passa_prefix = atoi(pp);
        const char *pf = getenv("MMIDX_PASSA_FILTER");
        h->passa_filter = pf && pf[0] == '1';
        const char *nsd = getenv("MMIDX_SEED");
        h->no_seed = !(nsd && nsd[0] == '1');
    }
    *out = h;
    return MMIDX_OK;
}

int mmidx_destroy(mmidx_index *h) {
    if (!h) return MMIDX_OK;
    if (h->grp) {  // a sharded parent owns nothing but its group
        sharded_destroy(h);
        delete h;
        return MMIDX_OK;
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->pin_hint) {
        (void)hipDeviceSynchronize();  // a read-back of the hint may still be queued on the caller's stream
        (void)hipHostFree(h->pin_hint);
        h->pin_hint = nullptr;
    }
    if (h->pin_stage) (void)hipHostFree(h->pin_stage);
    h->pin_stage = nullptr;
    for (auto &sl : h->host_slot) {
        if (sl.e_in) (void)hipEventDestroy(sl.e_in);
        if (sl.e_done) (void)hipEventDestroy(sl.e_done);
        if (sl.st) (void)hipStreamDestroy(sl.st);
        sl.e_in = sl.e_done = nullptr;
        sl.st = nullptr;
        sl.dQ.release();
        sl.out.release();
    }
    h->ws_out.release();
    void *ptrs[] = {h->d_Ch, h->d_Cl, h->d_cn_pad, h->d_cn, h->d_cnorm, h->d_coarseT32, h->d_coarse, h->d_coarseT, h->d_pq, h->d_pqT, h->d_rot, h->d_perm, h->d_off, h->d_codes,
                    h->d_ids,    h->d_pcell,   h->d_pid, h->d_pcodes};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    h->ws_Q.release();
    h->ws_cdist.release();
    h->ws_odist.release();
    h->ws_X.release();
    h->ws_qn.release();
    h->ws_Xa.release();
    h->ws_sdc.release();
    h->ws_fb.release();
    h->ws_Qh.release();
    h->ws_Ql.release();
    h->ws_gmin.release();
    h->ws_clist.release();
    h->ws_glut.release();
    h->ws_cdsel.release();
    h->ws_Q32.release();
    h->ws_S.release();
    h->ws_cells.release();
    h->ws_oiid.release();
    h->ws_ocnt.release();
    h->ws_flag.release();
    h->ws_ecell.release();
    h->ws_pcount.release();
    h->ws_pstart.release();
    h->ws_pcursor.release();
    h->ws_order.release();
    h->ws_T.release();
    h->ws_pkey.release();
    h->ws_pval.release();
    h->ws_pcnt.release();
    h->ws_ecode.release();
    h->ws_tmp.release();
    h->ws_keep.release();
    h->ws_amb.release();
    h->ws_aidx.release();
    h->ws_acell.release();
    h->ws_dest.release();
    h->ws_gdesc.release();
    h->ws_gfb.release();
    h->ws_flatoff.release();
    h->ws_flatlut.release();
    h->ws_ghist.release();
    if (h->d_grpx) (void)hipFree(h->d_grpx);
    if (h->pin_grpx) (void)hipHostFree(h->pin_grpx);
    h->ws_T0.release();
    h->ws_inv.release();
    for (int e = 0; e < 4; e++)
        if (h->host_ev[e]) (void)hipEventDestroy(h->host_ev[e]);
    if (h->d_pqstat) (void)hipFree(h->d_pqstat);
    if (h->d_pqT32) (void)hipFree(h->d_pqT32);
    if (h->d_pq32T) (void)hipFree(h->d_pq32T);
    if (h->d_pn32) (void)hipFree(h->d_pn32);
    if (h->d_pnmax) (void)hipFree(h->d_pnmax);
    if (h->d_coarseP) (void)hipFree(h->d_coarseP);
    if (h->d_zero) (void)hipFree(h->d_zero);
    if (h->d_pq16) (void)hipFree(h->d_pq16);
    if (h->d_pn64) (void)hipFree(h->d_pn64);
    h->xn.release();
    h->ws_surv.release();
    h->ws_R.release();
    h->ws_R16.release();
    h->ws_defer.release();
    h->ws_nrow.release();
    h->ws_lutpre.release();
    h->ws_acand.release();
    h->ws_arowc.release();
    h->ws_abm.release();
    h->ws_aicnt.release();
    h->ws_arec.release();
    h->ws_arows.release();
    h->ws_ameta.release();
    h->ws_ametaT.release();
    h->ws_arnd.release();
    for (hipEvent_t e : h->a_ev) (void)hipEventDestroy(e);
    h->a_ev.clear();
    h->ws_mfctl.release();
    h->ws_psnap.release();
    h->ws_redo.release();
    h->ws_Qp.release();
    for (auto &ev : h->evpool)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : h->mf_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (h->d_counters) (void)hipFree(h->d_counters);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MMIDX_OK;
}

int mmidx_set_coarse(mmidx_index *h, const double *coarse) {
    if (!h || !coarse) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->kind != MMIDX_KIND_IVFPQ) return fail(MMIDX_ERR_INVALID_ARG, "PQ index has no coarse quantizer");
    if (h->grp) {  // codebooks are replicated on every shard
        int rcs = sharded_for_each(h, [&](mmidx_index *s) { return mmidx_set_coarse(s, coarse); });
        if (!rcs) h->coarse_set = true;
        return rcs;
    }
    int rc = set_device(h);
    if (rc) return rc;
    const size_t n = (size_t)h->C * h->D;
    std::vector<double> T(n);
    for (int c = 0; c < h->C; c++)
        for (int j = 0; j < h->D; j++) T[(size_t)j * h->C + c] = coarse[(size_t)c * h->D + j];
    if (!h->d_coarse) HIPCK(hipMalloc((void **)&h->d_coarse, n * sizeof(double)));
    if (!h->d_coarseT) HIPCK(hipMalloc((void **)&h->d_coarseT, n * sizeof(double)));
    HIPCK(hipMemcpy(h->d_coarse, coarse, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_coarseT, T.data(), n * sizeof(double), hipMemcpyHostToDevice));
    // certified approximate coarse stage: |c|^2, |c| (fp64) and an fp32 transposed copy
    h->cn_max = 0.0;
    h->cnorm_max = 0.0;
    h->coarse_maxabs = 0.0;
    std::vector<double> cn((size_t)h->C), cnorm((size_t)h->C);
    std::vector<float> T32(n);
    for (int c = 0; c < h->C; c++) {
        double s2 = 0.0;
        for (int j = 0; j < h->D; j++) {
            const double v = coarse[(size_t)c * h->D + j];
            s2 += v * v;
            if (!(std::fabs(v) <= h->coarse_maxabs)) h->coarse_maxabs = v == v ? std::fabs(v) : INFINITY;
            T32[(size_t)j * h->C + c] = (float)v;
        }
        cn[(size_t)c] = s2;
        cnorm[(size_t)c] = std::sqrt(s2) * (1.0 + 1e-12);
        h->cn_max = std::max(h->cn_max, s2 * (1.0 + 1e-12));
        h->cnorm_max = std::max(h->cnorm_max, cnorm[(size_t)c]);
    }
    if (!h->d_cn) HIPCK(hipMalloc((void **)&h->d_cn, (size_t)h->C * 8));
    if (!h->d_cnorm) HIPCK(hipMalloc((void **)&h->d_cnorm, (size_t)h->C * 8));
    if (!h->d_coarseT32) HIPCK(hipMalloc((void **)&h->d_coarseT32, n * sizeof(float)));
    HIPCK(hipMemcpy(h->d_cn, cn.data(), (size_t)h->C * 8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_cnorm, cnorm.data(), (size_t)h->C * 8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_coarseT32, T32.data(), n * sizeof(float), hipMemcpyHostToDevice));
    // bf16 head / tail copies, rows padded to a multiple of 128 (zero rows, |c|^2 = +inf), k to a multiple of 32
    h->Cp = (h->C + 127) / 128 * 128;
    h->Dp = (h->D + 31) / 32 * 32;
    if (h->d_Ch) (void)hipFree(h->d_Ch);
    if (h->d_Cl) (void)hipFree(h->d_Cl);
    if (h->d_cn_pad) (void)hipFree(h->d_cn_pad);
    h->d_Ch = h->d_Cl = nullptr;
    h->d_cn_pad = nullptr;
    const size_t nb = (size_t)h->Cp * h->Dp;
    HIPCK(hipMalloc((void **)&h->d_Ch, nb * 2));
    HIPCK(hipMalloc((void **)&h->d_Cl, nb * 2));
    HIPCK(hipMalloc((void **)&h->d_cn_pad, (size_t)h->Cp * 8));
    HIPCK(hipMemset(h->d_Ch, 0, nb * 2));
    HIPCK(hipMemset(h->d_Cl, 0, nb * 2));
    {
        std::vector<double> cnp((size_t)h->Cp, std::numeric_limits<double>::infinity());
        for (int c = 0; c < h->C; c++) cnp[(size_t)c] = cn[(size_t)c];
        HIPCK(hipMemcpy(h->d_cn_pad, cnp.data(), (size_t)h->Cp * 8, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)((h->C + 3) / 4)), dim3(MMIDX_BLOCK), 0, h->stream, h->d_coarse, (__bf16 *)h->d_Ch,
                       (__bf16 *)h->d_Cl, (float *)nullptr, (double *)nullptr, h->D, h->Dp, (long long)h->C);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    h->coarse_set = true;
    h->grp_valid = false;  // (K3s's transformed copy of the centroids)
    return MMIDX_OK;
}

int mmidx_set_pq(mmidx_index *h, const double *pq) {
    if (!h || !pq) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->grp) {
        int rcs = sharded_for_each(h, [&](mmidx_index *s) { return mmidx_set_pq(s, pq); });
        if (!rcs) h->pq_set = true;
        return rcs;
    }
    int rc = set_device(h);
    if (rc) return rc;
    const size_t n = (size_t)h->m * h->ks * h->dsub;
    std::vector<double> T(n);
    for (int s = 0; s < h->m; s++)
        for (int j = 0; j < h->ks; j++)
            for (int t = 0; t < h->dsub; t++)
                T[((size_t)s * h->dsub + t) * h->ks + j] = pq[((size_t)s * h->ks + j) * h->dsub + t];
    if (!h->d_pq) HIPCK(hipMalloc((void **)&h->d_pq, n * sizeof(double)));
    if (!h->d_pqT) HIPCK(hipMalloc((void **)&h->d_pqT, n * sizeof(double)));
    HIPCK(hipMemcpy(h->d_pq, pq, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(h->d_pqT, T.data(), n * sizeof(double), hipMemcpyHostToDevice));
    double amax = 0.0;
    for (size_t i = 0; i < n; i++) amax = std::max(amax, std::fabs(pq[i]));
    h->pq_maxabs = amax;
    h->mfma_valid = false;
    h->xn_valid = false;
    double r2 = 0.0;
    for (int s = 0; s < h->m; s++) {
        double mx = 0.0;
        for (int j = 0; j < h->ks; j++) {
            double nn = 0.0;
            for (int t = 0; t < h->dsub; t++) {
                const double v = pq[((size_t)s * h->ks + j) * h->dsub + t];
                nn += v * v;
            }
            mx = std::max(mx, nn);
        }
        r2 += mx;
    }
    h->rmax = std::sqrt(r2) * (1.0 + 1e-12);
    {  // K3q's scale: per-row mean vector and mean squared norm of the codebook (a scale only: no result depends on them)
        std::vector<double> stt((size_t)h->m * h->dsub + h->m, 0.0);
        for (int s = 0; s < h->m; s++)
            for (int j = 0; j < h->ks; j++) {
                double nn = 0.0;
                for (int t = 0; t < h->dsub; t++) {
                    const double v = pq[((size_t)s * h->ks + j) * h->dsub + t];
                    stt[(size_t)s * h->dsub + t] += v / h->ks;
                    nn += v * v;
                }
                stt[(size_t)h->m * h->dsub + s] += nn / h->ks;
            }
        if (!h->d_pqstat) HIPCK(hipMalloc((void **)&h->d_pqstat, stt.size() * sizeof(double)));
        HIPCK(hipMemcpy(h->d_pqstat, stt.data(), stt.size() * sizeof(double), hipMemcpyHostToDevice));
        if (h->d_pqT32) (void)hipFree(h->d_pqT32);
        h->d_pqT32 = nullptr;
        if (h->ks == 256 && h->dsub % 2 == 0) {
            std::vector<float> t32(n);
            for (int s = 0; s < h->m; s++)
                for (int j = 0; j < 256; j++)
                    for (int t = 0; t < h->dsub; t++)
                        t32[(((size_t)s * (h->dsub / 2) + t / 2) * 256 + j) * 2 + (t & 1)] = (float)pq[((size_t)s * h->ks + j) * h->dsub + t];
            HIPCK(hipMalloc((void **)&h->d_pqT32, n * sizeof(float)));
            HIPCK(hipMemcpy(h->d_pqT32, t32.data(), n * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    h->pq_set = true;
    h->grp_valid = false;
    return MMIDX_OK;
}

int mmidx_set_w(mmidx_index *h, int w) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    h->w = w;  // validated at search time, as the reference does (IVFPQ.java:95-97)
    if (h->grp) return sharded_for_each(h, [&](mmidx_index *s) { return mmidx_set_w(s, w); });
    return MMIDX_OK;
}
int mmidx_get_w(const mmidx_index *h, int *w_out) {
    if (!h || !w_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *w_out = h->w;
    return MMIDX_OK;
}
int mmidx_size(const mmidx_index *h, int64_t *n_out) {
    if (!h || !n_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *n_out = h->grp ? sharded_total(h) : total_size(h);
    return MMIDX_OK;
}

int mmidx_sync_index(mmidx_index *h) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (h->grp) return sharded_sync(h);
    int rc = set_device(h);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    return build_csr(h);
}

int mmidx_list_sizes(mmidx_index *h, int32_t *sizes_out) {
    if (!h || !sizes_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->grp) return sharded_list_sizes(h, sizes_out);
    int rc = mmidx_sync_index(h);
    if (rc) return rc;
    for (int c = 0; c < h->nlists; c++) sizes_out[c] = (int32_t)(h->h_off[(size_t)c + 1] - h->h_off[(size_t)c]);
    return MMIDX_OK;
}

int mmidx_encode_device(mmidx_index *h, int64_t n, const double *dX, int32_t *d_cell_out, void *d_code_out, void *stream) {
    NOT_ON_SHARDED(h, "mmidx_encode_device");
    int rc = check_ready(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!dX || !d_cell_out || !d_code_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return MMIDX_OK;
    DeviceCall call(h, st);
    if (h->code_bytes == 1) {
        // kernels produce centroid indices; the boundary carries the stored form idx - 128
        HIPCK(h->ws_tmp.reserve((size_t)n * h->m));
        rc = encode_device(h, n, dX, d_cell_out, h->ws_tmp.p, st);
        if (rc) return rc;
        const long long tot = (long long)n * h->m;
        hipLaunchKernelGGL(k_bias_codes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->ws_tmp.p, (signed char *)d_code_out, tot);
        HIPCK(hipGetLastError());
        return MMIDX_OK;
    }
    return encode_device(h, n, dX, d_cell_out, d_code_out, st);
}

int mmidx_encode(mmidx_index *h, int64_t n, const double *X, int32_t *cell_out, void *code_out) {
    if (h && h->grp) {
        if (n < 0 || (n > 0 && (!X || !code_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
        return sharded_encode(h, n, X, cell_out, code_out);
    }
    int rc = check_ready(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!X || !code_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MMIDX_OK;
    rc = set_device(h);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t cb = (size_t)h->m * h->code_bytes;
    const int64_t B = 1 << 18;
    for (int64_t i0 = 0; i0 < n; i0 += B) {
        const int64_t nb = std::min(B, n - i0);
        HIPCK(h->ws_X.reserve((size_t)nb * h->D));
        HIPCK(h->ws_ecell.reserve((size_t)nb));
        HIPCK(h->ws_ecode.reserve((size_t)nb * cb));
        HIPCK(hipMemcpyAsync(h->ws_X.p, X + (size_t)i0 * h->D, (size_t)nb * h->D * 8, hipMemcpyHostToDevice, h->stream));
        rc = mmidx_encode_device(h, nb, h->ws_X.p, h->ws_ecell.p, h->ws_ecode.p, h->stream);
        if (rc) return rc;
        if (cell_out) HIPCK(hipMemcpyAsync(cell_out + i0, h->ws_ecell.p, (size_t)nb * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync((char *)code_out + (size_t)i0 * cb, h->ws_ecode.p, (size_t)nb * cb, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return MMIDX_OK;
}

// append n device-resident records; codes in stored form (int8 biased / int16)
int mmidx_add_codes_device(mmidx_index *h, int64_t n, const int32_t *d_iids, const int32_t *d_cells, const void *d_codes, void *stream) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    NOT_ON_SHARDED(h, "mmidx_add_codes_device");
    if (n < 0 || (n > 0 && (!d_iids || !d_codes))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->kind == MMIDX_KIND_IVFPQ && n > 0 && !d_cells) return fail(MMIDX_ERR_INVALID_ARG, "IVFPQ needs list ids");
    if (n == 0) return MMIDX_OK;
    if (total_size(h) + n > 2147483647LL) return fail(MMIDX_ERR_CAPACITY, "Maximum index capacity reached, no more vectors can be indexed!");
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::recursive_mutex> alk(h->add_mu);
    DeviceCall call(h, h->stream);  // (the encoder uses the search workspaces, on the handle's own stream)
    std::lock_guard<std::mutex> lk(h->mu);
    if (st != h->stream) HIPCK(hipStreamSynchronize(st));
    rc = ensure_pending(h, n);
    if (rc) return rc;
    const int64_t o = h->n_pend;
    HIPCK(hipMemcpyAsync(h->d_pid + o, d_iids, (size_t)n * 4, hipMemcpyDeviceToDevice, h->stream));
    if (h->kind == MMIDX_KIND_IVFPQ)
        HIPCK(hipMemcpyAsync(h->d_pcell + o, d_cells, (size_t)n * 4, hipMemcpyDeviceToDevice, h->stream));
    else
        HIPCK(hipMemsetAsync(h->d_pcell + o, 0, (size_t)n * 4, h->stream));
    const long long tot = (long long)n * h->m;
    if (h->code_bytes == 1) {
        hipLaunchKernelGGL(k_unbias_codes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (const signed char *)d_codes,
                           (unsigned char *)h->d_pcodes + (size_t)o * h->m, tot);
        HIPCK(hipGetLastError());
    } else {
        HIPCK(hipMemcpyAsync((char *)h->d_pcodes + (size_t)o * h->m * 2, d_codes, (size_t)tot * 2, hipMemcpyDeviceToDevice, h->stream));
    }
    // validate before committing: the records stay out of the index when a list id or a code value is out of range
    HIPCK(hipMemsetAsync(h->d_counters + 4, 0, sizeof(int32_t), h->stream));
    {
        const unsigned grid = (unsigned)((n + 255) / 256);
        const int32_t *pc = h->kind == MMIDX_KIND_IVFPQ ? h->d_pcell + o : nullptr;
        if (h->code_bytes == 1)
            hipLaunchKernelGGL(k_check_records<unsigned char>, dim3(grid), dim3(256), 0, h->stream, pc, h->nlists,
                               (const unsigned char *)h->d_pcodes + (size_t)o * h->m, h->ks, h->m, (long long)n, (int32_t *)(h->d_counters + 4));
        else
            hipLaunchKernelGGL(k_check_records<unsigned short>, dim3(grid), dim3(256), 0, h->stream, pc, h->nlists,
                               (const unsigned short *)h->d_pcodes + (size_t)o * h->m, h->ks, h->m, (long long)n, (int32_t *)(h->d_counters + 4));
        HIPCK(hipGetLastError());
    }
    int32_t bad = 0;
    HIPCK(hipMemcpyAsync(&bad, h->d_counters + 4, sizeof(bad), hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    if (bad & 1) return fail(MMIDX_ERR_INVALID_ARG, "list id outside 0..%d: the batch was not added", h->nlists - 1);
    if (bad & 2) return fail(MMIDX_ERR_INVALID_ARG, "code value outside 0..%d (numProductCentroids): the batch was not added", h->ks - 1);
    h->n_pend += n;
    return MMIDX_OK;
}

int mmidx_add_codes(mmidx_index *h, int64_t n, const int32_t *iids, const int32_t *cells, const void *codes) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (n < 0 || (n > 0 && (!iids || !codes))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->kind == MMIDX_KIND_IVFPQ && n > 0 && !cells) return fail(MMIDX_ERR_INVALID_ARG, "IVFPQ needs list ids");
    if (n == 0) return MMIDX_OK;
    if (h->grp) return sharded_add_codes(h, n, iids, cells, codes);
    if (h->kind == MMIDX_KIND_IVFPQ)
        for (int64_t i = 0; i < n; i++)
            if (cells[i] < 0 || cells[i] >= h->C) return fail(MMIDX_ERR_INVALID_ARG, "list id %d outside 0..%d", cells[i], h->C - 1);
    int rc = set_device(h);
    if (rc) return rc;
    const size_t cb = (size_t)h->m * h->code_bytes;
    int32_t *d_i = nullptr, *d_c = nullptr;
    void *d_k = nullptr;
    HIPCK(hipMalloc((void **)&d_i, (size_t)n * 4));
    HIPCK(hipMalloc((void **)&d_c, (size_t)n * 4));
    HIPCK(hipMalloc(&d_k, (size_t)n * cb));
    HIPCK(hipMemcpy(d_i, iids, (size_t)n * 4, hipMemcpyHostToDevice));
    if (cells) HIPCK(hipMemcpy(d_c, cells, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(d_k, codes, (size_t)n * cb, hipMemcpyHostToDevice));
    rc = mmidx_add_codes_device(h, n, d_i, cells ? d_c : nullptr, d_k, nullptr);
    (void)hipFree(d_i);
    (void)hipFree(d_c);
    (void)hipFree(d_k);
    return rc;
}

int mmidx_add_vectors_device(mmidx_index *h, int64_t n, const double *dX, const int32_t *d_iids, int32_t iid0, void *stream) {
    NOT_ON_SHARDED(h, "mmidx_add_vectors_device");
    int rc = check_ready(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !dX)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MMIDX_OK;
    if (total_size(h) + n > 2147483647LL) return fail(MMIDX_ERR_CAPACITY, "Maximum index capacity reached, no more vectors can be indexed!");
    rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::recursive_mutex> alk(h->add_mu);
    DeviceCall call(h, h->stream);  // (the encoder uses the search workspaces, on the handle's own stream)
    std::lock_guard<std::mutex> lk(h->mu);
    if (st != h->stream) HIPCK(hipStreamSynchronize(st));
    rc = ensure_pending(h, n);
    if (rc) return rc;
    const int64_t o = h->n_pend;
    rc = encode_device(h, n, dX, h->d_pcell + o, (char *)h->d_pcodes + (size_t)o * h->m * h->code_bytes, h->stream);
    if (rc) return rc;
    if (h->kind == MMIDX_KIND_PQ) HIPCK(hipMemsetAsync(h->d_pcell + o, 0, (size_t)n * 4, h->stream));
    if (d_iids)
        HIPCK(hipMemcpyAsync(h->d_pid + o, d_iids, (size_t)n * 4, hipMemcpyDeviceToDevice, h->stream));
    else
        hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_pid + o, iid0, (long long)n);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    h->n_pend += n;
    return MMIDX_OK;
}

int mmidx_add_vectors(mmidx_index *h, int64_t n, const double *X, const int32_t *iids, int32_t *cell_out, void *code_out) {
    if (h && h->grp) {
        if (n < 0 || (n > 0 && !X)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
        return sharded_add_vectors(h, n, X, nullptr, nullptr, iids, MMIDX_IID_AUTO, cell_out, code_out);
    }
    int rc = check_ready(h);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !X)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MMIDX_OK;
    rc = set_device(h);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> alk(h->add_mu);  // (the pending offset read below and the append are one step)
    const size_t cb = (size_t)h->m * h->code_bytes;
    const int64_t B = 1 << 18;
    for (int64_t i0 = 0; i0 < n; i0 += B) {
        const int64_t nb = std::min(B, n - i0);
        double *dX = nullptr;
        int32_t *d_i = nullptr;
        HIPCK(hipMalloc((void **)&dX, (size_t)nb * h->D * 8));
        HIPCK(hipMemcpy(dX, X + (size_t)i0 * h->D, (size_t)nb * h->D * 8, hipMemcpyHostToDevice));
        if (iids) {
            HIPCK(hipMalloc((void **)&d_i, (size_t)nb * 4));
            HIPCK(hipMemcpy(d_i, iids + i0, (size_t)nb * 4, hipMemcpyHostToDevice));
        }
        const int64_t o = h->n_pend;
        rc = mmidx_add_vectors_device(h, nb, dX, d_i, (int32_t)total_size(h), nullptr);
        if (!rc && cell_out) {
            if (h->kind == MMIDX_KIND_IVFPQ) HIPCK(hipMemcpy(cell_out + i0, h->d_pcell + o, (size_t)nb * 4, hipMemcpyDeviceToHost));
            else std::fill(cell_out + i0, cell_out + i0 + nb, -1);
        }
        if (!rc && code_out) {
            HIPCK(hipMemcpy((char *)code_out + (size_t)i0 * cb, (char *)h->d_pcodes + (size_t)o * cb, (size_t)nb * cb, hipMemcpyDeviceToHost));
            if (h->code_bytes == 1) {  // stored form idx - 128 (PQ.java:555)
                signed char *c = (signed char *)code_out + (size_t)i0 * cb;
                for (size_t t = 0; t < (size_t)nb * cb; t++) c[t] = (signed char)((int)(unsigned char)c[t] - 128);
            }
        }
        (void)hipFree(dX);
        if (d_i) (void)hipFree(d_i);
        if (rc) return rc;
    }
    return MMIDX_OK;
}

int mmidx_search_device(mmidx_index *h, int k, int64_t nq, const double *dQ, int32_t *d_iid_out, double *d_dist_out,
                        int32_t *d_count_out, void *stream) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nq > 0 && (!dQ || !d_iid_out || !d_dist_out || !d_count_out)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    NOT_ON_SHARDED(h, "mmidx_search_device");
    hipStream_t st = (hipStream_t)stream;
    DeviceCall call(h, st);
    return search_common(h, k, nq, dQ, nullptr, 0, d_iid_out, d_dist_out, d_count_out, nullptr, nullptr, st);
}

// Queries of several concurrent callers in one device batch.  The reference's API is one query per call
// (ASS.computeNearestNeighbors) from as many reader threads as the application likes; one call costs ~0.15 ms of
// launches and copies whatever its size, so callers that arrive while a batch is running are queued and served
// TOGETHER by the next leader: queries staged in pinned memory (one H2D), one search, one D2H of
// (distances | ids | counts), results scattered to the callers' buffers.  Every query's answer is the one it
// would get alone (queries of a batch are independent in every kernel).
#define MMIDX_COMB_MAX_Q 4096   // queries per combined batch; larger requests run alone with direct copies
static int search_host_direct(mmidx_index *h, const SearchReq &r) {
    const int k = r.k;
    const int64_t nq = r.nq;
    HIPCK(h->ws_Q.reserve((size_t)nq * h->D));
    HIPCK(h->ws_oiid.reserve((size_t)nq * k));
    HIPCK(h->ws_odist.reserve((size_t)nq * k));
    HIPCK(h->ws_ocnt.reserve((size_t)nq));
    HIPCK(hipMemcpyAsync(h->ws_Q.p, r.Q, (size_t)nq * h->D * 8, hipMemcpyHostToDevice, h->stream));
    int rc = search_common(h, k, nq, h->ws_Q.p, nullptr, 0, h->ws_oiid.p, h->ws_odist.p, h->ws_ocnt.p, nullptr, nullptr, h->stream);
    if (rc) return rc;
    HIPCK(hipMemcpyAsync(r.iid, h->ws_oiid.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipMemcpyAsync(r.dist, h->ws_odist.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipMemcpyAsync(r.cnt, h->ws_ocnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return MMIDX_OK;
}

// A request of more than MMIDX_COMB_MAX_Q queries, outside the combiner.  The caller takes one of the handle's slots (its own copy
// stream, query and answer buffers), sends its queries on that stream, holds search_mu only while its kernels are enqueued on the
// handle's stream (ordered behind its upload by an event), and fetches its answers on its own stream again.  With two or three
// reader threads -- the reference's model: any number of threads calling computeNearestNeighbors (AbstractSearchStructure.java:281-291)
// -- the uploads and downloads of one caller run under the kernels of another; a single caller gains nothing.
static int search_host_big(mmidx_index *h, const SearchReq &r) {
    int rc = set_device(h);
    if (rc) return rc;
    const int k = r.k;
    const int64_t nq = r.nq;
    mmidx_index::HostSlot *sl = nullptr;
    {
        std::unique_lock<std::mutex> lk(h->slot_mu);
        for (;;) {
            for (auto &c : h->host_slot)
                if (!c.busy) {
                    sl = &c;
                    break;
                }
            if (sl) break;
            h->slot_cv.wait(lk);
        }
        sl->busy = true;
    }
    struct Release {
        mmidx_index *h;
        mmidx_index::HostSlot *sl;
        ~Release() {
            {
                std::lock_guard<std::mutex> lk(h->slot_mu);
                sl->busy = false;
            }
            h->slot_cv.notify_one();
        }
    } release{h, sl};
    if (!sl->st) {
        HIPCK(hipStreamCreateWithFlags(&sl->st, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&sl->e_in, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&sl->e_done, hipEventDisableTiming));
    }
    const size_t od = (size_t)nq * k * 8, oi = (size_t)nq * k * 4, oc = (size_t)nq * 4;
    if ((size_t)nq * h->D > sl->dQ.cap || od + oi + oc > sl->out.cap) HIPCK(hipStreamSynchronize(sl->st));
    HIPCK(sl->dQ.reserve((size_t)nq * h->D));
    HIPCK(sl->out.reserve(od + oi + oc));
    double *d_dist = (double *)sl->out.p;
    int32_t *d_iid = (int32_t *)(sl->out.p + od), *d_cnt = (int32_t *)(sl->out.p + od + oi);
    HIPCK(hipMemcpyAsync(sl->dQ.p, r.Q, (size_t)nq * h->D * 8, hipMemcpyHostToDevice, sl->st));
    HIPCK(hipEventRecord(sl->e_in, sl->st));
    {
        DeviceCall call(h, h->stream);  // search_mu; a _device call on another stream is waited for
        HIPCK(hipStreamWaitEvent(h->stream, sl->e_in, 0));
        rc = search_common(h, k, nq, sl->dQ.p, nullptr, 0, d_iid, d_dist, d_cnt, nullptr, nullptr, h->stream);
        if (rc) {
            (void)hipStreamSynchronize(h->stream);
            return rc;
        }
        HIPCK(hipEventRecord(sl->e_done, h->stream));
    }
    HIPCK(hipStreamWaitEvent(sl->st, sl->e_done, 0));
    HIPCK(hipMemcpyAsync(r.dist, d_dist, od, hipMemcpyDeviceToHost, sl->st));
    HIPCK(hipMemcpyAsync(r.iid, d_iid, oi, hipMemcpyDeviceToHost, sl->st));
    HIPCK(hipMemcpyAsync(r.cnt, d_cnt, oc, hipMemcpyDeviceToHost, sl->st));
    HIPCK(hipStreamSynchronize(sl->st));
    return MMIDX_OK;
}

// serves batch[0..nb) (same k, sum of nq <= MMIDX_COMB_MAX_Q unless nb == 1); the caller holds search_mu
static int search_host_batch(mmidx_index *h, SearchReq *const *batch, size_t nb) {
    int rc = set_device(h);
    if (rc) return rc;
    const int k = batch[0]->k;
    int64_t tot = 0;
    for (size_t i = 0; i < nb; i++) tot += batch[i]->nq;
    if (tot > MMIDX_COMB_MAX_Q) return search_host_direct(h, *batch[0]);  // (nb == 1)
    const size_t in_bytes = (size_t)tot * h->D * 8;
    const size_t od = (size_t)tot * k * 8, oi = (size_t)tot * k * 4, oc = (size_t)tot * 4;
    const size_t need = in_bytes + od + oi + oc;
    if (need > h->pin_stage_cap) {
        if (h->pin_stage) (void)hipHostFree(h->pin_stage);
        h->pin_stage = nullptr;
        h->pin_stage_cap = 0;
        const size_t want = need + need / 4 + 4096;
        if (hipHostMalloc((void **)&h->pin_stage, want) == hipSuccess) {
            h->pin_stage_cap = want;
        } else {
            (void)hipGetLastError();
            h->pin_stage = nullptr;
        }
    }
    if (!h->pin_stage) {  // no pinned memory: every request on its own, pageable copies
        for (size_t i = 0; i < nb; i++) {
            rc = search_host_direct(h, *batch[i]);
            if (rc) return rc;
        }
        return MMIDX_OK;
    }
    unsigned char *hin = h->pin_stage, *hout = h->pin_stage + in_bytes;
    HIPCK(h->ws_Q.reserve((size_t)tot * h->D));
    HIPCK(h->ws_out.reserve(od + oi + oc));
    double *d_dist = (double *)h->ws_out.p;
    int32_t *d_iid = (int32_t *)(h->ws_out.p + od), *d_cnt = (int32_t *)(h->ws_out.p + od + oi);
    // stage in: the requests side by side in the pinned buffer; a DMA leaves whenever 4 MiB have gathered (it runs while the next
    // piece is copied) -- ONE for a batch of single-query callers
    {
        const size_t piece = (size_t)4 << 20;
        size_t off = 0, sent = 0;
        for (size_t i = 0; i < nb; i++) {
            const size_t b = (size_t)batch[i]->nq * h->D * 8;
            for (size_t p0 = 0; p0 < b; p0 += piece) {
                const size_t pb = std::min(piece, b - p0);
                memcpy(hin + off, (const unsigned char *)batch[i]->Q + p0, pb);
                off += pb;
                if (off - sent >= piece) {
                    HIPCK(hipMemcpyAsync((unsigned char *)h->ws_Q.p + sent, hin + sent, off - sent, hipMemcpyHostToDevice, h->stream));
                    sent = off;
                }
            }
        }
        if (off > sent) HIPCK(hipMemcpyAsync((unsigned char *)h->ws_Q.p + sent, hin + sent, off - sent, hipMemcpyHostToDevice, h->stream));
    }
    rc = search_common(h, k, tot, h->ws_Q.p, nullptr, 0, d_iid, d_dist, d_cnt, nullptr, nullptr, h->stream);
    if (rc) return rc;
    // stage out: the answers come back in up to four slices of whole requests; a slice is copied to the callers' arrays while the
    // next one is still on the bus (events between the slices; the handle keeps them)
    constexpr int NSL = 4;
    if (!h->host_ev[0]) {
        for (int e = 0; e < NSL; e++) HIPCK(hipEventCreateWithFlags(&h->host_ev[e], hipEventDisableTiming));
    }
    std::vector<int64_t> req_q0(nb + 1, 0);  // first query of every request in the combined batch
    for (size_t i = 0; i < nb; i++) req_q0[i + 1] = req_q0[i] + batch[i]->nq;
    // (slices of >= 2 MiB of answers: the combined single-query callers get ONE copy and one wait, as before round 6)
    const int nsl = (int)std::min<size_t>(NSL, std::max<size_t>(1, (od + oi + oc) / ((size_t)2 << 20)));
    const int64_t per = (tot + nsl - 1) / nsl;
    for (int e = 0; e < nsl; e++) {
        const int64_t a = std::min<int64_t>(tot, (int64_t)e * per), b = e == nsl - 1 ? tot : std::min<int64_t>(tot, (int64_t)(e + 1) * per);
        if (b > a) {
            HIPCK(hipMemcpyAsync(hout + (size_t)a * k * 8, (unsigned char *)d_dist + (size_t)a * k * 8, (size_t)(b - a) * k * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCK(hipMemcpyAsync(hout + od + (size_t)a * k * 4, (unsigned char *)d_iid + (size_t)a * k * 4, (size_t)(b - a) * k * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCK(hipMemcpyAsync(hout + od + oi + (size_t)a * 4, (unsigned char *)d_cnt + (size_t)a * 4, (size_t)(b - a) * 4, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCK(hipEventRecord(h->host_ev[e], h->stream));
    }
    size_t ri = 0;
    for (int e = 0; e < nsl; e++) {
        const int64_t a = std::min<int64_t>(tot, (int64_t)e * per), b = e == nsl - 1 ? tot : std::min<int64_t>(tot, (int64_t)(e + 1) * per);
        HIPCK(hipEventSynchronize(h->host_ev[e]));
        for (int64_t g0 = a; g0 < b;) {  // the part of [a, b) that belongs to request ri
            while (req_q0[ri + 1] <= g0) ri++;
            const int64_t g1 = std::min<int64_t>(b, req_q0[ri + 1]), lo = g0 - req_q0[ri], n = g1 - g0;
            memcpy(batch[ri]->dist + (size_t)lo * k, hout + (size_t)g0 * k * 8, (size_t)n * k * 8);
            memcpy(batch[ri]->iid + (size_t)lo * k, hout + od + (size_t)g0 * k * 4, (size_t)n * k * 4);
            memcpy(batch[ri]->cnt + lo, hout + od + oi + (size_t)g0 * 4, (size_t)n * 4);
            g0 = g1;
        }
    }
    return MMIDX_OK;
}

int mmidx_search(mmidx_index *h, int k, int64_t nq, const double *Q, int32_t *iid_out, double *dist_out, int32_t *count_out) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nq > 0 && (!Q || !iid_out || !dist_out || !count_out)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (k < 1 || k > MMIDX_K_MAX) return fail(MMIDX_ERR_INVALID_ARG, "k must be in 1..%d (got %d)", MMIDX_K_MAX, k);
    if (h->grp) return sharded_search(h, k, nq, Q, iid_out, dist_out, count_out);
    int rc = check_ready(h);
    if (rc) return rc;
    if (nq == 0) return MMIDX_OK;
    SearchReq me;
    me.k = k;
    me.nq = nq;
    me.Q = Q;
    me.iid = iid_out;
    me.dist = dist_out;
    me.cnt = count_out;
    if (nq > MMIDX_COMB_MAX_Q && h->host_slots_on) return search_host_big(h, me);
    return combiner_submit(h->comb, me, MMIDX_COMB_MAX_Q, [h](SearchReq *const *batch, size_t nb) {
        std::lock_guard<std::recursive_mutex> slk(h->search_mu);  // (id queries use the same workspaces and stream)
        return search_host_batch(h, batch, nb);
    });
}

// computeNearestNeighborsInternal(k, iid) for PQ: computeKnnSDC, PQ.java:334-374
int mmidx_search_sdc(mmidx_index *h, int k, int64_t nq, const int32_t *iids, int32_t *iid_out, double *dist_out, int32_t *count_out) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nq > 0 && (!iids || !iid_out || !dist_out || !count_out)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    NOT_ON_SHARDED(h, "mmidx_search_sdc");
    if (h->kind != MMIDX_KIND_PQ) return fail(MMIDX_ERR_UNSUPPORTED, "id queries: IVFPQ.computeKnnIVFSDC is unimplemented in the reference (IVFPQ.java:509-511)");
    if (h->code_bytes != 1) return fail(MMIDX_ERR_UNSUPPORTED, "SDC needs byte codes (the reference dereferences pqByteCodes unconditionally, PQ.java:350)");
    if (k < 1 || k > MMIDX_K_MAX) return fail(MMIDX_ERR_INVALID_ARG, "k must be in 1..%d (got %d)", MMIDX_K_MAX, k);
    int rc = check_ready(h);
    if (rc) return rc;
    if (nq == 0) return MMIDX_OK;
    std::lock_guard<std::recursive_mutex> slk(h->search_mu);
    rc = set_device(h);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        rc = build_csr(h);
        if (rc) return rc;
    }
    for (int64_t i = 0; i < nq; i++)
        if (iids[i] < 0 || iids[i] >= h->n_csr) return fail(MMIDX_ERR_INVALID_ARG, "internal id %d outside 0..%lld", iids[i], (long long)h->n_csr - 1);
    const size_t per = (size_t)h->m * h->ks * h->dsub;
    const int64_t QB = std::max<int64_t>(1, (int64_t)((1ull << 30) / (per * 8)));  // <= 1 GiB of term tables per round
    for (int64_t q0 = 0; q0 < nq; q0 += QB) {
        const int64_t nb = std::min(QB, nq - q0);
        HIPCK(h->ws_sdc.reserve((size_t)nb * per));
        HIPCK(h->ws_aidx.reserve((size_t)nb));
        HIPCK(h->ws_Q.reserve((size_t)nb * h->D));  // unused by the SDC kernels, but the scan parameters want a valid pointer
        HIPCK(h->ws_oiid.reserve((size_t)nb * k));
        HIPCK(h->ws_odist.reserve((size_t)nb * k));
        HIPCK(h->ws_ocnt.reserve((size_t)nb));
        HIPCK(hipMemcpyAsync(h->ws_aidx.p, iids + q0, (size_t)nb * 4, hipMemcpyHostToDevice, h->stream));
        const long long tot = (long long)nb * (long long)per;
        hipLaunchKernelGGL(k_sdc_terms<unsigned char>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->d_pq,
                           (const unsigned char *)h->d_codes, h->ws_aidx.p, h->ws_sdc.p, h->m, h->ks, h->dsub, tot);
        HIPCK(hipGetLastError());
        rc = search_common(h, k, nb, h->ws_Q.p, nullptr, 0, h->ws_oiid.p, h->ws_odist.p, h->ws_ocnt.p, nullptr, nullptr, h->stream, h->ws_sdc.p);
        if (rc) return rc;
        HIPCK(hipMemcpyAsync(iid_out + (size_t)q0 * k, h->ws_oiid.p, (size_t)nb * k * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(dist_out + (size_t)q0 * k, h->ws_odist.p, (size_t)nb * k * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipMemcpyAsync(count_out + q0, h->ws_ocnt.p, (size_t)nb * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCK(hipStreamSynchronize(h->stream));
    }
    return MMIDX_OK;
}

int mmidx_coarse_device(mmidx_index *h, int64_t nq, const double *dQ, int32_t *d_cells_out, double *d_cdist_out, void *stream) {
    NOT_ON_SHARDED(h, "mmidx_coarse_device");
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != MMIDX_KIND_IVFPQ) return fail(MMIDX_ERR_INVALID_ARG, "PQ index has no coarse quantizer");
    if (h->w < 1 || h->w > h->C) return fail(MMIDX_ERR_INVALID_ARG, "w = %d outside 1..%d (setW)", h->w, h->C);
    if (nq > 0 && (!dQ || !d_cells_out)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    DeviceCall call(h, st);
    const int64_t qb = std::max<int64_t>(1, (2ll << 30) / ((int64_t)h->C * 8));
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        const int64_t nb = std::min(qb, nq - q0);
        rc = run_coarse(h, nb, dQ + (size_t)q0 * h->D, d_cells_out + (size_t)q0 * h->w, st);
        if (rc) return rc;
        if (d_cdist_out) {
            double *dst = d_cdist_out + (size_t)q0 * h->w;
            if (h->cdsel_valid) {
                HIPCK(hipMemcpyAsync(dst, h->ws_cdsel.p, (size_t)nb * h->w * 8, hipMemcpyDeviceToDevice, st));
            } else {
                const long long tot = (long long)nb * h->w;
                hipLaunchKernelGGL(k_gather_cdist, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->ws_cdist.p,
                                   d_cells_out + (size_t)q0 * h->w, dst, h->C, h->w, tot);
                HIPCK(hipGetLastError());
            }
        }
    }
    return MMIDX_OK;
}

int mmidx_search_partial_device(mmidx_index *h, int k, int64_t nq, const double *dQ, const int32_t *d_cells, double *d_pdist,
                                int64_t *d_pkey, int32_t *d_pcount, void *stream) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    NOT_ON_SHARDED(h, "mmidx_search_partial_device");
    if (nq > 0 && (!dQ || !d_pdist || !d_pkey || !d_pcount)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->kind == MMIDX_KIND_IVFPQ && nq > 0 && !d_cells) return fail(MMIDX_ERR_INVALID_ARG, "IVFPQ partial search needs the probe cells");
    hipStream_t st = (hipStream_t)stream;
    DeviceCall call(h, st);
    return search_common(h, k, nq, dQ, h->kind == MMIDX_KIND_IVFPQ ? d_cells : nullptr, 1, nullptr, nullptr, d_pcount, d_pdist,
                         (long long *)d_pkey, st);
}

// two-phase sharded search: thresholds are exchanged between the phases (MIN all-reduce)
static int shard_phase(mmidx_index *h, int k, int64_t nq, const double *dQ, const int32_t *d_cells, const double *d_cdist, int phase,
                       double *d_T, double *d_pdist, int64_t *d_pkey, int32_t *d_pcount, void *stream) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    NOT_ON_SHARDED(h, "the shard phases");
    if (nq > 0 && (!dQ || !d_T)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (phase == 2 && nq > 0 && (!d_pdist || !d_pkey || !d_pcount)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int rc = check_ready(h);
    if (rc) return rc;
    const int ivf = h->kind == MMIDX_KIND_IVFPQ;
    if (ivf && nq > 0 && !d_cells) return fail(MMIDX_ERR_INVALID_ARG, "IVFPQ shard search needs the probe cells");
    if (ivf && (h->w < 1 || h->w > h->C)) return fail(MMIDX_ERR_INVALID_ARG, "w = %d outside 1..%d (setW)", h->w, h->C);
    if (nq == 0) return MMIDX_OK;
    rc = set_device(h);
    if (rc) return rc;
    DeviceCall call(h, (hipStream_t)stream);
    {
        std::lock_guard<std::mutex> lk(h->mu);
        rc = build_csr(h);
        if (rc) return rc;
        rc = build_grp_tables(h);
        if (rc) return rc;
        rc = build_mfma_tables(h);
        if (rc) return rc;
    }
    SearchPlan pl;
    rc = make_plan(h, k, nq, pl, false);
    if (rc) return rc;
    if (nq > pl.qb) return fail(MMIDX_ERR_UNSUPPORTED, "shard phases take at most %lld queries per call for this index", (long long)pl.qb);
    return search_batch_device(h, pl, k, nq, dQ, ivf ? d_cells : nullptr, 1, nullptr, nullptr, d_pcount, d_pdist, (long long *)d_pkey, phase,
                               d_T, (hipStream_t)stream, nullptr, ivf ? d_cdist : nullptr);
}

int mmidx_shard_pass_a_device(mmidx_index *h, int k, int64_t nq, const double *dQ, const int32_t *d_cells, double *d_T_out, void *stream) {
    return shard_phase(h, k, nq, dQ, d_cells, nullptr, 1, d_T_out, nullptr, nullptr, nullptr, stream);
}

int mmidx_shard_pass_b_device(mmidx_index *h, int k, int64_t nq, const double *dQ, const int32_t *d_cells, const double *d_cdist,
                              const double *d_T_in, double *d_pdist, int64_t *d_pkey, int32_t *d_pcount, void *stream) {
    return shard_phase(h, k, nq, dQ, d_cells, d_cdist, 2, (double *)d_T_in, d_pdist, d_pkey, d_pcount, stream);
}

int mmidx_merge_partials_device(int device, int k, int64_t nq, int nshards, const double *d_pdist, const int64_t *d_pkey,
                                const int32_t *d_pcount, const int64_t *d_poff, int32_t *d_iid_out, double *d_dist_out,
                                int32_t *d_count_out, int32_t *d_flag_out, void *stream) {
    if (k < 1 || k > MMIDX_K_MAX) return fail(MMIDX_ERR_INVALID_ARG, "k must be in 1..%d (got %d)", MMIDX_K_MAX, k);
    if (nshards < 1 || nq < 0) return fail(MMIDX_ERR_INVALID_ARG, "bad shard / query count");
    if (nq > 0 && (!d_pdist || !d_pkey || !d_pcount || !d_iid_out || !d_dist_out || !d_count_out))
        return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (mmidx_device_count() < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    if (nq == 0) return MMIDX_OK;
    HIPCK(hipSetDevice(device));
    return launch_merge_partials(k, nq, nshards, d_pdist, d_pkey, d_pcount, d_poff, d_iid_out, d_dist_out, d_count_out, d_flag_out, nullptr,
                                 (hipStream_t)stream);
}

// one pass of the cross-shard tie replay (k_shard_tie); counts / pB / tie_iids are reduced over ranks by the caller in between
int mmidx_shard_tie_phase_device(mmidx_index *h, int phase, int k, int64_t nf, const double *dQ, const int32_t *d_cells, const int32_t *d_fq,
                                 const double *d_tau, int32_t *d_counts, int32_t *d_pB, int32_t *d_tie_iids, void *stream) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    NOT_ON_SHARDED(h, "mmidx_shard_tie_phase_device");
    if (phase < 0 || phase > 2 || k < 1 || nf < 0) return fail(MMIDX_ERR_INVALID_ARG, "bad phase / k / count");
    if (nf > 0 && (!dQ || !d_cells || !d_fq || !d_tau || !d_counts || (phase >= 1 && !d_pB) || (phase == 2 && !d_tie_iids)))
        return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->kind != MMIDX_KIND_IVFPQ) return fail(MMIDX_ERR_INVALID_ARG, "sharded search needs an IVFPQ index");
    if (nf == 0) return MMIDX_OK;
    rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    DeviceCall call(h, st);
    {
        std::lock_guard<std::mutex> lk(h->mu);
        rc = build_csr(h);
        if (rc) return rc;
    }
    TieShardParams TP{};
    TP.S.Q = dQ;
    TP.S.coarse = h->d_coarse;
    TP.S.pqT = h->d_pqT;
    TP.S.perm = h->d_perm;
    TP.S.rot = h->d_rot;
    TP.S.cells = d_cells;
    TP.S.list_off = h->d_off;
    TP.S.codes = h->d_codes;
    TP.S.D = h->D;
    TP.S.m = h->m;
    TP.S.ks = h->ks;
    TP.S.dsub = h->dsub;
    TP.S.w = h->w;
    TP.S.transform = h->transform;
    TP.S.ivf = 1;
    TP.ids = h->d_ids;
    TP.fq = d_fq;
    TP.tau = d_tau;
    TP.counts = d_counts;
    TP.pB = d_pB;
    TP.tie_iids = d_tie_iids;
    TP.k = k;
    TP.phase = phase;
    size_t lds = (size_t)h->m * h->ks * 8 + 2 * (size_t)h->D * 8;
    if (lds > 160 * 1024) {  // the table in global scratch: one slot per flagged query
        HIPCK(h->ws_glut.reserve((size_t)nf * (size_t)h->m * h->ks));
        h->glut_slots = std::max(h->glut_slots, (size_t)nf);
        TP.S.glut = h->ws_glut.p;
        lds = 2 * (size_t)h->D * 8;
        if (h->code_bytes == 1) hipLaunchKernelGGL((k_shard_tie<unsigned char, true>), dim3((unsigned)nf), dim3(MMIDX_BLOCK), lds, st, TP);
        else hipLaunchKernelGGL((k_shard_tie<unsigned short, true>), dim3((unsigned)nf), dim3(MMIDX_BLOCK), lds, st, TP);
    } else if (h->code_bytes == 1) {
        HIPCK(hipFuncSetAttribute((const void *)k_shard_tie<unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_shard_tie<unsigned char>, dim3((unsigned)nf), dim3(MMIDX_BLOCK), lds, st, TP);
    } else {
        HIPCK(hipFuncSetAttribute((const void *)k_shard_tie<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_shard_tie<unsigned short>, dim3((unsigned)nf), dim3(MMIDX_BLOCK), lds, st, TP);
    }
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

int mmidx_assign_device(mmidx_index *h, int64_t n, const double *dX, int32_t *d_cell_out, void *stream) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    NOT_ON_SHARDED(h, "mmidx_assign_device");
    if (h->kind != MMIDX_KIND_IVFPQ) return fail(MMIDX_ERR_INVALID_ARG, "PQ index has no coarse quantizer");
    if (!h->coarse_set) return fail(MMIDX_ERR_NOT_READY, "coarse quantizer not loaded (loadCoarseQuantizer)");
    if (n < 0 || (n > 0 && (!dX || !d_cell_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int rc = set_device(h);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    DeviceCall call(h, st);
    const int64_t step = std::max<int64_t>(1 << 20, ((int64_t)1 << 29) / std::max(h->Dp, 32));  // bounded scratch for the certified approximate path
    for (int64_t i0 = 0; i0 < n; i0 += step) {
        const int64_t nb = std::min(step, n - i0);
        rc = assign_device(h, nb, dX + (size_t)i0 * h->D, d_cell_out + i0, st);
        if (rc) return rc;
    }
    return MMIDX_OK;
}

int mmidx_compact_partials_device(int device, int k, int64_t nq, const double *d_pdist, const int64_t *d_pkey, const int32_t *d_pcount,
                                  const int64_t *d_poff, double *d_out_dist, int64_t *d_out_key, void *stream) {
    if (k < 1 || k > MMIDX_K_MAX || nq < 0) return fail(MMIDX_ERR_INVALID_ARG, "bad k / query count");
    if (nq > 0 && (!d_pdist || !d_pkey || !d_pcount || !d_poff)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (mmidx_device_count() < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    if (nq == 0) return MMIDX_OK;
    HIPCK(hipSetDevice(device));
    hipLaunchKernelGGL(k_compact_partials, dim3((unsigned)((nq + 3) / 4)), dim3(MMIDX_BLOCK), 0, (hipStream_t)stream, d_pdist,
                       (const long long *)d_pkey, d_pcount, (const long long *)d_poff, d_out_dist, (long long *)d_out_key, k + 1,
                       (long long)nq);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

// runtime switches for measurements (same meaning as the MMIDX_* environment variables read at create)
int mmidx_set_option(mmidx_index *h, const char *name, int value) {
    if (!h || !name) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->grp) return sharded_set_option(h, name, value);
    const std::string n(name);
    if (n == "exhaustive") {  // every probed code is read and summed in fp64: no filter, no coarse bound
        h->no_filter = value != 0;
        h->no_bound = value != 0;
    } else if (n == "no_filter") {
        h->no_filter = value != 0;
    } else if (n == "no_bound") {
        h->no_bound = value != 0;
    } else if (n == "exact_coarse") {
        h->exact_coarse = value != 0;
    } else if (n == "combine") {
        h->comb.enabled = value != 0;
    } else if (n == "passb_main_grid") {
        h->passb_main_grid = value;
    } else if (n == "coarse_v1") {
        h->coarse_v1 = value != 0;
    } else if (n == "coarse_fused") {
        h->coarse_fused = value != 0;
    } else if (n == "coarse_dma_kc") {
        h->coarse_dma_kc = value != 0;
    } else if (n == "coarse_nodma") {
        h->coarse_nodma = value != 0;
    } else if (n == "no_item_compaction") {
        h->no_item_compaction = value != 0;
    } else if (n == "passa_item_min") {
        h->passa_item_min = value;
    } else if (n == "passa_item_margin") {
        h->passa_item_margin = value;
    } else if (n == "passa_hist") {
        h->passa_hist = value;
    } else if (n == "passa_wide") {
        h->passa_wide = value;
    } else if (n == "no_grp") {  // pass B through K3f (one block per (query, list)) instead of the grouped K3g
        h->no_grp = value != 0;
    } else if (n == "flat_chunk") {
        h->flat_chunk = value;
    } else if (n == "smin_bf16") {
        h->smin_bf16 = value != 0;
    } else if (n == "smin_valu") {
        h->smin_valu = value != 0;
    } else if (n == "smin_pre") {  // K3s in front of pass B: 1 always, 0 never, -1 by the device's figures of the call before
        h->smin_pre = value < 0 ? -1 : (value != 0);
    } else if (n == "no_union") {  // K3g without the per-query histogram that lowers thresholds from the union over lists
        h->no_union = value < 0 ? -1 : (value != 0);
    } else if (n == "host_slots") {  // large host-pointer requests: 1 = up to three callers in flight (default), 0 = one at a time
        h->host_slots_on = value != 0;
    } else if (n == "passa_q") {  // K3q: 1 always, 0 never, -1 by the batch (default)
        h->passa_q = value;
    } else if (n == "passa_mfma") {  // K3ma: 1 always, 0 never, -1 by the batch (default)
        h->passa_mfma = value;
    } else if (n == "passa_mfma_wide") {
        h->a_wide = value != 0;
    } else if (n == "no_split_table") {  // m = 128: the table-in-global kernels instead of k_scan_split
        h->no_split_table = value != 0;
    } else if (n == "no_mfma") {  // pass B through K3g / K3f instead of the matrix-core bound K3m
        h->no_mfma = value != 0;
    } else if (n == "mfma_sub") {
        h->mfma_sub = value > 0 ? value : 0;
    } else if (n == "mfma_qcap") {
        h->mfma_qcap = value > 0 ? value : 0;
    } else if (n == "coarse_wave_sel") {
        h->coarse_wave_sel = value != 0;
    } else if (n == "passb_small") {
        h->passb_small = value != 0;
    } else if (n == "mfma_kc_v1") {
        h->mfma_kc_v1 = value != 0;
    } else if (n == "mfma_kc_tpw") {
        h->mfma_kc_tpw = value == 16 ? 16 : 8;
    } else if (n == "lut_pre") {
        h->lut_pre = value < 0 ? -1 : (value != 0);
    } else if (n == "mfma_blocks") {
        h->mfma_blocks = value > 0 ? value : 0;
    } else if (n == "grp_blocks") {
        h->grp_blocks = value > 0 ? value : 0;
    } else if (n == "passa_prefix") {
        h->passa_prefix = value > 0 ? value : 0;
    } else {
        return fail(MMIDX_ERR_INVALID_ARG, "unknown option '%s'", name);
    }
    return MMIDX_OK;
}

int mmidx_set_profiling(mmidx_index *h, int enabled) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (h->grp) return sharded_for_each(h, [&](mmidx_index *s) { return mmidx_set_profiling(s, enabled); });
    int rc = set_device(h);
    if (rc) return rc;
    HIPCK(hipDeviceSynchronize());
    h->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    h->stats = mmidx_stats{};
    h->ev_used = 0;
    h->mf_ev_used = 0;
    h->a_ev_used = 0;
    h->a_launches = 0;
    h->host_codes = 0;
    h->launches = 0;
    h->passa_launches = 0;
    h->host_passa_codes = 0;
    HIPCK(hipMemset(h->d_counters, 0, 8 * sizeof(u64)));
    HIPCK(hipMemset(h->d_counters + 12, 0, 2 * sizeof(u64)));
    return MMIDX_OK;
}

// Resolves every event recorded since the last call (synchronises with the launch stream) and
// returns the accumulated statistics; the accumulation restarts afterwards.
int mmidx_get_stats(mmidx_index *h, mmidx_stats *out) {
    if (!h || !out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->grp) return sharded_get_stats(h, out);
    int rc = set_device(h);
    if (rc) return rc;
    mmidx_stats s{};
    if (h->ev_used) HIPCK(hipEventSynchronize(h->evpool[h->ev_used - 1]));
    HIPCK(hipDeviceSynchronize());
    for (size_t g = 0; g + 6 <= h->ev_used; g += 6) {
        hipEvent_t *ev = h->evpool.data() + g;
        float a = 0, b = 0, c = 0, t = 0, pa = 0;
        if (h->profiling == 2) {  // light: only the pass-A pair was recorded
            HIPCK(hipEventElapsedTime(&pa, ev[2], ev[5]));
            s.passa_ms += pa;
            continue;
        }
        HIPCK(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIPCK(hipEventElapsedTime(&b, ev[2], ev[3]));
        HIPCK(hipEventElapsedTime(&c, ev[3], ev[4]));
        HIPCK(hipEventElapsedTime(&t, ev[0], ev[4]));
        HIPCK(hipEventElapsedTime(&pa, ev[2], ev[5]));
        s.coarse_ms += a;
        s.scan_ms += b;
        s.merge_ms += c;
        s.total_ms += t;
        s.passa_ms += pa;
    }
    u64 cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCK(hipMemcpy(cnt, h->d_counters, sizeof(cnt), hipMemcpyDeviceToHost));
    s.scan_codes = (int64_t)cnt[0] + h->host_codes;
    s.tie_fallbacks = (int32_t)cnt[1];
    s.scan_launches = h->launches;
    s.passa_codes = (int64_t)cnt[2] + h->host_passa_codes;
    s.passa_launches = h->passa_launches;
    s.verified_codes = (int64_t)cnt[3];
    for (size_t g = 0; g + 3 <= h->mf_ev_used; g += 3) {
        float a = 0.f, b = 0.f;
        HIPCK(hipEventSynchronize(h->mf_ev[g + 2]));
        HIPCK(hipEventElapsedTime(&a, h->mf_ev[g], h->mf_ev[g + 1]));
        HIPCK(hipEventElapsedTime(&b, h->mf_ev[g + 1], h->mf_ev[g + 2]));
        s.mfma_scan_ms += a;
        s.mfma_verify_ms += b;
        s.mfma_launches += 1;
    }
    h->mf_ev_used = 0;
    for (size_t g = 0; g + 5 <= h->a_ev_used; g += 5) {
        float t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f;
        HIPCK(hipEventSynchronize(h->a_ev[g + 4]));
        HIPCK(hipEventElapsedTime(&t1, h->a_ev[g], h->a_ev[g + 1]));
        HIPCK(hipEventElapsedTime(&t2, h->a_ev[g + 1], h->a_ev[g + 2]));
        HIPCK(hipEventElapsedTime(&t3, h->a_ev[g + 2], h->a_ev[g + 3]));
        HIPCK(hipEventElapsedTime(&t4, h->a_ev[g + 3], h->a_ev[g + 4]));
        s.passa_mfma_sweep1_ms += t1;
        s.passa_mfma_select_ms += t2;
        s.passa_mfma_sweep2_ms += t3;
        s.passa_mfma_verify_ms += t4;
    }
    h->a_ev_used = 0;
    s.passa_mfma_launches = h->a_launches;
    h->a_launches = 0;
    {
        u64 c2[2] = {0, 0};
        HIPCK(hipMemcpy(c2, h->d_counters + 12, sizeof(c2), hipMemcpyDeviceToHost));
        HIPCK(hipMemset(h->d_counters + 12, 0, sizeof(c2)));
        s.mfma_survivors = (int64_t)c2[0];
        s.mfma_redo_queries = (int64_t)c2[1];
    }
    if (h->debug_sync || getenv("MMIDX_GRP_STATS"))
    {
        fprintf(stderr, "[mmidx] K3g: %llu pairs in groups, %llu alive after the table build (Smin < T), %llu codes verified\n", (unsigned long long)cnt[5],
                (unsigned long long)cnt[6], (unsigned long long)cnt[3]);
        u64 c2[2] = {0, 0};
        (void)hipMemcpy(c2, h->d_counters + 10, sizeof(c2), hipMemcpyDeviceToHost);
        (void)hipMemset(h->d_counters + 10, 0, sizeof(c2));
        if (c2[1]) fprintf(stderr, "[mmidx] K3g: %llu of %llu codes have a pair alive after half their sub-quantizers\n", (unsigned long long)c2[0], (unsigned long long)c2[1]);
    }
    s.passb_items_last = h->pin_hint ? *(volatile int32_t *)h->pin_hint : -1;
    *out = s;
    h->ev_used = 0;
    h->host_codes = 0;
    h->launches = 0;
    h->passa_launches = 0;
    h->host_passa_codes = 0;
    HIPCK(hipMemset(h->d_counters, 0, 8 * sizeof(u64)));
    return MMIDX_OK;
}

/* ---- per-id utilities (IVFPQ.java:464-497, :801-880) ------------------------------------------------------------ */
namespace {
// iid -> position map over the list-major arrays; h->mu held by the caller
int ensure_inverse(mmidx_index *h) {
    if (h->inv_valid) return MMIDX_OK;
    int32_t *d_max = nullptr;
    HIPCK(hipMalloc((void **)&d_max, sizeof(int32_t)));
    int32_t mx = -1;
    HIPCK(hipMemcpy(d_max, &mx, sizeof(mx), hipMemcpyHostToDevice));
    const long long n = h->n_csr;
    if (n > 0) {
        hipLaunchKernelGGL(k_iid_max, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_ids, n, d_max);
        HIPCK(hipStreamSynchronize(h->stream));
    }
    hipError_t e = hipMemcpy(&mx, d_max, sizeof(mx), hipMemcpyDeviceToHost);
    (void)hipFree(d_max);
    if (e != hipSuccess) return fail(MMIDX_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    h->inv_size = (int64_t)mx + 1;
    HIPCK(h->ws_inv.reserve((size_t)std::max<int64_t>(h->inv_size, 1)));
    HIPCK(hipMemsetAsync(h->ws_inv.p, 0xFF, (size_t)std::max<int64_t>(h->inv_size, 1) * sizeof(int32_t), h->stream));
    if (n > 0) hipLaunchKernelGGL(k_inv_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_ids, n, h->ws_inv.p);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(h->stream));
    h->inv_valid = true;
    return MMIDX_OK;
}

// device lookup of n records: d_pos[n], d_code[n][m] (stored form); host cells from the list offsets
int lookup_records(mmidx_index *h, int64_t n, const int32_t *iids, int32_t **d_pos_out, void **d_code_out, std::vector<int32_t> &pos,
                   std::vector<int32_t> &cells, bool allow_missing = false) {
    int rc = build_csr(h);
    if (rc) return rc;
    rc = ensure_inverse(h);
    if (rc) return rc;
    int32_t *d_iids = nullptr, *d_pos = nullptr;
    void *d_code = nullptr;
    const size_t cbytes = (size_t)n * h->m * h->code_bytes;
    HIPCK(hipMalloc((void **)&d_iids, std::max<size_t>((size_t)n * 4, 16)));
    if (hipMalloc((void **)&d_pos, std::max<size_t>((size_t)n * 4, 16)) != hipSuccess || hipMalloc(&d_code, std::max<size_t>(cbytes, 16)) != hipSuccess) {
        (void)hipFree(d_iids);
        if (d_pos) (void)hipFree(d_pos);
        return fail(MMIDX_ERR_HIP, "hipMalloc failed");
    }
    auto cleanup = [&]() {
        (void)hipFree(d_iids);
        (void)hipFree(d_pos);
        (void)hipFree(d_code);
    };
    pos.assign((size_t)n, -1);
    cells.assign((size_t)n, -1);
    hipError_t e = hipMemcpyAsync(d_iids, iids, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (h->code_bytes == 1)
            hipLaunchKernelGGL(k_lookup_codes<unsigned char>, dim3(grid), dim3(256), 0, h->stream, d_iids, (long long)n, h->ws_inv.p, (long long)h->inv_size,
                               (const unsigned char *)h->d_codes, h->m, d_pos, (unsigned char *)d_code);
        else
            hipLaunchKernelGGL(k_lookup_codes<unsigned short>, dim3(grid), dim3(256), 0, h->stream, d_iids, (long long)n, h->ws_inv.p, (long long)h->inv_size,
                               (const unsigned short *)h->d_codes, h->m, d_pos, (unsigned short *)d_code);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(pos.data(), d_pos, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        cleanup();
        return fail(MMIDX_ERR_HIP, "record lookup failed: %s", hipGetErrorString(e));
    }
    for (int64_t i = 0; i < n; i++) {
        if (pos[(size_t)i] < 0) {
            if (allow_missing) continue;  // (a shard of a sharded handle: the id lives on another shard)
            cleanup();
            return fail(MMIDX_ERR_INVALID_ARG, "Id does not exist!");  // IVFPQ.java:803-805, :868-870
        }
        // list of a position: the last list whose start is <= pos
        const auto it = std::upper_bound(h->h_off.begin(), h->h_off.end(), (int64_t)pos[(size_t)i]);
        cells[(size_t)i] = (int32_t)(it - h->h_off.begin()) - 1;
    }
    (void)hipFree(d_iids);
    *d_pos_out = d_pos;
    *d_code_out = d_code;
    return MMIDX_OK;
}
}  // namespace

int mmidx_get_dispatch(mmidx_index *h, char *out, int cap) {
    if (!h || !out || cap < 1) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    const mmidx_index *s = h->grp ? sharded_first(h) : h;
    // (launch_scan_filtered / launch_scan: K3f falls back to the exact scan K3 for shapes it has no instance of)
    const bool k3f_ok = s->code_bytes == 1 && s->ks <= 256 && (s->m == 8 || s->m == 16 || s->m == 32) && !s->no_filter;
    const char *pb = s->disp_passb;
    if (!k3f_ok && pb[0] == 'K' && pb[1] == '3' && pb[2] == 'f') pb = "K3(exact scan: no K3f instance for this shape)";
    snprintf(out, (size_t)cap, "coarse=%s;pass_a=%s;pre=%s;pass_b=%s", s->disp_coarse, s->disp_passa, s->disp_pre, pb);
    return MMIDX_OK;
}

int mmidx_get_dims(const mmidx_index *h, int *D, int *m, int *ks, int *C, int *code_bytes) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (D) *D = h->D;
    if (m) *m = h->m;
    if (ks) *ks = h->ks;
    if (C) *C = h->C;
    if (code_bytes) *code_bytes = (int)h->code_bytes;
    return MMIDX_OK;
}

int mmidx_get_codes(mmidx_index *h, int64_t n, const int32_t *iids, int32_t *cell_out, void *code_out) {
    if (!h || (n > 0 && !iids)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (n < 0) return fail(MMIDX_ERR_INVALID_ARG, "n < 0");
    if (n == 0) return MMIDX_OK;
    if (h->grp) return sharded_get_codes(h, n, iids, cell_out, code_out);
    int rc = set_device(h);
    if (rc) return rc;
    DeviceCall call(h, h->stream);  // (the CSR rebuild below must not move the arrays under a search another stream is still running)
    std::lock_guard<std::mutex> lk(h->mu);
    int32_t *d_pos = nullptr;
    void *d_code = nullptr;
    std::vector<int32_t> pos, cells;
    rc = lookup_records(h, n, iids, &d_pos, &d_code, pos, cells);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if (code_out) e = hipMemcpy(code_out, d_code, (size_t)n * h->m * h->code_bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_pos);
    (void)hipFree(d_code);
    if (e != hipSuccess) return fail(MMIDX_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    if (cell_out)
        for (int64_t i = 0; i < n; i++) cell_out[i] = h->kind == MMIDX_KIND_IVFPQ ? cells[(size_t)i] : -1;
    return MMIDX_OK;
}

int mmidx_distance(mmidx_index *h, int64_t n, const double *Q, const int32_t *iids, double *dist_out) {
    if (!h || (n > 0 && (!Q || !iids || !dist_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (n < 0) return fail(MMIDX_ERR_INVALID_ARG, "n < 0");
    if (h->grp) return n == 0 ? MMIDX_OK : sharded_distance(h, n, Q, iids, dist_out);
    int rc = check_ready(h);
    if (rc) return rc;
    if (n == 0) return MMIDX_OK;
    rc = set_device(h);
    if (rc) return rc;
    DeviceCall call(h, h->stream);
    std::lock_guard<std::mutex> lk(h->mu);
    int32_t *d_pos = nullptr;
    void *d_code = nullptr;
    std::vector<int32_t> pos, cells;
    rc = lookup_records(h, n, iids, &d_pos, &d_code, pos, cells);
    if (rc) return rc;
    (void)hipFree(d_code);
    double *dQ = nullptr, *d_out = nullptr;
    int32_t *d_cell = nullptr;
    hipError_t e = hipMalloc((void **)&dQ, (size_t)n * h->D * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cell, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(dQ, Q, (size_t)n * h->D * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cell, cells.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        ScanParams P{};
        P.Q = dQ;
        P.coarse = h->d_coarse;
        P.perm = h->d_perm;
        P.rot = h->d_rot;
        P.codes = h->d_codes;
        P.D = h->D;
        P.m = h->m;
        P.ks = h->ks;
        P.dsub = h->dsub;
        P.transform = h->transform;
        P.ivf = h->kind == MMIDX_KIND_IVFPQ;
        const size_t lds = 2 * (size_t)h->D * 8;
        if (h->code_bytes == 1)
            hipLaunchKernelGGL(k_pair_distance<unsigned char>, dim3((unsigned)n), dim3(MMIDX_BLOCK), lds, h->stream, P, h->d_pq, d_pos, d_cell, d_out);
        else
            hipLaunchKernelGGL(k_pair_distance<unsigned short>, dim3((unsigned)n), dim3(MMIDX_BLOCK), lds, h->stream, P, h->d_pq, d_pos, d_cell, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dist_out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_pos);
    if (dQ) (void)hipFree(dQ);
    if (d_out) (void)hipFree(d_out);
    if (d_cell) (void)hipFree(d_cell);
    if (e != hipSuccess) return fail(MMIDX_ERR_HIP, "mmidx_distance failed: %s", hipGetErrorString(e));
    return MMIDX_OK;
}

/* snapshot of the in-memory index, list-major (the layout loadIndexInMemory builds) */
int mmidx_export(mmidx_index *h, int64_t *list_off_out, int32_t *iids_out, void *codes_out) {
    if (!h || !list_off_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (h->grp) return sharded_export(h, list_off_out, iids_out, codes_out);
    int rc = mmidx_sync_index(h);
    if (rc) return rc;
    for (int c = 0; c <= h->nlists; c++) list_off_out[c] = h->h_off[(size_t)c];
    const int64_t n = h->n_csr;
    if (n == 0) return MMIDX_OK;
    if (iids_out) HIPCK(hipMemcpy(iids_out, h->d_ids, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (codes_out) {
        const size_t bytes = (size_t)n * h->m * h->code_bytes;
        HIPCK(hipMemcpy(codes_out, h->d_codes, bytes, hipMemcpyDeviceToHost));
        if (h->code_bytes == 1) {  // stored form idx - 128 (PQ.java:555)
            unsigned char *c = (unsigned char *)codes_out;
            for (size_t t = 0; t < bytes; t++) c[t] = (unsigned char)(c[t] ^ 0x80);
        }
    }
    return MMIDX_OK;
}

// ---- native flat snapshot (fast restart alongside BDB; the load path it replaces: IVFPQ.java:680-728, PQ.java:436-483) ----
// File: header, list_off[nlists + 1] (int64), iids[n] (int32), codes[n][m] in stored form -- exactly what mmidx_export returns, so a
// list keeps its arrival order (= the reference's offer order).  Little-endian, no padding between the arrays.
struct MmidxSnapHeader {
    char magic[8];  // "MMIDXSN1"
    uint32_t version, kind, D, m, ks, C, code_bytes, transform;
    uint64_t n, nlists;
};

int mmidx_save(mmidx_index *h, const char *path) {
    if (!h || !path) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int64_t n = 0;
    int rc = mmidx_size(h, &n);
    if (rc) return rc;
    int D = 0, m = 0, ks = 0, C = 0, cb = 0;
    rc = mmidx_get_dims(h, &D, &m, &ks, &C, &cb);
    if (rc) return rc;
    const int64_t nlists = h->kind == MMIDX_KIND_IVFPQ ? C : 1;
    std::vector<int64_t> off((size_t)nlists + 1, 0);
    std::vector<int32_t> iids((size_t)n);
    std::vector<unsigned char> codes((size_t)n * m * cb);
    rc = mmidx_export(h, off.data(), n ? iids.data() : nullptr, n ? (void *)codes.data() : nullptr);
    if (rc) return rc;
    if (off[(size_t)nlists] != n) return fail(MMIDX_ERR_HIP, "export returned %lld records, the index holds %lld", (long long)off[(size_t)nlists], (long long)n);
    MmidxSnapHeader hd{};
    memcpy(hd.magic, "MMIDXSN1", 8);
    hd.version = 1;
    hd.kind = (uint32_t)h->kind;
    hd.D = (uint32_t)D;
    hd.m = (uint32_t)m;
    hd.ks = (uint32_t)ks;
    hd.C = (uint32_t)C;
    hd.code_bytes = (uint32_t)cb;
    hd.transform = (uint32_t)h->transform;
    hd.n = (uint64_t)n;
    hd.nlists = (uint64_t)nlists;
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(MMIDX_ERR_INVALID_ARG, "cannot open %s for writing", tmp.c_str());
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1 && fwrite(off.data(), 8, off.size(), f) == off.size();
    ok = ok && (n == 0 || (fwrite(iids.data(), 4, iids.size(), f) == iids.size() && fwrite(codes.data(), 1, codes.size(), f) == codes.size()));
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        (void)remove(tmp.c_str());
        return fail(MMIDX_ERR_INVALID_ARG, "writing the snapshot %s failed", path);
    }
    return MMIDX_OK;
}

int mmidx_load(mmidx_index *h, const char *path) {
    if (!h || !path) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int64_t have = 0;
    int rc = mmidx_size(h, &have);
    if (rc) return rc;
    if (have != 0) return fail(MMIDX_ERR_INVALID_ARG, "mmidx_load needs an empty index (this one holds %lld records)", (long long)have);
    int D = 0, m = 0, ks = 0, C = 0, cb = 0;
    rc = mmidx_get_dims(h, &D, &m, &ks, &C, &cb);
    if (rc) return rc;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MMIDX_ERR_INVALID_ARG, "cannot open the snapshot %s", path);
    MmidxSnapHeader hd{};
    const bool hok = fread(&hd, sizeof(hd), 1, f) == 1 && memcmp(hd.magic, "MMIDXSN1", 8) == 0 && hd.version == 1;
    const int64_t nlists = h->kind == MMIDX_KIND_IVFPQ ? C : 1;
    if (!hok || hd.kind != (uint32_t)h->kind || hd.D != (uint32_t)D || hd.m != (uint32_t)m || hd.ks != (uint32_t)ks || hd.C != (uint32_t)C ||
        hd.code_bytes != (uint32_t)cb || hd.transform != (uint32_t)h->transform || hd.nlists != (uint64_t)nlists) {
        fclose(f);
        return fail(MMIDX_ERR_INVALID_ARG, "%s is not a snapshot of an index of this shape (kind / D / m / ks / C / code width / transform)", path);
    }
    std::vector<int64_t> off((size_t)nlists + 1);
    bool ok = fread(off.data(), 8, off.size(), f) == off.size() && off[0] == 0 && off[(size_t)nlists] == (int64_t)hd.n;
    for (int64_t c = 0; ok && c < nlists; c++) ok = off[(size_t)c] <= off[(size_t)c + 1];
    if (!ok) {
        fclose(f);
        return fail(MMIDX_ERR_INVALID_ARG, "%s: damaged list offsets", path);
    }
    const int64_t n = (int64_t)hd.n;
    const size_t rec = (size_t)m * cb;
    // records go in list-major, in pieces of <= 16 M (the lists keep their order: mmidx_add_codes appends in the order given)
    const int64_t piece = 16ll << 20;
    std::vector<int32_t> iids, cells;
    std::vector<unsigned char> codes;
    const long data0 = ftell(f);
    int64_t c = 0;
    for (int64_t p0 = 0; p0 < n && ok; p0 += piece) {
        const int64_t pn = std::min(piece, n - p0);
        iids.resize((size_t)pn);
        cells.resize((size_t)pn);
        codes.resize((size_t)pn * rec);
        ok = fseek(f, data0 + (long)(p0 * 4), SEEK_SET) == 0 && fread(iids.data(), 4, (size_t)pn, f) == (size_t)pn;
        ok = ok && fseek(f, data0 + (long)(n * 4) + (long)((size_t)p0 * rec), SEEK_SET) == 0 && fread(codes.data(), 1, codes.size(), f) == codes.size();
        if (!ok) break;
        for (int64_t i = 0; i < pn; i++) {
            while (p0 + i >= off[(size_t)c + 1]) c++;
            cells[(size_t)i] = (int32_t)c;
        }
        rc = mmidx_add_codes(h, pn, iids.data(), h->kind == MMIDX_KIND_IVFPQ ? cells.data() : nullptr, codes.data());
        if (rc) {
            fclose(f);
            return rc;
        }
    }
    fclose(f);
    if (!ok) return fail(MMIDX_ERR_INVALID_ARG, "%s: truncated snapshot", path);
    return mmidx_sync_index(h);
}

}  // extern "C"

// =================================================================================================
// Front end of BASELINE config 5: PCA projection (K7) and VLAD aggregation (K8)
// =================================================================================================
struct mmidx_pca {
    std::mutex mu;  // host-pointer calls share the workspaces
    int nc = 0, ss = 0, whitening = 0, device = 0;
    double *d_mu = nullptr, *d_Vt = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<double> ws_X, ws_Y;
};
struct mmidx_vlad {
    std::mutex mu;  // host-pointer calls share the workspaces
    int nvocab = 0, dl = 0, norms = 0, device = 0, veclen = 0;
    std::vector<int> nc;
    std::vector<size_t> cb_off;  // element offset of each codebook
    double *d_cb = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<double> ws_desc, ws_out;
    DevBuf<long long> ws_off;
    // K8': the assignment of every descriptor of a launch through the encoder's certified MFMA argmin -- one hidden index handle
    // per vocabulary whose "coarse quantizer" is the vocabulary
    std::vector<mmidx_index *> asg;
    DevBuf<int32_t> ws_nn;
    int exact = 0;  // option "exact" / MMIDX_VLAD_EXACT=1: the one-kernel form (k_vlad: fp64 brute-force assignment inside the block)
    int two_pass = 0;  // option "two_pass": K8' also where K8'' (k_vlad_fused) applies
};

extern "C" {

int mmidx_pca_create(int nc, int ss, int whitening, const double *means, const double *eig, const double *Vt, int device,
                     mmidx_pca **out) {
    if (!out) return fail(MMIDX_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (nc < 1 || ss < 1 || !means || !Vt) return fail(MMIDX_ERR_INVALID_ARG, "bad PCA shape or null matrix");
    if (whitening && !eig) return fail(MMIDX_ERR_INVALID_ARG, "whitening needs the eigenvalues line of the PCA file");
    const int ndev = mmidx_device_count();
    if (ndev < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MMIDX_ERR_NO_DEVICE, "device %d outside 0..%d", device, ndev - 1);
    HIPCK(hipSetDevice(device));
    mmidx_pca *p = new mmidx_pca();
    p->nc = nc;
    p->ss = ss;
    p->whitening = whitening ? 1 : 0;
    p->device = device;
    std::vector<double> V((size_t)nc * ss);
    for (int i = 0; i < nc; i++) {
        // W(i,i) = pow(eig_i, -0.5); V_t <- W * V_t  (PCA.java:283-285, :311): row i scaled by w_ii
        const double wv = whitening ? std::pow(eig[i], -0.5) : 1.0;
        for (int j = 0; j < ss; j++) V[(size_t)i * ss + j] = whitening ? wv * Vt[(size_t)i * ss + j] : Vt[(size_t)i * ss + j];
    }
    HIPCK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIPCK(hipMalloc((void **)&p->d_mu, (size_t)ss * 8));
    HIPCK(hipMalloc((void **)&p->d_Vt, (size_t)nc * ss * 8));
    HIPCK(hipMemcpy(p->d_mu, means, (size_t)ss * 8, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(p->d_Vt, V.data(), (size_t)nc * ss * 8, hipMemcpyHostToDevice));
    *out = p;
    return MMIDX_OK;
}

int mmidx_pca_get_dims(const mmidx_pca *p, int *nc_out, int *ss_out) {
    if (!p) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nc_out) *nc_out = p->nc;
    if (ss_out) *ss_out = p->ss;
    return MMIDX_OK;
}

int mmidx_pca_destroy(mmidx_pca *p) {
    if (!p) return MMIDX_OK;
    (void)hipSetDevice(p->device);
    if (p->d_mu) (void)hipFree(p->d_mu);
    if (p->d_Vt) (void)hipFree(p->d_Vt);
    p->ws_X.release();
    p->ws_Y.release();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return MMIDX_OK;
}

int mmidx_pca_project_device(mmidx_pca *p, int64_t n, const double *dX, double *dY, void *stream) {
    if (!p) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (n < 0 || (n > 0 && (!dX || !dY))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (n == 0) return MMIDX_OK;
    HIPCK(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((n + PCA_BM - 1) / PCA_BM), (unsigned)((p->nc + PCA_BN - 1) / PCA_BN));
    HIPCK(hipFuncSetAttribute((const void *)k_pca_project, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PCA_LDS_BYTES));
    hipLaunchKernelGGL(k_pca_project, grid, dim3(PCA_NT), PCA_LDS_BYTES, st, dX, p->d_mu, p->d_Vt, dY, (long long)n, p->nc, p->ss);
    if (p->whitening)
        hipLaunchKernelGGL(k_rows_normalize_l2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dY, (long long)n, p->nc);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

int mmidx_pca_project(mmidx_pca *p, int64_t n, const double *X, double *Y) {
    if (!p) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (n < 0 || (n > 0 && (!X || !Y))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIPCK(hipSetDevice(p->device));
    const int64_t B = std::max<int64_t>(1, (int64_t)(1ll << 28) / p->ss);  // <= 2 GiB of samples per round
    for (int64_t i0 = 0; i0 < n; i0 += B) {
        const int64_t nb = std::min(B, n - i0);
        HIPCK(p->ws_X.reserve((size_t)nb * p->ss));
        HIPCK(p->ws_Y.reserve((size_t)nb * p->nc));
        HIPCK(hipMemcpyAsync(p->ws_X.p, X + (size_t)i0 * p->ss, (size_t)nb * p->ss * 8, hipMemcpyHostToDevice, p->stream));
        int rc = mmidx_pca_project_device(p, nb, p->ws_X.p, p->ws_Y.p, p->stream);
        if (rc) return rc;
        HIPCK(hipMemcpyAsync(Y + (size_t)i0 * p->nc, p->ws_Y.p, (size_t)nb * p->nc * 8, hipMemcpyDeviceToHost, p->stream));
        HIPCK(hipStreamSynchronize(p->stream));
    }
    return MMIDX_OK;
}

int mmidx_vlad_create(int nvocab, const int32_t *ncent, int dl, const double *codebooks, int normalizations_on, int device,
                      mmidx_vlad **out) {
    if (!out) return fail(MMIDX_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (nvocab < 1 || !ncent || dl < 1 || !codebooks) return fail(MMIDX_ERR_INVALID_ARG, "bad codebook description");
    const int ndev = mmidx_device_count();
    if (ndev < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MMIDX_ERR_NO_DEVICE, "device %d outside 0..%d", device, ndev - 1);
    HIPCK(hipSetDevice(device));
    mmidx_vlad *v = new mmidx_vlad();
    v->nvocab = nvocab;
    v->dl = dl;
    v->norms = normalizations_on ? 1 : 0;
    v->device = device;
    size_t tot = 0;
    for (int i = 0; i < nvocab; i++) {
        if (ncent[i] < 1) {
            delete v;
            return fail(MMIDX_ERR_INVALID_ARG, "empty codebook");
        }
        v->nc.push_back(ncent[i]);
        v->cb_off.push_back(tot);
        tot += (size_t)ncent[i] * dl;
    }
    v->veclen = (int)tot;
    HIPCK(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    HIPCK(hipMalloc((void **)&v->d_cb, tot * 8));
    HIPCK(hipMemcpy(v->d_cb, codebooks, tot * 8, hipMemcpyHostToDevice));
    v->exact = getenv("MMIDX_VLAD_EXACT") ? 1 : 0;
    for (int i = 0; i < nvocab; i++) {  // (a vocabulary the assignment kernels cannot take leaves its slot empty: k_vlad serves it)
        mmidx_index *a = nullptr;
        if (ncent[i] >= 2) {
            int rca = mmidx_create(MMIDX_KIND_IVFPQ, dl, 1, 2, ncent[i], MMIDX_TR_NONE, nullptr, nullptr, device, &a);
            if (rca == MMIDX_OK) rca = mmidx_set_coarse(a, codebooks + v->cb_off[(size_t)i]);
            if (rca != MMIDX_OK) {  // the slot falls back to k_vlad: not an error of this call, so no stale message either
                if (a) mmidx_destroy(a);
                a = nullptr;
                g_err.clear();
            }
        }
        v->asg.push_back(a);
    }
    *out = v;
    return MMIDX_OK;
}

int mmidx_vlad_set_option(mmidx_vlad *v, const char *name, int value) {
    if (!v || !name) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (std::string(name) == "exact") {
        v->exact = value != 0;
        return MMIDX_OK;
    }
    if (std::string(name) == "two_pass") {  // K8' (assignment kernel + accumulation kernel) also where the one-kernel form K8'' applies (A/B switch)
        v->two_pass = value != 0;
        return MMIDX_OK;
    }
    return fail(MMIDX_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int mmidx_vlad_destroy(mmidx_vlad *v) {
    if (!v) return MMIDX_OK;
    (void)hipSetDevice(v->device);
    if (v->d_cb) (void)hipFree(v->d_cb);
    for (mmidx_index *a : v->asg)
        if (a) mmidx_destroy(a);
    v->ws_nn.release();
    v->ws_desc.release();
    v->ws_out.release();
    v->ws_off.release();
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
    return MMIDX_OK;
}

int mmidx_vlad_descriptor_length(const mmidx_vlad *v, int *dl_out) {
    if (!v || !dl_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *dl_out = v->dl;
    return MMIDX_OK;
}

int mmidx_vlad_vector_length(const mmidx_vlad *v, int *len_out) {
    if (!v || !len_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *len_out = v->veclen;
    return MMIDX_OK;
}

// max_desc: largest descriptor count of any image in the batch (sizes the LDS work lists)
int mmidx_vlad_aggregate_device(mmidx_vlad *v, int64_t nimg, const int64_t *d_desc_off, const double *d_descs, int max_desc,
                                double *d_out, void *stream) {
    if (!v) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nimg < 0 || (nimg > 0 && (!d_desc_off || !d_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (nimg == 0) return MMIDX_OK;
    HIPCK(hipSetDevice(v->device));
    hipStream_t st = (hipStream_t)stream;
    const int maxnd = (std::max(max_desc, 2) + 1) & ~1;
    long long ndesc = -1;  // descriptors of the launch (read back once when the assignment runs as its own stage)
    for (int i = 0; i < v->nvocab; i++) {
        const int nc = v->nc[(size_t)i];
        if (!v->exact && !v->two_pass && v->asg[(size_t)i] && d_descs && v->dl == 64 && nc <= 128) {
            // K8'': one kernel, one pass over the descriptors in HBM, no host synchronisation (the flagged descriptors are redone by the
            // image's own block): 64-dimensional descriptors, vocabularies of at most 128 centroids
            const mmidx_index *a = v->asg[(size_t)i];
            if (a->d_Ch && a->Cp == G16_BC && a->Dp >= 64 && a->Dp <= G16_KC) {
                const size_t lf = 2 * (size_t)G16_BC * (64 * 2 + 16) + 2 * (size_t)maxnd * 4 + (size_t)((nc + 2) & ~1) * 4 + (size_t)(VF_FLAG_CAP + 2) * 4 + 32;
                if (lf <= 160 * 1024 && a->d_coarseT) {
                    HIPCK(hipFuncSetAttribute((const void *)k_vlad_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf));
                    hipLaunchKernelGGL(k_vlad_fused, dim3((unsigned)nimg), dim3(256), lf, st, v->d_cb + v->cb_off[(size_t)i], (const double *)a->d_coarseT, nc, maxnd, (const __bf16 *)a->d_Ch,
                                       (const __bf16 *)a->d_Cl, a->d_cn_pad, a->cnorm_max, a->cn_max, a->Dp, (const long long *)d_desc_off, d_descs, d_out, v->veclen,
                                       (int)v->cb_off[(size_t)i], v->norms);
                    HIPCK(hipGetLastError());
                    continue;
                }
            }
        }
        if (!v->exact && v->asg[(size_t)i] && d_descs) {
            // K8': nearest centroid of EVERY descriptor on the matrix cores (certified, exact redo of the flagged few), then one
            // block per image for the ordered accumulation
            if (ndesc < 0) {
                HIPCK(hipMemcpyAsync(&ndesc, d_desc_off + nimg, sizeof(long long), hipMemcpyDeviceToHost, st));
                HIPCK(hipStreamSynchronize(st));
                HIPCK(v->ws_nn.reserve((size_t)std::max<long long>(ndesc, 1)));
            }
            if (ndesc > 0) {
                int rca = mmidx_assign_device(v->asg[(size_t)i], ndesc, d_descs, v->ws_nn.p, st);
                if (rca) return rca;
                HIPCK(hipSetDevice(v->device));
            }
            const size_t lds2 = 2 * (size_t)maxnd * 4 + (size_t)((nc + 2) & ~1) * 4 + 32;
            if (lds2 > 64 * 1024) return fail(MMIDX_ERR_UNSUPPORTED, "%d descriptors per image exceed the accumulation kernel's LDS", max_desc);
            if (v->dl == 64)
                hipLaunchKernelGGL(k_vlad_accum<64>, dim3((unsigned)nimg), dim3(256), lds2, st, v->d_cb + v->cb_off[(size_t)i], nc, v->dl, maxnd, v->ws_nn.p,
                                   (const long long *)d_desc_off, d_descs, d_out, v->veclen, (int)v->cb_off[(size_t)i], v->norms);
            else
                hipLaunchKernelGGL(k_vlad_accum<0>, dim3((unsigned)nimg), dim3(256), lds2, st, v->d_cb + v->cb_off[(size_t)i], nc, v->dl, maxnd, v->ws_nn.p,
                                   (const long long *)d_desc_off, d_descs, d_out, v->veclen, (int)v->cb_off[(size_t)i], v->norms);
            HIPCK(hipGetLastError());
            continue;
        }
        const size_t lds = (size_t)nc * v->dl * 8 + 2 * (size_t)maxnd * 4 + (size_t)((nc + 2) & ~1) * 4 + 32;
        if (lds > 160 * 1024)
            return fail(MMIDX_ERR_UNSUPPORTED, "codebook %d x %d plus %d descriptors per image exceed the 160 KiB LDS", nc, v->dl, max_desc);
        const int norms = v->norms;
        if (v->dl == 64) {
            HIPCK(hipFuncSetAttribute((const void *)k_vlad<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_vlad<64>, dim3((unsigned)nimg), dim3(256), lds, st, v->d_cb + v->cb_off[(size_t)i], nc, v->dl, maxnd,
                               (const long long *)d_desc_off, d_descs, d_out, v->veclen, (int)v->cb_off[(size_t)i], norms);
        } else {
            HIPCK(hipFuncSetAttribute((const void *)k_vlad<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_vlad<0>, dim3((unsigned)nimg), dim3(256), lds, st, v->d_cb + v->cb_off[(size_t)i], nc, v->dl, maxnd,
                               (const long long *)d_desc_off, d_descs, d_out, v->veclen, (int)v->cb_off[(size_t)i], norms);
        }
    }
    if (v->nvocab > 1 && v->norms)
        hipLaunchKernelGGL(k_rows_normalize_l2_block, dim3((unsigned)nimg), dim3(256), 0, st, d_out, v->veclen);
    HIPCK(hipGetLastError());
    return MMIDX_OK;
}

int mmidx_vlad_aggregate(mmidx_vlad *v, int64_t nimg, const int64_t *desc_off, const double *descs, double *out) {
    if (!v) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nimg < 0 || (nimg > 0 && (!desc_off || !out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (nimg == 0) return MMIDX_OK;
    std::lock_guard<std::mutex> lk(v->mu);
    HIPCK(hipSetDevice(v->device));
    const int64_t total = desc_off[nimg] - desc_off[0];
    if (total > 0 && !descs) return fail(MMIDX_ERR_INVALID_ARG, "null descriptors");
    int max_desc = 0;
    std::vector<long long> off((size_t)nimg + 1);
    for (int64_t i = 0; i <= nimg; i++) off[(size_t)i] = desc_off[i] - desc_off[0];
    for (int64_t i = 0; i < nimg; i++) max_desc = std::max<int>(max_desc, (int)(off[(size_t)i + 1] - off[(size_t)i]));
    HIPCK(v->ws_off.reserve((size_t)nimg + 1));
    HIPCK(v->ws_desc.reserve((size_t)std::max<int64_t>(total, 1) * v->dl));
    HIPCK(v->ws_out.reserve((size_t)nimg * v->veclen));
    HIPCK(hipMemcpyAsync(v->ws_off.p, off.data(), ((size_t)nimg + 1) * 8, hipMemcpyHostToDevice, v->stream));
    if (total > 0)
        HIPCK(hipMemcpyAsync(v->ws_desc.p, descs + (size_t)desc_off[0] * v->dl, (size_t)total * v->dl * 8, hipMemcpyHostToDevice, v->stream));
    int rc = mmidx_vlad_aggregate_device(v, nimg, (const int64_t *)v->ws_off.p, v->ws_desc.p, max_desc, v->ws_out.p, v->stream);
    if (rc) return rc;
    HIPCK(hipMemcpyAsync(out, v->ws_out.p, (size_t)nimg * v->veclen * 8, hipMemcpyDeviceToHost, v->stream));
    HIPCK(hipStreamSynchronize(v->stream));
    return MMIDX_OK;
}

// ImageVectorization.transformToVector (J/vectorization/ImageVectorization.java:169-208) for a batch: aggregate, then
// PCA.sampleToEigenSpace -- descriptors in, projected vectors out, the VLAD vectors never leave the device
int mmidx_vectorize_device(mmidx_vlad *v, mmidx_pca *p, int64_t nimg, const int64_t *d_desc_off, const double *d_descs, int max_desc,
                           double *d_out, void *stream) {
    if (!v || !p) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (v->veclen != p->ss)
        return fail(MMIDX_ERR_WRONG_DIM, "VLAD vector length %d does not match the PCA sample size %d", v->veclen, p->ss);
    if (v->device != p->device) return fail(MMIDX_ERR_INVALID_ARG, "aggregator and PCA live on different devices");
    if (nimg < 0 || (nimg > 0 && (!d_desc_off || !d_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (nimg == 0) return MMIDX_OK;
    HIPCK(hipSetDevice(v->device));
    const int64_t B = std::max<int64_t>(1, (int64_t)(1ll << 28) / v->veclen);  // <= 2 GiB of VLAD vectors per round
    for (int64_t i0 = 0; i0 < nimg; i0 += B) {
        const int64_t nb = std::min(B, nimg - i0);
        HIPCK(v->ws_out.reserve((size_t)nb * v->veclen));
        // (the offsets are absolute into d_descs: a sub-range of images needs no rebasing)
        int rc = mmidx_vlad_aggregate_device(v, nb, d_desc_off + i0, d_descs, max_desc, v->ws_out.p, stream);
        if (rc) return rc;
        rc = mmidx_pca_project_device(p, nb, v->ws_out.p, d_out + (size_t)i0 * p->nc, stream);
        if (rc) return rc;
    }
    return MMIDX_OK;
}

int mmidx_vectorize(mmidx_vlad *v, mmidx_pca *p, int64_t nimg, const int64_t *desc_off, const double *descs, double *out) {
    if (!v || !p) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (nimg < 0 || (nimg > 0 && (!desc_off || !out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (nimg == 0) return MMIDX_OK;
    std::lock_guard<std::mutex> lk(v->mu);
    std::lock_guard<std::mutex> lk2(p->mu);
    HIPCK(hipSetDevice(v->device));
    const int64_t total = desc_off[nimg] - desc_off[0];
    if (total > 0 && !descs) return fail(MMIDX_ERR_INVALID_ARG, "null descriptors");
    int max_desc = 0;
    std::vector<long long> off((size_t)nimg + 1);
    for (int64_t i = 0; i <= nimg; i++) off[(size_t)i] = desc_off[i] - desc_off[0];
    for (int64_t i = 0; i < nimg; i++) max_desc = std::max<int>(max_desc, (int)(off[(size_t)i + 1] - off[(size_t)i]));
    HIPCK(v->ws_off.reserve((size_t)nimg + 1));
    HIPCK(v->ws_desc.reserve((size_t)std::max<int64_t>(total, 1) * v->dl));
    HIPCK(p->ws_Y.reserve((size_t)nimg * p->nc));
    HIPCK(hipMemcpyAsync(v->ws_off.p, off.data(), ((size_t)nimg + 1) * 8, hipMemcpyHostToDevice, v->stream));
    if (total > 0)
        HIPCK(hipMemcpyAsync(v->ws_desc.p, descs + (size_t)desc_off[0] * v->dl, (size_t)total * v->dl * 8, hipMemcpyHostToDevice, v->stream));
    int rc = mmidx_vectorize_device(v, p, nimg, (const int64_t *)v->ws_off.p, v->ws_desc.p, max_desc, p->ws_Y.p, v->stream);
    if (rc) return rc;
    HIPCK(hipMemcpyAsync(out, p->ws_Y.p, (size_t)nimg * p->nc * 8, hipMemcpyDeviceToHost, v->stream));
    HIPCK(hipStreamSynchronize(v->stream));
    return MMIDX_OK;
}

// ---- Linear (exhaustive exact search, J/datastructures/Linear.java) -----------------------------------------------
// computeNearestNeighborsInternal (Linear.java:138-163) offers (i, sum_j (q_j - x_ij)^2) for every vector in index order
// to a bounded queue of size k: exactly what computeNearestCoarseIndices does with the coarse centroids and w, so the
// indexed vectors are handed to the coarse stage as "centroids" (certified bf16 / fp32 matrix-core filter + exact fp64 for
// the few candidates while n <= 16384, the plain exact kernels beyond that); (q - x)^2 and (x - q)^2 are the same bits.
struct mmidx_linear {
    std::mutex mu;
    int D = 0, device = 0;
    int64_t capacity = 0;
    std::vector<double> X;  // [n][D]; Linear keeps its vectors in memory too (TDoubleArrayList, Linear.java:45)
    mmidx_index *inner = nullptr;
    int64_t inner_n = -1;   // number of vectors the inner handle was built for
    DevBuf<double> ws_Q, ws_d;
    DevBuf<int32_t> ws_i;
    Combiner comb;              // concurrent one-query callers are served together, as in mmidx_search
    std::vector<double> cat_Q;  // their queries, concatenated
};

int mmidx_linear_create(int D, int64_t capacity, int device, mmidx_linear **out) {
    if (!out) return fail(MMIDX_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (D < 1 || capacity < 0) return fail(MMIDX_ERR_INVALID_ARG, "bad vector length / capacity");
    const int ndev = mmidx_device_count();
    if (ndev < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MMIDX_ERR_NO_DEVICE, "device %d outside 0..%d", device, ndev - 1);
    mmidx_linear *l = new mmidx_linear();
    l->D = D;
    l->device = device;
    l->capacity = capacity;
    *out = l;
    return MMIDX_OK;
}

int mmidx_linear_destroy(mmidx_linear *l) {
    if (!l) return MMIDX_OK;
    (void)hipSetDevice(l->device);
    if (l->inner) mmidx_destroy(l->inner);
    l->ws_Q.release();
    l->ws_d.release();
    l->ws_i.release();
    delete l;
    return MMIDX_OK;
}

int mmidx_linear_add(mmidx_linear *l, int64_t n, const double *X) {  // indexVectorInternal, Linear.java:111-122
    if (!l || n < 0 || (n > 0 && !X)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(l->mu);
    const int64_t have = (int64_t)(l->X.size() / (size_t)l->D);
    if (l->capacity > 0 && have + n > l->capacity) return fail(MMIDX_ERR_CAPACITY, "Maximum index capacity reached, no more vectors can be indexed!");
    l->X.insert(l->X.end(), X, X + (size_t)n * l->D);
    return MMIDX_OK;
}

int mmidx_linear_get_dim(const mmidx_linear *l, int *D_out) {
    if (!l || !D_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *D_out = l->D;
    return MMIDX_OK;
}

int mmidx_linear_size(const mmidx_linear *l, int64_t *n_out) {
    if (!l || !n_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *n_out = (int64_t)(l->X.size() / (size_t)l->D);
    return MMIDX_OK;
}

int mmidx_linear_get_vector(const mmidx_linear *l, int64_t iid, double *out) {  // Linear.getVector, Linear.java:253-263
    if (!l || !out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    const int64_t have = (int64_t)(l->X.size() / (size_t)l->D);
    if (iid < 0 || iid >= have) return fail(MMIDX_ERR_INVALID_ARG, "Internal id %lld is out of range!", (long long)iid);
    memcpy(out, l->X.data() + (size_t)iid * l->D, (size_t)l->D * 8);
    return MMIDX_OK;
}

// serves batch[0..nb) (same k) as one search over the concatenated queries; the caller holds l->mu
static int linear_search_batch(mmidx_linear *l, SearchReq *const *batch, size_t nb) {
    const int k = batch[0]->k;
    const int64_t n = (int64_t)(l->X.size() / (size_t)l->D);
    if (n > 0x7fffffff) return fail(MMIDX_ERR_CAPACITY, "internal ids are 32-bit, as in the reference");
    int64_t nq = 0;
    for (size_t b = 0; b < nb; b++) {
        SearchReq *r = batch[b];
        for (int64_t i = 0; i < r->nq * k; i++) {
            r->iid[i] = -1;
            r->dist[i] = std::numeric_limits<double>::infinity();
        }
        for (int64_t q = 0; q < r->nq; q++) r->cnt[q] = 0;
        nq += r->nq;
    }
    if (nq == 0 || n == 0) return MMIDX_OK;
    const double *Q = batch[0]->Q;
    if (nb > 1) {
        l->cat_Q.resize((size_t)nq * l->D);
        size_t off = 0;
        for (size_t b = 0; b < nb; b++) {
            memcpy(l->cat_Q.data() + off, batch[b]->Q, (size_t)batch[b]->nq * l->D * 8);
            off += (size_t)batch[b]->nq * l->D;
        }
        Q = l->cat_Q.data();
    }
    HIPCK(hipSetDevice(l->device));
    if (l->inner_n != n) {
        if (l->inner) mmidx_destroy(l->inner);
        l->inner = nullptr;
        l->inner_n = -1;
        int rc = mmidx_create(MMIDX_KIND_IVFPQ, l->D, 1, 2, (int)n, MMIDX_TR_NONE, nullptr, nullptr, l->device, &l->inner);
        if (rc) return rc;
        rc = mmidx_set_coarse(l->inner, l->X.data());
        if (rc) return rc;
        l->inner_n = n;
    }
    mmidx_index *h = l->inner;
    const int w = (int)std::min<int64_t>(k, n);
    h->w = w;
    hipStream_t st = h->stream;
    const int64_t qb = std::max<int64_t>(1, std::min<int64_t>(nq, (2ll << 30) / ((int64_t)n * 8)));
    HIPCK(l->ws_Q.reserve((size_t)qb * l->D));
    HIPCK(l->ws_i.reserve((size_t)qb * w));
    HIPCK(l->ws_d.reserve((size_t)qb * w));
    std::vector<int32_t> hi((size_t)qb * w);
    std::vector<double> hd((size_t)qb * w);
    size_t cur = 0;        // request that holds query q0 + q, and that query's position in it
    int64_t cur_q = 0;
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        const int64_t nbq = std::min(qb, nq - q0);
        HIPCK(hipMemcpyAsync(l->ws_Q.p, Q + (size_t)q0 * l->D, (size_t)nbq * l->D * 8, hipMemcpyHostToDevice, st));
        int rc = run_coarse(h, nbq, l->ws_Q.p, l->ws_i.p, st);
        if (rc) return rc;
        if (h->cdsel_valid) {
            HIPCK(hipMemcpyAsync(l->ws_d.p, h->ws_cdsel.p, (size_t)nbq * w * 8, hipMemcpyDeviceToDevice, st));
        } else {
            const long long tot = (long long)nbq * w;
            hipLaunchKernelGGL(k_gather_cdist, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->ws_cdist.p, l->ws_i.p, l->ws_d.p,
                               h->C, w, tot);
            HIPCK(hipGetLastError());
        }
        HIPCK(hipMemcpyAsync(hi.data(), l->ws_i.p, (size_t)nbq * w * 4, hipMemcpyDeviceToHost, st));
        HIPCK(hipMemcpyAsync(hd.data(), l->ws_d.p, (size_t)nbq * w * 8, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        for (int64_t q = 0; q < nbq; q++) {
            while (cur_q >= batch[cur]->nq) {
                cur++;
                cur_q = 0;
            }
            SearchReq *r = batch[cur];
            for (int t = 0; t < w; t++) {
                r->iid[(size_t)cur_q * k + t] = hi[(size_t)q * w + t];
                r->dist[(size_t)cur_q * k + t] = hd[(size_t)q * w + t];
            }
            r->cnt[cur_q] = w;
            cur_q++;
        }
    }
    return MMIDX_OK;
}

int mmidx_linear_search(mmidx_linear *l, int k, int64_t nq, const double *Q, int32_t *iid_out, double *dist_out, int32_t *count_out) {
    if (!l) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (k < 1) return fail(MMIDX_ERR_INVALID_ARG, "k must be positive (got %d)", k);
    if (nq < 0 || (nq > 0 && (!Q || !iid_out || !dist_out || !count_out))) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    if (nq == 0) return MMIDX_OK;
    SearchReq me;
    me.k = k;
    me.nq = nq;
    me.Q = Q;
    me.iid = iid_out;
    me.dist = dist_out;
    me.cnt = count_out;
    return combiner_submit(l->comb, me, MMIDX_COMB_MAX_Q, [l](SearchReq *const *batch, size_t nb) {
        std::lock_guard<std::mutex> lk(l->mu);
        return linear_search_batch(l, batch, nb);
    });
}

}  // extern "C"

#include "mmidx_sharded.h"
