// mmidx_scan_mfma.h -- K3m: pass B of the IVFADC search (and the far chunks of flat PQ) as a list-major CERTIFIED LOWER BOUND
// on the matrix cores (gfx950), exact fp64 only for the codes the bound cannot drop.
//
// Reference loop: the per-probe body of computeKnnIVFADC, J/datastructures/IVFPQ.java:414-447 (residual :417 -> :642-648,
// lookup table :427 -> :525-538, scan :429-446) and PQ.computeKnnADC, J/datastructures/PQ.java:290-322.  The reference sums
// m table entries per code; the table entry of sub-quantizer s is ||r_s - p_s,code_s||^2, so the distance of a code is
//         d = ||r - x||^2 = ||r||^2 + ||x||^2 - 2 r.x ,        x = the code's m chosen sub-centroids side by side,
// a dot product per (query, code) -- a GEMM [queries of a list x D] x [D x codes of the list].  K3g (mmidx_scan_grp.h) evaluates
// a quantised form of the reference's table with one LDS lookup and ~1.4 vector instructions per (query, sub-quantizer); here
//   * a block takes ONE inverted list (or a piece of it) and up to 64 queries that probe it (the pairs of pass B are sorted by
//     cell): their residuals r = c - q (exact fp64, IVFPQ.java:645, transformed) are rounded to fp16 and held in REGISTERS as the
//     A operands of v_mfma_f32_16x16x32_f16 for the whole item;
//   * a code is DECODED ONCE per item: the fp16 codebook lives in LDS as [8-dimension group][entry] rows of 16 bytes -- with
//     8-dimensional sub-quantizers a row IS a lane's B fragment (k = 8 (lane >> 4) .. + 7 of column lane & 15), so a tile of 16
//     codes costs ONE ds_read_b128 per lane and 32 dimensions, and that fragment is multiplied against all (up to 64) queries;
//   * the accumulators start at -||x||^2 / 2 (a 4-byte per-code array, built with the index): acc = r.x - ||x||^2 / 2, and
//         d <= T   <=>   acc >= (||r||^2 - T) / 2
//     is ONE v_cmp per accumulator register against a per-query constant that carries the certified error term: a code
//     with d <= T is never dropped.  fp16 inputs have 11 significant bits (bf16: 8), so the bound is within
//     ~2^-9 |r| |x| of the true distance -- about ten times tighter than K3g's 8-bit table rows;
//   * survivors (lower bound <= T) are appended to a global list with their UPPER bound counted in a per-query histogram:
//     k + 1 upper bounds at or below a bucket edge make that edge a valid threshold for every later item of the query
//     (thresholds tighten from the union of what all probed lists have shown so far, during the scan);
//   * k_mfma_verify computes the survivors' exact distances -- entries in the reference's order (t ascending from 0.0,
//     IVFPQ.java:531-534), added in sub-quantizer order (:435-438): the same bits as the fp64 table lookup -- and offers those
//     at or below the query's current threshold to its pool.
// Whatever the bound cannot serve (no finite threshold yet, magnitudes beyond fp16 scaling, a full survivor list or pool) marks
// the QUERY for redo: k_mfma_redo resets its pool to pass A's and hands its pairs to K3f, so results never depend on the heuristics.
#pragma once
#include "mmidx_kernels.h"

#include <type_traits>

typedef _Float16 mf_h8 __attribute__((ext_vector_type(8)));
typedef float mf_f4 __attribute__((ext_vector_type(4)));

#define MF_NT 256       // threads per block: four waves, each scanning its own code tiles against the item's queries
#define MF_QG 64        // queries per item (four 16-row tiles)
#define MF_ASTRIDE 272  // bytes per staged residual row: 256 + 16 (conflict-free 16-byte fragment reads)

struct MfmaParams {
    ScanParams S;              // Q, coarse (rows in TRANSFORMED order: the host passes permuted copies), cells, list_off, codes, order, T, pool_*, D, m, ks, w, ivf, poolq
    const unsigned short *pq16;  // [D / 8][256][8] fp16 codebook, scaled by 2^ep (0 beyond ks)
    const float *xn;           // [n] ||x||^2 of every stored code (list-major, as codes)
    const double *pq;          // [m][ks][dsub] (file order): the verification's entries
    const double *flat_lut;    // flat PQ: the queries' exact tables [nq][m][256] (k_flat_lut), or null
    const double *R;           // null, or the pairs' exact transformed residuals [pair slot in order[]][D] (rotation)
    const int4 *gdesc;         // per group: {cell, first index into order[], number of pairs (1..64), 0}
    const int32_t *n_groups;   // device-side count
    int sub, nsub;             // codes per item, items per group = ceil(longest list / sub)
    int ep;                    // exponent of the codebook's power-of-two scale
    double xmax;               // >= ||x|| of every code: sqrt(sum_s max_j ||p_sj||^2) (1 + 1e-12)
    u32 *ghist;                // [nq][256] upper bounds of the survivors by bucket floor(ub * 256 / T0)
    const u64 *T0;             // [nq] thresholds as the launch found them (the bucket map must not move)
    uint4 *surv;               // survivor records {slot of the pair in order[], position in the list, lower bound (float bits), 0}
    u32 *surv_cnt;             // appended so far (may exceed surv_cap: the excess was dropped and its queries marked)
    u32 surv_cap;
    unsigned char *redo;       // [nq] 1: the query's pass B is redone by K3f
    const u32 *pool_snap;      // [nq] pool counters as pass A left them
    u32 *work;                 // [8] per-XCD item cursors (zeroed before the launch)
    u32 *fb_count;             // the redo hand-back: (pair, chunk) items for K3f
    int32_t *fb_items, *fb_ch;
    int fb_chunk, fb_nchunks;  // K3f's chunking of a list
    long long npairs_flat;     // flat PQ: number of pairs in order[] (IVF: *S.n_order)
    unsigned long long *stat;  // null, or profiling counters: [0] += verified codes, [9] += survivors, [10] += queries handed back
    unsigned long long *nver;  // always: [0] += verified codes (host hint)
    // K3ma (pass A on the matrix cores, MODE 1 of k_scan_mfma + k_a1_select + k_a1_verify; see the block comment above a1_scan_tiles)
    float *a_cand;             // [pair slot][nsub][256] the (row, lane, wave) slots' largest accumulator values of the sweep (-inf: none)
    double2 *a_rowc;           // [pair slot] {||r||^2 + err, -2 / s^2}: upper bound of a code's distance = x + y acc (NaN: the bound cannot serve the row)
    unsigned char *a_bm;       // [item][a_bm_stride] sweep 2's compare masks: per tile pair and lane a u16 (<= 32 rows: 8 bits per tile) or u32
    size_t a_bm_stride;        // bytes per item: ceil(sub / 32) * 256
    u32 *a_work;               // [16] sweep 1's per-XCD item cursors (its own set: both sweeps are enqueued behind one k_mfma_prep; the
                               // eight-wave instances use words 8 .. 15 of a_work / work)
    int a_wide;                // 1: items of <= 32 rows go to the eight-wave instances
    int a_cstride;             // floats per (pair slot, piece) of a_cand: 256, or 512 with the eight-wave instances
    u32 *a_icnt;               // [items + 1] bits sweep 2 set per item (zeroed by sweep 1), then their exclusive prefix (k_a1_item_scan): [items] = total
    uint2 *a_rec;              // the flat record list {pair slot, position in the index} in item order (k_a1_records), a_rec_cap entries
    u32 a_rec_cap;
    double *a_rows;            // [pair slot][D] the pairs' exact residuals c - q (k_a1_rows; rotation: = R)
    int4 *a_meta;              // [pair slot] {query, probe rank, list start (low, high word)}
    u64 *a_metaT;              // [pair slot] the query's threshold as k_a1_select left it
    uint2 *a_rnd;              // [round of a_rnd_size records] {smallest, largest pair slot} (initialised by k_a1_item_scan, filled by k_a1_records)
    int a_rnd_size;            // records per round of k_a1_verify (a power of two)
};

#define MF_CHUNK 512    // survivor records a wave reserves at a time (one global atomic per chunk, not per tile)
#define MF_BUF 128      // survivor records a wave stages in LDS (its quarter of the residual staging area, idle during the scan);
                        // flushed in one burst of stores + histogram atomics when half full
struct MfmaRow {        // what the survivor path needs of a query row (32 bytes: two 16-byte LDS reads)
    float thr;          // a code survives iff acc >= thr
    float cd;           // lower bound of a survivor's distance = cd + kd acc  (cd = ||r||^2 - err, kd = -2 / s^2: the item's)
    float kq, cq;       // histogram bucket of its UPPER bound = floor(cq + kq acc)  (kq = 0: no histogram for this row)
    int q, slot;        // query, slot of the pair in order[]
    float inv0;         // 256 / T0 as a float (bucket edges are converted back in fp64)
    int pad;
};
struct MfmaLds {
    size_t cb, stage, row, misc, total;
    __host__ __device__ MfmaLds(int D) {
        size_t o = 0;
        cb = o; o += (size_t)D * 512;             // [D / 8][256] rows of 8 halfs
        stage = o; o += 32 * (size_t)MF_ASTRIDE;  // 32 residual rows at a time
        row = o; o += MF_QG * sizeof(MfmaRow);
        misc = o; o += 64;  // [0] item, [1..2] touched mask, [4..7] wave maxima, [8..11] the waves' staged-record counters
        total = (o + 15) & ~(size_t)15;
    }
};

// ---- index-side tables ----------------------------------------------------------------------------------------------------
// fp16 codebook: row (g8, e) = dimensions 8 g8 .. 8 g8 + 7 of the concatenated entry e, scaled by 2^ep (one thread per row)
__global__ void k_pq16_table(const double *__restrict__ pqT, unsigned short *__restrict__ pq16, int m, int ks, int dsub, double scale) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int KG = m * dsub / 8;
    if (idx >= KG * 256) return;
    const int g8 = idx >> 8, e = idx & 255;
    mf_h8 v;
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int d = g8 * 8 + u, s = d / dsub, t = d - s * dsub;
        const double p = e < ks ? pqT[((size_t)s * dsub + t) * ks + e] : 0.0;
        v[u] = (_Float16)(float)(p * scale);
    }
    *(mf_h8 *)(pq16 + (size_t)idx * 8) = v;
}
// pn64[s][j] = ||p_sj||^2 (fp64), block s, thread j
__global__ void k_pn64_table(const double *__restrict__ pqT, double *__restrict__ pn64, int m, int ks, int dsub) {
    const int s = blockIdx.x, j = threadIdx.x;
    double a = 0.0;
    if (j < ks)
        for (int t = 0; t < dsub; t++) {
            const double p = pqT[((size_t)s * dsub + t) * ks + j];
            a += p * p;
        }
    pn64[(size_t)s * 256 + j] = a;
}
// xn[i] = ||x_i||^2 = sum_s ||p_s,code_is||^2 (fp64 sum, one rounding to fp32): the start value of a code's accumulators
__global__ void k_code_norms(const unsigned char *__restrict__ codes, const double *__restrict__ pn64, float *__restrict__ xn, int m,
                             long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char *c = codes + (size_t)i * m;
    double a = 0.0;
    if ((m & 3) == 0) {
        for (int s = 0; s < m; s += 4) {
            const u32 w = *(const u32 *)(c + s);
            a += pn64[(size_t)s * 256 + (w & 0xFFu)];
            a += pn64[(size_t)(s + 1) * 256 + ((w >> 8) & 0xFFu)];
            a += pn64[(size_t)(s + 2) * 256 + ((w >> 16) & 0xFFu)];
            a += pn64[(size_t)(s + 3) * 256 + (w >> 24)];
        }
    } else {
        for (int s = 0; s < m; s++) a += pn64[(size_t)s * 256 + c[s]];
    }
    xn[i] = (float)a;
}

// RandomRotation (RandomRotation.java:44-49: out = v (1 x D) . R (D x D), sequential over the row index): the exact transformed
// residuals of the pairs pass B kept, R[slot][j] = sum_i (c - q)[i] rot[i][j], i ascending from 0.0 -- the order of the oracle and
// of query_vector() in mmidx_kernels.h.  Eight pairs per block share every load of the matrix (128 KiB at D = 128: L2-resident);
// thread j owns output column j.  K3m reads its fp16 operands AND the survivors' exact entries from these rows.
__global__ __launch_bounds__(128) void k_pair_rotate(const double *__restrict__ Q, const double *__restrict__ coarse, const double *__restrict__ rot,
                                                     const int32_t *__restrict__ cells, const int32_t *__restrict__ order, const int32_t *__restrict__ n_order,
                                                     long long n_flat, int w, int D, int ivf, double *__restrict__ R) {
    extern __shared__ double s_r[];  // [8][D]
    const long long n = n_order ? (long long)*n_order : n_flat;
    const long long s0 = (long long)blockIdx.x * 8;
    if (s0 >= n) return;
    const int np = (int)(n - s0 < 8 ? n - s0 : 8);
    for (int idx = threadIdx.x; idx < 8 * D; idx += 128) {
        const int pi = idx / D, i = idx - pi * D;
        const long long sl = s0 + (pi < np ? pi : np - 1);
        const int e = order[sl];
        const int q = e / w;
        const int cell = ivf ? cells[e] : 0;
        const double cv = coarse[(size_t)(cell < 0 ? 0 : cell) * D + i], qv = Q[(size_t)q * D + i];
        s_r[idx] = ivf ? cv - qv : qv - cv;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < D; j += 128) {
        double acc[8];
#pragma unroll
        for (int p = 0; p < 8; p++) acc[p] = 0.0;
        for (int i = 0; i < D; i++) {
            const double m = rot[(size_t)i * D + j];
#pragma unroll
            for (int p = 0; p < 8; p++) acc[p] += s_r[p * D + i] * m;
        }
#pragma unroll
        for (int p = 0; p < 8; p++)
            if (p < np) R[(size_t)(s0 + p) * D + j] = acc[p];
    }
}

// one launch instead of three memsets and two copies in front of the scan: the upper-bound histograms, the redo flags and the
// control words zeroed, the thresholds and pool counters as pass A left them saved -- and nothing at all when the coarse bound left
// pass B no pair (the separable benchmark: the 16 MB histogram memset alone was 7 us of a 1.2 ms step)
__global__ void k_mfma_prep(const int32_t *__restrict__ n_groups, u32 *__restrict__ ghist, unsigned char *__restrict__ redo, u32 *__restrict__ ctl,
                            const u64 *__restrict__ T, u64 *__restrict__ T0, const u32 *__restrict__ pool_cnt, u32 *__restrict__ snap, long long nq) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 < 64) ctl[i0] = 0;  // [0] survivor count, [8..15] / [24..31] / [32..39] item cursors (K3mc: one set per stage), [40] [44..45] K3mk's scale words
    if (*n_groups == 0) return;
    const long long stride = (long long)gridDim.x * blockDim.x;
    uint4 *h4 = (uint4 *)ghist;  // (null: K3ma -- no threshold exists when its launch starts, so no bucket map either)
    if (h4)
        for (long long i = i0; i < nq * 64; i += stride) h4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = i0; i < nq; i += stride) {
        redo[i] = 0;
        T0[i] = T[i];
        snap[i] = pool_cnt[i];
    }
}

// largest float <= x
__device__ __forceinline__ float mf_float_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = __int_as_float(__float_as_int(f) + (f > 0.f ? -1 : (f < 0.f ? 1 : (int)0x80000001)));
    return f;
}

// ---- K3m ----------------------------------------------------------------------------------------------------------------
// NJ = D / 32 (MFMA k chunks per code), DSUB = dimensions per sub-quantizer (8 or 16).  Lane (n = lane & 15, g = lane >> 4) of a
// code tile holds column n (code n of the tile) and the 8-dimension groups g8 = NJ g + j, j = 0 .. NJ-1: the code bytes it
// needs are contiguous (NB of them at byte offset NB g).
// wave-private survivor chunk: [base, base + cap) of P.surv, `used` records written (wave-uniform values)
struct MfmaChunk {
    u32 base, cap, used;
};
// make room for `tot` records: the rest of the current chunk is padded with invalid records, a new chunk reserved
__device__ __forceinline__ void mf_chunk_room(const MfmaParams &P, MfmaChunk &ck, const u32 tot, const int lane) {
    if (ck.used + tot <= ck.cap) return;
    for (u32 i = ck.used + (u32)lane; i < ck.cap; i += 64)
        if (ck.base + i < P.surv_cap) P.surv[ck.base + i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
    const u32 want = tot > MF_CHUNK ? tot : MF_CHUNK;
    u32 b = 0;
    if (lane == 0) b = atomicAdd(P.surv_cnt, want);
    ck.base = (u32)__builtin_amdgcn_readfirstlane((int)b);
    ck.cap = want;
    ck.used = 0;
}
__device__ __forceinline__ u32 mf_mbcnt(const u64 m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// the wave's staged survivor records -> the global list (its chunk) + the queries' histograms of upper bounds
__device__ __forceinline__ void mf_flush(const MfmaParams &P, MfmaChunk &ck, const uint4 *s_buf, const u32 nb, const MfmaRow *s_row, u32 *s_touch,
                                         const int first, const int lane) {
    mf_chunk_room(P, ck, nb, lane);
    for (u32 i = (u32)lane; i < nb; i += 64) {
        const uint4 r = s_buf[i];
        const int qs = (int)r.x - first;
        const int q = s_row[qs].q;
        const u32 off = ck.base + ck.used + i;
        if (off < P.surv_cap) P.surv[off] = make_uint4(r.x, r.y, r.z, 0u);
        else P.redo[q] = 1;
        if (r.w != 0xFFFFFFFFu) {
            atomicAdd(P.ghist + (size_t)q * 256 + r.w, 1u);
            atomicOr(s_touch + (qs >> 5), 1u << (qs & 31));
        }
    }
    ck.used += nb;
}

template <int NJ, int DSUB, int NTL>
__device__ __forceinline__ void mf_scan_tiles(const MfmaParams &P, const mf_h8 (&A)[4][NJ], const float (&thr)[4][4], const unsigned char *codes,
                                              const float *xn, const long long c0, const long long c1, const float kinit, const float kd,
                                              const u32 lds_cb, const MfmaRow *s_row, u32 *s_touch, uint4 *s_buf, u32 *s_wcnt, const int first, MfmaChunk &ck, const int lane,
                                              const int wv) {
    constexpr int D = NJ * 32, M = D / DSUB;
    constexpr int NB = (NJ * 8 >= DSUB) ? NJ * 8 / DSUB : 1;  // code bytes per lane
    const int n = lane & 15, g = lane >> 4;
    const u32 boff = (u32)((NJ * g * 8) / DSUB);              // first code byte of the lane
    const u32 lane_base = lds_cb + (u32)(NJ * g) * 4096u;
    const int ntiles = (int)((c1 - c0 + 15) >> 4);
    u32 bufn = 0;  // records staged in the wave's LDS buffer (wave-uniform)
    float thrmin = thr[0][0];
#pragma unroll
    for (int b = 1; b < NTL * 4; b++) thrmin = __builtin_fminf(thrmin, thr[b >> 2][b & 3]);
    // (32-bit lane offsets from the item's wave-uniform base pointers: an item spans at most 2^31 bytes of codes)
    const unsigned char *cbase = codes + (size_t)c0 * M + boff;
    const float *xbase = xn + c0;
    const u32 last = (u32)(c1 - c0 - 1);
    typedef typename std::conditional<NB == 8, u64, u32>::type CW;  // (4-dimensional sub-quantizers at D = 128: eight code bytes per lane)
    auto load_tile = [&](int t, CW &cw, float &xv) {
        u32 p = (u32)t * 16u + (u32)n;
        p = p < last ? p : last;
        const unsigned char *cp = cbase + p * (u32)M;
        if constexpr (NB == 8) cw = *(const u64 *)cp;
        else if constexpr (NB == 4) cw = *(const u32 *)cp;
        else if constexpr (NB == 2) cw = (u32) * (const unsigned short *)cp;
        else cw = (u32)*cp;
        xv = xbase[p];
    };
    CW cw[4];
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) load_tile(wv + 4 * u, cw[u], xv[u]);
    for (int t = wv; t < ntiles; t += 16) {
        // TWO tiles per step: both tiles' gathers are issued together, the second tile's matrix work runs under the first one's
        // compares -- one tile at a time the wave serialised gather latency, the MFMA chain, its drain and the compares
        // (truncated builds: 1.8 of pass B's 2.25 ms were neither MFMA issue nor bank conflicts)
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            const int tt = t + 4 * u;
            const CW c[2] = {cw[u], cw[u + 1]};
            const float x[2] = {xv[u], xv[u + 1]};
            load_tile(tt + 16, cw[u], xv[u]);  // (clamped to the item's last code: always a valid address)
            load_tile(tt + 20, cw[u + 1], xv[u + 1]);
            if (tt >= ntiles) continue;        // (wave-uniform; a second tile past the end computes on clamped codes and is masked below)
            mf_h8 B[2][NJ];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if constexpr (DSUB == 4) {
                        // two sub-quantizers per 8-dimension group: entry (2 g8, b0) is the low half of row b0, entry (2 g8 + 1, b1) the
                        // high half of row b1 -- two 8-byte gathers make the fragment
                        const u32 b0 = (u32)(c[h] >> (16 * j)) & 0xFFu, b1 = (u32)(c[h] >> (16 * j + 8)) & 0xFFu;
                        typedef u64 __attribute__((address_space(3))) lds_u64;
                        const u64 lo = *(const lds_u64 *)(size_t)(lane_base + (b0 << 4) + (u32)j * 4096u);
                        const u64 hi = *(const lds_u64 *)(size_t)(lane_base + (b1 << 4) + 8u + (u32)j * 4096u);
                        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
                        const u64x2 both = {lo, hi};
                        B[h][j] = __builtin_bit_cast(mf_h8, both);
                    } else {
                        const u32 byte = (u32)(c[h] >> (8 * (j / (DSUB >= 8 ? DSUB / 8 : 1)))) & 0xFFu;
                        const u32 addr = lane_base + (byte << 4);
                        B[h][j] = *(const __attribute__((address_space(3))) mf_h8 *)(size_t)(addr + (u32)j * 4096u);
                    }
                }
            // (all gathers of the step are in flight before the first MFMA: left to itself the scheduler issues each one right in
            //  front of its use and the wave waits out the LDS latency 2 NJ times per step)
            __builtin_amdgcn_sched_barrier(0);
            mf_f4 acc[2][NTL];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float ci = x[h] * kinit;
                const mf_f4 c4 = {ci, ci, ci, ci};
#pragma unroll
                for (int rt = 0; rt < NTL; rt++) acc[h][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[rt][0], B[h][0], c4, 0, 0, 0);
#pragma unroll
                for (int j = 1; j < NJ; j++)
#pragma unroll
                    for (int rt = 0; rt < NTL; rt++) acc[h][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[rt][j], B[h][j], acc[h][rt], 0, 0, 0);
            }
            // Pre-test: the largest of the lane's accumulators against the SMALLEST of its rows' constants -- v_max3 over the 4 NTL
            // registers and one compare (9 vector instructions per tile instead of 16 compares + 16 scalar ORs).  Conservative: a
            // tile that passes it is looked at register by register below (the exact compares), and survivors are rare.
            u64 any[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float mxa = acc[h][0][0];
#pragma unroll
                for (int b = 1; b + 1 < NTL * 4; b += 2) mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, acc[h][b >> 2][b & 3]), acc[h][(b + 1) >> 2][(b + 1) & 3]);
                mxa = __builtin_fmaxf(mxa, acc[h][NTL - 1][3]);
                any[h] = __builtin_amdgcn_ballot_w64(mxa >= thrmin);
            }
            if (any[0] | any[1]) {
                // ---- survivors (about one per tile where far probes feed the queue).  Kept SMALL: an unrolled branch per accumulator
                // register made this path ~400 instructions per tile (1250 cycles per survivor measured).  Each lane packs its
                // compares into a bit mask (two instructions per register), and only the lanes that hold a survivor walk their
                // bits: the accumulator by a select tree, the record into the wave's LDS buffer (slot from an LDS counter). ----
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (!any[h]) continue;  // (scalar branch)
                    const long long pos = c0 + (long long)(tt + 4 * h) * 16 + n;
                    u32 bits = 0;  // bit rt * 4 + i: the lane's code survives row rt * 16 + 4 g + i
#pragma unroll
                    for (int b = NTL * 4 - 1; b >= 0; b--)
                        asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(acc[h][b >> 2][b & 3]), "v"(thr[b >> 2][b & 3]) : "vcc");
                    if (pos >= c1) bits = 0;  // (only an item's last tile has lanes past the end)
                    // (a wave-uniform loop -- one pass per survivor of the busiest lane, usually one -- so that the buffer slots come from
                    //  a ballot and a counter in a scalar register: no LDS atomic, no round trip before the record is written)
                    u64 act = __builtin_amdgcn_ballot_w64(bits != 0);
                    while (act) {
                        if (bits) {
                            const int b = __ffs((int)bits) - 1;
                            bits &= bits - 1u;
                            // (bit selects -- v_bfi_b32 -- on purpose: written as ?: the compiler turns the tree into an indexed scratch array)
                            const u32 m0 = 0u - ((u32)b & 1u), m1 = 0u - (((u32)b >> 1) & 1u), m2 = 0u - (((u32)b >> 2) & 1u), m3 = 0u - (((u32)b >> 3) & 1u);
                            u32 v8[8], v4[4], v2[2];
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const u32 lo = 2 * j < NTL * 4 ? (u32)__float_as_int(acc[h][(2 * j) >> 2][(2 * j) & 3]) : 0u,
                                          hi = 2 * j + 1 < NTL * 4 ? (u32)__float_as_int(acc[h][(2 * j + 1) >> 2][(2 * j + 1) & 3]) : 0u;
                                v8[j] = (hi & m0) | (lo & ~m0);
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) v4[j] = (v8[2 * j + 1] & m1) | (v8[2 * j] & ~m1);
#pragma unroll
                            for (int j = 0; j < 2; j++) v2[j] = (v4[2 * j + 1] & m2) | (v4[2 * j] & ~m2);
                            const float a = __int_as_float((int)((v2[1] & m3) | (v2[0] & ~m3)));
                            const int qs = (b >> 2) * 16 + 4 * g + (b & 3);  // (a live row: rows past the group's pairs never survive)
                            const MfmaRow rw = s_row[qs];
                            // d~ = ||r||^2 - 2 acc / s^2 and |d - d~| <= err: lower bound cd + kd acc (kept with the record: the
                            // verification drops what the query's FINAL threshold has left behind), upper bound -> bucket
                            const float lbf = fmaf(a, kd, rw.cd);
                            const float xb = fmaf(a, rw.kq, rw.cq);
                            const u32 bk = (rw.kq != 0.f && xb < 255.f) ? (xb > 0.f ? (u32)(int)xb : 0u) : 0xFFFFFFFFu;
                            const u32 o = bufn + mf_mbcnt(act);
                            if (o < MF_BUF) s_buf[o] = make_uint4((u32)(first + qs), (u32)pos, (u32)__float_as_int(lbf), bk);
                            else P.redo[rw.q] = 1;  // (more than half a buffer from one step: a threshold far above what the lists hold -- K3f's case)
                        }
                        bufn += (u32)__popcll(act);
                        act = __builtin_amdgcn_ballot_w64(bits != 0);
                    }
                }
                // the wave's staged records: out in one burst when the buffer is half full
                if (bufn >= MF_BUF / 2) {
                    mf_flush(P, ck, s_buf, bufn < MF_BUF ? bufn : MF_BUF, s_row, s_touch, first, lane);
                    bufn = 0;
                }
            }
        }
    }
    if (bufn) mf_flush(P, ck, s_buf, bufn < MF_BUF ? bufn : MF_BUF, s_row, s_touch, first, lane);  // what is left at the end of the item
}

// ---- K3ma: PASS A through the same bound (round 5) -------------------------------------------------------------------------
// Pass A (the scan of every query's NEAREST list, IVFPQ.java:429-446 for probe 0) is what produces the thresholds everything above
// compares against, so it cannot start from one.  With >= 8 queries per nearest list (a shard of the 8-GPU configuration, one GPU at
// batches >= 65536) the list-major matrix-core form pays anyway, in two sweeps over the list and no bootstrap:
//   MODE 1 (sweep 1): the accumulators acc = (r.x - ||x||^2 / 2) s^2 of every (query row, code); each (row, lane, wave) slot -- 64
//     disjoint subsets of the list's codes per row -- keeps its FOUR largest in registers (one v_max + three v_med3 per accumulator,
//     no compare against anything).  The item leaves rows x 256 values; k_a1_select takes the K1-th largest a* of a pair's values:
//     K1 distinct codes have acc >= a*, hence d <= ||r||^2 + err - 2 a* / s^2 =: T, a valid threshold (rank ~ K1 + 3 of the list at
//     K1 = 101: a slot holds more than four of the best 105 with probability 0.08);
//   MODE 2 (sweep 2): the same accumulators against (||r||^2 - T - err) s^2 / 2; each lane packs its compares into a bit mask and
//     the masks go to a dense bitmap of the item (one store per lane and two tiles): NO survivor path in the loop;
//   k_a1_verify (mmidx_scan_mfma_a.h): a block per item turns the bitmap into (row, position) records and computes their exact
//     distances sub-quantizer by sub-quantizer with the codebook slice in LDS -- the reference's operations in its order.
// Two tiles of a step are neighbours (tiles 2 p, 2 p + 1 of the wave's pair p): their masks share a store.
template <int NJ, int DSUB, int NTL, int MODE, int NWV>
__device__ __forceinline__ u32 a_scan_tiles(const mf_h8 (&A)[4][NJ], const float (&thr)[4][4], const unsigned char *codes, const float *xn, const long long c0,
                                             const long long c1, const float kinit, const u32 lds_cb, float (&top)[NTL * 4][4], unsigned char *bm, const int lane,
                                             const int wv) {
    constexpr int D = NJ * 32, M = D / DSUB;
    constexpr int NB = (NJ * 8 >= DSUB) ? NJ * 8 / DSUB : 1;  // code bytes per lane
    const int n = lane & 15, g = lane >> 4;
    const u32 boff = (u32)((NJ * g * 8) / DSUB);
    const u32 lane_base = lds_cb + (u32)(NJ * g) * 4096u;
    const int ntiles = (int)((c1 - c0 + 15) >> 4);
    const int npt = (ntiles + 1) >> 1;  // tile pairs
    const unsigned char *cbase = codes + (size_t)c0 * M + boff;
    const float *xbase = xn + c0;
    const u32 last = (u32)(c1 - c0 - 1);
    typedef typename std::conditional<NB == 8, u64, u32>::type CW;
    // (a position past the item's last code: the load runs on the last code, the start value is -inf -- such a column never enters a
    //  slot's best four and never passes a compare)
    auto load_tile = [&](int t, CW &cw, float &xv) {
        const u32 p0 = (u32)t * 16u + (u32)n;
        const u32 p = p0 < last ? p0 : last;
        const unsigned char *cp = cbase + p * (u32)M;
        if constexpr (NB == 8) cw = *(const u64 *)cp;
        else if constexpr (NB == 4) cw = *(const u32 *)cp;
        else if constexpr (NB == 2) cw = (u32) * (const unsigned short *)cp;
        else cw = (u32)*cp;
        const float x = xbase[p];
        xv = p0 <= last ? x : __int_as_float(0x7F800000);  // (kinit < 0: the start value x kinit is -inf)
    };
    CW cw[4];
    float xv[4];
    u32 nbits = 0;  // MODE 2: set bits of this lane (the item's total sizes its share of the record list)
#pragma unroll
    for (int u = 0; u < 2; u++) {
        load_tile(2 * (wv + NWV * u), cw[2 * u], xv[2 * u]);
        load_tile(2 * (wv + NWV * u) + 1, cw[2 * u + 1], xv[2 * u + 1]);
    }
    for (int pp = wv; pp < npt; pp += 2 * NWV) {  // (the wave's tile pairs: wv, wv + NWV, ...; two of them per round of the loop)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int pr = pp + NWV * u;
            const CW c[2] = {cw[2 * u], cw[2 * u + 1]};
            const float x[2] = {xv[2 * u], xv[2 * u + 1]};
            load_tile(2 * (pr + 2 * NWV), cw[2 * u], xv[2 * u]);
            load_tile(2 * (pr + 2 * NWV) + 1, cw[2 * u + 1], xv[2 * u + 1]);
            if (pr >= npt) continue;  // (wave-uniform)
            // the B fragments of tile h: the decode gathers (as mf_scan_tiles)
            auto gather = [&](const CW ch, mf_h8 (&Bh)[NJ]) {
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    if constexpr (DSUB == 4) {
                        const u32 b0 = (u32)(ch >> (16 * j)) & 0xFFu, b1 = (u32)(ch >> (16 * j + 8)) & 0xFFu;
                        typedef u64 __attribute__((address_space(3))) lds_u64;
                        const u64 lo = *(const lds_u64 *)(size_t)(lane_base + (b0 << 4) + (u32)j * 4096u);
                        const u64 hi = *(const lds_u64 *)(size_t)(lane_base + (b1 << 4) + 8u + (u32)j * 4096u);
                        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
                        const u64x2 both = {lo, hi};
                        Bh[j] = __builtin_bit_cast(mf_h8, both);
                    } else {
                        const u32 byte = (u32)(ch >> (8 * (j / (DSUB >= 8 ? DSUB / 8 : 1)))) & 0xFFu;
                        const u32 addr = lane_base + (byte << 4);
                        Bh[j] = *(const __attribute__((address_space(3))) mf_h8 *)(size_t)(addr + (u32)j * 4096u);
                    }
                }
            };
            auto matmul = [&](const mf_h8 (&Bh)[NJ], const float xh, mf_f4 (&ah)[NTL]) {
                const float ci = xh * kinit;
                const mf_f4 c4 = {ci, ci, ci, ci};
#pragma unroll
                for (int rt = 0; rt < NTL; rt++) ah[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[rt][0], Bh[0], c4, 0, 0, 0);
#pragma unroll
                for (int j = 1; j < NJ; j++)
#pragma unroll
                    for (int rt = 0; rt < NTL; rt++) ah[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[rt][j], Bh[j], ah[rt], 0, 0, 0);
            };
            // the slot's four largest, kept sorted: with m1 >= m2 >= m3 >= m4 the new second is the median of {m1, m2, a} and so on
            // (all four from the OLD values: independent instructions)
            auto insert = [&](const mf_f4 (&ah)[NTL]) {
#pragma unroll
                for (int b = 0; b < NTL * 4; b++) {
                    const float a = ah[b >> 2][b & 3];
                    const float m1 = top[b][0], m2 = top[b][1], m3 = top[b][2], m4 = top[b][3];
                    top[b][0] = __builtin_fmaxf(m1, a);
                    top[b][1] = __builtin_amdgcn_fmed3f(m1, m2, a);
                    top[b][2] = __builtin_amdgcn_fmed3f(m2, m3, a);
                    top[b][3] = __builtin_amdgcn_fmed3f(m3, m4, a);
                }
            };
            if constexpr (MODE == 1 && NTL >= 3) {
                // (48 / 64 registers of kept values next to the 64 of the queries: one tile at a time, or the loop spills)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    mf_h8 B1[NJ];
                    mf_f4 a1[NTL];
                    gather(c[h], B1);
                    __builtin_amdgcn_sched_barrier(0);
                    matmul(B1, x[h], a1);
                    insert(a1);
                }
                continue;
            }
            mf_h8 B[2][NJ];
            gather(c[0], B[0]);
            gather(c[1], B[1]);
            __builtin_amdgcn_sched_barrier(0);
            mf_f4 acc[2][NTL];
            matmul(B[0], x[0], acc[0]);
            matmul(B[1], x[1], acc[1]);
            if constexpr (MODE == 1) {
                insert(acc[0]);
                insert(acc[1]);
            } else {
                u32 bits[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    // bit rt * 4 + i: the lane's code passes row rt * 16 + 4 g + i.  Plain C++ on purpose: mf_scan_tiles packs its masks with
                    // inline v_cmp + v_addc behind a v_max3 pre-test; HERE the compares are the first readers of the accumulators, and the
                    // compiler inserts the wait states between a matrix instruction and a read of its result only for instructions it
                    // knows -- inline assembly read registers 2 and 3 of the 8-pass v_mfma_f32_16x16x32_f16 before they were written
                    // (rows 4 g + 2 and 4 g + 3 lost a third of their survivors).
                    u32 bt = 0;
#pragma unroll
                    for (int b = 0; b < NTL * 4; b++) bt |= acc[h][b >> 2][b & 3] >= thr[b >> 2][b & 3] ? (1u << b) : 0u;
                    bits[h] = bt;
                }
                const u32 both = NTL <= 2 ? (bits[0] | (bits[1] << 8)) : (bits[0] | (bits[1] << 16));
                if constexpr (NTL <= 2) ((unsigned short *)bm)[(size_t)pr * 64 + lane] = (unsigned short)both;
                else ((u32 *)bm)[(size_t)pr * 64 + lane] = both;
                nbits += (u32)__popc(both);
            }
        }
    }
    return nbits;
}

// MODE 0: pass B (and flat PQ's far chunks) with survivor records; MODE 1 / 2: the two sweeps of K3ma (above).
// NWV = waves per block.  8 (K3ma only): the instance for items of at most 32 rows (NTL <= 2), which are most of pass A's -- their
// queries, kept values and fragments fit 128 registers, so sixteen waves run per CU instead of eight and the decode gathers (LDS), the
// matrix instructions and the slot updates / compares (VALU) of different waves overlap; it skips larger items, the four-wave
// instance (launched behind it when P.a_wide is set) skips the small ones.  Waves 4 .. 7 take no part in the residual phase (a).
template <int NJ, int DSUB, int MODE = 0, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, NWV == 8 ? 4 : 2) void k_scan_mfma(const MfmaParams P) {  // (second figure: waves per SIMD the registers must allow)
    constexpr int NT = NWV * 64;
    static_assert(NWV == 4 || (NWV == 8 && MODE != 0), "eight waves: K3ma's sweeps only");
    constexpr int D = NJ * 32;
    constexpr int DPT = D / 8;  // dimensions per thread in the residual phase: a thread is (row of 32, eighth of the dimensions)
    static_assert(NJ == 1 || NJ == 2 || NJ == 4, "D = 32, 64 or 128");
    static_assert(DSUB == 4 || DSUB == 8 || DSUB == 16, "a 16-byte codebook row is two, one or half a sub-quantizer entry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const MfmaLds L(D);
    unsigned char *s_stage = smem + L.stage;
    MfmaRow *s_row = (MfmaRow *)(smem + L.row);
    u32 *s_misc = (u32 *)(smem + L.misc);
    MfmaChunk ck{0u, 0u, 0u};
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 lds_cb = (u32)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + L.cb);

    const int nv = *P.n_groups * P.nsub;
    if (nv == 0) return;  // (the separable benchmark: the coarse bound left pass B nothing)
    // the fp16 codebook: once per block
    for (int i = tid; i < D * 32; i += NT) ((uint4 *)(smem + L.cb))[i] = ((const uint4 *)P.pq16)[i];
    const int per = (nv + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const double xmax = P.xmax;
    const double gam = (double)(D + 4) * 0x1p-23 * 1.01;  // fp32 accumulation of D products + the start value, any order, 2 u per step
    __syncthreads();

    for (;;) {
        if (tid == 0) {
            s_misc[0] = atomicAdd((MODE == 1 ? P.a_work : P.work) + (NWV == 8 ? 8 : 0) + xcd, 1u);
            s_misc[1] = 0;
            s_misc[2] = 0;
            s_misc[8] = s_misc[9] = s_misc[10] = s_misc[11] = 0;
        }
        __syncthreads();
        const int it = (int)s_misc[0];
        if (it >= per) break;
        const int v = xcd * per + it;  // consecutive items -- the groups of one list -- run on the same XCD
        if (v >= nv) break;
        const int gi = v / P.nsub, isub = v - gi * P.nsub;
        const int4 gd = P.gdesc[gi];
        const int cell = gd.x, first = gd.y, np = gd.z;
        const long long beg = P.S.list_off[cell];
        const long long len = P.S.list_off[cell + 1] - beg;
        const long long c0 = (long long)isub * P.sub;
        if (MODE != 0 && (NWV == 8 ? np > 32 : (P.a_wide && np <= 32))) {  // (block-uniform) the other instance's item
            __syncthreads();
            continue;
        }
        if (c0 >= len) {
            if (MODE == 1 && tid == 0) P.a_icnt[v] = 0;  // (an empty item of K3ma: no bits)
            __syncthreads();
            continue;
        }
        const long long c1 = (c0 + P.sub < len) ? c0 + P.sub : len;
        const unsigned char *codes = (const unsigned char *)P.S.codes + (size_t)beg * (D / DSUB);
        const float *xn = P.xn + beg;

        // ---- (a) residuals of the item's pairs: thread (row = tid >> 3 of a half, eighth = tid & 7) ----
        // (loads on a clamped pair index, stores predicated: see the note at pair_keep() in mmidx_kernels.h)
        const int ta = tid & 255;       // (eight waves: waves 4 .. 7 shadow waves 0 .. 3 here and store nothing)
        const bool pa_act = tid < 256;
        double rv[2][DPT];
        double nrp[2];
        int qq[2];
        float mx = 0.f;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int row = h * 32 + (ta >> 3);
            const int tl = row < np ? row : np - 1;
            const int e = P.S.order[first + tl];
            const int q = e / P.S.w;
            qq[h] = q;
            const int d0 = (ta & 7) * DPT;
            double nr = 0.0;
            if (P.R) {
                const double *rr = P.R + (size_t)(first + tl) * D + d0;
#pragma unroll
                for (int t = 0; t < DPT; t += 2) {
                    const double2 r2 = *(const double2 *)(rr + t);
                    rv[h][t] = r2.x;
                    rv[h][t + 1] = r2.y;
                }
            } else {
                const double *cc = P.S.coarse + (size_t)(P.S.ivf ? cell : 0) * D + d0, *qv = P.S.Q + (size_t)q * D + d0;
#pragma unroll
                for (int t = 0; t < DPT; t += 2) {
                    const double2 c2 = *(const double2 *)(cc + t), q2 = *(const double2 *)(qv + t);
                    // (flat PQ: the "centroid" is a zero vector and the sign turns round -- q - 0 = q exactly, PQ.java:294-300)
                    rv[h][t] = P.S.ivf ? c2.x - q2.x : q2.x - c2.x;
                    rv[h][t + 1] = P.S.ivf ? c2.y - q2.y : q2.y - c2.y;
                }
            }
#pragma unroll
            for (int t = 0; t < DPT; t++) {
                if (row >= np) rv[h][t] = 0.0;
                nr += rv[h][t] * rv[h][t];
                mx = fmaxf(mx, fabsf((float)rv[h][t]));
            }
            nr += __shfl_xor(nr, 1);
            nr += __shfl_xor(nr, 2);
            nr += __shfl_xor(nr, 4);
            nrp[h] = nr;
        }
        // the item's power-of-two scale: the largest |r_i| lands in [2^12, 2^13)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0 && wv < 4) ((float *)s_misc)[4 + wv] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(((float *)s_misc)[4], ((float *)s_misc)[5]), fmaxf(((float *)s_misc)[6], ((float *)s_misc)[7]));
        int er = 0;
        bool scale_ok = mx < 1e30f;  // (false for NaN / inf as well)
        if (mx > 0.f && scale_ok) {
            int ex;
            (void)frexpf(mx, &ex);  // mx = f 2^ex, f in [0.5, 1)
            er = 13 - ex;
        }
        // (s^2 / 2 a normal float; ||r||^2 s^2 and ||x||^2 s^2 -- 2^26 D at most in their own scales -- far inside its range)
        scale_ok = scale_ok && er + P.ep > -100 && er + P.ep < 100 && er - P.ep > -60 && er - P.ep < 60;
        const float s_r = scale_ok ? ldexpf(1.f, er) : 1.f;
        const double s2 = scale_ok ? ldexp(1.0, er + P.ep) : 1.0;
        const double inv_s2 = 1.0 / s2, inv_sr = scale_ok ? ldexp(1.0, -er) : 1.0, inv_sp = ldexp(1.0, -P.ep);
        const double sqrtD = 11.32;  // >= sqrt(128)
        const float kinit = (float)(-0.5 * s2);
        const float kd = (float)(-2.0 * inv_s2);  // (a power of two: exact)
        // per query: ||r||^2, the certified error term, the threshold constant, the histogram scale
        if (pa_act && (tid & 7) == 0) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int row = h * 32 + (tid >> 3);
                const int q = qq[h];
                const double nr = nrp[h], nrm = sqrt(nr) * (1.0 + 1e-12);
                if constexpr (MODE == 1) {  // (sweep 1 of K3ma: the pair's upper-bound map for k_a1_select -- the same error term as below)
                    const double err1 = nrm * xmax * (4.02 * 0x1p-11 + 2.004 * gam) + xmax * xmax * (1.001 * gam + 0x1p-24) +
                                        2.02 * sqrtD * 0x1p-14 * (xmax * inv_sr + nrm * inv_sp) + 2.0 * D * 0x1p-28 * inv_s2 +
                                        0x1p-19 * (nr + xmax * xmax + 2.0 * nrm * xmax) + 1e-300;
                    if (row < np && isub == 0) {
                        const bool ok = scale_ok && (err1 < 1e300) && (nr < 1e300);
                        P.a_rowc[first + row] = ok ? make_double2(nr + err1, -2.0 * inv_s2) : make_double2(__longlong_as_double(0x7FF8000000000000ll), 0.0);
                    }
                    continue;
                }
                // |acc / s^2 - (r.x - ||x||^2 / 2)| <= |r| xmax (2.01 2^-11 + 1.002 gam) + xmax^2 (0.5005 gam + 2^-25)
                //                                       + 1.01 sqrt(D) 2^-14 (xmax / s_r + |r| / s_p) + D 2^-28 / s^2 :
                //   fp16 inputs (an element's error is at most 2^-11 of itself -- after the fp32 step 1.001 of that -- or, where the
                //   scaled element is subnormal or flushed, 2^-14 / scale), exact products, fp32 accumulation (gam), the fp32 start
                //   value.  d = ||r||^2 - 2 (...); the reference's own fp64 roundings and the survivor path's fp32 arithmetic (lower and
                //   upper bounds are single fma's of quantities no larger than ||r||^2 + ||x||^2 + 2 |r||x|) are inside the 2^-19 term:
                const double err = nrm * xmax * (4.02 * 0x1p-11 + 2.004 * gam) + xmax * xmax * (1.001 * gam + 0x1p-24) +
                                   2.02 * sqrtD * 0x1p-14 * (xmax * inv_sr + nrm * inv_sp) + 2.0 * D * 0x1p-28 * inv_s2 +
                                   0x1p-19 * (nr + xmax * xmax + 2.0 * nrm * xmax) + 1e-300;
                const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float th = __int_as_float(0x7F800000);  // +inf: nothing survives (rows past np, queries handed to the redo)
                double inv0 = 0.0;
                if (row < np) {
                    const bool fin = T < 0x7FF0000000000000ull;
                    if (!fin || !scale_ok || !(err < 1e300)) {
                        P.redo[q] = 1;
                    } else if (!P.redo[q]) {
                        // d <= T  =>  acc >= (||r||^2 - T - err) s^2 / 2, rounded down
                        th = mf_float_down((nr - keyd(T) - err) * (0.5 * s2));
                        if (!(th < 3e38f)) {  // (cannot happen inside the scale limits; never drop on an overflow)
                            P.redo[q] = 1;
                            th = __int_as_float(0x7F800000);
                        }
                        const u64 t0 = P.T0[q];
                        if (t0 < 0x7FF0000000000000ull && t0 > 0) {
                            const double iv = 256.0 / keyd(t0);
                            if (iv < 1e300) inv0 = iv;
                        }
                    }
                }
                // the survivor path works in fp32: err carries 2^-20 of the magnitudes for its roundings (cd, kd acc, the fma's)
                MfmaRow rw;
                rw.thr = th;
                rw.cd = (float)(nr - err);
                rw.kq = 0.f;
                rw.cq = 0.f;
                rw.inv0 = 0.f;
                if (inv0 > 0.0 && inv0 < 1e30 && inv0 * inv_s2 < 1e30) {
                    // bucket of the upper bound (nr + err - 2 acc / s^2) * inv0, shifted up against the fp32 roundings of cq, kq and the
                    // fma (2^-21 of the magnitudes that cancel in it, + 1e-3 of a bucket)
                    const double mag = ((nr + err) + 2.02 * (nrm * xmax + 0.5 * xmax * xmax)) * inv0;
                    rw.kq = (float)(-2.0 * inv_s2 * inv0);
                    rw.cq = (float)((nr + err) * inv0 + 1e-3 + 0x1p-21 * mag);
                    rw.inv0 = (float)inv0;
                    if (!(rw.kq != 0.f) || !(rw.cq < 3e38f)) rw.kq = 0.f;
                }
                rw.q = q;
                rw.slot = first + row;
                rw.pad = 0;
                s_row[row] = rw;
            }
        }
        // fp16 A operands, 32 rows at a time through LDS
        const int ntl = (np + 15) >> 4;
        mf_h8 A[4][NJ];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h * 2 < ntl && pa_act) {  // (block-uniform but for the shadow waves)
                unsigned char *dst = s_stage + (size_t)(tid >> 3) * MF_ASTRIDE + (size_t)(tid & 7) * DPT * 2;
#pragma unroll
                for (int t = 0; t < DPT; t += 2) {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    h2 p2;
                    p2[0] = (_Float16)((float)rv[h][t] * s_r);
                    p2[1] = (_Float16)((float)rv[h][t + 1] * s_r);
                    *(h2 *)(dst + 2 * t) = p2;
                }
            }
            __syncthreads();
            if (h * 2 < ntl) {
#pragma unroll
                for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
                        A[h * 2 + r2][j] = *(const mf_h8 *)(s_stage + (size_t)(r2 * 16 + (lane & 15)) * MF_ASTRIDE + (size_t)(NJ * (lane >> 4) + j) * 16);
            } else {
#pragma unroll
                for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) A[h * 2 + r2][j] = mf_h8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            __syncthreads();
        }
        float thr[4][4];
#pragma unroll
        for (int rt = 0; rt < 4; rt++)
#pragma unroll
            for (int i = 0; i < 4; i++) thr[rt][i] = MODE == 1 ? 0.f : s_row[rt * 16 + 4 * (lane >> 4) + i].thr;

        // ---- (b) the scan: every wave its own code tiles against all row tiles ----
        if constexpr (MODE == 0) {
            uint4 *s_buf = (uint4 *)(s_stage + (size_t)wv * (MF_BUF * 16));  // (the staging area is idle until the next item)
            u32 *s_wcnt = s_misc + 8 + wv;
            switch (ntl) {
                case 1: mf_scan_tiles<NJ, DSUB, 1>(P, A, thr, codes, xn, c0, c1, kinit, kd, lds_cb, s_row, s_misc + 1, s_buf, s_wcnt, first, ck, lane, wv); break;
                case 2: mf_scan_tiles<NJ, DSUB, 2>(P, A, thr, codes, xn, c0, c1, kinit, kd, lds_cb, s_row, s_misc + 1, s_buf, s_wcnt, first, ck, lane, wv); break;
                case 3: mf_scan_tiles<NJ, DSUB, 3>(P, A, thr, codes, xn, c0, c1, kinit, kd, lds_cb, s_row, s_misc + 1, s_buf, s_wcnt, first, ck, lane, wv); break;
                default: mf_scan_tiles<NJ, DSUB, 4>(P, A, thr, codes, xn, c0, c1, kinit, kd, lds_cb, s_row, s_misc + 1, s_buf, s_wcnt, first, ck, lane, wv); break;
            }
        } else {
            // K3ma.  MODE 1: the slots' best four -> a_cand[pair slot][piece][wave * 64 + 4 n ..]; MODE 2: the item's bitmap (a_bm_stride bytes per item)
            unsigned char *bm = MODE == 2 ? P.a_bm + (size_t)v * P.a_bm_stride : nullptr;
            const float ninf = -__int_as_float(0x7F800000);
            auto run = [&](auto ntl_c) {
                constexpr int NTL = decltype(ntl_c)::value;
                float top[NTL * 4][4];
#pragma unroll
                for (int b = 0; b < NTL * 4; b++) top[b][0] = top[b][1] = top[b][2] = top[b][3] = ninf;
                u32 nb = a_scan_tiles<NJ, DSUB, NTL, MODE, NWV>(A, thr, codes, xn, c0, c1, kinit, lds_cb, top, bm, lane, wv);
                if constexpr (MODE == 2) {  // the item's bit count: one atomic per wave
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) nb += __shfl_xor(nb, off);
                    if (lane == 0 && nb) atomicAdd(P.a_icnt + v, nb);
                }
                if constexpr (MODE == 1) {
                    if (tid == 0) P.a_icnt[v] = 0;  // (sweep 2 counts its bits here)
#pragma unroll
                    for (int b = 0; b < NTL * 4; b++) {
                        const int row = (b >> 2) * 16 + 4 * (lane >> 4) + (b & 3);
                        if (row < np) {
                            float *cp = P.a_cand + ((size_t)(first + row) * P.nsub + isub) * (size_t)P.a_cstride + wv * 64 + (lane & 15) * 4;
                            *(float4 *)cp = make_float4(top[b][0], top[b][1], top[b][2], top[b][3]);
                            // (four waves where the eight-wave instance sets the stride: the other half of the pair's slots stays empty)
                            if (NWV == 4 && P.a_cstride > 256) *(float4 *)(cp + 256) = make_float4(ninf, ninf, ninf, ninf);
                        }
                    }
                }
            };
            if constexpr (NWV == 8) {  // (items of at most 32 rows)
                if (ntl == 1) run(std::integral_constant<int, 1>{});
                else run(std::integral_constant<int, 2>{});
            } else {
                switch (ntl) {
                    case 1: run(std::integral_constant<int, 1>{}); break;
                    case 2: run(std::integral_constant<int, 2>{}); break;
                    case 3: run(std::integral_constant<int, 3>{}); break;
                    default: run(std::integral_constant<int, 4>{}); break;
                }
            }
        }
        __syncthreads();
        // ---- (c) thresholds from the union of the survivors' upper bounds: K1 of them at or below a bucket's upper edge make
        //      that edge a valid threshold (K1 offers lie at or below it) for every later item of the query ----
        if constexpr (MODE == 0) {
            const u64 tall = ((u64)s_misc[2] << 32) | (u64)s_misc[1];
            for (int qs = wv; qs < MF_QG; qs += MF_NT / 64) {
                if (!((tall >> qs) & 1ull)) continue;  // (wave-uniform)
                const int q = s_row[qs].q;
                if (!(s_row[qs].kq != 0.f)) continue;
                const double inv0q = 256.0 / keyd(P.T0[q]);  // (as phase (a) computed it)
                const u32 *hq = P.ghist + (size_t)q * 256 + 4 * lane;
                const u32 h0 = __hip_atomic_load(hq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h1 = __hip_atomic_load(hq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                          h2 = __hip_atomic_load(hq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h3 = __hip_atomic_load(hq + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32 incl = wave_incl_scan_u32(h0 + h1 + h2 + h3);
                const u64 reached = __builtin_amdgcn_ballot_w64(incl >= (u32)P.S.K1);
                if (reached) {
                    const int Lr = __ffsll((long long)reached) - 1;
                    if (lane == Lr) {
                        u32 c = incl - (h0 + h1 + h2 + h3) + h0;
                        int b = 4 * Lr;
                        if (c < (u32)P.S.K1) { c += h1; b++; }
                        if (c < (u32)P.S.K1) { c += h2; b++; }
                        if (c < (u32)P.S.K1) { c += h3; b++; }
                        // every counted survivor has d <= ub < (b + 1) / inv0
                        atomicMin(P.S.T + q, dkey((double)(b + 1) / inv0q * (1.0 + 1e-12)));
                    }
                }
            }
        }
        __syncthreads();  // LDS is reused by the next item
    }
    // the rest of the wave's last chunk: invalid records (k_mfma_verify skips them)
    if constexpr (MODE == 0)
        for (u32 i = ck.used + (u32)lane; i < ck.cap; i += 64)
            if (ck.base + i < P.surv_cap) P.surv[ck.base + i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
}

// ---- exact distances of the survivors -------------------------------------------------------------------------------------
// Phase 1, a thread per record: the record's lower bound against the query's threshold as it stands NOW (the scan is over: the
// final one) -- thresholds tighten while the scan runs, and most of what survived an early item lies above the final threshold;
// those records are dropped without looking at their codes.  Phase 2, LPS = M lanes per remaining survivor (IVF, and flat PQ
// without tables): lane s computes the table entry of sub-quantizer s in the reference's order (t ascending from 0.0,
// IVFPQ.java:531-534) and the entries are added in sub-quantizer order (:435-438, ((0 + e_0) + e_1) + ...) down the lanes by DPP
// -- the bits of a lookup in the fp64 table.  FLAT: one lane per survivor reads the query's own table (k_flat_lut; PQ.java:308-311).
#define MF_VLIST 1024  // records a block filters per round
template <int M, int DSUB, bool FLAT>
__global__ __launch_bounds__(256) void k_mfma_verify(const MfmaParams P) {
    // lanes per survivor and sub-quantizers per lane: up to 16 lanes (one DPP row), each with SPL consecutive sub-quantizers
    constexpr int LPS = FLAT ? 1 : (M <= 16 ? M : 16), SPL = FLAT ? M : M / LPS, D = M * DSUB;
    __shared__ uint2 s_rec[MF_VLIST];
    __shared__ int s_pair[MF_VLIST];
    __shared__ u32 s_n, s_nv;
    const u32 cnt = *P.surv_cnt;
    const u32 ns = cnt < P.surv_cap ? cnt : P.surv_cap;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_nv = 0;
    for (u32 base = blockIdx.x * MF_VLIST; base < ns; base += gridDim.x * MF_VLIST) {
        if (tid == 0) s_n = 0;
        __syncthreads();
        // ---- phase 1 ----
#pragma unroll
        for (int r = 0; r < MF_VLIST / 256; r++) {
            const u32 si_raw = base + (u32)(r * 256 + tid);
            const u32 si = si_raw < ns ? si_raw : ns - 1u;  // (loads run on a clamped index)
            uint4 rec = P.surv[si];
            const bool valid = si_raw < ns && rec.x != 0xFFFFFFFFu;  // (padding of the waves' chunks)
            if (!valid) rec = make_uint4(0u, 0u, 0u, 0u);
            const int e = P.S.order[rec.x];
            const int q = e / P.S.w;
            const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool alive = valid && !((double)__int_as_float((int)rec.z) > keyd(T));
            const u64 mk = __builtin_amdgcn_ballot_w64(alive);
            u32 b = 0;
            if (mk && lane == 0) b = atomicAdd(&s_n, (u32)__popcll(mk));
            b = (u32)__builtin_amdgcn_readfirstlane((int)b);
            if (alive) {
                const u32 o = b + mf_mbcnt(mk);
                s_rec[o] = make_uint2(rec.x, rec.y);
                s_pair[o] = e;
            }
        }
        __syncthreads();
        const u32 na = s_n;
        if (tid == 0 && na) atomicAdd(&s_nv, na);
        // ---- phase 2 ----
        for (u32 r0 = 0; r0 < na; r0 += 256 / LPS) {
            const u32 li_raw = r0 + (u32)(tid / LPS);
            const bool act = li_raw < na;
            const u32 li = act ? li_raw : na - 1u;
            const uint2 rec = s_rec[li];
            const int slot = (int)rec.x;
            const u32 pos = rec.y;
            const int e = s_pair[li];
            const int q = e / P.S.w, rank = e - q * P.S.w;
            const int cell = P.S.ivf ? P.S.cells[e] : rank;
            const long long beg = P.S.list_off[cell];
            const unsigned char *code = (const unsigned char *)P.S.codes + (size_t)(beg + pos) * M;
            double d = 0.0;
            if constexpr (FLAT) {
                const double *lq = P.flat_lut + (size_t)q * (size_t)(M * 256);
                double en[M];
#pragma unroll
                for (int s = 0; s < M; s++) en[s] = lq[s * 256 + (int)code[s]];
#pragma unroll
                for (int s = 0; s < M; s++) d += en[s];
            } else {
                const int ls = tid & (LPS - 1);
                double en[SPL];  // the lane's table entries, each in the reference's order (t ascending from 0.0)
#pragma unroll
                for (int i = 0; i < SPL; i++) {
                    const int s = ls * SPL + i;
                    const u32 cs = (u32)code[s];
                    double tv[DSUB];
                    if (P.R) {
                        const double *rr = P.R + (size_t)slot * D + s * DSUB;
#pragma unroll
                        for (int t = 0; t < DSUB; t += 2) {
                            const double2 r2 = *(const double2 *)(rr + t);
                            tv[t] = r2.x;
                            tv[t + 1] = r2.y;
                        }
                    } else {
                        const double *cc = P.S.coarse + (size_t)(P.S.ivf ? cell : 0) * D + s * DSUB, *qv = P.S.Q + (size_t)q * D + s * DSUB;
#pragma unroll
                        for (int t = 0; t < DSUB; t += 2) {
                            const double2 c2 = *(const double2 *)(cc + t), q2 = *(const double2 *)(qv + t);
                            tv[t] = P.S.ivf ? c2.x - q2.x : q2.x - c2.x;
                            tv[t + 1] = P.S.ivf ? c2.y - q2.y : q2.y - c2.y;
                        }
                    }
                    const double *pp = P.pq + ((size_t)s * P.S.ks + cs) * DSUB;
                    double pv[DSUB];
#pragma unroll
                    for (int t = 0; t < DSUB; t += 2) {
                        const double2 v2 = *(const double2 *)(pp + t);
                        pv[t] = v2.x;
                        pv[t + 1] = v2.y;
                    }
                    double e1 = 0.0;
#pragma unroll
                    for (int t = 0; t < DSUB; t++) {
                        const double df = tv[t] - pv[t];
                        e1 += df * df;
                    }
                    en[i] = e1;
                }
                // ((0 + e_0) + e_1) + ... in sub-quantizer order: the running sum travels down the survivor's lanes, every lane adds its own
                d = en[0];  // lane 0: 0.0 + e_0 = e_0
#pragma unroll
                for (int i = 1; i < SPL; i++) d += en[i];
#pragma unroll
                for (int st = 1; st < LPS; st++) {
                    const u64 b = (u64)__double_as_longlong(d);
                    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x111, 0xf, 0xf, false);  // row_shr:1
                    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x111, 0xf, 0xf, false);
                    const double prev = __longlong_as_double((long long)(((u64)hi << 32) | lo));
                    if (ls == st) {
                        d = prev;
#pragma unroll
                        for (int i = 0; i < SPL; i++) d += en[i];
                    }
                }
            }
            const bool last = FLAT ? true : ((tid & (LPS - 1)) == LPS - 1);
            if (act && last) {
                const u64 key = dkey(d);
                const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (key <= T) {
                    const u32 slotp = atomicAdd(P.S.pool_cnt + q, 1u);
                    if (slotp < (u32)P.S.poolq) {
                        P.S.pool_key[(size_t)q * P.S.poolq + slotp] = key;
                        // (flat PQ: positions in the pool are those of the single list, as K3f writes them)
                        P.S.pool_val[(size_t)q * P.S.poolq + slotp] = ((u64)rank << 32) | (u64)(pos + (P.S.ivf ? 0u : (u32)beg));
                    } else {
                        P.redo[q] = 1;
                    }
                }
            }
        }
        __syncthreads();  // the list is reused by the next round
    }
    __syncthreads();
    if (tid == 0 && s_nv) {  // statistics: codes whose exact distance was computed
        if (P.stat) atomicAdd(P.stat, (unsigned long long)s_nv);
        atomicAdd(P.nver, (unsigned long long)s_nv);
    }
}
// survivors of the scan (valid records), for the statistics
__global__ void k_mfma_count(const MfmaParams P) {
    const u32 cnt = *P.surv_cnt;
    const u32 ns = cnt < P.surv_cap ? cnt : P.surv_cap;
    u32 n = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) n += P.surv[i].x != 0xFFFFFFFFu;
    const u64 mk = __builtin_amdgcn_ballot_w64(n != 0);
    (void)mk;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(P.stat + 9, (unsigned long long)n);  // (mmidx_stats::mfma_survivors)
}

// ---- redo: the queries K3m could not serve go to K3f with pass A's pool ---------------------------------------------------
__global__ void k_mfma_redo(const MfmaParams P, long long nq) {
    if (*P.n_groups == 0) return;  // (nothing was scanned; the flags were not even cleared: k_mfma_prep)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = P.S.n_order ? (long long)*P.S.n_order : P.npairs_flat;
    if (i < nq && P.redo[i]) {
        P.S.pool_cnt[i] = P.pool_snap[i];
        if (P.stat) atomicAdd(P.stat + 10, 1ull);  // (mmidx_stats::mfma_redo_queries)
    }
    const long long ic = i < n ? i : (n > 0 ? n - 1 : 0);
    const int e = n > 0 ? P.S.order[ic] : 0;
    const int q = e / P.S.w, rank = e - q * P.S.w;
    const bool go = i < n && P.redo[q];
    const int cell = P.S.ivf ? P.S.cells[e] : 0;
    const unsigned cu = cell >= 0 ? (unsigned)cell : 0u;
    const long long len = P.S.ivf ? P.S.list_off[cu + 1] - P.S.list_off[cu] : 1;
    if (go) {
        if (P.S.ivf) {
            for (int ch = 0; ch < P.fb_nchunks && (long long)ch * P.fb_chunk < len; ch++) {
                const u32 f = atomicAdd(P.fb_count, 1u);
                P.fb_items[f] = e;
                P.fb_ch[f] = ch;
            }
        } else {  // (flat PQ: K3f's chunk is the pair's rank)
            const u32 f = atomicAdd(P.fb_count, 1u);
            P.fb_items[f] = e;
            P.fb_ch[f] = rank;
        }
    }
}
