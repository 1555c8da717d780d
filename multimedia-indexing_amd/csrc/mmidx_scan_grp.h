// mmidx_scan_grp.h -- K3g: pass B of the IVFADC search as a list-major, multi-query filtered scan (gfx950).
//
// Reference loop: the per-probe body of computeKnnIVFADC, J/datastructures/IVFPQ.java:414-447 (residual :417,
// lookup table :427 -> :525-538, scan :429-446).  Results are those of K3 / K3f bit for bit; what changes is who
// shares what:
//   * a block takes ONE inverted list (or one chunk of it) and a GROUP of up to G queries that probe it (the pairs of
//     pass B are already sorted by cell).  The quantised lower-bound rows are entry-major -- the G queries' bytes of entry j
//     of sub-quantizer s are adjacent -- so a code costs ONE ds_read_b64 per sub-quantizer for the whole group (address
//     8 x byte by one SDWA shift, row in the immediate offset); bytes are <= 127, two reads add byte-wise, and the sums are
//     spread into 16-bit fields: 6 instructions per sub-quantizer and code for eight queries (K3f: 5 per query).
//   * the rows (the q8 rows of K3f) are built from an fp32 evaluation of the entries instead of the exact fp64
//     table: with the exact residual r = c - q held in LDS,
//         ||r_s - p||^2 = ||r_s||^2 + ||p||^2 - 2 r_s.p
//     the dot products run on an fp32 copy of the codebook (128 KiB for 16 x 256 x 8, L2-resident, transposed so that a
//     lane's four entries are one 16-byte load per dimension), ||p||^2 comes from a 16 KiB table, ||r_s||^2 is computed in
//     fp64 once per (query, sub-quantizer).  The quantisation step is T / 126, known before any entry is, so a row is
//     quantised in the pass that finds its minimum; the sum of the minima becomes a per-query bound on the byte sums of a
//     survivor.  The fp32 evaluation is certified: a rigorous error term is subtracted from the minima and every rounding
//     of the quantisation is allowed for in the bound, so the filter can only under-estimate (it never drops a code with
//     d <= T).  Nothing is precomputed per cell or per query.
//   * the EXACT distance is computed only for the filter's survivors, four lanes per survivor: the quad reads every
//     table entry's codebook row together (64 contiguous bytes per load) and hands the entry's running sum round the
//     quad in the reference's order (t ascending from 0.0, IVFPQ.java:531-534); the entries are added in sub-quantizer
//     order (s ascending, :435-438) -- same bits as a lookup in the fp64 table.  (dsub not 8 or 16: each lane builds the
//     m/4 entries it owns and the sum is passed down the quad.)
// Items the kernel does not handle (no finite threshold yet, degenerate or huge magnitudes) are handed to K3f through
// a device-side list, so results never depend on the heuristics.
#pragma once
#include "mmidx_kernels.h"

#define GRP_NT 512     // threads per block (8 wave64: 4 per SIMD at two blocks per CU)
#define GRP_QCAP 2048  // survivor queue entries (>= 64 lanes x GRP_SEGU codes x 8 queries = one wave's worst case, GRP_SEGU <= 4)
#define GRP_VR 128     // survivors verified per round: 4 lanes each
#ifndef GRP_SEGU
#define GRP_SEGU 2     // codes per thread per segment
#endif
#define GRP_SEG (GRP_NT * GRP_SEGU)
#ifndef GRP_EPOCH
#define GRP_EPOCH 8  // segments between two block barriers of the scan (8: hard 2.84 -> 2.91 M q/s against 4, spread unchanged; 2: -3 %)
#endif
#ifndef GRP_QMAX
#define GRP_QMAX 127  // largest table byte: two bytes add without a carry into the neighbouring query's byte (255: every read spread on its own)
#endif
#ifndef GRP_QMAX_UNION
#define GRP_QMAX_UNION 255  // largest table byte of the IVF UNION instances (0: GRP_QMAX): half the filter slack -- spread verifies 991
                            // codes per query instead of 2,279 -- at 8 instead of 6 instructions per sub-quantizer and code: spread
                            // 0.66 -> 0.71 M q/s.  The flat-PQ instances keep 127 (cfg2: 1.52 M against 1.47: their survivors cost
                            // m table reads, not m x dsub codebook terms)
#endif
#ifndef GRP_BQ
#define GRP_BQ 2  // queries whose row arithmetic the table build lets the scheduler interleave
#endif
#ifndef GRP_RW
#define GRP_RW 4  // table reads in flight per code in the interleaved scan (2, 4 or 8)
#endif
#ifndef GRP_EARLY
#define GRP_EARLY 1  // 1: the plain instances look after every quarter of a code's sub-quantizers whether any (code, query) pair of
                     // the wave is still at or below its bound, and skip the rest of the lookups when none is (hard workload: pass B
                     // 4.86 -> 4.58 ms; 0: off; n > 1: every n sub-quantizers)
#endif
#ifndef GRP_WPS
#define GRP_WPS 4  // waves per SIMD the register allocation is held to (blocks per CU x 2)
#endif

struct GrpExtra {
    // flat PQ only (PQ.computeKnnADC builds ONE table per query, PQ.java:300 -> :387-399, whatever part of the list is scanned):
    // [nq][M][256] exact fp64 tables written by k_flat_lut before the scan; a survivor's distance is then M table reads summed in
    // sub-quantizer order -- the reference's own loop (PQ.java:308-311) -- instead of M x dsub codebook terms.  null: IVF.
    const double *flat_lut;
    // Thresholds from the UNION of what the items of a query have verified so far (a block alone lowers T[q] only when ONE list
    // fills its candidate buffer): ghist[q][256] counts the accepted candidates of query q by bucket floor(d * 256 / T0[q]), T0 =
    // the threshold pass A left (copied before the launch: the map must not move).  K1 accepted candidates at or below a bucket
    // mean K1 offers below that bucket's upper edge, so the edge is a valid threshold.  null: off.
    u32 *ghist;
    const u64 *T0;
    // always on: codes verified exactly since k_group_build last looked (it reports the figure to the host through the pinned
    // hint, and the host picks the UNION instance for the next call when the figure says verification is where the time goes)
    unsigned long long *nver;  // (three words: [1] pairs grouped, [2] pairs alive after the table build, same life cycle)
};

struct GrpParams {
    ScanParams S;             // Q, coarse, perm, list_off, codes, order (sorted pairs), T, pool_*, D, m, ks, dsub, w, transform, chunk, K1, poolq
    const double *pq;         // [m][ks][dsub] (file order)
    const float *pq32T;       // [m][dsub][256] fp32 copy of the codebook, entry index innermost (0 beyond ks)
    const float *pn32;        // [m][256]       ||p_sj||^2 (+inf beyond ks: such an entry never wins a minimum)
    const double *pnmax;      // [2][m]: max_j ||p_sj||, max_j ||p_sj||^2
    const int4 *gdesc;        // per group: {cell, first index into order[], number of pairs (1..G), 0}
    const int32_t *n_groups;  // device-side count
    int nchunks;              // chunks per list (list position < 2^24)
    u32 *fb_count;            // items handed to K3f: (pair, chunk)
    int32_t *fb_items, *fb_ch;
    int cb;                   // candidate buffer entries per query: power of two >= K1 + GRP_VR
    unsigned long long *stat; // null, or: [0] += filter survivors that were verified exactly
    const struct GrpExtra *extra;  // the pointers only a few instructions off the hot path use (the kernel argument registers are full)
};

// flat PQ: the queries' exact lookup tables, block (s, q) <-> sub-quantizer s of query q, thread j <-> entry j
// (computeLookupADC PQ.java:387-399: t ascending from 0.0; the permutation of PQ.java:294-298 applied to the query first)
template <int DSUB>
__global__ __launch_bounds__(256) void k_flat_lut(const double *__restrict__ Q, const int32_t *__restrict__ perm, const double *__restrict__ pqT,
                                                  double *__restrict__ lut, int D, int m, int ks, int dsub) {
    __shared__ double tr[64];  // the sub-vector (dsub <= 64: the host checks)
    const int s = blockIdx.x, j = threadIdx.x;
    const long long q = blockIdx.y;
    if (j < dsub) {
        const int d = s * dsub + j;
        tr[j] = Q[(size_t)q * D + (perm ? perm[d] : d)];
    }
    __syncthreads();
    // (lut_entry indexes the vector by s * dsub: hand it the sub-vector's base shifted back)
    const double e = j < ks ? lut_entry<DSUB>(tr - s * (DSUB > 0 ? DSUB : dsub), pqT, s, j, ks, dsub) : __longlong_as_double(0x7FF0000000000000ll);
    lut[((size_t)q * m + s) * 256 + j] = e;
}

// LDS layout, shared by host (size) and device (offsets)
struct GrpLds {
    size_t lut8, tr, tr32, ckey, cpos, queue, sT, err, nrf, mn, inv, misc, total;
    __host__ __device__ GrpLds(int M, int G, int D, int cb) {
        size_t o = 0;
        lut8 = o; o += (size_t)G * M * 256;
        tr = o; o += (size_t)G * D * 8;
        tr32 = o; o += (size_t)G * D * 4;  // fp32 copy of the residuals (table build)
        ckey = o; o += (size_t)G * cb * 8;
        sT = o; o += (size_t)G * 8;
        err = o; o += (size_t)G * M * 8;
        cpos = o; o += (size_t)G * cb * 4;
        queue = o; o += (size_t)GRP_QCAP * 4;
        nrf = o; o += (size_t)G * M * 4;
        mn = o; o += (size_t)G * M * 4;
        inv = o; o += (size_t)G * 4;
        misc = o; o += 64 * 4;  // [0..G) q, [G..2G) rank, [2G..3G) state, [3G..4G) candidate counts, [32..35) new, [36..39) valid, [40..40+G) survivor bounds
        total = (o + 15) & ~(size_t)15;
    }
};

__device__ __forceinline__ float wave_min_f32(float x) {
    float v = x;
#define GRP_DPP_MIN(ctrl, rmask)                                                                                              \
    {                                                                                                                         \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false)); \
        v = fminf(o, v);                                                                                                      \
    }
    GRP_DPP_MIN(0x111, 0xf) GRP_DPP_MIN(0x112, 0xf) GRP_DPP_MIN(0x114, 0xf) GRP_DPP_MIN(0x118, 0xf)
    GRP_DPP_MIN(0x142, 0xa) GRP_DPP_MIN(0x143, 0xc)
#undef GRP_DPP_MIN
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// value of the previous lane of the quad (lane 0 of a quad keeps its own) / of the quad's last lane
__device__ __forceinline__ double quad_prev_f64(double x) {
    const u64 b = (u64)__double_as_longlong(x);
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x90, 0xf, 0xf, false);  // quad_perm [0,0,1,2]
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x90, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
// value of the previous lane of the quad, lane 0 takes lane 3's (a rotation)
__device__ __forceinline__ double quad_rot_f64(double x) {
    const u64 b = (u64)__double_as_longlong(x);
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x93, 0xf, 0xf, false);  // quad_perm [3,0,1,2]
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x93, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ double quad_last_f64(double x) {
    const u64 b = (u64)__double_as_longlong(x);
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0xFF, 0xf, 0xf, false);  // quad_perm [3,3,3,3]
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0xFF, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

// ---- index-side tables: fp32 codebook, entry index innermost, and ||p_sj||^2 (fp64 sum, one rounding) -------------
__global__ __launch_bounds__(256) void k_pq32_table(const double *__restrict__ pqT, float *__restrict__ pq32T, float *__restrict__ pn32,
                                                    int m, int ks, int dsub) {
    const int s = blockIdx.x, j = threadIdx.x;
    double pn = 0.0;
    for (int t = 0; t < dsub; t++) {
        const double p = j < ks ? pqT[((size_t)s * dsub + t) * ks + j] : 0.0;
        pq32T[((size_t)s * dsub + t) * 256 + j] = (float)p;
        pn += p * p;
    }
    pn32[(size_t)s * 256 + j] = j < ks ? (float)pn : __int_as_float(0x7F800000);
}
// pnmax[s] = max_j ||p_sj|| * (1 + 1e-12), pnmax[m + s] = max_j ||p_sj||^2 * (1 + 1e-12)
__global__ __launch_bounds__(256) void k_pn_max(const double *__restrict__ pqT, double *__restrict__ pnmax, int m, int ks, int dsub) {
    __shared__ double s_w[4];
    const int s = blockIdx.x, j = threadIdx.x;
    double pn = 0.0;
    for (int jj = j; jj < ks; jj += 256) {
        double a = 0.0;
        for (int t = 0; t < dsub; t++) {
            const double p = pqT[((size_t)s * dsub + t) * ks + jj];
            a += p * p;
        }
        pn = a > pn ? a : pn;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(pn, off);
        pn = o > pn ? o : pn;
    }
    if ((j & 63) == 0) s_w[j >> 6] = pn;
    __syncthreads();
    if (j == 0) {
        double a = s_w[0];
        for (int i = 1; i < 4; i++) a = s_w[i] > a ? s_w[i] : a;
        pnmax[s] = sqrt(a) * (1.0 + 1e-12);
        pnmax[m + s] = a * (1.0 + 1e-12);
    }
}
// ---- flat PQ (PQ.computeKnnADC, PQ.java:290-322): the chunks 1 .. nch-1 of the single list play the role of inverted lists,
// every query "probes" all of them: pair id q * nch + c at order[(c - 1) * nq + q], counts / starts per chunk, and the
// chunk offsets in place of list offsets (chunk 0 belongs to pass A)
__global__ void k_flat_pairs(long long nq, int nch, long long chunk, long long n, int32_t *__restrict__ order, int32_t *__restrict__ cnt,
                             int32_t *__restrict__ start, int64_t *__restrict__ off) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long np = nq * (long long)(nch - 1);
    if (i < np) {
        const long long c = 1 + i / nq, q = i - (c - 1) * nq;
        order[i] = (int32_t)(q * nch + c);
    }
    if (i <= nch) {
        cnt[i] = (i >= 1 && i < nch) ? (int32_t)nq : (i == nch ? (int32_t)np : 0);
        start[i] = i >= 1 ? (int32_t)((i - 1) * nq) : 0;
        const long long o = i * chunk;
        off[i] = o < n ? o : n;
    }
}
// ---- groups: the pairs of a cell (contiguous in order[]) in runs of G -----------------------------------------------
// single block; also zeroes the hand-back counter of this step
__global__ __launch_bounds__(1024) void k_group_build(const int32_t *__restrict__ cnt, const int32_t *__restrict__ start, int C, int G,
                                                      int4 *__restrict__ gdesc, int32_t *__restrict__ n_groups, u32 *__restrict__ fb_count,
                                                      unsigned long long *__restrict__ nver, int32_t *host_hint_ver) {
    __shared__ u32 s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0 && nver) {  // what the previous launch of k_scan_grp verified, for the host's choice of instance
        const unsigned long long v = *nver;
        *nver = 0;
        if (host_hint_ver) *host_hint_ver = v > 0x7fffffffull ? 0x7fffffff : (int32_t)v;
        // ... and how many of its pairs were still alive after their table build (nver[1] grouped, nver[2] alive): the host's
        // figures for K3s (k_pair_smin), hint words [4] and [5]
        const unsigned long long a = nver[1], b = nver[2];
        nver[1] = 0;
        nver[2] = 0;
        if (host_hint_ver) {
            host_hint_ver[3] = a > 0x7fffffffull ? 0x7fffffff : (int32_t)a;
            host_hint_ver[4] = b > 0x7fffffffull ? 0x7fffffff : (int32_t)b;
        }
    }
    if (start[C] == 0) {  // no pair survived the coarse bound (the separable benchmark): nothing to read, nothing to scan
        if (tid == 0) {
            *n_groups = 0;
            *fb_count = 0u;
        }
        return;
    }
    const int per = (C + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < C) ? lo + per : C;
    u32 sum = 0;
    for (int c = lo; c < hi; c++) sum += ((u32)cnt[c] + (u32)G - 1u) / (u32)G;
    const u32 incl = wave_incl_scan_u32(sum);
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) base += (i < wv) ? s_wave[i] : 0u;
    u32 run = base + incl - sum;
    for (int c = lo; c < hi; c++) {
        const int n = cnt[c], st = start[c];
        for (int o = 0; o < n; o += G) gdesc[run++] = make_int4(c, st + o, (n - o < G) ? n - o : G, 0);
    }
    if (tid == 1023) *n_groups = (int32_t)(base + incl);
    if (tid == 0) *fb_count = 0u;
}

// prune query i's candidate buffer to the K1 smallest by (distance, list position); publishes the threshold
__device__ __forceinline__ void grp_prune(u64 *ckey, u32 *cpos, u32 *s_cnt, u64 *s_T, int K1, u64 *Tq) {
    scan_prune(ckey, cpos, s_cnt, K1, Tq);  // (barriers inside; every thread of the block calls it)
    if (threadIdx.x == 0) *s_T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
}

// fp32 table entries of sub-quantizer s for the G queries of a group: Av[i][k] = ||r_s||^2 + ||p||^2 - 2 r_s.p for the
// entries j = 4 lane + k (r from the fp32 copy of the residuals in LDS, broadcast reads)
template <int M, int G, int DSUB>
__device__ __forceinline__ void grp_entries(float (&Av)[G][4], const int s, const int lane, const float *__restrict__ pq32T,
                                            const float *__restrict__ pn32, const float *s_tr32, const float *s_nrf, const int D,
                                            const int dsub) {
    const u32 col = (u32)s * 256u + 4u * (u32)lane;
    const float4 pn4 = *(const float4 *)(pn32 + col);
    float4 dot[G];
#pragma unroll
    for (int i = 0; i < G; i++) dot[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const u32 pbase = (u32)s * (u32)dsub * 256u + 4u * (u32)lane;
    const float *rrow = s_tr32 + s * dsub;
#pragma unroll 2
    for (int t = 0; t < (DSUB > 0 ? DSUB : dsub); t++) {
        const float4 p4 = *(const float4 *)(pq32T + (pbase + (u32)t * 256u));
#pragma unroll
        for (int i = 0; i < G; i++) {
            const float r = rrow[i * D + t];  // (broadcast read)
            dot[i].x = fmaf(r, p4.x, dot[i].x);
            dot[i].y = fmaf(r, p4.y, dot[i].y);
            dot[i].z = fmaf(r, p4.z, dot[i].z);
            dot[i].w = fmaf(r, p4.w, dot[i].w);
        }
    }
#pragma unroll
    for (int i = 0; i < G; i++) {
        const float nrf = s_nrf[i * M + s];
        Av[i][0] = fmaf(-2.f, dot[i].x, nrf + pn4.x);
        Av[i][1] = fmaf(-2.f, dot[i].y, nrf + pn4.y);
        Av[i][2] = fmaf(-2.f, dot[i].z, nrf + pn4.z);
        Av[i][3] = fmaf(-2.f, dot[i].w, nrf + pn4.w);
    }
}

// One sub-quantizer's u8 rows for the whole group, entry-major (the G queries' bytes of entry j side by side), written by one
// wave: lane l owns entries 4l .. 4l+3.  Per query: the fp32 entries (as in grp_entries: fl(fl(nrf + pn) - 2 dot), t ascending),
// the row minimum (reported in s_mn), and q8 = RNE-and-saturate((entry - mn) * inv - 0.5) through v_cvt_pk_u8_f32 -- never
// above floor((entry - mn) * inv) + 1e-3 (the bound th of phase (e) allows for it); NaN -> 0 or the largest byte, +inf (entries beyond ks) -> the largest byte GRP_QMAX.
// The codebook columns are loaded once and stay in registers for the G queries.
template <int QMAX>
__device__ __forceinline__ float grp_q(const float a, const float inv, const float c) {
    const float x = fmaf(a, inv, c);
    return QMAX < 255 ? fminf(x, (float)QMAX) : x;  // (v_cvt_pk_u8_f32 saturates at 255 itself; NaN -> GRP_QMAX or 0: both valid)
}
template <int M, int G, int DSUB, int QMAX>
__device__ __forceinline__ void grp_rows(unsigned char *lut8, const int s, const int lane, const float *__restrict__ pq32T,
                                         const float *__restrict__ pn32, const float *s_tr32, const float *s_nrf, const float *s_inv,
                                         float *s_mn, const int D) {
    static_assert(DSUB > 0, "compile-time dsub");
    const u32 col = (u32)s * 256u + 4u * (u32)lane;
    const float4 pn4 = *(const float4 *)(pn32 + col);
    const u32 pbase = (u32)s * (u32)DSUB * 256u + 4u * (u32)lane;
    float4 p4[DSUB];
#pragma unroll
    for (int t = 0; t < DSUB; t++) p4[t] = *(const float4 *)(pq32T + (pbase + (u32)t * 256u));
    u32 w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < G; i++) {
        const float *rr = s_tr32 + i * D + s * DSUB;  // (broadcast reads)
        // (two v_pk_fma_f32 per dimension: the same fused operations on the same operands as four v_fma_f32)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < DSUB; t++) {
            const float r = rr[t];
            const f32x2 r2 = {r, r}, p01 = {p4[t].x, p4[t].y}, p23 = {p4[t].z, p4[t].w};
            d01 = __builtin_elementwise_fma(r2, p01, d01);
            d23 = __builtin_elementwise_fma(r2, p23, d23);
        }
        const float nrf = s_nrf[i * M + s];
        const float a0 = fmaf(-2.f, d01.x, nrf + pn4.x), a1 = fmaf(-2.f, d01.y, nrf + pn4.y), a2 = fmaf(-2.f, d23.x, nrf + pn4.z),
                    a3 = fmaf(-2.f, d23.y, nrf + pn4.w);
        const float mn = wave_min_f32(fminf(fminf(a0, a1), fminf(a2, a3)));
        if (lane == 0) s_mn[i * M + s] = mn;
        const float inv = s_inv[i];
        const float c = fmaf(-mn, inv, -0.5f);
        if (i < 4) {
            w0[0] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a0, inv, c), (u32)i, w0[0]);
            w0[1] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a1, inv, c), (u32)i, w0[1]);
            w0[2] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a2, inv, c), (u32)i, w0[2]);
            w0[3] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a3, inv, c), (u32)i, w0[3]);
        } else {
            w1[0] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a0, inv, c), (u32)(i - 4), w1[0]);
            w1[1] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a1, inv, c), (u32)(i - 4), w1[1]);
            w1[2] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a2, inv, c), (u32)(i - 4), w1[2]);
            w1[3] = __builtin_amdgcn_cvt_pk_u8_f32(grp_q<QMAX>(a3, inv, c), (u32)(i - 4), w1[3]);
        }
        if ((i % GRP_BQ) == GRP_BQ - 1) __builtin_amdgcn_sched_barrier(0);  // (GRP_BQ queries at a time: the scheduler otherwise interleaves all G and spills)
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned char *dst = lut8 + s * (256 * G) + (4 * lane + k) * G;
        if constexpr (G == 8) *(uint2 *)dst = make_uint2(w0[k], w1[k]);
        else *(u32 *)dst = w0[k];
    }
}

// ---- K3s: a certified lower bound of Smin for every (query, far probe) pair BEFORE any row is built ----------------------
// K3g drops a pair whose Smin = sum_s min_j ||r_s - p_sj||^2 reaches the query's threshold only after it has built the pair's
// whole table -- G x m rows from m x dsub x 256 codebook floats that every item re-reads from L2 (1 MiB at 64 x 256 x 16).  On the
// reference's flagship shape (1024 dimensions, 64 x 256: YFCC100MExample.java:85-90) EVERY far pair of the synthetic workload ends
// there, so pass B was nothing but table builds.  This kernel computes the same row minima (same fp32 evaluation, same certified
// error term as phase (c) / grp_rows) with the codebook held in REGISTERS: a wave owns one sub-quantizer (its lane's four entries
// in all DSUB dimensions), a block NW of them, and the pairs stream through in batches of SMIN_NP -- the codebook is read once per
// block instead of once per item.  Output: smin[ci] (+)= the block's part of sum_s (mn_s - err_s), every term rounded down;
// k_pair_recount drops the pairs with smin >= T before the counting sort.  Dead pairs contribute nothing in K3g either (state 1).
// out[i][d] = X[i][perm[d]]: the rows K3s reads (centroids once per index, queries once per call) in transformed order, so that a
// thread's dimensions are contiguous -- gathered 8 bytes at a time through the permutation, the residual phase of k_pair_smin_mfma
// was bound by the number of cache-line requests (64 per wave-level load), not by anything it computes
__global__ void k_permute_cols(const double *__restrict__ X, const int32_t *__restrict__ perm, double *__restrict__ out, int D, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const long long row = i / D;
    const int d = (int)(i - row * D);
    out[i] = X[row * D + perm[d]];
}
struct SminParams {
    const double *Q, *coarse;
    const int32_t *perm;
    const int32_t *cells;  // [nq][w]
    const int32_t *cand;   // pair ids (q * w + rank) that survived the coarse bound, any order
    const int32_t *ncand;  // device-side count
    const float *pq32T, *pn32;
    const double *pnmax;
    double *smin;          // [ncand]
    int D, w, M;
};
#define SMIN_NP 32  // pairs per batch (a multiple of 4, at most 64: their results travel in the lanes of one register)
#ifndef SMIN_NPM
#define SMIN_NPM 64  // pairs per batch of the matrix-core form (32 or 64)
#endif
#define SMIN_NW 8   // waves = sub-quantizers per block (two blocks per CU: one stages residuals while the other multiplies)
template <int DSUB>
__global__ __launch_bounds__(SMIN_NW * 64, 4) void k_pair_smin(const SminParams P) {
    constexpr int NW = SMIN_NW, BD = NW * DSUB;  // BD: dimensions this block looks at
    static_assert((SMIN_NP * BD) % (NW * 64) == 0 && (DSUB == 4 || DSUB == 8 || DSUB == 16), "whole waves in the residual phase");
    __shared__ __attribute__((aligned(16))) float s_r[SMIN_NP][BD];
    __shared__ float s_nrf[SMIN_NP][NW];
    __shared__ double s_err[SMIN_NP][NW];
    __shared__ double s_part[SMIN_NP][NW];
    __shared__ int s_q[SMIN_NP], s_cell[SMIN_NP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = blockIdx.x, sl = sg * NW + wv;  // slice group of the block, sub-quantizer of the wave
    const int nc = *P.ncand;
    float4 p4[DSUB];
#pragma unroll
    for (int t = 0; t < DSUB; t++) p4[t] = *(const float4 *)(P.pq32T + ((size_t)sl * DSUB + t) * 256 + 4 * lane);
    const float4 pn4 = *(const float4 *)(P.pn32 + (size_t)sl * 256 + 4 * lane);
    for (int base = (int)blockIdx.y * SMIN_NP; base < nc; base += (int)gridDim.y * SMIN_NP) {
        const int nb = nc - base < SMIN_NP ? nc - base : SMIN_NP;
        if (tid < SMIN_NP) {  // (slots past the end repeat the last pair: every row of the batch holds valid numbers)
            const int e = P.cand[base + (tid < nb ? tid : nb - 1)];
            s_q[tid] = e / P.w;
            s_cell[tid] = P.cells[e];
        }
        __syncthreads();
        // (i) residuals of the block's dimensions (exact: centroid - q, IVFPQ.java:645, then the permutation), their fp32 copies,
        //     ||r_s||^2 and the error term of the fp32 entries (as phase (c) of k_scan_grp)
#pragma unroll 2
        for (int idx = tid; idx < SMIN_NP * BD; idx += NW * 64) {
            const int pi = idx / BD, dd = idx - pi * BD;
            const int d = sg * BD + dd;
            const int src = P.perm ? P.perm[d] : d;
            const double r = P.coarse[(size_t)s_cell[pi] * P.D + src] - P.Q[(size_t)s_q[pi] * P.D + src];
            s_r[pi][dd] = (float)r;
            double nr = r * r;
#pragma unroll
            for (int off = 1; off < DSUB; off <<= 1) nr += __shfl_xor(nr, off);
            if ((dd & (DSUB - 1)) == 0) {
                const int sw = dd / DSUB;
                const double pm = P.pnmax[sg * NW + sw], pm2 = P.pnmax[P.M + sg * NW + sw];
                s_err[pi][sw] = 0x1p-24 * 1.01 * (3.0 * (nr + pm2) + (2.0 * DSUB + 9.0) * sqrt(nr) * pm) + 1e-30;
                s_nrf[pi][sw] = (float)nr;
            }
        }
        __syncthreads();
        // (ii) the wave's row minimum for every pair of the batch: fl(fl(nrf + pn) - 2 dot), t ascending, fused -- grp_rows' entries.
        //      Four pairs at a time: their wave-level minima run as four interleaved chains of v_min_f32_dpp (one instruction per
        //      step and pair; the other three chains fill the two wait states a DPP read needs behind a write), and pair p's
        //      minimum lands in lane p of one register, so that the fp64 tail runs once per batch, a lane per pair.
        float mres = 0.f;
#pragma unroll 1
        for (int p0 = 0; p0 < SMIN_NP; p0 += 4) {
            float ml[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const float4 *rr = (const float4 *)&s_r[p0 + u][wv * DSUB];  // (broadcast reads)
                f32x2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
#pragma unroll
                for (int t4 = 0; t4 < DSUB / 4; t4++) {
                    const float4 r4 = rr[t4];
                    const float rv[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const f32x2 r2 = {rv[k], rv[k]}, p01 = {p4[t4 * 4 + k].x, p4[t4 * 4 + k].y}, p23 = {p4[t4 * 4 + k].z, p4[t4 * 4 + k].w};
                        d01 = __builtin_elementwise_fma(r2, p01, d01);
                        d23 = __builtin_elementwise_fma(r2, p23, d23);
                    }
                }
                const float nrf = s_nrf[p0 + u][wv];
                const float a0 = fmaf(-2.f, d01.x, nrf + pn4.x), a1 = fmaf(-2.f, d01.y, nrf + pn4.y), a2 = fmaf(-2.f, d23.x, nrf + pn4.z),
                            a3 = fmaf(-2.f, d23.y, nrf + pn4.w);
                ml[u] = fminf(fminf(a0, a1), fminf(a2, a3));
            }
#define SMIN_STEP(pre, ctrl)                                                                                                         \
    asm volatile(pre "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_min_f32_dpp %1, %1, %1 " ctrl "\n\tv_min_f32_dpp %2, %2, %2 " ctrl           \
                     "\n\tv_min_f32_dpp %3, %3, %3 " ctrl                                                                              \
                 : "+v"(ml[0]), "+v"(ml[1]), "+v"(ml[2]), "+v"(ml[3]))
            SMIN_STEP("s_nop 1\n\t", "row_shr:1 row_mask:0xf bank_mask:0xf");
            SMIN_STEP("", "row_shr:2 row_mask:0xf bank_mask:0xf");
            SMIN_STEP("", "row_shr:4 row_mask:0xf bank_mask:0xf");
            SMIN_STEP("", "row_shr:8 row_mask:0xf bank_mask:0xf");
            SMIN_STEP("", "row_bcast:15 row_mask:0xa bank_mask:0xf");
            SMIN_STEP("", "row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SMIN_STEP
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const float sv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ml[u]), 63));
                mres = lane == p0 + u ? sv : mres;
            }
        }
        if (lane < nb) {
            const double err = s_err[lane][wv];
            double term = (double)mres - err;
            term -= fabs(term) * 0x1p-40;  // (the additions below and in k_pair_recount round by far less)
            if (!(err < 1e22)) term = -__longlong_as_double(0x7FF0000000000000ll);  // magnitudes beyond what fp32 carries: never dropped
            s_part[lane][wv] = term;
        }
        __syncthreads();
        if (tid < nb) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NW; i++) s += s_part[tid][i];
            if (gridDim.x == 1) P.smin[base + tid] = s;
            else atomicAdd(P.smin + base + tid, s);  // (the other slice groups' parts; zeroed before the launch)
        }
    }
}
// K3s on the matrix cores.  The dot products of a batch are a small GEMM per sub-quantizer -- [32 pairs x DSUB] x [DSUB x 256
// entries] -- and v_mfma_f32_16x16x4_f32 computes it as the same fused chain (t ascending, exact f32) at the rate of the packed
// VALU FMAs but on the other pipe: the vector ALUs only keep a running maximum.  Per entry the chain starts at -||p||^2 / 2, so
// that  entry = fl(nrf - 2 acc)  and  min_j entry_j = fl(nrf - 2 max_j acc_j)  (rounding is monotone): one maximum per (pair,
// sub-quantizer) instead of 256 entries.  Error of an entry against the exact one, with u = 2^-24:  u nr (nrf) + u pn (pn32) +
// 2 (2.01 u |r_s||p|) (inputs of the products) + 2 dsub 1.01 u (pn / 2 + |r_s||p|) (the chain) + u (nr + pn + 2 |r_s||p|) (the
// last rounding)  <=  2^-24 1.01 [3 nr + (dsub + 3) max pn + (2 dsub + 9) |r_s| max|p|]: the term phase (i) stores.
// Operand layouts (cdna_hip_programming.md): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], C/D register v of lane l =
// row 4 (l >> 4) + v, column l & 15.  Rows = 16 pairs (two row tiles per batch of SMIN_NP = 32), columns = 16 entries.
template <int DSUB>
__global__ __launch_bounds__(SMIN_NW * 64, 4) void k_pair_smin_mfma(const SminParams P) {
    constexpr int NW = SMIN_NW, BD = NW * DSUB, KS = DSUB / 4, RS = BD + 4;  // (RS: the A-operand reads -- 16 rows x 4 k -- hit 64 different banks)
    constexpr int VPT = DSUB / 2;  // dimensions per thread in the residual phase: a thread is (pair, sub-quantizer, half)
    constexpr int NP = SMIN_NPM;   // pairs per batch: NP / 16 row tiles, two at a time
    static_assert(NP % 32 == 0 && NP <= 64 && NW == 8 && (DSUB == 4 || DSUB == 8 || DSUB == 16), "MFMA row tiles of pairs in twos; 32 x 8 x 2 threads stage 32 pairs' residuals at a time");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float s_r[NP * RS];
    __shared__ float s_pnh[NW][256];
    __shared__ float s_nrf[NP][NW];
    __shared__ float s_mx[NP][NW];
    __shared__ double s_err[NP][NW];
    __shared__ double s_part[NP][NW];
    __shared__ int s_q[NP], s_cell[NP], s_src[BD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = blockIdx.x, sl = sg * NW + wv;
    const int nc = *P.ncand;
    const int lj = lane & 15, lk = lane >> 4;
    float bq[16][KS];  // the wave's codebook slice as B operands: column block cb, k-step ks
#pragma unroll
    for (int cb = 0; cb < 16; cb++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) bq[cb][ks] = P.pq32T[((size_t)sl * DSUB + 4 * ks + lk) * 256 + cb * 16 + lj];
#pragma unroll
    for (int j = lane; j < 256; j += 64) s_pnh[wv][j] = -0.5f * P.pn32[(size_t)sl * 256 + j];  // (+inf beyond ks: -inf never wins a maximum)
    // the source dimension of every dimension of the block (the permutation does not depend on the pair)
    for (int j = tid; j < BD; j += NW * 64) s_src[j] = P.perm ? P.perm[sg * BD + j] : sg * BD + j;
    // the (query, cell) of a batch's pairs are fetched one batch ahead: cand -> cells is two dependent round trips
    int nxt_q = 0, nxt_cell = 0;
    if (tid < NP && (int)blockIdx.y * NP < nc) {
        const int b0 = (int)blockIdx.y * NP, n0 = nc - b0 < NP ? nc - b0 : NP;
        const int e = P.cand[b0 + (tid < n0 ? tid : n0 - 1)];
        nxt_q = e / P.w;
        nxt_cell = P.cells[e];
    }
    for (int base = (int)blockIdx.y * NP; base < nc; base += (int)gridDim.y * NP) {
        const int nb_ = nc - base < NP ? nc - base : NP;
        if (tid < NP) {  // (slots past the end repeat the last pair: every row of the batch holds valid numbers)
            s_q[tid] = nxt_q;
            s_cell[tid] = nxt_cell;
            const int b1 = base + (int)gridDim.y * NP;
            const int n1 = nc - b1 < NP ? nc - b1 : NP;
            const int i1 = b1 < nc ? b1 + (tid < n1 ? tid : n1 - 1) : base;  // (a valid index whatever happens: loads are unconditional)
            const int e = P.cand[i1];
            nxt_q = e / P.w;
            nxt_cell = P.cells[e];
        }
        __syncthreads();
        {   // (i) residuals (exact: centroid - q, IVFPQ.java:645, then the permutation), their fp32 copies, ||r_s||^2 and this
            //     kernel's error term.  A thread is (pair, sub-quantizer of the block, half of its dimensions): 32 x 8 x 2.
            for (int pi = tid >> 4; pi < NP; pi += 32) {
            const int sw = (tid >> 1) & 7, half = tid & 1;
            const int dd0 = sw * DSUB + half * VPT, d0 = sg * BD + dd0;
            const double *cc = P.coarse + (size_t)s_cell[pi] * P.D, *qq = P.Q + (size_t)s_q[pi] * P.D;
            double nr = 0.0;
            float rf[VPT];
            if (P.perm) {
#pragma unroll
                for (int t = 0; t < VPT; t++) {
                    const int src = s_src[dd0 + t];
                    const double r = cc[src] - qq[src];
                    rf[t] = (float)r;
                    nr += r * r;
                }
            } else {
#pragma unroll
                for (int t = 0; t < VPT; t += 2) {
                    const double2 c2 = *(const double2 *)(cc + d0 + t), q2 = *(const double2 *)(qq + d0 + t);
                    const double r0 = c2.x - q2.x, r1 = c2.y - q2.y;
                    rf[t] = (float)r0;
                    rf[t + 1] = (float)r1;
                    nr += r0 * r0;
                    nr += r1 * r1;
                }
            }
#pragma unroll
            for (int t = 0; t < VPT; t++) s_r[pi * RS + dd0 + t] = rf[t];
            nr += __shfl_xor(nr, 1);  // (the other half)
            if (half == 0) {
                const double pm = P.pnmax[sg * NW + sw], pm2 = P.pnmax[P.M + sg * NW + sw];
                s_err[pi][sw] = 0x1p-24 * 1.01 * (3.0 * nr + (DSUB + 3.0) * pm2 + (2.0 * DSUB + 9.0) * sqrt(nr) * pm) + 1e-30;
                s_nrf[pi][sw] = (float)nr;
            }
            }
        }
        __syncthreads();
        // (ii) 16 column blocks x 2 row tiles x KS k-steps of v_mfma_f32_16x16x4_f32; running maximum over the column blocks
#pragma unroll 1
        for (int r0 = 0; r0 < NP; r0 += 32) {  // (two row tiles = 32 pairs at a time: the accumulators of four would not fit)
        float av[2][KS];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) av[rt][ks] = s_r[(r0 + rt * 16 + lj) * RS + wv * DSUB + 4 * ks + lk];
        f32x4 mx[2];
#pragma unroll
        for (int cb = 0; cb < 16; cb++) {
            const float ph = s_pnh[wv][cb * 16 + lj];
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                f32x4 acc = {ph, ph, ph, ph};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][ks], bq[cb][ks], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++) mx[rt][v] = cb == 0 ? acc[v] : __builtin_fmaxf(mx[rt][v], acc[v]);
            }
        }
        // maximum over the 16 columns = the 16 lanes of each row of the wave: lanes 15, 31, 47, 63 end up with their row's
        float m_[8];
#pragma unroll
        for (int v = 0; v < 8; v++) m_[v] = mx[v >> 2][v & 3];
#define SMAX_STEP(pre, q, ctrl)                                                                                                       \
    asm volatile(pre "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\tv_max_f32_dpp %2, %2, %2 " ctrl           \
                     "\n\tv_max_f32_dpp %3, %3, %3 " ctrl                                                                              \
                 : "+v"(m_[q]), "+v"(m_[q + 1]), "+v"(m_[q + 2]), "+v"(m_[q + 3]))
#pragma unroll
        for (int q4 = 0; q4 < 8; q4 += 4) {
            SMAX_STEP("s_nop 1\n\t", q4, "row_shr:1 row_mask:0xf bank_mask:0xf");
            SMAX_STEP("", q4, "row_shr:2 row_mask:0xf bank_mask:0xf");
            SMAX_STEP("", q4, "row_shr:4 row_mask:0xf bank_mask:0xf");
            SMAX_STEP("", q4, "row_shr:8 row_mask:0xf bank_mask:0xf");
        }
#undef SMAX_STEP
        if (lj == 15) {
#pragma unroll
            for (int v = 0; v < 8; v++) s_mx[r0 + (v >> 2) * 16 + 4 * lk + (v & 3)][wv] = m_[v];
        }
        }
        // (the wave reads back what its own lanes wrote: the LDS operations of one wave complete in order)
        if (lane < nb_) {
            const float mn = fmaf(-2.f, s_mx[lane][wv], s_nrf[lane][wv]);
            const double err = s_err[lane][wv];
            double term = (double)mn - err;
            term -= fabs(term) * 0x1p-40;
            if (!(err < 1e22)) term = -__longlong_as_double(0x7FF0000000000000ll);
            s_part[lane][wv] = term;
        }
        __syncthreads();
        if (tid < nb_) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NW; i++) s += s_part[tid][i];
            if (gridDim.x == 1) P.smin[base + tid] = s;
            else atomicAdd(P.smin + base + tid, s);
        }
    }
}
// K3s, first stage for 64 x 16-shaped codebooks: the same bound from ONE v_mfma_f32_16x16x32_bf16 per (row tile, column block)
// instead of four fp32 MFMAs at a sixteenth of the rate.  k = 32 holds the residual's bf16 head and tail against the codebook's
// bf16 head twice -- [r_hi | r_lo] x [p_hi | p_hi] -- so the products carry r to 2^-17 and p to 2^-9 (products of bf16 numbers are
// exact in fp32, the accumulation is fp32).  Error of an entry against the exact one, u = 2^-24:
//   u nr + u pn  (nrf, pn32)  +  2 [2^-9 1.01 |r_s||p|  (p's head; r's tail and fp32 rounding are inside the 1.01)
//   + 33 u (pn / 2 + 1.01 |r_s||p|)]  (33 fp32 accumulation steps)  +  u (nr + pn + 2 |r_s||p|)  (last rounding)
//   <=  2^-24 1.01 [3 nr + 36 max pn + 72 |r_s| max|p|] + 2^-8 1.02 |r_s| max|p| + 1e-30.
// The bound is ~0.4 % of |r_s||p| looser than the fp32 one: pairs it cannot drop (Smin within that of T) go on to
// k_pair_smin_mfma, pairs it drops are certified by it alone.  Layouts: A[i = l & 15][k = 8 (l >> 4) .. + 7], B[k][j = l & 15]
// likewise, C/D as the fp32 form.  DSUB = 16 only (the shape that needs it: YFCC100MExample.java:85-90).
__global__ __launch_bounds__(SMIN_NW * 64, 4) void k_pair_smin_bf16(const SminParams P) {
    constexpr int DSUB = 16, NW = SMIN_NW, BD = NW * DSUB, RS = BD + 8, VPT = DSUB / 2, NP = SMIN_NPM;  // (RS in 2-byte units: rows 4 banks apart)
    static_assert(NP % 32 == 0 && NP <= 64 && NW == 8, "MFMA row tiles of pairs in twos; 32 x 8 x 2 threads stage 32 pairs' residuals at a time");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned short s_rh[NP * RS], s_rl[NP * RS];
    __shared__ float s_pnh[NW][256];
    __shared__ float s_nrf[NP][NW];
    __shared__ float s_mx[NP][NW];
    __shared__ double s_err[NP][NW];
    __shared__ double s_part[NP][NW];
    __shared__ int s_q[NP], s_cell[NP], s_src[BD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sg = blockIdx.x, sl = sg * NW + wv;
    const int nc = *P.ncand;
    const int lj = lane & 15, lk = lane >> 4;
    // B operands: column block cb, lane (j = lj, k = 8 (lk & 1) .. + 7 of the sub-quantizer's 16 dimensions -- the same head for lk and lk + 2)
    bf16x8 bq[16];
#pragma unroll
    for (int cb = 0; cb < 16; cb++)
#pragma unroll
        for (int t = 0; t < 8; t++) bq[cb][t] = (__bf16)P.pq32T[((size_t)sl * DSUB + 8 * (lk & 1) + t) * 256 + cb * 16 + lj];
#pragma unroll
    for (int j = lane; j < 256; j += 64) s_pnh[wv][j] = -0.5f * P.pn32[(size_t)sl * 256 + j];
    for (int j = tid; j < BD; j += NW * 64) s_src[j] = P.perm ? P.perm[sg * BD + j] : sg * BD + j;
    int nxt_q = 0, nxt_cell = 0;
    if (tid < NP && (int)blockIdx.y * NP < nc) {
        const int b0 = (int)blockIdx.y * NP, n0 = nc - b0 < NP ? nc - b0 : NP;
        const int e = P.cand[b0 + (tid < n0 ? tid : n0 - 1)];
        nxt_q = e / P.w;
        nxt_cell = P.cells[e];
    }
    for (int base = (int)blockIdx.y * NP; base < nc; base += (int)gridDim.y * NP) {
        const int nb_ = nc - base < NP ? nc - base : NP;
        if (tid < NP) {
            s_q[tid] = nxt_q;
            s_cell[tid] = nxt_cell;
            const int b1 = base + (int)gridDim.y * NP;
            const int n1 = nc - b1 < NP ? nc - b1 : NP;
            const int i1 = b1 < nc ? b1 + (tid < n1 ? tid : n1 - 1) : base;
            const int e = P.cand[i1];
            nxt_q = e / P.w;
            nxt_cell = P.cells[e];
        }
        __syncthreads();
        for (int pi = tid >> 4; pi < NP; pi += 32) {  // (i) residuals, split into bf16 head and tail; ||r_s||^2; this kernel's error term
            const int sw = (tid >> 1) & 7, half = tid & 1;
            const int dd0 = sw * DSUB + half * VPT, d0 = sg * BD + dd0;
            const double *cc = P.coarse + (size_t)s_cell[pi] * P.D, *qq = P.Q + (size_t)s_q[pi] * P.D;
            double nr = 0.0;
            bf16x8 rh, rl;
            if (P.perm) {
#pragma unroll
                for (int t = 0; t < VPT; t++) {
                    const int src = s_src[dd0 + t];
                    const double r = cc[src] - qq[src];
                    const float f = (float)r;
                    rh[t] = (__bf16)f;
                    rl[t] = (__bf16)(f - (float)rh[t]);  // (f - head is exact in fp32)
                    nr += r * r;
                }
            } else {
#pragma unroll
                for (int t = 0; t < VPT; t += 2) {
                    const double2 c2 = *(const double2 *)(cc + d0 + t), q2 = *(const double2 *)(qq + d0 + t);
                    const double r0 = c2.x - q2.x, r1 = c2.y - q2.y;
                    const float f0 = (float)r0, f1 = (float)r1;
                    rh[t] = (__bf16)f0;
                    rl[t] = (__bf16)(f0 - (float)rh[t]);
                    rh[t + 1] = (__bf16)f1;
                    rl[t + 1] = (__bf16)(f1 - (float)rh[t + 1]);
                    nr += r0 * r0;
                    nr += r1 * r1;
                }
            }
            *(bf16x8 *)(s_rh + pi * RS + dd0) = rh;
            *(bf16x8 *)(s_rl + pi * RS + dd0) = rl;
            nr += __shfl_xor(nr, 1);
            if (half == 0) {
                const double pm = P.pnmax[sg * NW + sw], pm2 = P.pnmax[P.M + sg * NW + sw];
                const double cross = sqrt(nr) * pm;
                s_err[pi][sw] = 0x1p-24 * 1.01 * (3.0 * nr + 36.0 * pm2 + 72.0 * cross) + 0x1p-8 * 1.02 * cross + 1e-30;
                s_nrf[pi][sw] = (float)nr;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int r0 = 0; r0 < NP; r0 += 32) {
            // A operands: rows r0 + rt * 16 + lj; lanes lk < 2 take the head's dimensions 8 lk .. + 7, lanes lk >= 2 the tail's
            bf16x8 av[2];
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
                av[rt] = *(const bf16x8 *)((lk < 2 ? s_rh : s_rl) + (r0 + rt * 16 + lj) * RS + wv * DSUB + 8 * (lk & 1));
            f32x4 mx[2];
#pragma unroll
            for (int cb = 0; cb < 16; cb++) {
                const float ph = s_pnh[wv][cb * 16 + lj];
#pragma unroll
                for (int rt = 0; rt < 2; rt++) {
                    f32x4 acc = {ph, ph, ph, ph};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bq[cb], acc, 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < 4; v++) mx[rt][v] = cb == 0 ? acc[v] : __builtin_fmaxf(mx[rt][v], acc[v]);
                }
            }
            float m_[8];
#pragma unroll
            for (int v = 0; v < 8; v++) m_[v] = mx[v >> 2][v & 3];
#define SMAX_STEP(pre, q, ctrl)                                                                                                       \
    asm volatile(pre "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\tv_max_f32_dpp %2, %2, %2 " ctrl           \
                     "\n\tv_max_f32_dpp %3, %3, %3 " ctrl                                                                              \
                 : "+v"(m_[q]), "+v"(m_[q + 1]), "+v"(m_[q + 2]), "+v"(m_[q + 3]))
#pragma unroll
            for (int q4 = 0; q4 < 8; q4 += 4) {
                SMAX_STEP("s_nop 1\n\t", q4, "row_shr:1 row_mask:0xf bank_mask:0xf");
                SMAX_STEP("", q4, "row_shr:2 row_mask:0xf bank_mask:0xf");
                SMAX_STEP("", q4, "row_shr:4 row_mask:0xf bank_mask:0xf");
                SMAX_STEP("", q4, "row_shr:8 row_mask:0xf bank_mask:0xf");
            }
#undef SMAX_STEP
            if (lj == 15) {
#pragma unroll
                for (int v = 0; v < 8; v++) s_mx[r0 + (v >> 2) * 16 + 4 * lk + (v & 3)][wv] = m_[v];
            }
        }
        if (lane < nb_) {
            const float mn = fmaf(-2.f, s_mx[lane][wv], s_nrf[lane][wv]);
            const double err = s_err[lane][wv];
            double term = (double)mn - err;
            term -= fabs(term) * 0x1p-40;
            if (!(err < 1e22)) term = -__longlong_as_double(0x7FF0000000000000ll);
            s_part[lane][wv] = term;
        }
        __syncthreads();
        if (tid < nb_) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NW; i++) s += s_part[tid][i];
            if (gridDim.x == 1) P.smin[base + tid] = s;
            else atomicAdd(P.smin + base + tid, s);
        }
    }
}
// between the two stages of K3s: the pairs the bf16 bound drops lose their keep flag, the others form the second stage's list
__global__ void k_pair_filter1(const int32_t *__restrict__ cand, const int32_t *__restrict__ ncand, int w, const u64 *__restrict__ T,
                               const double *__restrict__ smin, unsigned char *__restrict__ keep, int32_t *__restrict__ cand2,
                               int32_t *__restrict__ ncand2, int32_t *host_hint_cand) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int nc = *ncand;
    if (i == 0 && host_hint_cand) *host_hint_cand = nc;
    const int ic = i < nc ? i : (nc > 0 ? nc - 1 : 0);
    const int e = nc > 0 ? cand[ic] : 0;
    const u64 Tq = T[e / w];
    const double s = nc > 0 ? smin[ic] : 0.0;
    const double s_lo = s - fabs(s) * 0x1p-40;
    const double Td = keyd(Tq);
    const bool dead = (Tq < 0x7FF0000000000000ull) & (Td > 0.0) & (s_lo >= Td);
    const bool go = (i < nc) & !dead;
    if ((i < nc) & dead) keep[e] = 0;
    const u64 mk = __builtin_amdgcn_ballot_w64(go);
    const int lane = (int)(threadIdx.x & 63);
    const int leader = mk ? __ffsll((long long)mk) - 1 : 0;
    u32 base = 0;
    if (mk && lane == leader) base = (u32)atomicAdd(ncand2, (int)__popcll(mk));
    base = wave_read_u32(base, leader);
    if (go) cand2[base + (u32)__popcll(mk & ((1ull << lane) - 1ull))] = e;
}
// the pairs K3s found dead leave pass B; the others are counted per cell as k_pair_hist would have (cnt[C] = their total)
__global__ void k_pair_recount(const int32_t *__restrict__ cand, const int32_t *__restrict__ ncand, const int32_t *__restrict__ cells, int w,
                               const u64 *__restrict__ T, const double *__restrict__ smin, unsigned char *__restrict__ keep,
                               int32_t *__restrict__ cnt, int C, int32_t *host_hint_cand) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int nc = *ncand;
    if (i == 0 && host_hint_cand) *host_hint_cand = nc;
    const int ic = i < nc ? i : (nc > 0 ? nc - 1 : 0);  // (loads on a clamped index, stores predicated: see pair_keep())
    const int e = nc > 0 ? cand[ic] : 0;
    const int c = cells[e];
    const unsigned cu = c >= 0 ? (unsigned)c : 0u;
    const u64 Tq = T[e / w];
    const double s = nc > 0 ? smin[ic] : 0.0;
    const double s_lo = s - fabs(s) * 0x1p-40;
    const double Td = keyd(Tq);
    // every code of the list has d >= Smin >= s_lo: with s_lo >= T > 0 (finite) none can enter the query's k + 1 best
    const bool dead = (Tq < 0x7FF0000000000000ull) & (Td > 0.0) & (s_lo >= Td);
    const bool alive = (i < nc) & !dead;
    if ((i < nc) & dead) keep[e] = 0;
    if (alive) atomicAdd(cnt + (size_t)cu, 1);
    const u64 mk = __builtin_amdgcn_ballot_w64(alive);
    if (mk && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)mk) - 1)) atomicAdd(cnt + C, (int)__popcll(mk));
}

template <int DSUB>
__device__ __forceinline__ double grp_exact_entry(const double *tv, const double *__restrict__ pp, int dsub_rt) {
    double acc = 0.0;
    if constexpr (DSUB > 0) {
        double pv[DSUB];
#pragma unroll
        for (int t = 0; t < DSUB; t += 2) {
            const double2 v = *(const double2 *)(pp + t);
            pv[t] = v.x;
            pv[t + 1] = v.y;
        }
#pragma unroll
        for (int t = 0; t < DSUB; t++) {
            const double df = tv[t] - pv[t];
            acc += df * df;
        }
    } else {
        for (int t = 0; t < dsub_rt; t++) {
            const double df = tv[t] - pp[t];
            acc += df * df;
        }
    }
    return acc;
}

// DSUB = dimensions per sub-quantizer as a compile-time constant (4 / 8 / 16), or 0 = run-time value
// (m = 64: the rows of a group and its residuals -- 1024 dimensions in the reference's flagship shape -- leave room for one block
//  per CU, so the register allocation may use what two waves per SIMD leave)
// FLAT: the flat-PQ instance (survivors verified from the queries' exact tables, P.flat_lut) -- a template parameter because
// the IVF instances sit at their register budget and must not carry that branch
// UNION: the instance that lowers thresholds from the union of the verified candidates (GrpExtra::ghist).  Also a template
// parameter: with it compiled in, the plain instance spilled 32 bytes more per lane and pass B of the hard workload (where
// nothing is verified at all) ran 6 % slower; the host picks it only when the previous call verified enough to matter.
template <int M, int G, int DSUB, bool FLAT = false, bool UNION = false>
__global__ __launch_bounds__(GRP_NT, M >= 64 ? 2 : GRP_WPS) void k_scan_grp(const GrpParams P) {
    static_assert(M % 8 == 0 && G <= 8 && G * M * 256 <= 65536, "imm offsets of the table reads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int QMAX = (UNION && !FLAT && GRP_QMAX_UNION > 0) ? GRP_QMAX_UNION : GRP_QMAX;  // largest table byte of this instance
    constexpr int SPW = M / 8;   // sub-quantizers per wave in the table build
    constexpr int EPL = M / 4;   // exact entries per lane of a verifying quad
    const int D = P.S.D, ks = P.S.ks, dsub = DSUB > 0 ? DSUB : P.S.dsub, cb = P.cb, K1 = P.S.K1;
    const GrpLds L(M, G, D, cb);
    unsigned char *lut8 = smem + L.lut8;
    double *s_tr = (double *)(smem + L.tr);
    float *s_tr32 = (float *)(smem + L.tr32);
    u64 *ckey = (u64 *)(smem + L.ckey);
    u32 *cpos = (u32 *)(smem + L.cpos);
    u32 *s_queue = (u32 *)(smem + L.queue);
    u64 *s_T = (u64 *)(smem + L.sT);
    double *s_err = (double *)(smem + L.err);
    float *s_nrf = (float *)(smem + L.nrf);
    float *s_mn = (float *)(smem + L.mn);
    float *s_inv = (float *)(smem + L.inv);
    int *s_q = (int *)(smem + L.misc);
    int *s_pr = s_q + G;
    int *s_state = s_q + 2 * G;
    u32 *s_ccnt = (u32 *)(s_q + 3 * G);
    u32 *s_new = (u32 *)(s_q + 32);
    u32 *s_qvalid = (u32 *)(s_q + 36);
    u32 *s_th = (u32 *)(s_q + 40);  // [G] survivor bound of the query: sum of its u8 lower bounds <= th
    [[maybe_unused]] double *s_inv0 = (double *)(s_q + 48);  // [G] 256 / T0 of the query (0: no union histogram for it); UNION only

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u64 lane_lt = (1ull << lane) - 1ull;
    const int nv = *P.n_groups * P.nchunks;
    const int per = (nv + 7) >> 3;
    const int xcd = blockIdx.x & 7, nj = gridDim.x >> 3;

    for (int it = (int)(blockIdx.x >> 3); it < per; it += nj) {
        const int v = xcd * per + it;  // consecutive items -- the groups of one list -- run on the same XCD
        if (v >= nv) break;
        const int g = v / P.nchunks, ch = v - g * P.nchunks;
        const int4 gd = P.gdesc[g];
        const int cell = gd.x, first = gd.y, np = gd.z;
        const int64_t beg = P.S.list_off[cell];
        const int64_t len = P.S.list_off[cell + 1] - beg;
        const int64_t c0 = (int64_t)ch * P.S.chunk;
        if (c0 >= len) continue;
        const int64_t c1 = (c0 + P.S.chunk < len) ? c0 + P.S.chunk : len;
        const unsigned char *codes = (const unsigned char *)P.S.codes + (size_t)beg * M;

        // ---- (a) the group's pairs, thresholds, counters ---------------------------------------------------------
        // (Loads whose 64-bit address depends on an index are kept out of divergent regions throughout this kernel: every
        //  thread computes a valid address -- clamped to the group's last pair -- and only the stores are predicated.  hipcc
        //  (ROCm 7.2) otherwise produced wild addresses here, as described at pair_keep() in mmidx_kernels.h.)
        {
            const int tl = tid < np ? tid : np - 1;
            const int e = P.S.order[first + tl];
            const int q = e / P.S.w;
            const int pr = e - q * P.S.w;
            const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < G) {
                s_q[tid] = q;  // (slots past np repeat the last pair; their state is "nothing to do")
                s_pr[tid] = pr;
                s_T[tid] = tid < np ? T : 0;
                s_ccnt[tid] = 0;
                // quantisation step of the query's u8 rows: T / (GRP_QMAX - 1), known before any entry is (0: no finite positive
                // threshold -- the pair goes back to K3f in (e))
                float inv = 0.f;
                if (tid < np && T < 0x7FF0000000000000ull) {
                    const double iv = (double)(QMAX - 1) / keyd(T);  // (T = 0: inf)
                    if (iv < 1e30) inv = (float)iv;
                }
                s_inv[tid] = inv;
                if constexpr (UNION) {
                    double inv0 = 0.0;
                    if (P.extra->ghist && tid < np) {
                        const u64 t0 = P.extra->T0[q];
                        if (t0 < 0x7FF0000000000000ull && t0 > 0) {
                            const double iv0 = 256.0 / keyd(t0);
                            if (iv0 < 1e300) inv0 = iv0;
                        }
                    }
                    s_inv0[tid] = inv0;
                }
            }
        }
        if (tid < 3) {
            s_new[tid] = 0;
            s_qvalid[tid] = 0xFFFFFFFFu;
        }
        if (tid == 3) s_q[35] = 0;  // (UNION instances: the verification's overflow bits)
        __syncthreads();
        // ---- (b) transformed residuals (exact: centroid - q, IVFPQ.java:645, then the permutation) + fp32 copies -------
        for (int idx = tid; idx < G * D; idx += GRP_NT) {
            const int i = idx / D, d = idx - i * D;
            const int src = P.S.perm ? P.S.perm[d] : d;
            // (flat PQ: the "centroid" is a zero vector and the sign turns round -- q - 0 = q exactly, PQ.java:294-300; both loads
            //  unconditional)
            const double cv = P.S.coarse[(size_t)(P.S.ivf ? cell : 0) * D + src], qv = P.S.Q[(size_t)s_q[i] * D + src];
            const double r = P.S.ivf ? cv - qv : qv - cv;
            s_tr[idx] = r;
            s_tr32[idx] = (float)r;
        }
        __syncthreads();
        // ---- (c) per (query, sub-quantizer): ||r_s||^2 and the error term of the fp32 table ----------------------
        if (tid < G * M) {
            const int i = tid / M, s = tid - i * M;
            double nr = 0.0;
            for (int t = 0; t < dsub; t++) {
                const double r = s_tr[i * D + s * dsub + t];
                nr += r * r;
            }
            // fp32 entry = fl(fl(nrf + pn32) - 2 dot32(r32, p32)): input roundings 2^-24 each, dsub fused steps, two final
            // roundings:  |entry - exact| <= 2^-24 [3 (nr + pn) + (2 dsub + 8.1) |r_s| |p|]; 1.01 and the absolute term cover
            // the fp64 evaluation order of nr / of the exact entry and flushed denormals
            const double pm = P.pnmax[s], pm2 = P.pnmax[M + s];
            s_err[tid] = 0x1p-24 * 1.01 * (3.0 * (nr + pm2) + (2.0 * dsub + 9.0) * sqrt(nr) * pm) + 1e-30;  // (inf / nan for huge inputs: caught in (e))
            s_nrf[tid] = (float)nr;
        }
        __syncthreads();
        // ---- (d) u8 rows, ONE pass over the fp32 entries: q8 = min(255, floor((entry - row minimum) * 254 / T)).  The step
        //      depends on the query's threshold only, so a row is quantised as soon as its minimum is known (a wave owns
        //      whole rows: no barrier); what depends on the other rows -- Smin = the sum of the minima -- moves into the
        //      query's survivor bound th (e): d <= T  =>  sum_s q8 <= (T - Smin) * 254 / T.  Entry-major rows: the G queries'
        //      bytes of entry j are adjacent, one ds_read serves the whole group.
        u32 alive0 = 0;
#pragma unroll 1
        for (int ss = 0; ss < SPW; ss++) {
            if constexpr (DSUB > 0) {
                grp_rows<M, G, DSUB, QMAX>(lut8, wv * SPW + ss, lane, P.pq32T, P.pn32, s_tr32, s_nrf, s_inv, s_mn, D);
                continue;
            }
            float Av[G][4];  // (run-time dsub: the generic form)
            grp_entries<M, G, DSUB>(Av, wv * SPW + ss, lane, P.pq32T, P.pn32, s_tr32, s_nrf, D, dsub);
            u32 w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < G; i++) {
                const float m4 = fminf(fminf(Av[i][0], Av[i][1]), fminf(Av[i][2], Av[i][3]));
                const float mn = wave_min_f32(m4);
                if (lane == 0) s_mn[i * M + wv * SPW + ss] = mn;
                const float inv = s_inv[i];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 b = (u32)fminf((Av[i][k] - mn) * inv, (float)QMAX);  // >= 0; +inf beyond ks and NaN -> the largest byte
                    if (i < 4) w0[k] |= b << (8 * i);
                    else w1[k] |= b << (8 * (i - 4));
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned char *dst = lut8 + (wv * SPW + ss) * (256 * G) + (4 * lane + k) * G;
                if constexpr (G == 8) *(uint2 *)dst = make_uint2(w0[k], w1[k]);
                else *(u32 *)dst = w0[k];
            }
        }
        __syncthreads();
        // ---- (e) per query: lower bound of the sum of minima, state, survivor bound --------------------------------------
        if (tid < G * M) {  // M lanes per query: the sums over the sub-quantizers by butterfly (M is a power of two <= 64 = a wave)
            const int qi = tid / M;
            double smin = (double)s_mn[tid] - s_err[tid], es = s_err[tid];
#pragma unroll
            for (int off = M / 2; off > 0; off >>= 1) {
                smin += __shfl_xor(smin, off);
                es += __shfl_xor(es, off);
            }
            int state = 1;  // 0 scan, 1 nothing to do (no pair / list exhausted: Smin >= T), 2 hand back to K3f
            u32 th = 0;
            if (qi < np) {
                const double smin_lo = smin - fabs(smin) * 0x1p-40;  // (also covers the order of the butterfly's additions)
                const u64 T = s_T[qi];
                const double Td = keyd(T);
                const float invf = s_inv[qi];
                if (!(T < 0x7FF0000000000000ull) || !(es < 1e24) || !(fabs(smin_lo) < 1e30) || !(invf > 0.f)) {
                    state = 2;  // no finite positive threshold yet, or magnitudes beyond what fp32 carries
                } else if (!(smin_lo < Td)) {
                    state = 1;
                } else {
                    // a code with d <= T: sum_s (entry32_s - mn_s) <= d + es - sum_s mn_s <= T - Smin, every product rounded once in
                    // fp32 and truncated: sum_s q8 <= (T - Smin) * inv * (1 + 2^-23); 2^-20 covers the fp64 evaluation here
                    const double t = (Td - smin_lo) * (double)invf * (1.0 + 0x1p-20) + 0.02;  // (+: the fused roundings of grp_rows)
                    state = 0;
                    th = t < 32000.0 ? (u32)t : 32000u;  // (the packed 16-bit test treats the fields as signed)
                }
            }
            if (tid % M == 0) {
                if (state == 2) {
                    const u32 f = atomicAdd(P.fb_count, 1u);
                    P.fb_items[f] = s_q[qi] * P.S.w + s_pr[qi];
                    P.fb_ch[f] = P.S.ivf ? ch : s_pr[qi];  // (flat PQ: K3f's chunk is the pair's rank)
                }
                s_state[qi] = state;
                s_th[qi] = th;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < G; i++)
            if (s_state[i] == 0) alive0 |= 1u << i;
        alive0 = (u32)__builtin_amdgcn_readfirstlane((int)alive0);
        if (tid == 0 && ch == 0) {  // (always: two fire-and-forget atomics per item) the figures K3s is switched on and off by
            unsigned long long *nv = P.extra->nver;
            atomicAdd(nv + 1, (unsigned long long)np);
            atomicAdd(nv + 2, (unsigned long long)__popc(alive0));
        }
        if (tid == 0 && P.stat) {  // (profiling runs only) pairs of this item, pairs that survive the table build's Smin >= T test
            atomicAdd(P.stat + 2, (unsigned long long)np);
            atomicAdd(P.stat + 3, (unsigned long long)__popc(alive0));
        }
        if (alive0 == 0) continue;

        // ---- (g) filter scan ---------------------------------------------------------------------------------------
        u32 thr[G];  // (scalar registers)
#pragma unroll
        for (int i = 0; i < G; i++) thr[i] = (u32)__builtin_amdgcn_readfirstlane((int)s_th[i]);
        u32 thrp[G / 2];  // the same as packed 16-bit fields in the layout of the sums; -1: the query is not scanned
#pragma unroll
        for (int r = 0; r < G / 2; r++) {
            const int i0 = (r >> 1) * 4 + (r & 1), i1 = i0 + 2;
            const u32 t0 = ((alive0 >> i0) & 1u) ? thr[i0] : 0xFFFFu, t1 = ((alive0 >> i1) & 1u) ? thr[i1] : 0xFFFFu;
            thrp[r] = (t0 & 0xFFFFu) | (t1 << 16);
        }
        CodeVec<M, unsigned char> cur[GRP_SEGU], nxt[GRP_SEGU];
#pragma unroll
        for (int u = 0; u < GRP_SEGU; u++) {
            const int64_t p = c0 + u * GRP_NT + tid;
            cur[u].load(codes + (size_t)(p < c1 ? p : c1 - 1) * M);
        }
        u32 carried = 0;  // queue entries carried over from earlier segments (block-uniform)
        u32 n_verified = 0;
        int par = 0;
        // reserve room in the queue for this wave's survivors of one segment and write them; on failure (the queue is full) the
        // bits stay set and the lowest failing base marks where the valid entries of this round end
        auto append_try = [&](u32 &pend, const u32 segbase) {
            if constexpr (UNION) {
                // the instances for workloads where nearly every wave finds survivors: a prefix sum over the lanes' own survivor
                // counts and a short loop over a lane's bits (usually one) instead of a ballot per (query, code) bit -- about
                // 35 instructions against 150.  Survivor bit b = u * G + i here (code u of the lane, query i).
                const u32 mine = (u32)__popc(pend);
                const u32 incl = wave_incl_scan_u32(mine);
                const u32 nwu = wave_read_u32(incl, 63);
                if (nwu == 0) return;
                u32 baseu = 0;
                if (lane == 0) baseu = atomicAdd(s_new + par, nwu);
                baseu = carried + (u32)__builtin_amdgcn_readfirstlane((int)baseu);
                if (baseu + nwu <= GRP_QCAP) {
                    u32 off = baseu + incl - mine, pb = pend;
                    while (pb) {
                        const int b = __ffs((int)pb) - 1;
                        pb &= pb - 1u;
                        s_queue[off++] = ((u32)(b & (G - 1)) << 24) | (segbase + (u32)((b / G) * GRP_NT) + (u32)tid);
                    }
                    pend = 0;
                } else if (lane == 0) {
                    atomicMin(s_qvalid + par, baseu);
                }
                return;
            }
            u32 nw = 0;
#pragma unroll
            for (int b = 0; b < G * GRP_SEGU; b++) nw += (u32)__popcll(__builtin_amdgcn_ballot_w64((pend >> b) & 1u));
            if (nw == 0) return;
            u32 base = 0;
            if (lane == 0) base = atomicAdd(s_new + par, nw);
            base = carried + (u32)__builtin_amdgcn_readfirstlane((int)base);
            if (base + nw <= GRP_QCAP) {
                u32 off = base;
#pragma unroll
                for (int b = 0; b < G * GRP_SEGU; b++) {
                    const bool mine = (pend >> b) & 1u;
                    const u64 mk = __builtin_amdgcn_ballot_w64(mine);
                    if (mine) s_queue[off + (u32)__popcll(mk & lane_lt)] = ((u32)(b / GRP_SEGU) << 24) | (segbase + (u32)((b % GRP_SEGU) * GRP_NT) + (u32)tid);
                    off += (u32)__popcll(mk);
                }
                pend = 0;
            } else if (lane == 0) {
                atomicMin(s_qvalid + par, base);  // everything from here on in this round failed
            }
        };
        // The block synchronises once per EPOCH of GRP_EPOCH segments, not per segment: a wave whose codes die early does not
        // wait for the wave next to it after every 128 codes.  Survivors are appended to the queue right away (atomic
        // reservation); a wave whose reservation does not fit keeps the survivor bits of that segment (pendq) until the epoch's
        // barrier, where the queue is verified and emptied and the reservation retried.
        for (int64_t eseg = c0; eseg < c1; eseg += (int64_t)GRP_EPOCH * GRP_SEG) {
          u32 pendq[GRP_EPOCH];
#pragma unroll
          for (int ej = 0; ej < GRP_EPOCH; ej++) {
            pendq[ej] = 0;
            const int64_t seg = eseg + (int64_t)ej * GRP_SEG;
            if (seg >= c1) continue;  // (uniform)
            const bool more = seg + GRP_SEG < c1;
            if (more) {
#pragma unroll
                for (int u = 0; u < GRP_SEGU; u++) {
                    const int64_t p = seg + GRP_SEG + u * GRP_NT + tid;
                    nxt[u].load(codes + (size_t)(p < c1 ? p : c1 - 1) * M);
                }
            }
            // one code of the lane at a time: per sub-quantizer ONE table read (address = 8 x byte from one SDWA shift, row in
            // the immediate offset) brings the byte of every query of the group; the bytes are spread into 16-bit fields
            // (queries 0|2, 1|3, 4|6, 5|7: sums stay below 2^16, so plain 32-bit adds carry nothing across fields) -- about
            // 1.1 instructions per (query, sub-quantizer) instead of 3, and an eighth of the LDS requests
            u32 pend = 0;  // bit i * GRP_SEGU + u: code u of this lane survives query i's filter
#pragma unroll
            for (int u = 0; u < GRP_SEGU; u++) {
                const int64_t p = seg + u * GRP_NT + tid;
                u32 acc[G / 2];
#pragma unroll
                for (int i = 0; i < G / 2; i++) acc[i] = 0;
                // GRP_RW table reads are issued together.  With bytes <= 127 two reads are first added byte-wise (no carry can
                // leave a byte), then spread and added to the fields: 6 instructions per sub-quantizer and code for the group
                static_assert(GRP_RW % 4 == 0 || QMAX > 127, "byte-wise pre-add takes the reads in pairs of pairs");
                auto spread = [&](const u32 lo, const u32 hi) {
                    acc[0] += lo & 0x00FF00FFu;
                    acc[1] += __builtin_amdgcn_perm(0u, lo, 0x0C030C01u);
                    if constexpr (G == 8) {
                        acc[2] += hi & 0x00FF00FFu;
                        acc[3] += __builtin_amdgcn_perm(0u, hi, 0x0C030C01u);
                    }
                };
                auto spread2 = [&](const u32 lo0, const u32 hi0, const u32 lo1, const u32 hi1) {  // (two spreads, v_add3)
                    acc[0] += (lo0 & 0x00FF00FFu) + (lo1 & 0x00FF00FFu);
                    acc[1] += __builtin_amdgcn_perm(0u, lo0, 0x0C030C01u) + __builtin_amdgcn_perm(0u, lo1, 0x0C030C01u);
                    if constexpr (G == 8) {
                        acc[2] += (hi0 & 0x00FF00FFu) + (hi1 & 0x00FF00FFu);
                        acc[3] += __builtin_amdgcn_perm(0u, hi0, 0x0C030C01u) + __builtin_amdgcn_perm(0u, hi1, 0x0C030C01u);
                    }
                };
#pragma unroll
                for (int sq = 0; sq < M; sq += GRP_RW) {
                    u32 lo[GRP_RW], hi[GRP_RW];
#pragma unroll
                    for (int k = 0; k < GRP_RW; k++) {
                        const u32 off8 = byte_x8(cur[u].wd[(sq + k) >> 2], (sq + k) & 3);
                        if constexpr (G == 8) {
                            const u64 v = *(const __attribute__((address_space(3))) u64 *)(size_t)(off8 + (u32)(sq + k) * 2048u);
                            lo[k] = (u32)v;
                            hi[k] = (u32)(v >> 32);
                        } else {
                            lo[k] = *(const __attribute__((address_space(3))) u32 *)(size_t)((off8 >> 1) + (u32)(sq + k) * 1024u);
                            hi[k] = 0;
                        }
                    }
                    if constexpr (QMAX <= 63 && GRP_RW == 8) {  // four reads add byte-wise without a carry
                        spread2(lo[0] + lo[1] + lo[2] + lo[3], hi[0] + hi[1] + hi[2] + hi[3], lo[4] + lo[5] + lo[6] + lo[7],
                                hi[4] + hi[5] + hi[6] + hi[7]);
                    } else if constexpr (QMAX <= 63) {
#pragma unroll
                        for (int k = 0; k < GRP_RW; k += 4) spread(lo[k] + lo[k + 1] + lo[k + 2] + lo[k + 3], hi[k] + hi[k + 1] + hi[k + 2] + hi[k + 3]);
                    } else if constexpr (QMAX <= 127) {
#pragma unroll
                        for (int k = 0; k < GRP_RW; k += 4)
                            spread2(lo[k] + lo[k + 1], hi[k] + hi[k + 1], lo[k + 2] + lo[k + 3], hi[k + 2] + hi[k + 3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < GRP_RW; k += 2) spread2(lo[k], hi[k], lo[k + 1], hi[k + 1]);
                    }
#if GRP_EARLY > 0
                    // The byte sums only grow: when every (code, query) pair of the wave is already above its bound, the rest of
                    // the code's lookups cannot change the outcome (the final test below fails on the partial sums just the same).
                    constexpr int EI = GRP_EARLY > 1 ? GRP_EARLY : (M / 4 >= GRP_RW ? M / 4 : GRP_RW);
                    if (!UNION && sq + GRP_RW < M && (sq + GRP_RW) % EI == 0) {
                        u32 ng = 0xFFFFFFFFu;
#pragma unroll
                        for (int r = 0; r < G / 2; r++) {
                            typedef short s16x2 __attribute__((ext_vector_type(2)));
                            union { u32 w; s16x2 v; } ta, aa, dd;
                            ta.w = thrp[r];
                            aa.w = acc[r];
                            dd.v = ta.v - aa.v;
                            ng &= dd.w;
                        }
                        if (!__builtin_amdgcn_ballot_w64((~ng & 0x80008000u) != 0u)) break;
                    }
#endif
                }
                (void)spread;
                // survivors: sum of lower bounds <= the query's bound th (e).  Query i sits in field (i & 2) >> 1 of register
                // (i >> 2) * 2 + (i & 1).  First one packed test for the whole group (thrp - acc per 16-bit field: a field
                // that stays >= 0 survives; queries that are not scanned carry -1): survivors are rare -- about one (code,
                // query) pair in 2500 -- so most waves leave here
                const bool valid = p < c1;
                u32 neg = 0xFFFFFFFFu;
                [[maybe_unused]] u32 sbits = 0;  // UNION: bit i = query i survives (a 16-bit field thr - sum that stays >= 0)
#pragma unroll
                for (int r = 0; r < G / 2; r++) {
                    typedef short s16x2 __attribute__((ext_vector_type(2)));
                    union { u32 w; s16x2 v; } ta, aa, dd;
                    ta.w = thrp[r];
                    aa.w = acc[r];
                    dd.v = ta.v - aa.v;
                    neg &= dd.w;
                    if constexpr (UNION) {
                        const int i0 = (r >> 1) * 4 + (r & 1);
                        sbits |= (((~dd.w) >> 15) & 1u) << i0 | (((~dd.w) >> 31) & 1u) << (i0 + 2);
                    }
                }
                if constexpr (UNION) {
                    // (queries that are not scanned carry thr = -1: their fields are negative whatever the sum)
                    if (valid) pend |= sbits << (u * G);
                } else
                if (__builtin_amdgcn_ballot_w64(valid && (~neg & 0x80008000u) != 0u)) {
#pragma unroll
                    for (int i = 0; i < G; i++) {
                        const u32 f = (acc[(i >> 2) * 2 + (i & 1)] >> (8 * (i & 2))) & 0xFFFFu;
                        if (((alive0 >> i) & 1u) && valid && f <= thr[i]) pend |= 1u << (i * GRP_SEGU + u);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- survivors go to the queue at once when they fit; what does not fit waits for the epoch's barrier ----
            pendq[ej] = pend;
            if (__builtin_amdgcn_ballot_w64(pend != 0)) append_try(pendq[ej], (u32)seg);
            if (more) {
#pragma unroll
                for (int u = 0; u < GRP_SEGU; u++) cur[u] = nxt[u];
            }
          }  // segments of the epoch
            const bool more = eseg + (int64_t)GRP_EPOCH * GRP_SEG < c1;
            // ---- epoch end: verify when a round is full, at the end of the list, or when the queue overflowed ----
            for (bool first = true;; first = false) {
                if (!first) {  // retry what did not fit (the queue has just been emptied)
#pragma unroll
                    for (int ej = 0; ej < GRP_EPOCH; ej++)
                        if (__builtin_amdgcn_ballot_w64(pendq[ej] != 0)) append_try(pendq[ej], (u32)(eseg + (int64_t)ej * GRP_SEG));
                }
                __syncthreads();
                const u32 cnt = carried + s_new[par], qv = s_qvalid[par];
                // three sets of round words: the set of round r + 2 is cleared here, behind the barrier of round r (every
                // thread has read it -- it was round r - 1's -- before arriving) and ahead of the barrier of round r + 1
                const int par2 = par == 0 ? 2 : par - 1;
                if (tid == 0) {
                    s_new[par2] = 0;
                    s_qvalid[par2] = 0xFFFFFFFFu;
                }
                par = par == 2 ? 0 : par + 1;
                const bool failed = qv != 0xFFFFFFFFu;
                if (!failed && more && cnt < (u32)(2 * GRP_VR)) {
                    carried = cnt;
                    break;
                }
                const u32 nvalid = failed ? (qv < cnt ? qv : cnt) : cnt;
                n_verified += nvalid;
                if constexpr (UNION) {
                    // ---- the instances for workloads that verify thousands of codes per query: a LANE per survivor, GRP_NT
                    // survivors per round (the quad form below: 128).  Every lane computes its survivor's distance alone --
                    // the entries in the reference's order (t ascending from 0.0, IVFPQ.java:531-534), added in sub-quantizer
                    // order (:435-438): the same additions as the quad form -- or, flat PQ, reads it from the query's exact
                    // table.  A candidate buffer cannot promise room for a whole round any more, so appends are bounded: a
                    // lane whose slot lies beyond the buffer raises the query's overflow bit, the buffer is pruned to k + 1
                    // (publishing the threshold), and the lane tries again under the tighter threshold.
                    u32 *s_ovf = (u32 *)(s_q + 35);
                    for (u32 r0 = 0; r0 < nvalid; r0 += GRP_NT) {
                        const u32 e_raw = r0 + (u32)tid;
                        const bool act = e_raw < nvalid;
                        const u32 e = act ? e_raw : nvalid - 1u;  // (loads run on a clamped index: see the note below)
                        const u32 ent = s_queue[e];
                        const int i = (int)(ent >> 24);
                        const u32 pos = ent & 0xFFFFFFu;
                        u32 cw[M / 4];
                        if constexpr (M % 16 == 0) {
#pragma unroll
                            for (int x = 0; x < M / 16; x++) {
                                const uint4 v = *(const uint4 *)(codes + pos * (u32)M + 16 * x);
                                cw[4 * x] = v.x, cw[4 * x + 1] = v.y, cw[4 * x + 2] = v.z, cw[4 * x + 3] = v.w;
                            }
                        } else {
                            const uint2 v = *(const uint2 *)(codes + pos * (u32)M);
                            cw[0] = v.x, cw[1] = v.y;
                        }
                        double d = 0.0;
                        if constexpr (FLAT) {
                            const double *lq = P.extra->flat_lut + (size_t)s_q[i] * (size_t)(M * 256);
                            double en[M];
#pragma unroll
                            for (int sx = 0; sx < M; sx++) en[sx] = lq[sx * 256 + (int)((cw[sx >> 2] >> (8 * (sx & 3))) & 0xFFu)];
#pragma unroll
                            for (int sx = 0; sx < M; sx++) d += en[sx];
                        } else {
                            const double *tvq = s_tr + i * D;
#pragma unroll 2
                            for (int sx = 0; sx < M; sx++) {
                                const u32 cs = (cw[sx >> 2] >> (8 * (sx & 3))) & 0xFFu;
                                d += grp_exact_entry<DSUB>(tvq + sx * dsub, P.pq + (u32)((sx * ks + (int)cs) * dsub), dsub);
                            }
                        }
                        const u64 key = dkey(d);
                        bool want = act && key <= s_T[i];
                        for (;;) {
                            if (want) {
                                const u32 slot = atomicAdd(s_ccnt + i, 1u);
                                if (slot < (u32)cb) {
                                    ckey[(size_t)i * cb + slot] = key;
                                    cpos[(size_t)i * cb + slot] = pos;
                                    want = false;
                                } else {
                                    atomicOr(s_ovf, 1u << i);
                                }
                            }
                            __syncthreads();
                            const u32 ovf = *s_ovf;  // (block-uniform)
                            if (!ovf) break;
                            __syncthreads();
                            if (tid == 0) *s_ovf = 0;
                            if (tid < G && ((ovf >> tid) & 1u)) s_ccnt[tid] = (u32)cb;  // (the slots beyond the buffer were never written)
                            __syncthreads();
#pragma unroll 1
                            for (int qi = 0; qi < G; qi++)
                                if ((ovf >> qi) & 1u) grp_prune(ckey + (size_t)qi * cb, cpos + (size_t)qi * cb, s_ccnt + qi, s_T + qi, K1, P.S.T + s_q[qi]);
                            want = want && key <= s_T[i];
                        }
                    }
                } else
                // ---- exact verification, GRP_VR survivors per round, four lanes each --------------------------
                for (u32 r0 = 0; r0 < nvalid; r0 += GRP_VR) {
                    __syncthreads();  // candidate counts of the previous round are final
                    u32 full = 0;     // queries whose buffer may not take another round
#pragma unroll
                    for (int i = 0; i < G; i++) full |= (s_ccnt[i] > (u32)(cb - GRP_VR)) ? 1u << i : 0u;
                    __syncthreads();  // every thread has its snapshot before this round's appends start
#pragma unroll 1
                    for (int i = 0; i < G; i++)
                        if ((full >> i) & 1u) grp_prune(ckey + (size_t)i * cb, cpos + (size_t)i * cb, s_ccnt + i, s_T + i, K1, P.S.T + s_q[i]);
                    // Written without a divergent region on purpose: every quad runs the whole chain (quads past the end
                    // redo the round's last survivor) and only the final append is predicated.  With the loads inside
                    // `if (e < nvalid)` hipcc (ROCm 7.2) produced a kernel that died with a memory aperture violation on
                    // valid entries -- the same family as the note at pair_keep() in mmidx_kernels.h.
                    const u32 e_raw = r0 + (u32)(tid >> 2);
                    const bool act = e_raw < nvalid;
                    const u32 e = act ? e_raw : nvalid - 1u;
                    const int ql = tid & 3;
                    const u32 ent = s_queue[e];
                    const int i = (int)(ent >> 24);
                    const u32 pos = ent & 0xFFFFFFu;
                    double d = 0.0;
                    if constexpr (FLAT) {
                        // flat PQ: the query's own table (k_flat_lut), EPL reads per lane, the sum handed down the quad in
                        // sub-quantizer order: ((0 + e_0) + e_1) + ...
                        const u32 coff = pos * (u32)M + (u32)(ql * EPL);
                        u32 cw[(EPL + 3) / 4];
                        if constexpr (EPL >= 4) {
#pragma unroll
                            for (int x = 0; x < EPL / 4; x++) cw[x] = *(const u32 *)(codes + coff + 4 * x);
                        } else {
                            cw[0] = (u32) * (const unsigned short *)(codes + coff);
                        }
                        const double *lq = P.extra->flat_lut + (size_t)s_q[i] * (size_t)(M * 256);
                        double en[EPL];
#pragma unroll
                        for (int k = 0; k < EPL; k++) {
                            const int s = ql * EPL + k;
                            const u32 cs = (cw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                            en[k] = lq[s * 256 + (int)cs];
                        }
#pragma unroll
                        for (int ph = 0; ph < 4; ph++) {
                            const double din = quad_prev_f64(d);
                            if (ql == ph) {
                                d = ph ? din : 0.0;
#pragma unroll
                                for (int k = 0; k < EPL; k++) d += en[k];
                            }
                        }
                    } else if constexpr (DSUB == 8 || DSUB == 16) {
                        // Quad-wide entries: the four lanes read the survivor's whole code (one request per quad) and every
                        // codebook entry TOGETHER -- lane ql holds doubles 8r + 2ql, 8r + 2ql + 1 of round r, 64 contiguous bytes
                        // per quad and load -- instead of each lane fetching its own entries 16 bytes at a time (64 separate
                        // lines per wave instruction: the verification rounds were bound by the address unit, 4 x this).  The
                        // running sum of an entry travels round the quad in dimension order (t ascending from 0.0,
                        // IVFPQ.java:531-534): at hop h lane h & 3 adds its two terms to what lane (h - 1) & 3 handed on; the
                        // other lanes compute along on values nobody reads.  After the last hop lane 3 holds the entry and adds it
                        // to its d: ((0 + e_0) + e_1) + ... in sub-quantizer order (:435-438).
                        constexpr int R = DSUB / 8;
                        u32 cw[M / 4];
                        if constexpr (M % 16 == 0) {
#pragma unroll
                            for (int x = 0; x < M / 16; x++) {
                                const uint4 v = *(const uint4 *)(codes + pos * (u32)M + 16 * x);
                                cw[4 * x] = v.x, cw[4 * x + 1] = v.y, cw[4 * x + 2] = v.z, cw[4 * x + 3] = v.w;
                            }
                        } else {
                            const uint2 v = *(const uint2 *)(codes + pos * (u32)M);
                            cw[0] = v.x, cw[1] = v.y;
                        }
                        // (eight entries of the code in a rolled loop -- unrolled over all M the loads are hoisted together and the
                        //  kernel spills 100+ registers into the scan loop -- with the next entry's codebook doubles requested before
                        //  the current entry's hops)
#pragma unroll
                        for (int x = 0; x < M / 8; x++) {
                            u64 wrem = ((u64)cw[2 * x + 1] << 32) | (u64)cw[2 * x];
                            // (Tried for the UNION instances, which verify thousands of codes per query: all eight codebook rows of a half
                            //  requested at once -- 180 bytes per lane spilled into the scan loop, 30.7 -> 39.9 ms on the spread workload --
                            //  and two rows in flight instead of one -- no change.  Timing builds say where that workload's pass B goes:
                            //  14 of 29 ms with verification compiled out -- nearly every wave of the scan finds a survivor among its
                            //  64 x 8 (code, query) pairs and takes the extraction path -- 1 ms of pruning, the rest verification rounds.)
                            double2 pvn[R];
                            {
                                const double *pp = P.pq + ((u32)((8 * x * ks + (int)((u32)wrem & 0xFFu)) * DSUB) + 2u * (u32)ql);
#pragma unroll
                                for (int r = 0; r < R; r++) pvn[r] = *(const double2 *)(pp + 8 * r);
                            }
#pragma unroll 1
                            for (int s = 8 * x; s < 8 * x + 8; s++) {
                                double2 pv[R];
#pragma unroll
                                for (int r = 0; r < R; r++) pv[r] = pvn[r];
                                wrem >>= 8;
                                {   // (the entry after the half's last one repeats it: a valid address, the value is not used)
                                    const int sn = s + 1 < 8 * x + 8 ? s + 1 : s;
                                    const u32 cn = s + 1 < 8 * x + 8 ? (u32)wrem & 0xFFu : 0u;
                                    const double *pp = P.pq + ((u32)((sn * ks + (int)cn) * DSUB) + 2u * (u32)ql);
#pragma unroll
                                    for (int r = 0; r < R; r++) pvn[r] = *(const double2 *)(pp + 8 * r);
                                }
                                const double *tv = s_tr + i * D + s * DSUB + 2 * ql;
                                double t0[R], t1[R];
#pragma unroll
                                for (int r = 0; r < R; r++) {
                                    const double a = tv[8 * r] - pv[r].x, b2 = tv[8 * r + 1] - pv[r].y;
                                    t0[r] = a * a;
                                    t1[r] = b2 * b2;
                                }
                                double acc = 0.0;
#pragma unroll
                                for (int h = 0; h < 4 * R; h++) acc = (quad_rot_f64(acc) + t0[h >> 2]) + t1[h >> 2];
                                d += acc;
                            }
                        }
                    } else {
                    // the lane's EPL code bytes (sub-quantizers ql*EPL ..): one aligned load, no indexed register array
                    const u32 coff = pos * (u32)M + (u32)(ql * EPL);  // (< 2^24 * M: list positions are below 2^24)
                    u32 cw[(EPL + 3) / 4];
                    if constexpr (EPL >= 4) {
#pragma unroll
                        for (int x = 0; x < EPL / 4; x++) cw[x] = *(const u32 *)(codes + coff + 4 * x);
                    } else {
                        cw[0] = (u32) * (const unsigned short *)(codes + coff);
                    }
                    double en[EPL];
#pragma unroll
                    for (int k = 0; k < EPL; k++) {
                        const int s = ql * EPL + k;
                        const u32 cs = (cw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                        const double *pp = P.pq + (u32)((s * ks + (int)cs) * dsub);
                        const double *tv = s_tr + i * D + s * dsub;
                        // (t ascending from 0.0: IVFPQ.java:531-534; the loads of an entry are issued together)
                        en[k] = grp_exact_entry<DSUB>(tv, pp, dsub);
                    }
                    // d = ((0 + e_0) + e_1) + ... in sub-quantizer order: lane 0's partial sum moves down the quad
#pragma unroll
                    for (int ph = 0; ph < 4; ph++) {
                        const double din = quad_prev_f64(d);
                        if (ql == ph) {
                            d = ph ? din : 0.0;
#pragma unroll
                            for (int k = 0; k < EPL; k++) d += en[k];
                        }
                    }
                    }
                    const u64 key = dkey(d);
                    if (act && ql == 3 && key <= s_T[i]) {
                        const u32 slot = atomicAdd(s_ccnt + i, 1u);
                        ckey[(size_t)i * cb + slot] = key;
                        cpos[(size_t)i * cb + slot] = pos;
                    }
                }
                __syncthreads();
                carried = 0;
                if (!failed) break;
            }
        }  // epochs
        // ---- (h) hand the candidates to the queries' pools (at most K1 per item) ------------------------------------
        __syncthreads();
        if (tid == 0 && P.stat) atomicAdd(P.stat, (unsigned long long)n_verified);
        if (tid == 0 && n_verified) atomicAdd(P.extra->nver, (unsigned long long)n_verified);
#pragma unroll 1
        for (int i = 0; i < G; i++) {
            if (!((alive0 >> i) & 1u)) continue;
            if (s_ccnt[i] > (u32)K1) grp_prune(ckey + (size_t)i * cb, cpos + (size_t)i * cb, s_ccnt + i, s_T + i, K1, P.S.T + s_q[i]);
            const int n = (int)s_ccnt[i];
            if (n == 0) continue;
            const int q = s_q[i];
            if constexpr (UNION) if (s_inv0[i] > 0.0) {
                // The union histogram of the query: the (at most K1) candidates this item keeps are counted by bucket -- a code
                // beyond its own list's K1 best cannot be among the union's -- and one wave then looks whether the union holds K1
                // candidates at or below some bucket: its upper edge is a valid threshold for every later item of the query.
                // (Here, not in the verification rounds: the scan loop has no register to spare.)
                if (tid < n) {
                    const double x = keyd(ckey[(size_t)i * cb + tid]) * s_inv0[i];  // monotone map: truncation, clamped
                    const int b = x >= 255.0 ? 255 : (x > 0.0 ? (int)x : 0);
                    atomicAdd(P.extra->ghist + (size_t)q * 256 + b, 1u);
                }
                __syncthreads();  // (uniform: i, n and s_inv0[i] are the block's)
                if (wv == 0) {
                    const u32 *hq = P.extra->ghist + (size_t)q * 256 + 4 * lane;
                    const u32 h0 = __hip_atomic_load(hq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h1 = __hip_atomic_load(hq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                              h2 = __hip_atomic_load(hq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h3 = __hip_atomic_load(hq + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const u32 incl = wave_incl_scan_u32(h0 + h1 + h2 + h3);
                    const u64 reached = __builtin_amdgcn_ballot_w64(incl >= (u32)K1);
                    if (reached) {
                        const int Lr = __ffsll((long long)reached) - 1;
                        if (lane == Lr) {
                            u32 c = incl - (h0 + h1 + h2 + h3) + h0;
                            int b = 4 * Lr;
                            if (c < (u32)K1) { c += h1; b++; }
                            if (c < (u32)K1) { c += h2; b++; }
                            if (c < (u32)K1) { c += h3; b++; }
                            // every counted candidate has d * inv0 < b + 1 (the clamped last bucket's edge is not used)
                            if (b < 255) atomicMin(P.S.T + q, dkey((double)(b + 1) / s_inv0[i] * (1.0 + 1e-12)));
                        }
                    }
                }
            }
            const u64 Tfin = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // per-lane use only
            const bool pass = tid < n && ckey[(size_t)i * cb + tid] <= Tfin;
            const u64 mask = __builtin_amdgcn_ballot_w64(pass);
            if (mask) {
                u32 base = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if (lane == leader) base = atomicAdd(P.S.pool_cnt + q, (u32)__popcll(mask));
                base = wave_read_u32(base, leader);
                if (pass) {
                    const u32 slot = base + (u32)__popcll(mask & lane_lt);
                    if (slot < (u32)P.S.poolq) {
                        P.S.pool_key[(size_t)q * P.S.poolq + slot] = ckey[(size_t)i * cb + tid];
                        // (flat PQ: positions in the pool are those of the single list, as K3f writes them)
                        P.S.pool_val[(size_t)q * P.S.poolq + slot] = ((u64)s_pr[i] << 32) | (u64)(cpos[(size_t)i * cb + tid] + (P.S.ivf ? 0u : (u32)beg));
                    }
                }
            }
        }
        __syncthreads();  // LDS is reused by the next item
    }
}
