// mmidx_scan_grp.h -- K3g: pass B of the IVFADC search as a list-major, multi-query filtered scan (gfx950).
//
// Reference loop: the per-probe body of computeKnnIVFADC, J/datastructures/IVFPQ.java:414-447 (residual :417,
// lookup table :427 -> :525-538, scan :429-446).  Results are those of K3 / K3f bit for bit; what changes is who
// shares what:
//   * a block takes ONE inverted list (or one chunk of it) and a GROUP of up to G queries that probe it (the pairs of
//     pass B are already sorted by cell).  A code is loaded once, its m bytes are split into (8-byte slot, byte
//     selector) once, and every query of the group pays only one conflict-free ds_read_b64 + v_perm + v_add per
//     sub-quantizer (K3f: one block per (query, list), 5 VALU per lookup).
//   * the quantised lower-bound table (the q8 rows of K3f) is built from an fp32 evaluation of the entries
//     instead of the exact fp64 table: with the exact residual r = c - q held in LDS,
//         ||r_s - p||^2 = ||r_s||^2 + ||p||^2 - 2 r_s.p
//     the dot products run on an fp32 copy of the codebook (128 KiB for 16 x 256 x 8, L2-resident, transposed so that a
//     lane's four entries are one 16-byte load per dimension), ||p||^2 comes from a 16 KiB table, ||r_s||^2 is computed in
//     fp64 once per (query, sub-quantizer).  The fp32 evaluation is certified: a rigorous error term is subtracted from
//     the minima, every rounding of the quantisation is directed downwards, so the filter can only under-estimate (it
//     never drops a code with d <= T).  Nothing is precomputed per cell or per query.
//   * the EXACT distance is computed only for the filter's survivors, four lanes per survivor, each lane building the
//     m/4 table entries it needs in the reference's order (t ascending, IVFPQ.java:531-534) and the sum passed
//     through the quad in sub-quantizer order (s ascending, :435-438) -- same bits as a lookup in the fp64 table.
//   * a wave stops looking a query's bytes up as soon as none of its 128 codes can still pass (checked after every
//     four sub-quantizers): far lists die after 4-8 of the m lookups.
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
#ifndef GRP_BIS
#define GRP_BIS 0
#endif
#ifndef GRP_EPOCH
#define GRP_EPOCH 4  // segments between two block barriers of the scan
#endif
#ifndef GRP_INTERLEAVED
#define GRP_INTERLEAVED 1  // u8 rows entry-major: the group's bytes of one entry side by side, one table read per code and sub-quantizer
#endif
#ifndef GRP_RW
#define GRP_RW 4  // table reads in flight per code in the interleaved scan (2, 4 or 8)
#endif
#ifndef GRP_WPS
#define GRP_WPS 4  // waves per SIMD the register allocation is held to (blocks per CU x 2)
#endif

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
};

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
        misc = o; o += 64 * 4;  // [0..G) q, [G..2G) rank, [2G..3G) state, [3G..4G) candidate counts, [32..35) new, [36..39) valid
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
// ---- groups: the pairs of a cell (contiguous in order[]) in runs of G -----------------------------------------------
// single block; also zeroes the hand-back counter of this step
__global__ __launch_bounds__(1024) void k_group_build(const int32_t *__restrict__ cnt, const int32_t *__restrict__ start, int C, int G,
                                                      int4 *__restrict__ gdesc, int32_t *__restrict__ n_groups, u32 *__restrict__ fb_count) {
    __shared__ u32 s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
template <int M, int G, int DSUB>
__global__ __launch_bounds__(GRP_NT, GRP_WPS) void k_scan_grp(const GrpParams P) {
    static_assert(M % 8 == 0 && G <= 8 && G * M * 256 <= 65536, "imm offsets of the table reads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    [[maybe_unused]] constexpr int LQ = M * 256;  // bytes of one query's u8 table (query-major layout)
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
            }
        }
        if (tid < 3) {
            s_new[tid] = 0;
            s_qvalid[tid] = 0xFFFFFFFFu;
        }
        __syncthreads();
        // ---- (b) transformed residuals (exact: centroid - q, IVFPQ.java:645, then the permutation) + fp32 copies -------
        for (int idx = tid; idx < G * D; idx += GRP_NT) {
            const int i = idx / D, d = idx - i * D;
            const int src = P.S.perm ? P.S.perm[d] : d;
            const double r = P.S.coarse[(size_t)cell * D + src] - P.S.Q[(size_t)s_q[i] * D + src];
            s_tr[idx] = r;
            s_tr32[idx] = (float)r;
        }
        __syncthreads();
#if GRP_BIS == 1
        continue;
#endif
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
#if GRP_BIS == 2
        continue;
#endif
        // ---- (d)-(f) u8 rows.  Two passes over the fp32 entries (minima, then quantisation): recomputing an entry is 2 dsub
        //      flops and one L2-resident 16-byte load per dimension, cheaper than carrying G x M x 256 floats across the barriers
        u32 alive0 = 0;
        float mnv[G][SPW];
        // (d) per-sub-quantizer minima
#pragma unroll
        for (int ss = 0; ss < SPW; ss++) {
            float Av[G][4];
            grp_entries<M, G, DSUB>(Av, wv * SPW + ss, lane, P.pq32T, P.pn32, s_tr32, s_nrf, D, dsub);
#pragma unroll
            for (int i = 0; i < G; i++) {
                const float m4 = fminf(fminf(Av[i][0], Av[i][1]), fminf(Av[i][2], Av[i][3]));
                mnv[i][ss] = wave_min_f32(m4);
                if (lane == 0) s_mn[i * M + wv * SPW + ss] = mnv[i][ss];
            }
        }
        __syncthreads();
#if GRP_BIS == 3
        continue;
#endif
        // (e) per query: lower bound of the sum of minima, state, quantisation step
        if (tid < G) {
            int state = 1;  // 0 scan, 1 nothing to do (no pair / list exhausted: Smin >= T), 2 hand back to K3f
            float inv = 0.f;
            if (tid < np) {
                double smin = 0.0, es = 0.0;
                for (int s = 0; s < M; s++) {
                    smin += (double)s_mn[tid * M + s] - s_err[tid * M + s];
                    es += s_err[tid * M + s];
                }
                const double smin_lo = smin - fabs(smin) * 0x1p-40;
                const u64 T = s_T[tid];
                const double Td = keyd(T);
                if (!(T < 0x7FF0000000000000ull) || !(es < 1e24) || !(fabs(smin_lo) < 1e30)) {
                    state = 2;  // no finite threshold yet, or magnitudes beyond what fp32 carries
                } else if (!(smin_lo < Td)) {
                    state = 1;
                } else if (!((Td - smin_lo) > Td * 0x1p-20)) {
                    state = 2;  // degenerate step
                } else {
                    state = 0;
                    inv = (float)((254.0 / (Td - smin_lo)) * (1.0 - 0x1p-18));
                }
                if (state == 2) {
                    const u32 f = atomicAdd(P.fb_count, 1u);
                    P.fb_items[f] = s_q[tid] * P.S.w + s_pr[tid];
                    P.fb_ch[f] = ch;
                }
            }
            s_state[tid] = state;
            s_inv[tid] = inv;
        }
        __syncthreads();
#if GRP_BIS == 4
        continue;
#endif
        // (f) u8 rows: q8 = min(255, floor((entry - min) * inv)), packed four per store
#pragma unroll
        for (int i = 0; i < G; i++)
            if (s_state[i] == 0) alive0 |= 1u << i;
        if (alive0) {
#pragma unroll
            for (int ss = 0; ss < SPW; ss++) {
                float Av[G][4];
                grp_entries<M, G, DSUB>(Av, wv * SPW + ss, lane, P.pq32T, P.pn32, s_tr32, s_nrf, D, dsub);
#if GRP_INTERLEAVED
                // entry-major rows: the G queries' bytes of entry j are adjacent (one ds_read serves the whole group);
                // a query that is not scanned gets 255 everywhere -- its sums can never pass
                u32 w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < G; i++) {
                    const bool on = (alive0 >> i) & 1u;
                    const float inv = s_inv[i];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float x = fminf((Av[i][k] - mnv[i][ss]) * inv, 255.f);  // >= 0; +inf beyond ks -> 255
                        const u32 b = on ? (u32)x : 255u;
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
#else
#pragma unroll
                for (int i = 0; i < G; i++) {
                    if ((alive0 >> i) & 1u) {
                        const float inv = s_inv[i];
                        u32 pk = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float x = fminf((Av[i][k] - mnv[i][ss]) * inv, 255.f);  // >= 0; +inf beyond ks -> 255
                            pk |= (u32)x << (8 * k);
                        }
                        *(u32 *)(lut8 + i * LQ + (wv * SPW + ss) * 256 + 4 * lane) = pk;
                    }
                }
#endif
            }
        }
        alive0 = (u32)__builtin_amdgcn_readfirstlane((int)alive0);
        __syncthreads();
        if (tid == 0 && P.stat) {  // (profiling runs only) pairs of this item, pairs that survive the table build's Smin >= T test
            atomicAdd(P.stat + 2, (unsigned long long)np);
            atomicAdd(P.stat + 3, (unsigned long long)__popc(alive0));
        }
        if (alive0 == 0) continue;
#if defined(GRP_TIMING_STOP_AFTER_BUILD) || GRP_BIS == 5  // (timing experiments only: results are wrong)
        continue;
#endif

        // ---- (g) filter scan ---------------------------------------------------------------------------------------
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
#if GRP_INTERLEAVED
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
                // GRP_RW table reads are issued together, then spread and added: the wave has GRP_RW LDS requests in flight
                // instead of waiting for every pair
#pragma unroll
                for (int sq = 0; sq < M; sq += GRP_RW) {
                    if constexpr (G == 8) {
                        u64 v[GRP_RW];
#pragma unroll
                        for (int k = 0; k < GRP_RW; k++)
                            v[k] = *(const __attribute__((address_space(3))) u64 *)(size_t)(byte_x8(cur[u].wd[(sq + k) >> 2], (sq + k) & 3) + (u32)(sq + k) * 2048u);
#pragma unroll
                        for (int k = 0; k < GRP_RW; k += 2) {
                            const u32 v0x = (u32)v[k], v0y = (u32)(v[k] >> 32), v1x = (u32)v[k + 1], v1y = (u32)(v[k + 1] >> 32);
                            acc[0] += (v0x & 0x00FF00FFu) + (v1x & 0x00FF00FFu);
                            acc[1] += __builtin_amdgcn_perm(0u, v0x, 0x0C030C01u) + __builtin_amdgcn_perm(0u, v1x, 0x0C030C01u);
                            acc[2] += (v0y & 0x00FF00FFu) + (v1y & 0x00FF00FFu);
                            acc[3] += __builtin_amdgcn_perm(0u, v0y, 0x0C030C01u) + __builtin_amdgcn_perm(0u, v1y, 0x0C030C01u);
                        }
                    } else {
                        u32 v[GRP_RW];
#pragma unroll
                        for (int k = 0; k < GRP_RW; k++)
                            v[k] = *(const __attribute__((address_space(3))) u32 *)(size_t)((byte_x8(cur[u].wd[(sq + k) >> 2], (sq + k) & 3) >> 1) + (u32)(sq + k) * 1024u);
#pragma unroll
                        for (int k = 0; k < GRP_RW; k += 2) {
                            acc[0] += (v[k] & 0x00FF00FFu) + (v[k + 1] & 0x00FF00FFu);
                            acc[1] += __builtin_amdgcn_perm(0u, v[k], 0x0C030C01u) + __builtin_amdgcn_perm(0u, v[k + 1], 0x0C030C01u);
                        }
                    }
                }
                // survivors: sum of lower bounds <= 254 (255 already certifies d > T); query i sits in field (i & 2) >> 1 of
                // register (i >> 2) * 2 + (i & 1)
                const bool valid = p < c1;
#pragma unroll
                for (int i = 0; i < G; i++) {
                    const u32 f = (acc[(i >> 2) * 2 + (i & 1)] >> (8 * (i & 2))) & 0xFFFFu;
                    if (valid && f <= 254u) pend |= 1u << (i * GRP_SEGU + u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            // one code of the lane at a time (64 codes per wave and early-out test); only the survivor bits outlive a code
            u32 pend = 0;  // bit i * GRP_SEGU + u: code u of this lane survives query i's filter
#pragma unroll
            for (int u = 0; u < GRP_SEGU; u++) {
                const int64_t p = seg + u * GRP_NT + tid;
                const u32 a0 = p < c1 ? 0u : 0x10000u;
                u32 acc[G];
#pragma unroll
                for (int i = 0; i < G; i++) acc[i] = a0;
                u32 al = alive0;
#pragma unroll
                for (int sb = 0; sb < M / 4; sb++) {
                    u32 slot[4], sel[4];
                    const u32 wd = cur[u].wd[sb];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const u32 b = (wd >> (8 * k)) & 0xFFu;
                        slot[k] = b & 0xF8u;
                        sel[k] = (b & 7u) | 0x0C0C0C00u;
                    }
#pragma unroll
                    for (int i = 0; i < G; i++) {
                        if ((al >> i) & 1u) {  // wave-uniform.  (One branch per query also keeps the compiler from fusing two queries' rows
                                            //  into ds_read2st64_b64 -- 8 LDS cycles against 2 x 2: 13.6 vs 11.1 ms per step.  Sixteen hand-issued
                                            //  reads behind ONE wait for the first four sub-quantizers measured no faster: the loop is bound by
                                            //  instruction issue -- ~475 instructions per 64 codes and wave, 140 of them scalar -- not by LDS latency.)
                            uint2 rv[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) rv[k] = *(const uint2 *)(lut8 + i * LQ + (sb * 4 + k) * 256 + slot[k]);
#ifdef GRP_ONE_WAIT
                            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): one wait for the four reads instead of one per byte
#endif
#pragma unroll
                            for (int k = 0; k < 4; k++) acc[i] += __builtin_amdgcn_perm(rv[k].y, rv[k].x, sel[k]);
                            if (sb + 1 < M / 4) {
                                if (__builtin_amdgcn_ballot_w64(acc[i] <= 254u) == 0) al &= ~(1u << i);
                            }
                        }
                    }
                }
                // survivors: sum of lower bounds <= 254 (255 already certifies d > T)
#pragma unroll
                for (int i = 0; i < G; i++)
                    if (((al >> i) & 1u) && acc[i] <= 254u) pend |= 1u << (i * GRP_SEGU + u);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
#ifdef GRP_TIMING_NO_VERIFY
            if (pend == 0x2345u) s_queue[1] = 1;  // (keeps the scan alive)
            pend = 0;
#endif
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
                    double d = 0.0;
#pragma unroll
                    for (int ph = 0; ph < 4; ph++) {
                        const double din = quad_prev_f64(d);
                        if (ql == ph) {
                            d = ph ? din : 0.0;
#pragma unroll
                            for (int k = 0; k < EPL; k++) d += en[k];
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
#pragma unroll 1
        for (int i = 0; i < G; i++) {
            if (!((alive0 >> i) & 1u)) continue;
            if (s_ccnt[i] > (u32)K1) grp_prune(ckey + (size_t)i * cb, cpos + (size_t)i * cb, s_ccnt + i, s_T + i, K1, P.S.T + s_q[i]);
            const int n = (int)s_ccnt[i];
            if (n == 0) continue;
            const int q = s_q[i];
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
                        P.S.pool_val[(size_t)q * P.S.poolq + slot] = ((u64)s_pr[i] << 32) | (u64)cpos[(size_t)i * cb + tid];
                    }
                }
            }
        }
        __syncthreads();  // LDS is reused by the next item
    }
}
