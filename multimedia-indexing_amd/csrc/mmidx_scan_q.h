// mmidx_scan_q.h -- K3q: pass A decided on integers, up to four queries of a nearest list per block (round 6).
//
// K3h (k_scan_hist, mmidx_kernels.h) adds up sixteen random 8-byte fp64 table entries for EVERY code of EVERY query's nearest list:
// 32 lanes of a ds_read_b64 over 32 bank pairs collide ~3 x, 6.5 LDS cycles per wave-gather, 0.64 ms of LDS time per 16384 lists of
// 12 k codes -- the kernel sat at 0.48 of the HBM peak for five rounds whatever was done around the gather.  But only ~1 % of a list
// can be among a query's k+1 nearest, and a batch brings several queries to the same list (two on average at 16384 queries over 8192
// lists).  K3q finds out WHICH codes can matter from a quantised table that four queries share, and sums only those exactly:
//
//   tab[s][j] = { q_0, q_1, q_2, q_3 },   q_i = min(4095, floor(LUT_i[s][j] * 4000 / qr_i))        (4 x u16 = one 8-byte word)
//
// LUT_i = query i's fp64 table (IVFPQ.java:525-538, t ascending from 0.0; with fused multiply-adds: see the table build), qr_i a
// scale of the order of the query's typical distance to a code (closed form from the residual and the codebook's per-row mean vector
// and mean squared norm: the mean distance to a code with independent uniform entries).  ONE ds_read_b64 per (code, sub-quantizer)
// at the address K3h reads its fp64 entry from -- row + 8 x code byte -- serves four queries; the four partial sums ride in two
// 32-bit accumulators (16 x 4095 < 65536: the halves never carry).  Per (query, code): 1.6 LDS cycles and 0.75 vector instructions
// instead of 6.5 and 2; and a list is read from HBM once per block, not once per query.
//
// Why integers decide.  For a code c and a query let a(c) = sum_s q[s][code_s] and Y(c) = D(c) * 4000 / qr, D the real-number sum
// of the code's table entries.  floor() loses less than one unit per entry and saturation only lowers:
//       Y(c) - 16 - 1e-9 < a(c) <= Y(c) + 1e-9      if no entry of c is saturated (a(c) < 4095 guarantees that),
//                           a(c) <= Y(c) + 1e-9      always,
// so   a(c2) >= a(c1) + 18, c1 unsaturated   =>   D(c2) - D(c1) > 2 qr / 4000   =>   d(c2) > d(c1) for the fp64 sums as well (their
// rounding is 2^-49 relative).  Hence: if K1 codes of the list have a <= a* < 4095, every code with a > a* + 17 is STRICTLY farther
// than K1 codes of the list -- it cannot be among the k+1 nearest under any tie rule -- and never needs to be summed.
//
// The scan has no threshold, no branch and no bookkeeping in memory: every lane keeps, per query, the SIX smallest keys
// (a << 16 | step) of the codes it has seen (position = 256 step + thread) in registers -- a sorted insertion is one v_min and five
// v_med3.  A lane sees 1 / 256 of the list, ~0.4 of a query's ~110 candidates; that a lane holds more than six of them has
// probability ~1e-5 per query, is detected (its sixth key is within the candidate range) and sends the query to the exact kernel.
// After the scan wave i takes query i: a* = the K1-th smallest a among the block's 1536 kept keys (a bisection with ballots), the
// candidates = the kept keys with a <= a* + 17.  They are re-loaded and summed exactly, a thread per candidate, straight from the
// fp64 codebook in the reference's order (IVFPQ.java:531-534 inside :435-438 -- the bits of the table lookup).  T = the largest sum
// among the candidates with a <= a* (at least K1 of them); every candidate with d <= T goes to the query's pool.  A code that is not
// a candidate is strictly farther than all of those, so the pool holds exactly the codes with d <= T, with their exact distances:
// what K3h publishes, for a threshold that is as valid and about as tight.
//
// Handed back to the exact kernel (K3, as K3h does): a query with more than Q_HKQ candidates (massive ties), one whose K1-th
// smallest a is saturated (a scale that turned out too small), one with a lane that may have dropped a candidate, or whose scale is
// unusable (zero / non-finite).  Results never depend on the scale or on any of the heuristics.
#pragma once
#include <type_traits>
#ifndef Q_STOP
#define Q_STOP 0
#endif

#define MMIDX_Q_G 4        // queries per block
#define MMIDX_Q_SLOTS 6    // keys a lane keeps per query
#define MMIDX_Q_HKQ 192    // most candidates of one query the final phase takes (< 255: an emitted candidate's pool slot rides in a byte)
#ifndef MMIDX_Q_U
#define MMIDX_Q_U 8
#endif
//        // codes per lane per round of the scan loop
#define MMIDX_Q_SCALE 4000.0
#ifndef Q_STAGSEL
#define Q_STAGSEL (blockIdx.x >> 8)
#endif
#ifndef Q_WPS
#define Q_WPS 4
#endif
#ifndef Q_TU
#define Q_TU 4  // table rows in flight per thread (fp32 pairs: 4 registers a row at dsub = 8; 2, 4 and 8 measured the same)
#endif
#ifndef Q_CH
#define Q_CH 4  // independent lookup chains of the scan (2 / 4 / 8: 0.558 / 0.565 / 0.564 ms per 7525 groups -- no difference)
#endif
#ifndef Q_FR
#define Q_FR 2  // codebook rows in flight per candidate (1 / 2 / 4: 0.567 / 0.550 / 0.556 ms per 7525 groups)
#endif

struct QParams {
    ScanParams S;             // Q, coarse, pqT, perm, rot, cells, list_off, codes, order (pairs sorted by cell), T, pool_*, fb_*, D, m, dsub, w, transform, K1, poolq
    const int4 *gdesc;        // per group: {cell, first index into order[], number of pairs (1 .. G), 0}
    const int32_t *n_groups;  // device-side count
    const double *pq;         // [m][ks][dsub] (file order): a candidate's rows
    const double *pqstat;     // [m * dsub] mean_j p_sj[t], then [m] mean_j ||p_sj||^2
    const float *pqT32;       // the transposed codebook in fp32, two dimensions side by side: [m][dsub / 2][256] x {p[2 t2], p[2 t2 + 1]} (the table's input)
    double pmax;              // >= ||x|| of every code: sqrt(sum_s max_j ||p_sj||^2) (the fp32 table's error bound)
    unsigned long long *timing;  // Q_TIMING builds: cycles per phase, summed over the blocks' thread 0
};

typedef unsigned int q_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const q_u32x2 lds_cuint2;
typedef __attribute__((address_space(3))) const unsigned int lds_cu32;

// LDS layout, shared by host (size) and device (offsets)
struct QLds {
    size_t tab, res, misc, cent, sel, total;
    __host__ __device__ QLds(int M, int D) {
        size_t o = 0;
        tab = o; o += (size_t)M * 2048;                               // [M][256] x {4 x u16}; LDS address 0 (byte_x8 addressing).  After the scan: sel (below),
                                                                      // then the candidates' exact sums [G * HKQ] u64
        res = o; o += (size_t)MMIDX_Q_G * D * 8;                      // the queries' residuals, transformed (fp64); first: raw residuals (rotation)
        misc = o; o += 64 * 8;                                        // see the kernel
        cent = o;                                                     // the candidates: (evidence << 31 | position), later (pool slot << 24 | position);
        {                                                             // before the scan: the residuals in fp32, [D] x {r0, r1, r2, r3} (the table's input)
            const size_t a = (size_t)MMIDX_Q_G * MMIDX_Q_HKQ * 4, b = (size_t)D * 16;
            o += a > b ? a : b;
        }
        sel = tab;                                                    // the kept keys' a values (u16) [G][SLOTS * 256]: the selection's input, OVER the table
                                                                      // (12 KiB of its 32; a block with a lane to rescue builds the table again)
        total = (o + 15) & ~(size_t)15;
    }
};

__device__ __forceinline__ u32 q_med3(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef float q_f2 __attribute__((ext_vector_type(2)));
typedef float q_f4 __attribute__((ext_vector_type(4)));
// packed fp32 (two queries per instruction): r - p.lo / r - p.hi in both halves, fused multiply-add, product
__device__ __forceinline__ q_f2 q_pk_sub_lo(q_f2 r, q_f2 p) {
    q_f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(r), "v"(p));
    return d;
}
__device__ __forceinline__ q_f2 q_pk_sub_hi(q_f2 r, q_f2 p) {
    q_f2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(r), "v"(p));
    return d;
}
__device__ __forceinline__ q_f2 q_pk_fma(q_f2 a, q_f2 b, q_f2 c) {
    q_f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ q_f2 q_pk_mul(q_f2 a, q_f2 b) {
    q_f2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// (byte b of w) << SH in one VALU instruction (SDWA byte select; see byte_x8 in mmidx_kernels.h)
template <int SH>
__device__ __forceinline__ u32 q_byte_shl(u32 w, int b) {
    u32 r;
    const u32 sh = (u32)SH;
    switch (b & 3) {
        case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(sh), "v"(w)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(sh), "v"(w)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(sh), "v"(w)); break;
        default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(sh), "v"(w)); break;
    }
    return r;
}
__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// K3q's pair sort, middle step: the exclusive prefix of the per-cell pair counts (start[], cursor[] = 0: k_pair_scan's work) AND the
// groups of <= G pairs per cell (gdesc[], their number: k_group_build's work) in ONE single-block launch -- each of the two walks the
// same C counters and costs ~12 us of latency on its own.  It leaves the counters ZEROED.
__global__ __launch_bounds__(1024) void k_q_scan_groups(int32_t *__restrict__ cnt, int C, int G, int32_t *__restrict__ start, int32_t *__restrict__ cursor,
                                                        int4 *__restrict__ gdesc, int32_t *__restrict__ n_groups) {
    __shared__ u32 s_wa[16], s_wb[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (C + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < C) ? lo + per : C;
    // COMPACT code: a single block runs every instruction once, from a cold instruction cache -- the form with eight unrolled loads and
    // the counters in registers measured 20 us against 16 for the plain loops.  Where every thread has a whole multiple of four
    // counters (C = 4096 k: the headline's 8192 cells) they are read as 16-byte vectors, two in flight.
    const bool vec = C == per * 1024 && (per & 3) == 0;
    auto groups_of = [&](const int n) -> u32 { return ((u32)n + (u32)G - 1u) / (u32)G; };
    u32 sa = 0, sb = 0;
    if (vec) {
        const int4 *p4 = (const int4 *)(cnt + lo);
#pragma unroll 2
        for (int i4 = 0; i4 < per / 4; i4++) {
            const int4 a4 = p4[i4];
            sa += (u32)a4.x + (u32)a4.y + (u32)a4.z + (u32)a4.w;
            sb += groups_of(a4.x) + groups_of(a4.y) + groups_of(a4.z) + groups_of(a4.w);
        }
    } else {
        for (int c = lo; c < hi; c++) {
            const int n = cnt[c];
            sa += (u32)n;
            sb += groups_of(n);
        }
    }
    const u32 ia = wave_incl_scan_u32(sa), ib = wave_incl_scan_u32(sb);
    if (lane == 63) {
        s_wa[wv] = ia;
        s_wb[wv] = ib;
    }
    __syncthreads();
    u32 ba = 0, bb = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        ba += (i < wv) ? s_wa[i] : 0u;
        bb += (i < wv) ? s_wb[i] : 0u;
    }
    u32 ra = ba + ia - sa, rb = bb + ib - sb;  // exclusive prefixes of this thread's range
    auto emit = [&](const int c, const int n) {
        start[c] = (int32_t)ra;
        cursor[c] = 0;
        for (int o = 0; o < n; o += G) gdesc[rb++] = make_int4(c, (int)ra + o, (n - o < G) ? n - o : G, 0);
        ra += (u32)n;
    };
    // (the counters are left zeroed: pass B's pair sort counts in the same array and finds it clean, no memset in between)
    if (vec) {
        int4 *p4 = (int4 *)(cnt + lo);
#pragma unroll 1
        for (int i4 = 0; i4 < per / 4; i4++) {
            const int4 a4 = p4[i4];
            p4[i4] = make_int4(0, 0, 0, 0);
            const int c = lo + 4 * i4;
            emit(c, a4.x);
            emit(c + 1, a4.y);
            emit(c + 2, a4.z);
            emit(c + 3, a4.w);
        }
    } else {
        for (int c = lo; c < hi; c++) {
            const int n = cnt[c];
            cnt[c] = 0;
            emit(c, n);
        }
    }
    if (tid == 1023) {
        cnt[C] = 0;  // (the pairs' total, k_a1_pair_count's: k_pair_scan leaves at once when pass B's sort finds it zero)
        start[C] = (int32_t)(ba + ia);
        *n_groups = (int32_t)(bb + ib);
    }
}

#ifdef Q_TIMING
#define Q_T(n) do { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(QP.timing + (n), t_ - qt_last); qt_last = t_; } } while (0)
#else
#define Q_T(n) do {} while (0)
#endif
// One group.  GA = the queries the block is compiled for: 4, or 2 for a group of one or two pairs (six in ten at the headline batch:
// two queries per list on average) -- half the table (4-byte entries, ds_read_b32), half the insertions, half the table's arithmetic.
template <int M, int DSUB, int GA>
__device__ __forceinline__ void q_scan_group(const QParams &QP, const int4 gd, unsigned char *smem) {
    constexpr int G = MMIDX_Q_G, NT = 256, WV = 4, U = MMIDX_Q_U, HKQ = MMIDX_Q_HKQ, NS = MMIDX_Q_SLOTS;
    static_assert(G == WV, "wave i takes query i in the selection");
    static_assert(M * 2048 >= G * MMIDX_Q_HKQ * 8, "the candidates' exact sums re-use the table");
    const ScanParams &P = QP.S;
    const int D = P.D;
    const QLds L(M, D);
    double *s_r = (double *)(smem + L.res);        // [G][D]
    // misc (64 x 8 bytes): words [0..3] a* per query (int; -1: no evidence), [4..7] hand-back flags, [8..11] candidates, [12..15] entries to emit,
    // [16..19] pool bases, [24..27] a* + 17; doubles [16..19] 4000 / qr; u64 [20..23] largest evidence sum
    int *s_astar = (int *)(smem + L.misc);
    u32 *s_flag = (u32 *)(smem + L.misc) + 4;
    u32 *s_kept = (u32 *)(smem + L.misc) + 8;
    u32 *s_emit = (u32 *)(smem + L.misc) + 12;
    u32 *s_base = (u32 *)(smem + L.misc) + 16;
    double *s_inv = (double *)(smem + L.misc) + 16;
    u64 *s_max = (u64 *)(smem + L.misc) + 20;
    u32 *s_cut = (u32 *)(smem + L.misc) + 24;      // [24..27]: a* + 17 per query
    u32 *cent = (u32 *)(smem + L.cent);            // [G][HKQ]

    const int cell = gd.x, ng = gd.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef Q_TIMING
    unsigned long long qt_last = __builtin_readcyclecounter();
#endif
    int qid[G];
#pragma unroll
    for (int i = 0; i < G; i++) qid[i] = P.order[gd.y + (i < ng ? i : 0)] / P.w;
    const int64_t beg = P.list_off[cell];
    const u32 n_seg = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(P.list_off[cell + 1] - beg));  // (< 2^24: the host checks)
    if (n_seg == 0) return;
    const unsigned char *codes0;  // (rebuilt from the kernel argument: a pointer made from integers would be FLAT-addressed)
    {
        const u64 e0 = (u64)beg;
        const u32 lo32 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)e0), hi32 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(e0 >> 32));
        codes0 = (const unsigned char *)P.codes + (size_t)(((u64)hi32 << 32) | lo32) * M;
    }
    auto fetch = [&](CodeVec<M, unsigned char> &dst, const u32 i) {
        const u32 ic = min(i, n_seg - 1u);
        dst.load(codes0 + ic * (u32)M);  // (32-bit offset from a uniform base; past the end: the list's last code, never used)
    };
    // the first round's codes leave now: they are back before the table is built
    CodeVec<M, unsigned char> X0[U], X1[U];
#pragma unroll
    for (int u = 0; u < U; u++) fetch(X0[u], (u32)u * NT + (u32)tid);

    // the table's input: Q_TU rows of the transposed codebook at a time (thread j <-> entry j of every row)
    static_assert(M % Q_TU == 0, "the table is built Q_TU rows at a time");
    static_assert(DSUB % 2 == 0, "two dimensions per load");
    auto load_rows = [&](q_f2 (&p)[Q_TU][DSUB / 2], const int s0) {
#pragma unroll
        for (int j = 0; j < Q_TU; j++)
#pragma unroll
            for (int t2 = 0; t2 < DSUB / 2; t2++) p[j][t2] = ((const q_f2 *)QP.pqT32)[((size_t)(s0 + j) * (DSUB / 2) + t2) * 256 + tid];
    };
    float *r32 = (float *)(smem + L.cent);  // [D][4]: the residuals of the block's queries in fp32 (until the scan is over)

    for (int i = tid; i < 128; i += NT) ((u32 *)(smem + L.misc))[i] = 0;
    __syncthreads();
    // ---- residuals centroid - q (IVFPQ.java:645), permuted / rotated as the index says; wave i <-> query i ------------------
    if (wv < GA) {
        const int i = wv;
        const double *qrow = P.Q + (size_t)qid[i] * D, *crow = P.coarse + (size_t)cell * D;
        if (P.transform == 1) {  // RandomRotation.rotate: out = v (1 x D) . R (D x D), sequential over the row index (RandomRotation.java:44-49)
            double *raw = (double *)smem + (size_t)i * D;  // (the table region: not built yet)
            for (int d = lane; d < D; d += 64) raw[d] = crow[d] - qrow[d];
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            for (int j = lane; j < D; j += 64) {
                double total = 0.0;
                for (int k = 0; k < D; k++) total += raw[k] * P.rot[(size_t)k * D + j];
                s_r[(size_t)i * D + j] = total;
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                const int src = P.transform == 2 ? P.perm[d] : d;  // RandomPermutation.permute: out[i] = v[perm[i]]
                s_r[(size_t)i * D + d] = crow[src] - qrow[src];
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // scale: the mean distance to a code with independent uniform entries, sum_d (r_d^2 - 2 r_d mu_d) + sum_s nu_s
        double part = 0.0, rr = 0.0;
        for (int d = lane; d < D; d += 64) {
            const double r = s_r[(size_t)i * D + d];
            part += r * r - 2.0 * r * QP.pqstat[d];
            rr += r * r;
            r32[d * 4 + i] = (float)r;
        }
        if (lane < M) part += QP.pqstat[M * DSUB + lane];
        const double qr = wave_sum_f64(part);
        // the table is computed in fp32 from fp32 copies of residual and codebook: per code that matters (scaled distance <= 4128) its
        // scaled entries are off by at most 2 u sqrt(4128 scale) (||r|| + max ||x||) + 0.004 in total, u = 2^-24; a
        // query for which that is not under 0.25 units -- or whose magnitudes leave fp32's comfortable range -- goes to the exact kernel
        const double rnx = sqrt(wave_sum_f64(rr)) + QP.pmax;
        const bool fits = qr > 1e-24 && qr < 1e24 && rnx < 1e15 && 2.0 * 5.9604644775390625e-8 * sqrt(4128.0 * (MMIDX_Q_SCALE / qr)) * rnx <= 0.25;
        const bool ok = i < ng && fits;
        if (lane == 0) {
            s_inv[i] = ok ? MMIDX_Q_SCALE / qr : 0.0;
            if (i < ng && !ok) s_flag[i] = 1u;
        }
    }
    __syncthreads();
    Q_T(0);
    // ---- the table: thread j <-> entry j of every row; the exact entries of the four queries, quantised and packed -----------------
    // fp32, two queries per instruction: df = r - p (v_pk_add_f32, the codebook value broadcast to both halves), acc = df df + acc
    // (v_pk_fma_f32), x = acc (4000 / qr), q = min(4095, floor(x)).  FROM_LDS: the residuals from their fp32 copies (the first build);
    // else converted from the fp64 residuals on the way (the rescue's rebuild, the copies' place taken: the same values, the same table)
    auto calc_rows = [&](const q_f2 (&p)[Q_TU][DSUB / 2], const int s0, auto from_lds) {
        constexpr bool FROM_LDS = decltype(from_lds)::value;
        q_f2 inv2[GA / 2];
#pragma unroll
        for (int h = 0; h < GA / 2; h++) inv2[h] = q_f2{(float)s_inv[2 * h], (float)s_inv[2 * h + 1]};
#pragma unroll
        for (int j = 0; j < Q_TU; j++) {
            const int s = s0 + j;
            q_f2 acc[GA / 2];
#pragma unroll
            for (int h = 0; h < GA / 2; h++) acc[h] = q_f2{0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < DSUB; t++) {
                q_f2 rv[GA / 2];
                if constexpr (FROM_LDS) {
                    if constexpr (GA == 4) {
                        const q_f4 r4 = *(const q_f4 *)(r32 + (size_t)(s * DSUB + t) * 4);
                        rv[0] = q_f2{r4.x, r4.y};
                        rv[1] = q_f2{r4.z, r4.w};
                    } else {
                        rv[0] = *(const q_f2 *)(r32 + (size_t)(s * DSUB + t) * 4);
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < GA / 2; h++) rv[h] = q_f2{(float)s_r[(size_t)(2 * h) * D + s * DSUB + t], (float)s_r[(size_t)(2 * h + 1) * D + s * DSUB + t]};
                }
#pragma unroll
                for (int h = 0; h < GA / 2; h++) {
                    const q_f2 df = (t & 1) ? q_pk_sub_hi(rv[h], p[j][t >> 1]) : q_pk_sub_lo(rv[h], p[j][t >> 1]);
                    acc[h] = q_pk_fma(df, df, acc[h]);
                }
            }
            u32 qv[GA];
#pragma unroll
            for (int h = 0; h < GA / 2; h++) {
                const q_f2 x = q_pk_mul(acc[h], inv2[h]);  // >= 0 (or +inf: saturated)
                qv[2 * h] = (u32)__builtin_fminf(x.x, 4095.0f);  // (a NaN -- inf x 0, a flagged query's column -- gives 4095)
                qv[2 * h + 1] = (u32)__builtin_fminf(x.y, 4095.0f);
            }
            if constexpr (GA == 4) *(uint2 *)(smem + (size_t)s * 2048 + (size_t)tid * 8) = make_uint2(qv[0] | (qv[1] << 16), qv[2] | (qv[3] << 16));
            else *(u32 *)(smem + (size_t)s * 1024 + (size_t)tid * 4) = qv[0] | (qv[1] << 16);
        }
    };
    auto build_table = [&](auto from_lds) {
#pragma unroll 1
        for (int s0 = 0; s0 < M; s0 += Q_TU) {
            q_f2 pB[Q_TU][DSUB / 2];
            load_rows(pB, s0);
            calc_rows(pB, s0, from_lds);
        }
    };
    // (the first rows requested before the residuals -- one round trip less on paper -- measured slower twice: 0.648 against 0.635 ms
    //  with fp64 rows, 0.576 against 0.563 ms with the fp32 pairs)
    build_table(std::true_type{});
    __syncthreads();
#if Q_STOP == 1
    return;
#endif

    Q_T(1);
    // ---- scan: lookups and sorted insertions, nothing else --------------------------------------------------------------------
    u32 slot[GA][NS];
#pragma unroll
    for (int i = 0; i < GA; i++)
#pragma unroll
        for (int k = 0; k < NS; k++) slot[i][k] = 0xFFFFFFFFu;
    auto lookup = [&](const CodeVec<M, unsigned char> &cv, u32 &lo, u32 &hi) {
        lo = 0;
        hi = 0;
#pragma unroll
        for (int s = 0; s < M; s++) {
            if constexpr (GA == 4) {
                const q_u32x2 e = *(lds_cuint2 *)(size_t)(q_byte_shl<3>(cv.wd[s >> 2], s & 3) + (u32)s * 2048u);
                lo += e.x;
                hi += e.y;
            } else {
                lo += *(lds_cu32 *)(size_t)(q_byte_shl<2>(cv.wd[s >> 2], s & 3) + (u32)s * 1024u);
            }
        }
    };
    auto insert = [&](u32 (&sl)[NS], const u32 k) {  // sl ascending; k in, the largest out
        u32 prev = sl[0];
        sl[0] = prev < k ? prev : k;
#pragma unroll
        for (int j = 1; j < NS; j++) {
            const u32 cur = sl[j];
            sl[j] = q_med3(prev, cur, k);
            prev = cur;
        }
    };
    auto keep = [&](const u32 lo, const u32 hi, const u32 step, const bool inb) {
        // keys (a << 16 | step); a code past the end of the list must not enter (its key: all ones)
        // (step is wave-uniform and < 2^16: v_bfi_b32 takes it from its SGPR -- (mask & a) | (~mask & step))
        auto bfi = [](const u32 a, const u32 st) {
            u32 r;
            const u32 mask = 0xFFFF0000u;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "s"(st));
            return r;
        };
        const u32 k0 = inb ? ((lo << 16) | step) : 0xFFFFFFFFu, k1 = inb ? bfi(lo, step) : 0xFFFFFFFFu;
        insert(slot[0], k0);
        insert(slot[1], k1);
        if constexpr (GA == 4) {
            const u32 k2 = inb ? ((hi << 16) | step) : 0xFFFFFFFFu, k3 = inb ? bfi(hi, step) : 0xFFFFFFFFu;
            insert(slot[2], k2);
            insert(slot[3], k3);
        }
    };
    // full rounds: every position is inside the list.  Two rounds per trip, the two code buffers changing roles (no register copies)
    u32 seg = 0;
    auto round = [&](CodeVec<M, unsigned char> (&cur)[U], CodeVec<M, unsigned char> (&nxt)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) fetch(nxt[u], seg + (u32)(U + u) * NT + (u32)tid);
#pragma unroll
        for (int h4 = 0; h4 < U; h4 += Q_CH) {  // Q_CH independent lookup chains at a time
            u32 lo[Q_CH], hi[Q_CH];
#pragma unroll
            for (int u = 0; u < Q_CH; u++) lookup(cur[h4 + u], lo[u], hi[u]);
#pragma unroll
            for (int u = 0; u < Q_CH; u++) keep(lo[u], hi[u], (seg >> 8) + (u32)(h4 + u), true);
        }
        seg += (u32)U * NT;
    };
#pragma unroll 1
    while (seg + 2u * U * NT <= n_seg) {
        round(X0, X1);
        round(X1, X0);
    }
    if (seg + (u32)U * NT <= n_seg) {
        round(X0, X1);
#pragma unroll
        for (int u = 0; u < U; u++) X0[u] = X1[u];
    }
    // the rest of the list (fewer than U codes per lane; X0 holds them, clamped past the end)
#pragma unroll 1
    for (int u = 0; u < U; u++) {
        if (seg + (u32)u * NT >= n_seg) break;
        u32 lo, hi;
        CodeVec<M, unsigned char> cv = X0[0];
#pragma unroll
        for (int v = 1; v < U; v++)
            if (v == u) cv = X0[v];
        lookup(cv, lo, hi);
        keep(lo, hi, (seg >> 8) + (u32)u, seg + (u32)u * NT + (u32)tid < n_seg);
    }
    Q_T(2);
#if Q_STOP == 2
    return;
#endif
    // ---- selection: wave i takes query i -- a* = the K1-th smallest a among the 256 x NS kept keys (their a values, u16, side by side
    //      in LDS, over the table), candidates = every code with a <= a* + 17.  A lane that may have dropped a candidate (its sixth key
    //      is within the range, ~1e-5 of the lanes) walks its own codes again -- against the table, which its block builds once more. ----
    static_assert(M * 2048 >= G * MMIDX_Q_SLOTS * 256 * 2, "the selection's input re-uses the table");
    unsigned short *sa = (unsigned short *)(smem + L.sel);  // [G][NS * 256]
    u32 *s_resc = (u32 *)(smem + L.misc) + 28;              // some lane of the block has codes to walk again
    __syncthreads();  // (every wave is past its last lookup)
#pragma unroll
    for (int i = 0; i < GA; i++)
#pragma unroll
        for (int k = 0; k < NS; k++) sa[(size_t)i * NS * 256 + (size_t)k * 256 + tid] = (unsigned short)(slot[i][k] >> 16);
    __syncthreads();
    if (wv < ng && s_flag[wv] == 0u) {
        const int i = wv;
        constexpr int WPL = NS * 2;  // dwords (pairs of values) per lane
        u32 kk[WPL];
#pragma unroll
        for (int k = 0; k < WPL; k++) kk[k] = ((const u32 *)(sa + (size_t)i * NS * 256))[k * 64 + lane];
        // (a = 0xFFFF only in the all-ones key of an empty slot; a real sum is at most 16 x 4095)
        u32 nvalid = 0;
#pragma unroll
        for (int k = 0; k < WPL; k++)
            nvalid += (u32)__popcll(__builtin_amdgcn_ballot_w64((kk[k] & 0xFFFFu) != 0xFFFFu)) + (u32)__popcll(__builtin_amdgcn_ballot_w64((kk[k] >> 16) != 0xFFFFu));
        int astar = -1;
        u32 cut = 0xFFFEu;  // (no evidence: every kept key of a list this short)
        bool bad = false;
        if (nvalid >= (u32)P.K1) {
            u32 lo_a = 0, hi_a = 0xFFFEu;  // smallest A with at least K1 values <= A
            while (lo_a < hi_a) {
                const u32 mid = lo_a + ((hi_a - lo_a) >> 1);
                u32 c = 0;
#pragma unroll
                for (int k = 0; k < WPL; k++)
                    c += (u32)__popcll(__builtin_amdgcn_ballot_w64((kk[k] & 0xFFFFu) <= mid)) + (u32)__popcll(__builtin_amdgcn_ballot_w64((kk[k] >> 16) <= mid));
                if (c >= (u32)P.K1) hi_a = mid;
                else lo_a = mid + 1;
            }
            if (lo_a >= 4095u) bad = true;  // the evidence would include saturated codes: the scale was too small
            astar = (int)lo_a;
            cut = lo_a + 17u;
        }
        if (lane == 0) {
            s_astar[i] = astar;
            s_cut[i] = cut;
            if (bad) s_flag[i] = 1u;
        }
    }
    __syncthreads();
    // every lane: its kept keys within the range -> the query's candidate list
    auto push = [&](const int i, const u32 a, const u32 pos) {
        const u32 at = atomicAdd(s_kept + i, 1u);
        if (at < (u32)HKQ) cent[i * HKQ + at] = pos | (((int)a <= s_astar[i]) ? 0x80000000u : 0u);
    };
    u32 rmask = 0;  // the queries this lane has to walk its codes again for
#pragma unroll
    for (int i = 0; i < GA; i++) {
        if (i >= ng || s_flag[i] != 0u) continue;  // block-uniform
        const u32 cut = s_cut[i];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const u32 key = slot[i][k];
            if (key != 0xFFFFFFFFu && (key >> 16) <= cut) push(i, key >> 16, ((key & 0xFFFFu) << 8) | (u32)tid);
        }
        const u32 last = slot[i][NS - 1];
        if (last != 0xFFFFFFFFu && (last >> 16) <= cut) rmask |= 1u << i;
    }
    if (rmask) *s_resc = 1u;
    __syncthreads();
    if (*s_resc) {  // block-uniform, about one block in a hundred
        build_table(std::false_type{});  // (the same instructions on the same values: the same table)
        __syncthreads();
        if (rmask) {
#pragma unroll
            for (int i = 0; i < GA; i++) {
                if (!((rmask >> i) & 1u)) continue;
                const u32 cut = s_cut[i], last = slot[i][NS - 1];
                for (u32 step = 0; step * NT + (u32)tid < n_seg; step++) {
                    CodeVec<M, unsigned char> cv;
                    cv.load(codes0 + (step * NT + (u32)tid) * (u32)M);
                    u32 lo, hi;
                    lookup(cv, lo, hi);
                    const u32 a = (i & 1) ? ((i < 2 ? lo : hi) >> 16) : ((i < 2 ? lo : hi) & 0xFFFFu);
                    if (a <= cut && ((a << 16) | step) > last) push(i, a, (step << 8) | (u32)tid);  // (keys up to `last` are in the slots)
                }
            }
        }
    }
    __syncthreads();
    Q_T(3);
    bool live[G];
    u32 first[G + 1];
    first[0] = 0;
#pragma unroll
    for (int i = 0; i < G; i++) {
        live[i] = i < ng && s_flag[i] == 0u && s_kept[i] <= (u32)HKQ;
        if (i < ng && !live[i] && tid == 0) {  // this query's pair goes to the exact kernel, chunk by chunk of the list
            const u32 nch = (n_seg + (u32)P.chunk - 1u) / (u32)P.chunk;
            const u32 sl = atomicAdd(P.fb_count, nch);
            for (u32 c = 0; c < nch; c++) {
                P.fb_items[sl + c] = qid[i] * P.w;
                P.fb_ch[sl + c] = (int32_t)c;
            }
        }
        first[i + 1] = first[i] + (live[i] ? s_kept[i] : 0u);
    }
    const u32 ncand = first[G];
    if (ncand == 0) return;  // block-uniform
    // ---- exact sums, a thread per candidate: d = sum_s [ sum_t (r[s dsub + t] - pq[s][code_s][t])^2 ], t then s ascending from 0.0.
    //      Two candidates per thread side by side, their codes requested together, Q_FR codebook rows of each in flight ---------------
    u64 *ckey = (u64 *)smem;  // [ncand] (the table is dead: every wave passed the barrier above after its last lookup)
    {
        const u32 ia = (u32)tid, ib = (u32)tid + NT, ic = (u32)tid + 2 * NT;
        auto which = [&](const u32 idx) -> int {
            int i = 0;
#pragma unroll
            for (int k = 1; k < G; k++) i += (idx >= first[k]) ? 1 : 0;
            return i;
        };
        auto entry_of = [&](const u32 idx, const int i) -> u32 { return cent[i * HKQ + (idx - (i == 0 ? first[0] : (i == 1 ? first[1] : (i == 2 ? first[2] : first[3]))))]; };
        auto sum2 = [&](const u32 i0, const u32 i1) {  // candidates i0 and i1 (i1 may be past the end)
            const bool h0 = i0 < ncand, h1 = i1 < ncand;
            if (!__builtin_amdgcn_ballot_w64(h0)) return;
            const int q0 = h0 ? which(i0) : 0, q1 = h1 ? which(i1) : 0;
            const u32 v0 = h0 ? entry_of(i0, q0) : 0u, v1 = h1 ? entry_of(i1, q1) : 0u;
            CodeVec<M, unsigned char> c0v, c1v;
            c0v.load(codes0 + (v0 & 0xFFFFFFu) * (u32)M);
            c1v.load(codes0 + (v1 & 0xFFFFFFu) * (u32)M);
            const double *t0 = s_r + (size_t)q0 * D, *t1 = s_r + (size_t)q1 * D;
            double d0 = 0.0, d1 = 0.0;
#pragma unroll 1
            for (int s0 = 0; s0 < M; s0 += Q_FR) {
                double p0[Q_FR][DSUB], p1[Q_FR][DSUB];
#pragma unroll
                for (int j = 0; j < Q_FR; j++) {
                    const u32 b0 = (c0v.wd[(s0 + j) >> 2] >> (8 * ((s0 + j) & 3))) & 0xffu, b1 = (c1v.wd[(s0 + j) >> 2] >> (8 * ((s0 + j) & 3))) & 0xffu;
                    const double *r0 = QP.pq + ((size_t)(s0 + j) * 256 + (size_t)b0) * DSUB;
                    const double *r1 = QP.pq + ((size_t)(s0 + j) * 256 + (size_t)b1) * DSUB;
#pragma unroll
                    for (int t = 0; t < DSUB; t++) {
                        p0[j][t] = r0[t];
                        p1[j][t] = r1[t];
                    }
                }
#pragma unroll
                for (int j = 0; j < Q_FR; j++) {
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int t = 0; t < DSUB; t++) {
                        const double f0 = t0[(s0 + j) * DSUB + t] - p0[j][t], f1 = t1[(s0 + j) * DSUB + t] - p1[j][t];
                        a0 += f0 * f0;
                        a1 += f1 * f1;
                    }
                    d0 = (s0 + j) == 0 ? a0 : d0 + a0;  // (0.0 + x == x: the first entry itself)
                    d1 = (s0 + j) == 0 ? a1 : d1 + a1;
                }
            }
            if (h0) {
                ckey[i0] = dkey(d0);
                if (v0 & 0x80000000u) atomicMax(s_max + q0, dkey(d0));  // evidence codes: at least K1 per query
            }
            if (h1) {
                ckey[i1] = dkey(d1);
                if (v1 & 0x80000000u) atomicMax(s_max + q1, dkey(d1));
            }
        };
        // (two candidates side by side measured no faster than one after the other: 0.78 against 0.73 ms per 7525 groups)
        sum2(ia, 0xFFFFFFFFu);
        sum2(ib, 0xFFFFFFFFu);
        sum2(ic, 0xFFFFFFFFu);
    }
    __syncthreads();
    Q_T(4);
    // T_i = the largest evidence sum; publish every candidate at or under it (and under a threshold another shard may have published)
    u64 Te[G];
#pragma unroll
    for (int i = 0; i < G; i++) {
        const bool has_T = live[i] && s_astar[i] >= 0;
        const u64 Tl = has_T ? s_max[i] : MMIDX_KEY_MAX;  // (a list of fewer than K1 codes: everything, no threshold)
        const u64 Tg = __hip_atomic_load(P.T + qid[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Te[i] = Tl < Tg ? Tl : Tg;
        if (has_T && tid == 0) atomicMin(P.T + qid[i], Tl);
    }
    // (an emitted candidate keeps its pool slot in the entry's top byte, 0xFF = not emitted)
#pragma unroll 1
    for (u32 idx = (u32)tid; idx < ncand; idx += NT) {
        int i = 0;
#pragma unroll
        for (int k = 1; k < G; k++) i += (idx >= first[k]) ? 1 : 0;
        const u64 te = i == 0 ? Te[0] : (i == 1 ? Te[1] : (i == 2 ? Te[2] : Te[3]));
        u32 *ce = cent + i * HKQ + (idx - (i == 0 ? first[0] : (i == 1 ? first[1] : (i == 2 ? first[2] : first[3]))));
        const u32 v = *ce & 0xFFFFFFu;
        const u32 sl = ckey[idx] <= te ? atomicAdd(s_emit + i, 1u) : 0xFFu;  // (< HKQ = 192 < 255)
        *ce = v | (sl << 24);
    }
    __syncthreads();
    if (tid < G && s_emit[tid]) s_base[tid] = atomicAdd(P.pool_cnt + qid[tid], s_emit[tid]);
    __syncthreads();
#pragma unroll 1
    for (u32 idx = (u32)tid; idx < ncand; idx += NT) {
        int i = 0;
#pragma unroll
        for (int k = 1; k < G; k++) i += (idx >= first[k]) ? 1 : 0;
        const u32 v = cent[i * HKQ + (idx - (i == 0 ? first[0] : (i == 1 ? first[1] : (i == 2 ? first[2] : first[3]))))];
        if ((v >> 24) == 0xFFu) continue;
        const int qq = i == 0 ? qid[0] : (i == 1 ? qid[1] : (i == 2 ? qid[2] : qid[3]));
        const u32 sl = s_base[i] + (v >> 24);
        if (sl < (u32)P.poolq) {
            P.pool_key[(size_t)qq * P.poolq + sl] = ckey[idx];
            P.pool_val[(size_t)qq * P.poolq + sl] = (u64)(v & 0xFFFFFFu);  // (probe rank 0)
        }
    }
    Q_T(5);
}

template <int M, int DSUB>
__global__ __launch_bounds__(256, Q_WPS) void k_scan_q(const QParams QP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // block -> group, in order.  (Dealing the groups to the XCDs in eight contiguous ranges -- so that the two to four blocks of one list
    // share an L2 -- was measured: 0.765 against 0.750 ms per 7525 groups, 3.52 against 3.47 ms per 35796; the kernel is not bound by
    // the code stream, and neighbouring blocks in one phase of the kernel on one XCD cost more than the L2 hits bring.)
    const int grp = (int)blockIdx.x;
    if (grp >= *QP.n_groups) return;
    const int4 gd = QP.gdesc[grp];
#ifndef Q_NO_GA2
    if (gd.z <= 2) q_scan_group<M, DSUB, 2>(QP, gd, smem);
    else
#endif
        q_scan_group<M, DSUB, 4>(QP, gd, smem);
}
