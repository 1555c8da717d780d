// mmidx_scan_mfma_cr.h -- K3mc: the matrix-core lower bound of pass B (mmidx_scan_mfma.h) with the CODES resident and the queries
// streaming, for lists that MANY groups of queries probe -- flat PQ first of all (PQ.computeKnnADC, PQ.java:290-322: every query
// scans every code; cfg2: 4096 queries = 64 groups per chunk of the list).
//
// K3m decodes a code tile once per (group of <= 64 queries, tile): four random 16-byte LDS gathers feed 16 matrix instructions, and
// the gathers' bank conflicts (three of four LDS cycles) co-limit the kernel at 0.3-0.4 of the fp16 peak.  With 64 groups on the
// same codes the decode is the same 64 times over.  Here a wave DECODES ITS FOUR TILES ONCE per item -- the B fragments of 64 codes
// (64 registers), gathered straight from the fp16 codebook in global memory (L2) -- and then walks the list's groups: a group's 64
// fp16 residual rows (16 KiB, XOR-swizzled units) and its 64 row records (2 KiB: threshold, bound constants, query, slot) arrive by
// LDS-DMA in the buffer pair that is not being read, its A fragments come from there by 16 conflict-free reads per lane, and 64
// matrix instructions follow.  No random gather in the loop.
//
// STATE (end of round 4): parity-green, OFF by default (option "mfma_cr").  cfg2 (4096 queries, 1 M codes, m = 8): 1.56 ms per batch
// against K3m's 1.21.  Timing builds: without survivors and DMA 0.90 ms (the 64 matrix instructions per wave and group are 45 % of a
// step: barrier skew of four-wave blocks, LDS latency of the fragment reads), + DMA 1.04, + survivors 1.56 (≈ 14 survivors per
// 64 x 256 block of pairs: most tiles take the slow path), and the stage-wise thresholds verify 2.6 x more codes than K3m's
// continuously tightened ones (verification 0.25 against 0.12 ms).  What it took to get the loop free of full vector-memory waits is in
// the comments below (explicit partial waits, bare barriers, LDS reads by hand, all register loads complete before the first DMA).
//   * k_cr_rows, a thread per pair slot: the row records for the thresholds as they stand (everything K3m's phase (a) computes per
//     row).  Thresholds therefore move BETWEEN launches, not inside one: the scan runs in three stages over disjoint pieces of every
//     list (1/8, 2/8, 5/8 of the 256-code pieces), k_ghist_tighten (K3m's phase (c): k + 1 survivors' upper bounds at or below a
//     bucket edge make that edge a threshold) and k_cr_rows run between them.
//   * residual rows, their norms and the launch's power-of-two scale: k_pair_resid16 / k_resid_scale of K3mk (mmidx_scan_mfma_kc.h).
//   * compares, survivor records, upper-bound histogram, k_mfma_verify, k_mfma_redo: K3m's.  Same certificate (D + 4 accumulation steps).
#pragma once
#include "mmidx_scan_mfma_kc.h"

#define MFC_NT 256
#define MFC_TPW 4                         // code tiles per wave
#define MFC_PIECE (MFC_TPW * 4 * 16)      // codes per item: 256
#define MFC_MAXG 256                      // groups of a list the kernel keeps the descriptors of (the host checks)
#define MFC_BUF 256                       // survivor records a wave stages between flushes

struct MfmaCrParams {
    MfmaKcParams K;          // (K.M.work is unused: the stages have their own cursors)
    const int2 *lgrp;        // per list: {first group in gdesc, number of groups}
    const MfmaRow *rows;     // [rows + 1] row records (k_cr_rows); the last one never survives
    long long nrows;
    int nlists, npiece;      // pieces of MFC_PIECE codes in the longest list
    int st_lo, st_hi;        // this stage takes the pieces p with p % 8 in [st_lo, st_hi)
    int per_list_rows;       // 1: R16 / rows are indexed by the pair's position WITHIN its list (flat PQ: every chunk sees the same queries in
                             // the same order, so one copy of the rows serves all chunks and stays in L2), 0: by pair slot
    u32 *cursor;             // [8] per-XCD item cursors of this stage (zeroed by k_mfma_prep)
};

// row records of all pair slots (K3m's phase (a) without the residuals: k_pair_resid16 left their norms in nrow[])
__global__ __launch_bounds__(256) void k_cr_rows(const MfmaKcParams K, MfmaRow *__restrict__ rows, long long n_flat) {
    const MfmaParams &P = K.M;
    const long long n = P.S.n_order ? (long long)*P.S.n_order : n_flat;
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot == 0) {  // the record behind the last one: "never survives" (the rows of a group past its pairs are staged from it)
        MfmaRow z;
        z.thr = __int_as_float(0x7F800000);
        z.cd = 0.f;
        z.kq = 0.f;
        z.cq = 0.f;
        z.q = 0;
        z.slot = 0;
        z.inv0 = 0.f;
        z.pad = 0;
        rows[n_flat] = z;
    }
    if (slot >= n) return;
    const int D = K.D;
    const double xmax = P.xmax;
    const double gam = (double)(D + 4) * 0x1p-23 * 1.01;
    const int er = K.scale[0];
    const bool scale_ok = K.scale[1] != 0;
    const double s2 = scale_ok ? ldexp(1.0, er + P.ep) : 1.0, inv_s2 = 1.0 / s2, inv_sr = scale_ok ? ldexp(1.0, -er) : 1.0, inv_sp = ldexp(1.0, -P.ep);
    const double sqrtD = sqrt((double)D) * 1.001;
    const int e = P.S.order[slot];
    const int q = e / P.S.w;
    const double nr = K.nrow[slot], nrm = sqrt(nr) * (1.0 + 1e-12);
    const double err = nrm * xmax * (4.02 * 0x1p-11 + 2.004 * gam) + xmax * xmax * (1.001 * gam + 0x1p-24) +
                       2.02 * sqrtD * 0x1p-14 * (xmax * inv_sr + nrm * inv_sp) + 2.0 * D * 0x1p-28 * inv_s2 +
                       0x1p-19 * (nr + xmax * xmax + 2.0 * nrm * xmax) + 1e-300;
    const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float th = __int_as_float(0x7F800000);
    double inv0 = 0.0;
    const bool fin = T < 0x7FF0000000000000ull;
    if (!fin || !scale_ok || !(err < 1e300)) {
        P.redo[q] = 1;
    } else if (!P.redo[q]) {
        th = mf_float_down((nr - keyd(T) - err) * (0.5 * s2));
        if (!(th < 3e38f)) {
            P.redo[q] = 1;
            th = __int_as_float(0x7F800000);
        }
        const u64 t0 = P.T0[q];
        if (t0 < 0x7FF0000000000000ull && t0 > 0) {
            const double iv = 256.0 / keyd(t0);
            if (iv < 1e300) inv0 = iv;
        }
    }
    MfmaRow rw;
    rw.thr = th;
    rw.cd = (float)(nr - err);
    rw.kq = 0.f;
    rw.cq = 0.f;
    if (inv0 > 0.0 && inv0 < 1e30 && inv0 * inv_s2 < 1e30) {
        const double mag = ((nr + err) + 2.02 * (nrm * xmax + 0.5 * xmax * xmax)) * inv0;
        rw.kq = (float)(-2.0 * inv_s2 * inv0);
        rw.cq = (float)((nr + err) * inv0 + 1e-3 + 0x1p-21 * mag);
        if (!(rw.kq != 0.f) || !(rw.cq < 3e38f)) rw.kq = 0.f;
    }
    rw.q = q;
    rw.slot = (int)slot;
    rw.inv0 = 0.f;
    rw.pad = 0;
    rows[slot] = rw;
}

// thresholds from the survivors' upper bounds so far, one wave per query (K3m's phase (c) over all queries)
__global__ __launch_bounds__(256) void k_ghist_tighten(const MfmaParams P, long long nq) {
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int lane = threadIdx.x & 63;
    const u64 t0 = P.T0[q];
    if (!(t0 < 0x7FF0000000000000ull && t0 > 0)) return;
    const double inv0q = 256.0 / keyd(t0);
    if (!(inv0q < 1e300)) return;
    const u32 *hq = P.ghist + (size_t)q * 256 + 4 * lane;
    const u32 h0 = hq[0], h1 = hq[1], h2 = hq[2], h3 = hq[3];
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
            atomicMin(P.S.T + q, dkey((double)(b + 1) / inv0q * (1.0 + 1e-12)));
        }
    }
}

// LDS reads the compiler does not see as memory operations.  Its waitcnt pass puts a FULL vector-memory wait in front of an LDS read it
// cannot prove disjoint from the LDS-DMA in flight -- and here DMA for the next groups is always in flight.  The reads below are
// ordered by hand: lds_wait*() names the registers, so that their users cannot move above it.
__device__ __forceinline__ mf_h8 lds_h8(const u32 addr, const int off) {
    mf_h8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}
__device__ __forceinline__ float lds_f32(const u32 addr, const int off) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
    return v;
}
__device__ __forceinline__ void lds_wait4(mf_h8 &a, mf_h8 &b, mf_h8 &c, mf_h8 &d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_wait16(float (&t)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]),
                   "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
}

// the wave's staged records -> its chunk of the global list + the queries' histograms of upper bounds.  A staged record carries its
// query (w = q << 9 | bucket, bucket 511 = none): the rows it came from may have left the LDS by the time it is flushed.
__device__ __forceinline__ void mfc_flush(const MfmaParams &P, MfmaChunk &ck, const uint4 *s_buf, const u32 nb, const int lane) {
    mf_chunk_room(P, ck, nb, lane);
    for (u32 i = (u32)lane; i < nb; i += 64) {
        const uint4 r = s_buf[i];
        const u32 q = r.w >> 9, bk = r.w & 0x1FFu;
        const u32 off = ck.base + ck.used + i;
        if (off < P.surv_cap) P.surv[off] = make_uint4(r.x, r.y, r.z, 0u);
        else P.redo[q] = 1;
        if (bk != 0x1FFu) atomicAdd(P.ghist + (size_t)q * 256 + bk, 1u);
    }
    ck.used += nb;
}

template <int DSUB>
__global__ __launch_bounds__(MFC_NT, 2) void k_scan_mfma_cr(const MfmaCrParams C) {
    static_assert(DSUB == 8 || DSUB == 16, "sub-quantizers of 8 or 16 dimensions");
    constexpr int NTL = 4, NJ = 4, D = 128, M = D / DSUB, NB = 32 / DSUB, TPW = MFC_TPW, NW = MFC_NT / 64;
    // three buffer sets (a group's 64 residual rows + 64 row records): the DMA runs two groups ahead of the matrix cores
    // (one array per set: the compiler tells LDS-DMA targets apart by variable, and only a handful of them)
    __shared__ __attribute__((aligned(1024))) unsigned char bs0[16384 + 2048];
    __shared__ __attribute__((aligned(1024))) unsigned char bs1[16384 + 2048];
    __shared__ __attribute__((aligned(1024))) unsigned char bs2[16384 + 2048];
    __shared__ uint4 s_bufs[NW * MFC_BUF];
    __shared__ int2 s_gd[MFC_MAXG];  // the list's groups: {first pair slot, pairs}
    __shared__ u32 s_misc[8];
    const MfmaParams &P = C.K.M;
    MfmaChunk ck{0u, 0u, 0u};
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint4 *s_buf = s_bufs + (size_t)wv * MFC_BUF;
    const int nsel = C.st_hi - C.st_lo;
    const int pg = (C.npiece + 7) >> 3;
    const int ipl = pg * nsel;  // virtual items per list in this stage
    const int nv = *P.n_groups > 0 ? C.nlists * ipl : 0;
    if (nv == 0) return;
    const int per = (nv + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int er = C.K.scale[0];
    const bool scale_ok = C.K.scale[1] != 0;
    const double s2 = scale_ok ? ldexp(1.0, er + P.ep) : 1.0;
    const float kinit = (float)(-0.5 * s2), kd = (float)(-2.0 / s2);
    const unsigned char *pqb = (const unsigned char *)P.pq16;

    for (;;) {
        __syncthreads();  // (the previous item's LDS reads are done)
        if (tid == 0) {
            s_misc[0] = atomicAdd(C.cursor + xcd, 1u);
            s_misc[1] = 0;
            s_misc[2] = 0;
        }
        __syncthreads();
        const int it = (int)s_misc[0];
        if (it >= per) break;
        const int v = xcd * per + it;
        if (v >= nv) break;
        const int cell = v / ipl, kk = v - cell * ipl;
        const int piece = (kk / nsel) * 8 + C.st_lo + (kk % nsel);
        if (piece >= C.npiece) continue;
        const int2 lg = C.lgrp[cell];
        const int g0 = lg.x, ng = lg.y;
        if (ng <= 0) continue;
        const long long beg = P.S.list_off[cell];
        const long long len = P.S.list_off[cell + 1] - beg;
        const long long c0 = (long long)piece * MFC_PIECE;
        if (c0 >= len) continue;
        const long long c1 = (c0 + MFC_PIECE < len) ? c0 + MFC_PIECE : len;
        const int ntiles = (int)((c1 - c0 + 15) >> 4);
        const u32 last = (u32)(c1 - c0 - 1);
        const unsigned char *cbase = (const unsigned char *)P.S.codes + (size_t)(beg + c0) * M + (u32)(NB * g);
        const float *xn = P.xn + beg + c0;

        // DMA of a group: its 64 residual rows (rows 16 i' .. of DMA instruction i' = 4 i + wv: unit u of row r in slot u ^ (r & 15))
        // and its row records (two kibibytes: waves 0 and 1)
        int rbase = 0;
        auto stage = [&](unsigned char *ab, const int first, const int np) {
            unsigned char *rowb = ab + 16384;
            const int fr = first - rbase;  // (row index of the group's first pair in R16 / rows)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ii = i * NW + wv;  // kibibyte of the buffer: rows 4 ii .. 4 ii + 3
                const int row = 4 * ii + (lane >> 4);
                const int u = (lane & 15) ^ (row & 15);
                const unsigned char *src = (const unsigned char *)(C.K.R16 + (size_t)(fr + (row < np ? row : np - 1)) * D) + u * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(ab + ii * 1024), 16, 0, 0);
            }
            {   // (every wave issues FIVE instructions per group -- waves 2 and 3 repeat the records' halves -- so that one partial
                //  vector-memory wait, 5 instructions per group still in flight, fits all waves)
                const int w2 = wv & 1;
                const int row = w2 * 32 + (lane >> 1);
                const unsigned char *src = (const unsigned char *)(C.rows + (row < np ? (size_t)(fr + row) : (size_t)C.nrows)) + (lane & 1) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(rowb + w2 * 1024), 16, 0, 0);
            }
        };
        // (the list's group descriptors go to LDS once: a global load in front of a group's DMA would be waited for right there -- the
        //  vector-memory counter is in order -- a whole L2 round trip per group on every wave)
        for (int i = tid; i < ng; i += MFC_NT) {
            const int4 t = P.gdesc[g0 + i];
            s_gd[i] = make_int2(t.y, t.z);
        }
        // ---- the wave's four tiles, decoded once: B fragments (unit NJ g + j of the code's concatenated centroids) and start values
        mf_h8 B[TPW][NJ];
        float ci[TPW];
#pragma unroll
        for (int ti = 0; ti < TPW; ti++) {
            u32 pp = (u32)(ti * NW + wv) * 16u + (u32)n;  // (tiles beyond the piece decode its last code: never compared)
            pp = pp < last ? pp : last;
            const unsigned char *cp = cbase + (size_t)pp * M;
            u32 c;
            if constexpr (NB == 4) c = *(const u32 *)cp;
            else c = (u32) * (const unsigned short *)cp;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const u32 byte = (c >> (8 * (j / (DSUB / 8)))) & 0xFFu;
                B[ti][j] = *(const mf_h8 *)(pqb + ((size_t)(NJ * g + j) * 256 + byte) * 16);
            }
            ci[ti] = xn[pp] * kinit;
        }

        // (every register load of the item is complete before the first DMA is issued: with DMA, stores and loads mixed in flight the
        //  compiler treats the vector-memory counter as out of order and would put a full wait in front of every use of B)
        __builtin_amdgcn_s_waitcnt(0x0070);
        const int4 gd0 = P.gdesc[g0];
        rbase = C.per_list_rows ? gd0.y : 0;
        stage(bs0, gd0.y, gd0.z);
        if (ng > 1) {
            const int4 t = P.gdesc[g0 + 1];
            stage(bs1, t.y, t.z);
        }
        u32 bufn = 0;
        // one group from a buffer pair
        auto group = [&](const unsigned char *ab, const int first, const int np) {
            const u32 ab_lds = (u32)(size_t)(__attribute__((address_space(3))) const unsigned char *)ab;
            const u32 a_addr = ab_lds + (u32)n * 256u;            // + rt 4096, unit (NJ g + j) ^ n
            const u32 r_addr = ab_lds + 16384u + (u32)(4 * g) * 32u;  // the lane's rows 4 g .. 4 g + 3 of a row tile (+ rt 512)
            // the lane's 16 row constants and their minimum (the pre-test); rows past the group's pairs carry +inf
            float th[16];
#pragma unroll
            for (int b = 0; b < 16; b++) th[b] = lds_f32(r_addr, (b >> 2) * 512 + (b & 3) * 32);
            const int ntl = (np + 15) >> 4;
            mf_h8 A[2][NTL];
#pragma unroll
            for (int rt = 0; rt < NTL; rt++) A[0][rt] = lds_h8(a_addr + (u32)(((NJ * g + 0) ^ n) << 4), rt * 4096);
            lds_wait16(th);
            float thrmin = th[0];
#pragma unroll
            for (int b = 1; b < 16; b++) thrmin = __builtin_fminf(thrmin, th[b]);
            // all four tiles at once, the k steps outermost: a step's four A fragments (16 registers) feed 16 matrix instructions; the
            // next step's fragments are requested before this step's matrix instructions
            mf_f4 acc[TPW][NTL];
#pragma unroll
            for (int ti = 0; ti < TPW; ti++) {
                const mf_f4 c4 = {ci[ti], ci[ti], ci[ti], ci[ti]};
#pragma unroll
                for (int rt = 0; rt < NTL; rt++) acc[ti][rt] = c4;
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                lds_wait4(A[j & 1][0], A[j & 1][1], A[j & 1][2], A[j & 1][3]);
                if (j + 1 < NJ) {
#pragma unroll
                    for (int rt = 0; rt < NTL; rt++) A[(j + 1) & 1][rt] = lds_h8(a_addr + (u32)(((NJ * g + j + 1) ^ n) << 4), rt * 4096);
                }
#pragma unroll
                for (int rt = 0; rt < NTL; rt++) {
                    if (rt < ntl) {  // (wave-uniform)
#pragma unroll
                        for (int ti = 0; ti < TPW; ti++) acc[ti][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][rt], B[ti][j], acc[ti][rt], 0, 0, 0);
                    }
                }
            }
            {
                constexpr int t2 = 0;
#pragma unroll
                for (int h = 0; h < TPW; h++) {
                    const int tt = (t2 + h) * NW + wv;
                    if (tt >= ntiles) continue;
                    float mxa = acc[h][0][0];
#pragma unroll
                    for (int b = 1; b + 1 < NTL * 4; b += 2) mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, acc[h][b >> 2][b & 3]), acc[h][(b + 1) >> 2][(b + 1) & 3]);
                    mxa = __builtin_fmaxf(mxa, acc[h][NTL - 1][3]);
                    if (!__builtin_amdgcn_ballot_w64(mxa >= thrmin)) continue;
                    // ---- survivors: K3m's lane-level path ----
                    const long long pos = c0 + (long long)tt * 16 + n;
                    u32 bits = 0;
#pragma unroll
                    for (int b = NTL * 4 - 1; b >= 0; b--)
                        asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(acc[h][b >> 2][b & 3]), "v"(th[b]) : "vcc");
                    if (pos >= c1) bits = 0;
                    u64 act = __builtin_amdgcn_ballot_w64(bits != 0);
                    while (act) {
                        if (bufn > (u32)(MFC_BUF - 64)) {  // (wave-uniform) room for one record per lane
                            mfc_flush(P, ck, s_buf, bufn, lane);
                            bufn = 0;
                        }
                        if (bits) {
                            const int b = __ffs((int)bits) - 1;
                            bits &= bits - 1u;
                            const u32 m0 = 0u - ((u32)b & 1u), m1 = 0u - (((u32)b >> 1) & 1u), m2 = 0u - (((u32)b >> 2) & 1u), m3 = 0u - (((u32)b >> 3) & 1u);
                            u32 v8[8], v4[4], v2[2];
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const u32 lo = (u32)__float_as_int(acc[h][(2 * j) >> 2][(2 * j) & 3]), hi = (u32)__float_as_int(acc[h][(2 * j + 1) >> 2][(2 * j + 1) & 3]);
                                v8[j] = (hi & m0) | (lo & ~m0);
                            }
#pragma unroll
                            for (int j = 0; j < 4; j++) v4[j] = (v8[2 * j + 1] & m1) | (v8[2 * j] & ~m1);
#pragma unroll
                            for (int j = 0; j < 2; j++) v2[j] = (v4[2 * j + 1] & m2) | (v4[2 * j] & ~m2);
                            const float a = __int_as_float((int)((v2[1] & m3) | (v2[0] & ~m3)));
                            const int qs = (b >> 2) * 16 + 4 * g + (b & 3);
                            MfmaRow rw;
                            {   // (the row record, by hand as well: two 16-byte reads)
                                uint4 r0, r1;
                                const u32 ra = ab_lds + 16384u + (u32)qs * 32u;
                                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(ra));
                                rw.thr = __int_as_float((int)r0.x);
                                rw.cd = __int_as_float((int)r0.y);
                                rw.kq = __int_as_float((int)r0.z);
                                rw.cq = __int_as_float((int)r0.w);
                                rw.q = (int)r1.x;
                            }
                            const float lbf = fmaf(a, kd, rw.cd);
                            const float xb = fmaf(a, rw.kq, rw.cq);
                            const u32 bk = (rw.kq != 0.f && xb < 255.f) ? (xb > 0.f ? (u32)(int)xb : 0u) : 0x1FFu;
                            s_buf[bufn + mf_mbcnt(act)] = make_uint4((u32)(first + qs), (u32)pos, (u32)__float_as_int(lbf), ((u32)rw.q << 9) | bk);
                        }
                        bufn += (u32)__popcll(act);
                        act = __builtin_amdgcn_ballot_w64(bits != 0);
                    }
                }
            }
        };
        // step gi: wait until at most the DMA of group gi + 1 is in flight (5 instructions; the counter is in order, and younger stores of
        // the survivor path only make the wait stricter), barrier -- group gi has landed, nobody reads the buffer of group gi - 1 any
        // more --, request group gi + 2 into that buffer, then the matrix work of group gi
#define MFC_STEP(BA, NA, GI)                                                                                  \
        {                                                                                                     \
            if ((GI) + 1 < ng) __builtin_amdgcn_s_waitcnt(0x0075); /* vmcnt(5) lgkmcnt(0) */                    \
            else __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) */                                  \
            __builtin_amdgcn_s_barrier(); /* (the bare barrier: __syncthreads() drains the vector-memory counter) */ \
            if ((GI) + 2 < ng) {                                                                \
                const int2 t = s_gd[(GI) + 2];                                                                \
                stage(NA, __builtin_amdgcn_readfirstlane(t.x), __builtin_amdgcn_readfirstlane(t.y));          \
            }                                                                                                 \
            const int2 tc = s_gd[GI];                                                                         \
            group(BA, __builtin_amdgcn_readfirstlane(tc.x), __builtin_amdgcn_readfirstlane(tc.y));            \
        }
        for (int gi = 0; gi < ng; gi += 3) {
            MFC_STEP(bs0, bs2, gi)
            if (gi + 1 >= ng) break;
            MFC_STEP(bs1, bs0, gi + 1)
            if (gi + 2 >= ng) break;
            MFC_STEP(bs2, bs1, gi + 2)
        }
#undef MFC_STEP
        if (bufn) mfc_flush(P, ck, s_buf, bufn, lane);
    }
    for (u32 i = ck.used + (u32)lane; i < ck.cap; i += 64)
        if (ck.base + i < P.surv_cap) P.surv[ck.base + i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
}
