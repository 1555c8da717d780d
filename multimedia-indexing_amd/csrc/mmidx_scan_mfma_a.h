// mmidx_scan_mfma_a.h -- K3ma: PASS A of the IVFADC search on the matrix cores, for batches that bring >= 8 queries to a nearest list
// (a shard of the 8-GPU configuration; one GPU at batches >= 65536).  The two sweeps are MODE 1 / MODE 2 of k_scan_mfma
// (mmidx_scan_mfma.h, block comment above a_scan_tiles); this file holds what stands around them:
//   k_a1_pair_count / k_a1_pair_scatter   the (query, probe 0) pairs sorted by cell (counting sort; k_pair_scan between them)
//   k_a1_select                           a pair's threshold from the K1-th largest of the accumulator values sweep 1 kept
//   k_a1_verify                           sweep 2's bitmap -> records -> exact fp64 distances in the reference's order -> the pools
// Reference: the probe-0 iteration of computeKnnIVFADC, J/datastructures/IVFPQ.java:414-447 (table :525-538, sum :435-438).
// Everything the bound cannot serve (lists shorter than K1, magnitudes beyond the fp16 scale, a full record list or pool) marks the
// QUERY in P.redo: k_mfma_redo hands its pair to K3f, which scans it exactly from T = +inf.
#pragma once
#include "mmidx_scan_mfma.h"

// cnt[c] += queries whose nearest cell is c and whose list is non-empty here (a shard holds some of the lists); cnt[C] = their number
// (branch-free index arithmetic: see the note at pair_keep() in mmidx_kernels.h)
__global__ void k_a1_pair_count(const int32_t *__restrict__ cells, int w, long long nq, const int64_t *__restrict__ list_off, int32_t *__restrict__ cnt, int C) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long qc = q < nq ? q : nq - 1;
    const int c = cells[(size_t)qc * w];
    const unsigned cu = c >= 0 ? (unsigned)c : 0u;
    const bool own = (q < nq) & (c >= 0) & (list_off[cu + 1] > list_off[cu]);
    if (own) atomicAdd(cnt + (size_t)cu, 1);
    const u64 mk = __builtin_amdgcn_ballot_w64(own);
    const int lane = (int)(threadIdx.x & 63);
    const int leader = mk ? __ffsll((long long)mk) - 1 : 0;
    if (mk && lane == leader) atomicAdd(cnt + C, (int)__popcll(mk));
}
__global__ void k_a1_pair_scatter(const int32_t *__restrict__ cells, int w, long long nq, const int64_t *__restrict__ list_off, const int32_t *__restrict__ start,
                                  int32_t *__restrict__ cursor, int32_t *__restrict__ order) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long qc = q < nq ? q : nq - 1;
    const int c = cells[(size_t)qc * w];
    const unsigned cu = c >= 0 ? (unsigned)c : 0u;
    const bool own = (q < nq) & (c >= 0) & (list_off[cu + 1] > list_off[cu]);
    if (own) {
        const int pos = start[cu] + atomicAdd(cursor + (size_t)cu, 1);
        order[pos] = (int32_t)(q * w);  // (query, probe rank 0)
    }
}

// float bits <-> keys whose unsigned order is the floats' order (-inf -> 0x007FFFFF: "no value")
__device__ __forceinline__ u32 a1_key(float f) {
    const u32 b = (u32)__float_as_int(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float a1_unkey(u32 k) { return __int_as_float((int)((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k)); }

// One wave per pair: the K1-th largest of the (up to NS x 256) accumulator values sweep 1 kept for it, to 2^-13 relative (a bisection
// on the key bits with ballots; the low ten bits stay zero, which can only lower the value).  K1 DISTINCT codes of the list have
// acc >= a*, and a code's distance is at most ||r||^2 + err - 2 acc / s^2 (k_scan_mfma's certificate): that bound at a* is a valid
// threshold.  Pairs with fewer than K1 values (short lists) or an unusable scale keep T = +inf and go to the redo.
template <int NS>
__global__ __launch_bounds__(256) void k_a1_select(const MfmaParams P) {
    const long long slot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    const long long npairs = *P.S.n_order;
    if (slot >= npairs) return;
    const int e = P.S.order[slot];
    const int q = e / P.S.w;
    const int cell = P.S.cells[e];
    const long long len = P.S.list_off[cell + 1] - P.S.list_off[cell];
    const long long pieces_l = (len + P.sub - 1) / P.sub;
    const int pieces = (int)(pieces_l < (long long)P.nsub ? pieces_l : (long long)P.nsub);
    const double2 rc = P.a_rowc[slot];
    if (!(rc.x == rc.x)) return;
    u32 key[4 * NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        if (j < pieces) {
            const float4 v = *(const float4 *)(P.a_cand + ((size_t)slot * P.nsub + j) * 256 + lane * 4);
            key[4 * j] = a1_key(v.x);
            key[4 * j + 1] = a1_key(v.y);
            key[4 * j + 2] = a1_key(v.z);
            key[4 * j + 3] = a1_key(v.w);
        } else {
            key[4 * j] = key[4 * j + 1] = key[4 * j + 2] = key[4 * j + 3] = 0u;
        }
    }
    u32 K = 0;
    for (int bit = 31; bit >= 10; bit--) {
        const u32 t = K | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 4 * NS; i++) cnt += (int)__popcll(__builtin_amdgcn_ballot_w64(key[i] >= t));
        if (cnt >= P.S.K1) K = t;  // (wave-uniform)
    }
    if (K <= 0x007FFFFFu) return;  // fewer than K1 values
    const float a = a1_unkey(K);
    if (!(a == a) || !(fabsf(a) < 3e38f)) return;
    const double ub = (rc.x + rc.y * (double)a) * (1.0 + 1e-12);
    if (!(ub >= 0.0) || !(ub < 1e300)) return;
    if (lane == 0) atomicMin(P.S.T + q, dkey(ub));
}

// ---- exact distances of what sweep 2 let through --------------------------------------------------------------------------------
// A block per item (list piece x group of <= 64 pairs).  Phase 1: the item's bitmap -- per tile pair and lane the packed compares of
// a_scan_tiles<MODE 2> -- becomes (row, position) records in LDS.  Phase 2, A1V_R records per thread and round: sub-quantizer by
// sub-quantizer the codebook slice pq[s] (ks x dsub doubles) and the rows' residual sub-vectors (c - q, IVFPQ.java:645; rotated rows
// from P.R) are staged in LDS, and every record adds its table entry sum_t (r[t] - pq[s][code_s][t])^2, t ascending from 0.0
// (IVFPQ.java:531-534), to its running sum in sub-quantizer order (:435-438): the bits of the fp64 table lookup.  A thread reads a
// random 64-byte entry from LDS instead of from L2 (one exact distance is m x dsub x 8 bytes of codebook: 1 KiB at 16 x 8, and
// ~110 of them per query), and a slice is loaded once per 2048 records.
#define A1V_NT 256
#define A1V_R 8
#define A1V_CAP 8192
struct A1VLds {
    size_t pq, r, rec, q, misc, total;
    __host__ __device__ A1VLds(int dsub) {
        size_t o = 0;
        pq = o; o += 256 * (size_t)dsub * 8;
        r = o; o += MF_QG * (size_t)dsub * 8;
        rec = o; o += (size_t)A1V_CAP * 4;
        q = o; o += MF_QG * 4;
        misc = o; o += 16;
        total = (o + 15) & ~(size_t)15;
    }
};
template <int M, int DSUB>
__global__ __launch_bounds__(A1V_NT) void k_a1_verify(const MfmaParams P) {
    constexpr int D = M * DSUB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const A1VLds L(DSUB);
    double *s_pq = (double *)(smem + L.pq);
    double *s_r = (double *)(smem + L.r);
    u32 *s_rec = (u32 *)(smem + L.rec);
    int *s_q = (int *)(smem + L.q);
    u32 *s_n = (u32 *)(smem + L.misc);
    const int tid = threadIdx.x;
    const int nv = *P.n_groups * P.nsub;
    const int ks = P.S.ks;
    u32 nver = 0;
    for (int v = blockIdx.x; v < nv; v += gridDim.x) {
        const int gi = v / P.nsub, isub = v - gi * P.nsub;
        const int4 gd = P.gdesc[gi];
        const int cell = gd.x, first = gd.y, np = gd.z;
        const long long beg = P.S.list_off[cell];
        const long long len = P.S.list_off[cell + 1] - beg;
        const long long c0 = (long long)isub * P.sub;
        if (c0 >= len) continue;  // (block-uniform)
        const long long c1 = (c0 + P.sub < len) ? c0 + P.sub : len;
        const int ntiles = (int)((c1 - c0 + 15) >> 4), npt = (ntiles + 1) >> 1;
        const int ntl = (np + 15) >> 4;
        __syncthreads();  // (the previous item's LDS is done with)
        if (tid < MF_QG) s_q[tid] = tid < np ? P.S.order[first + tid] / P.S.w : -1;
        if (tid == 0) *s_n = 0;
        __syncthreads();
        // ---- phase 1: set bits -> records (row << 24 | position in the piece) ----
        {
            const unsigned char *bm = P.a_bm + (size_t)v * P.a_bm_stride;
            const int nwords = npt * 64;
            for (int x = tid; x < nwords; x += A1V_NT) {
                const u32 wb = ntl <= 2 ? (u32)((const unsigned short *)bm)[x] : ((const u32 *)bm)[x];
                if (!wb) continue;
                const int ln = x & 63, pr = x >> 6, n = ln & 15, g = ln >> 4;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    u32 bits = ntl <= 2 ? ((wb >> (8 * h)) & 0xFFu) : ((wb >> (16 * h)) & 0xFFFFu);
                    const u32 pos = (u32)((2 * pr + h) * 16 + n);
                    while (bits) {
                        const int b = __ffs((int)bits) - 1;
                        bits &= bits - 1u;
                        const u32 row = (u32)((b >> 2) * 16 + 4 * g + (b & 3));
                        const u32 o = atomicAdd(s_n, 1u);
                        if (o < (u32)A1V_CAP) s_rec[o] = (row << 24) | pos;
                    }
                }
            }
        }
        __syncthreads();
        const u32 nrec = *s_n;
        if (nrec > (u32)A1V_CAP) {  // (block-uniform) a threshold far above what the list holds: the exact kernels' case
            if (tid < np) P.redo[s_q[tid]] = 1;
            continue;
        }
        if (tid == 0) nver += nrec;
        const unsigned char *codes = (const unsigned char *)P.S.codes + (size_t)(beg + c0) * M;
        // ---- phase 2 ----
        for (u32 r0 = 0; r0 < nrec; r0 += A1V_NT * A1V_R) {
            // (the code bytes as 64-bit words: byte s is picked by selects and a shift -- an array indexed by the loop counter would
            //  live in scratch memory, and unrolling the loop over s lets the compiler hoist every slice's loads)
            constexpr int NW = (M + 7) / 8;
            u32 rec[A1V_R];
            u64 cw[A1V_R][NW];
            double d[A1V_R];
#pragma unroll
            for (int j = 0; j < A1V_R; j++) {
                const u32 idx = r0 + (u32)(j * A1V_NT + tid);
                rec[j] = s_rec[idx < nrec ? idx : nrec - 1u];  // (loads on a clamped index; the emit below is predicated)
                CodeVec<M, unsigned char> cv;
                cv.load(codes + (size_t)(rec[j] & 0xFFFFFFu) * M);
#pragma unroll
                for (int i = 0; i < NW; i++) cw[j][i] = (u64)cv.wd[2 * i] | (2 * i + 1 < CodeVec<M, unsigned char>::WORDS ? (u64)cv.wd[2 * i + 1] << 32 : 0ull);
                d[j] = 0.0;
            }
            for (int s = 0; s < M; s++) {
                __syncthreads();  // (the slice of s - 1 has been consumed)
                {
                    const double2 *src = (const double2 *)(P.pq + (size_t)s * ks * DSUB);
                    for (int i = tid; i < (ks * DSUB) >> 1; i += A1V_NT) ((double2 *)s_pq)[i] = src[i];
                    for (int i = tid; i < np * DSUB; i += A1V_NT) {
                        const int row = i / DSUB, t = i - row * DSUB;
                        double rv;
                        if (P.R) rv = P.R[(size_t)(first + row) * D + s * DSUB + t];
                        else rv = P.S.coarse[(size_t)cell * D + s * DSUB + t] - P.S.Q[(size_t)s_q[row] * D + s * DSUB + t];
                        s_r[i] = rv;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < A1V_R; j++) {
                    u64 wsel = cw[j][0];
#pragma unroll
                    for (int i = 1; i < NW; i++) wsel = (s >> 3) == i ? cw[j][i] : wsel;
                    const u32 byte = (u32)(wsel >> (8 * (s & 7))) & 0xFFu;
                    const double *pe = s_pq + (size_t)byte * DSUB;
                    const double *re = s_r + (size_t)(rec[j] >> 24) * DSUB;
                    double e1 = 0.0;
#pragma unroll
                    for (int t = 0; t < DSUB; t++) {
                        const double df = re[t] - pe[t];
                        e1 += df * df;
                    }
                    d[j] = s == 0 ? e1 : d[j] + e1;  // (0.0 + e_0 = e_0: IVFPQ.java:435-438)
                }
            }
#pragma unroll
            for (int j = 0; j < A1V_R; j++) {
                const u32 idx = r0 + (u32)(j * A1V_NT + tid);
                if (idx >= nrec) continue;
                const int row = (int)(rec[j] >> 24);
                const int q = s_q[row];
                const u64 key = dkey(d[j]);
                const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (key <= T) {
                    const int e = P.S.order[first + row];
                    const u32 slotp = atomicAdd(P.S.pool_cnt + q, 1u);
                    if (slotp < (u32)P.S.poolq) {
                        P.S.pool_key[(size_t)q * P.S.poolq + slotp] = key;
                        P.S.pool_val[(size_t)q * P.S.poolq + slotp] = ((u64)(u32)(e - q * P.S.w) << 32) | (u64)((u32)c0 + (rec[j] & 0xFFFFFFu));
                    } else {
                        P.redo[q] = 1;
                    }
                }
            }
        }
    }
    if (tid == 0 && nver) {
        if (P.stat) {
            atomicAdd(P.stat, (unsigned long long)nver);       // (mmidx_stats::verified_codes)
            atomicAdd(P.stat + 9, (unsigned long long)nver);   // (mmidx_stats::mfma_survivors)
        }
    }
}
