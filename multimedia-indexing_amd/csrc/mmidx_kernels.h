// mmidx_kernels.h -- gfx950 (MI355X / CDNA4) device kernels of the PQ / IVFPQ search path.
//
// Everything here computes in IEEE binary64 in the reference's left-to-right order (build with
// -ffp-contract=off): the returned distances are bit-equal to the Java arithmetic of
//   J/datastructures/IVFPQ.java:408-450, :525-538, :547-601, :613-648 and
//   J/datastructures/PQ.java:290-322, :387-429          (J/ = src/main/java/gr/iti/mklab/visual/)
// Distances are non-negative sums of squares, so their binary64 bit patterns order exactly like
// the values; kernels compare / sort them as u64 keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define MMIDX_BLOCK 256
#define MMIDX_KEY_MAX 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ u64 dkey(double d) { return (u64)__double_as_longlong(d); }
__device__ __forceinline__ double keyd(u64 k) { return __longlong_as_double((long long)k); }

// The timing-experiment switches of rounds 1-3 (MMIDX_HIST_STOP / MMIDX_SCAN_STOP / MMIDX_SEL_STOP truncated builds, GRP_BIS,
// GRP_TOUCH, GRP_TIMING_*, GRP_HALF_STATS, SMIN_TIMING) were removed from the kernels in round 4: what they measured is in
// DESIGN_NOTEBOOK.md, the code in the history before commit "housekeeping: experiment switches".  What is left under #ifndef are
// tunables with measured defaults (GRP_EPOCH, GRP_QMAX, MMIDX_K3H_PAIR ...); K3m's MF_TIMING went the same way at the end of round 4.
// ------------------------------------------------------------------------------------------------
// Block-wide bitonic sort of (key, val) pairs held in LDS, ascending by (key, val).  n must be a
// power of two; every thread of the block must call it.
// ------------------------------------------------------------------------------------------------
template <typename V>
__device__ __forceinline__ void block_bitonic_sort(u64 *key, V *val, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
                int lo = 2 * t - (t & (stride - 1));  // index with the `stride` bit clear
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                u64 ka = key[lo], kb = key[hi];
                V va = val[lo], vb = val[hi];
                bool gt = (ka > kb) || (ka == kb && va > vb);
                if (gt == up) {
                    key[lo] = kb;
                    key[hi] = ka;
                    val[lo] = vb;
                    val[hi] = va;
                }
            }
        }
    }
    __syncthreads();
}

// wave64 helpers that stay in the VALU (DPP / readlane): inside K3h's scan loop the LDS pipe is saturated
// by the table gather, and every ds_bpermute-based __shfl would queue behind it
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 x) {
    u32 v = x;  // Hillis-Steele inside each row of 16 lanes, then the row totals (gfx9 row broadcasts)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ u32 wave_read_u32(u32 x, int l) { return (u32)__builtin_amdgcn_readlane((int)x, l); }
// minimum / maximum over the wave (same DPP ladder; lanes a step does not reach keep their own value), wave-uniform result
__device__ __forceinline__ u32 wave_min_u32(u32 x) {
    u32 v = x;
#define MMIDX_DPP_MIN(ctrl, rmask)                                                              \
    {                                                                                           \
        const u32 o = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false); \
        v = o < v ? o : v;                                                                      \
    }
    MMIDX_DPP_MIN(0x111, 0xf) MMIDX_DPP_MIN(0x112, 0xf) MMIDX_DPP_MIN(0x114, 0xf) MMIDX_DPP_MIN(0x118, 0xf)
    MMIDX_DPP_MIN(0x142, 0xa) MMIDX_DPP_MIN(0x143, 0xc)
#undef MMIDX_DPP_MIN
    return wave_read_u32(v, 63);
}
__device__ __forceinline__ u32 wave_max_u32(u32 x) { return ~wave_min_u32(~x); }
__device__ __forceinline__ double wave_read_f64(double x, int l) {
    const u64 b = (u64)__double_as_longlong(x);
    const u32 lo_ = wave_read_u32((u32)b, l), hi_ = wave_read_u32((u32)(b >> 32), l);
    return __longlong_as_double((long long)(((u64)hi_ << 32) | lo_));
}

__device__ __forceinline__ int pow2ceil(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ------------------------------------------------------------------------------------------------
// K1a: coarse distances.  dist[q][c] = sum_j (coarse[c][j] - Q[q][j])^2, j ascending
// (computeNearestCoarseIndices IVFPQ.java:579-588; the early-abandon at :584 only skips
// candidates the bounded queue would reject, so the full sum is equivalent).
// One thread per centroid (coalesced over the transposed table), QT queries per block; the
// query values are wave-uniform and travel through scalar loads.
// ------------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_dist(const double *__restrict__ coarseT,
                                                             const double *__restrict__ Q,
                                                             double *__restrict__ out, int C, int D,
                                                             int nq) {
    const int c = blockIdx.x * MMIDX_BLOCK + threadIdx.x;
    const int q0 = blockIdx.y * QT;
    const int cc = c < C ? c : C - 1;
    double acc[QT];
#pragma unroll
    for (int t = 0; t < QT; t++) acc[t] = 0.0;
    int j = 0;
    for (; j + 8 <= D; j += 8) {  // eight rows of the table in flight (one at a time, every step was a memory round trip: 1024-d, 430 us)
        double cj[8];
#pragma unroll
        for (int u = 0; u < 8; u++) cj[u] = coarseT[(size_t)(j + u) * C + cc];
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int t = 0; t < QT; t++) {
                const int q = (q0 + t < nq) ? q0 + t : nq - 1;
                const double df = cj[u] - Q[(size_t)q * D + j + u];
                acc[t] += df * df;
            }
        }
    }
    for (; j < D; j++) {
        const double cj = coarseT[(size_t)j * C + cc];
#pragma unroll
        for (int t = 0; t < QT; t++) {
            const int q = (q0 + t < nq) ? q0 + t : nq - 1;
            const double df = cj - Q[(size_t)q * D + j];
            acc[t] += df * df;
        }
    }
    if (c < C) {
#pragma unroll
        for (int t = 0; t < QT; t++)
            if (q0 + t < nq) out[(size_t)(q0 + t) * C + c] = acc[t];
    }
}

// lexicographic (key, idx) min across the block; result broadcast to every thread
__device__ __forceinline__ void block_min_pair(u64 &k, int &i, u64 *s_k, int *s_i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u64 ok = __shfl_xor(k, off);
        int oi = __shfl_xor(i, off);
        if (ok < k || (ok == k && oi < i)) {
            k = ok;
            i = oi;
        }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // protect s_k / s_i reuse
    if ((threadIdx.x & 63) == 0) {
        s_k[wave] = k;
        s_i[wave] = i;
    }
    __syncthreads();
    k = s_k[0];
    i = s_i[0];
    for (int wv = 1; wv < (int)(blockDim.x >> 6); wv++) {
        u64 ok = s_k[wv];
        int oi = s_i[wv];
        if (ok < k || (ok == k && oi < i)) {
            k = ok;
            i = oi;
        }
    }
}

// final step of K1b (one thread): bounded-queue tie rule + output order
__device__ __forceinline__ void coarse_emit(const double *row, int C, int w, int rounds, u64 *sel_k, int *sel_i,
                                            int32_t *out) {
    int n = rounds < w ? rounds : w;
    if (rounds == w + 1 && sel_k[w - 1] == sel_k[w]) {
        // tie straddles the boundary: closed-form replay of the bounded queue
        const u64 tau = sel_k[w - 1];
        int b = 0;
        while (b < w && sel_k[b] < tau) b++;
        int nonjunk = 0, p = 0;
        for (int c = 0; c < C; c++) {
            const u64 k = dkey(row[c]);
            if (k <= tau) {
                nonjunk++;
                if (k == tau) p++;
                if (nonjunk == w) break;
            }
        }
        const int e = b - (w - p);
        int rank = 0, pos = b;
        for (int c = 0; c < C && rank < p; c++) {
            if (dkey(row[c]) == tau) {
                if (rank >= e) sel_i[pos++] = c;  // kept ties, arrival order
                rank++;
            }
        }
    }
    // emit nearest first; inside a run of equal distances later arrival (larger index) first
    int a = 0;
    while (a < n) {
        int bnd = a;
        while (bnd + 1 < n && sel_k[bnd + 1] == sel_k[a]) bnd++;
        for (int t = a; t <= bnd; t++) out[t] = sel_i[bnd - (t - a)];
        a = bnd + 1;
    }
    for (int t = n; t < w; t++) out[t] = -1;
}

// ------------------------------------------------------------------------------------------------
// K1b: top-w selection with the bounded queue's semantics (IVFPQ.java:576-600; LingPipe queue,
// assumption A1).  One block per query.  w+1 rounds of lexicographic (dist, index) argmin give
// the sorted head; a tie straddling position w is resolved by the queue's closed form (see
// DESIGN.md "bounded queue"): with tau the w-th distance, p = ties among the first w arrivals
// with d <= tau, e = (#d < tau) - (w - p); kept ties are those with tie rank e..p-1 in arrival
// (= index) order.  Output nearest first; equal distances later-arrival first.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select(const double *__restrict__ dist, int C,
                                                               int w, int32_t *__restrict__ cells) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *sel_k = (u64 *)smem;                   // [w+1]
    int *sel_i = (int *)(sel_k + (w + 1));      // [w+1]
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    const int q = blockIdx.x;
    const double *row = dist + (size_t)q * C;
    const int rounds = (w + 1 < C) ? w + 1 : C;
    u64 lastk = 0;
    int lasti = -1;
    for (int r = 0; r < rounds; r++) {
        u64 bk = MMIDX_KEY_MAX;
        int bi = 0x7fffffff;
        for (int c = threadIdx.x; c < C; c += MMIDX_BLOCK) {
            const u64 k = dkey(row[c]);
            const bool elig = (k > lastk) || (k == lastk && c > lasti);
            if (elig && (k < bk || (k == bk && c < bi))) {
                bk = k;
                bi = c;
            }
        }
        block_min_pair(bk, bi, s_k, s_i);
        lastk = bk;
        lasti = bi;
        if (threadIdx.x == 0) {
            sel_k[r] = bk;
            sel_i[r] = bi;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) coarse_emit(row, C, w, rounds, sel_k, sel_i, cells + (size_t)q * w);
}

// Register-resident variant of K1b for C <= 256 * PER: every thread keeps its PER distances in
// VGPRs (coalesced load, once), so a selection round is PER compares + one block reduction instead
// of a sweep over global memory.  Same result, including the tie rule (the rare straddling-tie
// replay re-reads the row from memory).
template <int PER>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select_reg(const double *__restrict__ dist, int C, int w,
                                                                   int32_t *__restrict__ cells) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *sel_k = (u64 *)smem;                   // [w+1]
    int *sel_i = (int *)(sel_k + (w + 1));      // [w+1]
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    const int q = blockIdx.x, tid = threadIdx.x;
    const double *row = dist + (size_t)q * C;
    u64 key[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = tid + i * MMIDX_BLOCK;
        key[i] = (c < C) ? dkey(row[c]) : MMIDX_KEY_MAX;
    }
    const int rounds = (w + 1 < C) ? w + 1 : C;
    for (int r = 0; r < rounds; r++) {
        u64 bk = MMIDX_KEY_MAX;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int c = tid + i * MMIDX_BLOCK;
            if (c < C && key[i] < bk) {  // c ascending in i: strict '<' keeps the lowest index among equals
                bk = key[i];
                bi = c;
            }
        }
        // a selected entry is retired by setting its key to KEY_MAX; a genuine KEY_MAX distance
        // (NaN with all mantissa bits) cannot be told apart and is never selected
        block_min_pair(bk, bi, s_k, s_i);
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (tid + i * MMIDX_BLOCK == bi) key[i] = MMIDX_KEY_MAX;
        if (tid == 0) {
            sel_k[r] = bk;
            sel_i[r] = bi;
        }
    }
    __syncthreads();
    if (tid == 0) coarse_emit(row, C, w, rounds, sel_k, sel_i, cells + (size_t)q * w);
}

// Fast variant of K1b (C >= 256, w + 1 <= 256): instead of w+1 block-wide argmin rounds, bound the
// (w+1)-th smallest distance from above by the (w+1)-th smallest of the 256 per-thread minima
// (those are w+1 distinct elements, so at least w+1 elements are <= tau), gather every element
// <= tau (a few dozen for w = 32, C = 8192) and sort just those.  Same output as the round-based
// kernels -- the sorted head by (distance, index) -- so the tie rule in coarse_emit is unchanged.
// If the gather overflows (massive ties) the round-based selection runs on the registers instead.
#ifndef MMIDX_CSEL_CAP
#define MMIDX_CSEL_CAP 1024
#endif
#ifndef MMIDX_CAND_CHUNK
#define MMIDX_CAND_CHUNK 24
#endif
#ifndef MMIDX_LUT_PAIRS
#define MMIDX_LUT_PAIRS 1  // table build: two adjacent entries per thread (16-byte loads)
#endif
#ifndef MMIDX_TERM_PAD
#define MMIDX_TERM_PAD 1  // row stride D + 1 doubles: the per-candidate sequential sums read LDS conflict-free
#endif
template <int PER>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select_fast(const double *__restrict__ dist, int C, int w,
                                                                    int32_t *__restrict__ cells) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *ckey = (u64 *)smem;                       // [CSEL_CAP] candidates / thread minima
    u32 *cidx = (u32 *)(ckey + MMIDX_CSEL_CAP);    // [CSEL_CAP]
    u64 *sel_k = (u64 *)(cidx + MMIDX_CSEL_CAP);   // [w+1]
    int *sel_i = (int *)(sel_k + (w + 1));         // [w+1]
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    __shared__ u32 s_n4[4];  // statics total 64 B: keeps the dynamic LDS base 16-byte aligned
    u32 &s_n = s_n4[0];
    const int q = blockIdx.x, tid = threadIdx.x;
    const double *row = dist + (size_t)q * C;
    u64 key[PER];
    u64 lk = MMIDX_KEY_MAX;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = tid + i * MMIDX_BLOCK;
        key[i] = (c < C) ? dkey(row[c]) : MMIDX_KEY_MAX;
        lk = key[i] < lk ? key[i] : lk;
    }
    const int R = w + 1;  // host guarantees R <= 256 <= C
    ckey[tid] = lk;
    cidx[tid] = (u32)tid;
    if (tid == 0) s_n = 0;
    block_bitonic_sort<u32>(ckey, cidx, MMIDX_BLOCK);
    const u64 tau = ckey[R - 1];
    __syncthreads();
    // gather all elements <= tau
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const bool pass = key[i] <= tau && (tid + i * MMIDX_BLOCK) < C;
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(&s_n, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < MMIDX_CSEL_CAP) {
                    ckey[slot] = key[i];
                    cidx[slot] = (u32)(tid + i * MMIDX_BLOCK);
                }
            }
        }
    }
    __syncthreads();
    const int n = (int)s_n;
    if (n <= MMIDX_CSEL_CAP) {
        const int Pn = pow2ceil(n < 2 ? 2 : n);
        for (int i = n + tid; i < Pn; i += MMIDX_BLOCK) {
            ckey[i] = MMIDX_KEY_MAX;
            cidx[i] = 0xFFFFFFFFu;
        }
        block_bitonic_sort<u32>(ckey, cidx, Pn);
        for (int i = tid; i < R; i += MMIDX_BLOCK) {
            sel_k[i] = ckey[i];
            sel_i[i] = (int)cidx[i];
        }
    } else {
        for (int r = 0; r < R; r++) {  // overflow: round-based selection (as k_coarse_select_reg)
            u64 bk = MMIDX_KEY_MAX;
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                const int c = tid + i * MMIDX_BLOCK;
                if (c < C && key[i] < bk) {
                    bk = key[i];
                    bi = c;
                }
            }
            block_min_pair(bk, bi, s_k, s_i);
#pragma unroll
            for (int i = 0; i < PER; i++)
                if (tid + i * MMIDX_BLOCK == bi) key[i] = MMIDX_KEY_MAX;
            if (tid == 0) {
                sel_k[r] = bk;
                sel_i[r] = bi;
            }
        }
    }
    __syncthreads();
    if (tid == 0) coarse_emit(row, C, w, R, sel_k, sel_i, cells + (size_t)q * w);
}

// ------------------------------------------------------------------------------------------------
// K1 (certified approximate path): the exact coarse stage costs C*D fp64 (sub, mul, add) triples
// per query although only w+1 of the C distances matter.  Here the C*D work is one fp32 FMA each
// (K1c, dot products q.c), and fp64 is spent only on the few centroids that can still be among the
// w+1 nearest given a rigorous error bound (K1d):
//
//   d~(c) = |c|^2 + |q|^2 - 2 S(c),  S = fp32 FMA chain over fl32(q), fl32(c)
//   |d~ - d| <= eps(c) = 2 (D + 3) 2^-24 (1.01) |q| |c| + 1e-12 (|c|^2 + |q|^2)
//       (input rounding 2^-24 each, D fused multiply-adds; Cauchy-Schwarz on sum |q_j c_j|)
//   tau = some upper bound on the (w+1)-th smallest of d~ + eps  (thread-minima trick of K1b-fast)
//   candidates = { c : d~(c) - eps(c) <= tau }: every centroid outside has d > tau >= the (w+1)-th
//   smallest exact distance, so it is neither selected nor tied with a selected one.
//   The candidates get the exact sequential fp64 distance (the reference's arithmetic), are sorted
//   by (distance, index), and the bounded-queue rule is applied to them.
// If the candidate buffer overflows, the block computes the exact row and runs the round-based
// selection (same code as K1b).
// ------------------------------------------------------------------------------------------------
// K1c: S[q][c] = sum_j Q32[q][j] * CT32[j][c] on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: an
// exact k-ordered fp32 FMA chain, so the error bound above applies as written).  Block = 4 waves,
// tile 64 queries x 256 centroids, K in steps of 32 through LDS; wave w owns 16 query rows and all
// 16 column tiles.  A fragment: A[i = lane & 15][k = lane >> 4], B fragment: B[k = lane >> 4][j = lane & 15];
// C/D (f32): col = lane & 15, row = 4 * (lane >> 4) + reg.
#define DOT_BM 64
#define DOT_BN 256
#define DOT_BK 32
#define DOT_LDA 34  // padded A row stride (floats): (2*row + k) mod 32 is conflict-free for the fragment read
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_dot32(const float *__restrict__ CT32, const float *__restrict__ Q32,
                                                              const double *__restrict__ cn, const double *__restrict__ qn,
                                                              float *__restrict__ S, int C, int D, int nq) {
    __shared__ float As[DOT_BM * DOT_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[DOT_BK * DOT_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * DOT_BN, q0 = blockIdx.y * DOT_BM;
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[2], rb[8];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; u++) {  // A: 64 rows x 32 k = 512 float4
            const int p = tid + u * 256, r = p >> 3, c4 = (p & 7) * 4;
            const int q = q0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < nq) {
                const float *src = Q32 + (size_t)q * D + k0 + c4;
                if (k0 + c4 + 3 < D) {
                    v = *(const float4 *)src;
                } else {
                    float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int e = 0; e < 4; e++)
                        if (k0 + c4 + e < D) tmp[e] = src[e];
                    v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {  // B: 32 k x 256 c = 2048 float4
            const int p = tid + u * 256, r = p >> 6, c4 = (p & 63) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + r < D) {
                const float *src = CT32 + (size_t)(k0 + r) * C + c0 + c4;
                if (c0 + c4 + 3 < C) {
                    v = *(const float4 *)src;
                } else {
                    float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int e = 0; e < 4; e++)
                        if (c0 + c4 + e < C) tmp[e] = src[e];
                    v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            rb[u] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int p = tid + u * 256, r = p >> 3, c4 = (p & 7) * 4;
            float *dst = As + r * DOT_LDA + c4;
            dst[0] = ra[u].x;
            dst[1] = ra[u].y;
            dst[2] = ra[u].z;
            dst[3] = ra[u].w;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int p = tid + u * 256, r = p >> 6, c4 = (p & 63) * 4;
            *(float4 *)(Bs + r * DOT_BN + c4) = rb[u];
        }
    };
    load_tiles(0);
    for (int k0 = 0; k0 < D; k0 += DOT_BK) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + DOT_BK < D) load_tiles(k0 + DOT_BK);
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < DOT_BK / 4; kk++) {
            const float a = As[(wave * 16 + fr) * DOT_LDA + kk * 4 + fk];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const float b = Bs[(kk * 4 + fk) * DOT_BN + t * 16 + fr];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int c = c0 + t * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = q0 + wave * 16 + 4 * (lane >> 4) + r;
            // epilogue: d~ = |c|^2 + |q|^2 - 2 S, stored as fp32 (its rounding is part of the bound)
            if (q < nq && c < C) S[(size_t)q * C + c] = (float)((cn[c] + qn[q]) - 2.0 * (double)acc[t][r]);
        }
    }
}

// fl32 copy of the queries and their fp64 squared norms
__global__ __launch_bounds__(MMIDX_BLOCK) void k_query_prep(const double *__restrict__ Q, float *__restrict__ Q32,
                                                            double *__restrict__ qn, int D, long long nq) {
    const long long q = (long long)blockIdx.x * (MMIDX_BLOCK / 64) + (threadIdx.x >> 6);  // one wave per query
    if (q >= nq) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int j = lane; j < D; j += 64) {
        const double v = Q[(size_t)q * D + j];
        Q32[(size_t)q * D + j] = (float)v;
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) qn[q] = s * (1.0 + 1e-12);  // only ever used inside the error bound: round up
}

struct ApproxSel {
    const float *S;          // [nq][C] d~ = |c|^2 + |q|^2 - 2 <q, c>  (fp32, from K1c)
    const double *qn;        // [nq] |q|^2 (rounded up)
    const double *Q;         // [nq][D]
    const double *coarse;    // [C][D]
    const double *coarseT;   // [D][C]
    double *row_scratch;     // [nq][C] exact rows, written only by the overflow fallback
    int32_t *cells;          // [nq][w]
    double *cdsel;           // [nq][w] exact distance of every selected cell (probe-bound input)
    double cnorm_max, cn_max;  // max |c| and max |c|^2 (rounded up): one error bound per query
    int C, D, w;
    // K1f (group minima front end)
    const float2 *gpair;     // [nq][G] (smallest, runner-up | position) of d~ over each group of 8 centroids (K1e)
    int G, Dp;               // groups per query (Cp / 8), k padded to a multiple of 32
    // K1f split in two (k_coarse_front -> k_coarse_select_list): the certified candidates of every query
    u32 *clist;              // [nq][MMIDX_CLIST]: entry 0 = number of candidates (MMIDX_CSEL_CAP + 1: the exact-row path), then the candidates
    u32 *defer;              // k_coarse_front_sel: [0] number of queries it left to k_coarse_select_defer, [1 ..] those queries
    int nq;
    int cand_chunk;          // candidates per round of the exact stage (<= MMIDX_CAND_CHUNK; the launch's LDS holds that many rows of terms)
};
#define MMIDX_CLIST 256

// Second half of the certified coarse selection, shared by the two front ends (K1d over the full d~ row,
// K1f over group minima): the n candidates in cidx[] get the exact sequential fp64 distance, are ordered
// by (distance, index), and the bounded-queue rule picks the w cells.  Every thread of the block calls it
// after a barrier that made cidx[0..n) and n visible.
template <int PER>
__device__ __forceinline__ void coarse_select_finish(const ApproxSel &A, const int q, const int n, u64 *ckey, u32 *cidx,
                                                     u64 *sel_k, int *sel_i, u64 *s_k, int *s_i) {
    const int tid = threadIdx.x, C = A.C, D = A.D, w = A.w;
    const int R = w + 1;
    const double *qv = A.Q + (size_t)q * D;
    // candidates whose terms are staged in LDS at a time: MMIDX_CAND_CHUNK where they fit the block's 64 KiB, fewer for long
    // vectors (D = 1024, the reference's flagship shape: 6), chosen by the host with the launch's LDS size
    const int CH = A.cand_chunk;
    if (n <= MMIDX_CSEL_CAP) {
        // exact fp64 distance of every candidate in the reference's order (IVFPQ.java:583).  The
        // per-dimension terms (c_j - q_j)^2 are independent and are computed by all threads with
        // coalesced loads; only their summation is order-sensitive and runs sequentially, one lane
        // per candidate, over the terms staged in LDS.
        double *terms = (double *)(sel_i + ((A.w + 2) & ~1));  // [CH][D + pad]
        // the ordered sum of one chunk: lane c adds the D staged terms of candidate base + c, j ascending (between barriers)
        auto sum_chunk = [&](const int base, const int nc_) {
            __syncthreads();
            if (tid < nc_) {
                const double *tt = terms + (size_t)tid * (D + MMIDX_TERM_PAD);
                double acc = 0.0;
                int j = 0;
                for (; j + 16 <= D; j += 16) {  // 16 LDS reads in flight, then the ordered adds
                    double b[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) b[u] = tt[j + u];
#pragma unroll
                    for (int u = 0; u < 16; u++) acc += b[u];
                }
                for (; j < D; j++) acc += tt[j];
                ckey[base + tid] = dkey(acc);
            }
            __syncthreads();
        };
        const int hD0 = D >> 1;
        constexpr int PUP = 6;  // candidate rows in flight per thread in the pipelined form
        const bool pipelined = (D & 1) == 0 && (hD0 & (hD0 - 1)) == 0 && hD0 <= MMIDX_BLOCK && CH <= PUP * (MMIDX_BLOCK / hD0);
        if (CH < MMIDX_CAND_CHUNK && (D & 1) == 0 && n > 4 * CH) {
            // Long vectors and many candidates (D = 1024, w = 64 -- YFCC100MExample.java:85, Example.java:96: only six rows of
            // terms fit the LDS, and six lanes summing 1024 terms each left the other 250 idle, 14 rounds of ~8 us per query):
            // a LANE per candidate instead.  Every lane streams its own centroid row from L2 in dimension order and adds
            // (c_j - q_j)^2 as it goes -- the same operations in the same order (IVFPQ.java:583), 256 candidates at a time,
            // eight 16-byte loads in flight per lane; the query sits in LDS (broadcast reads).  The rows are the traffic that
            // is left: (w + ~20) x 8 KiB per query.  (Loads run on a clamped index, only the store is predicated: see the
            // note at pair_keep().)  Few candidates (w = 2) keep the staged form below: a handful of lanes streaming whole rows
            // is a chain of cache misses.
            for (int j = tid; j < D; j += MMIDX_BLOCK) terms[j] = qv[j];
            __syncthreads();
            for (int base = 0; base < n; base += MMIDX_BLOCK) {
                const int ci = base + tid;
                const int cc = ci < n ? ci : n - 1;
                const double *row = A.coarse + (size_t)cidx[cc] * (u32)D;
                double acc = 0.0;
                int j = 0;
                for (; j + 16 <= D; j += 16) {
                    double2 cv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) cv[u] = *(const double2 *)(row + j + 2 * u);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const double2 qq = *(const double2 *)(terms + j + 2 * u);
                        const double d0 = cv[u].x - qq.x, d1 = cv[u].y - qq.y;
                        acc += d0 * d0;
                        acc += d1 * d1;
                    }
                }
                for (; j < D; j++) {
                    const double df = row[j] - terms[j];
                    acc += df * df;
                }
                if (ci < n) ckey[ci] = dkey(acc);
            }
            __syncthreads();
        } else if (pipelined) {
            // D/2 a power of two and a whole chunk in one round of loads (D = 128: 4 candidates per 256 threads, 6 rounds):
            // a thread owns ONE pair of coordinates (j, j + 1) -- no division per element, the query pair is loaded once --
            // and the rows of the NEXT chunk are requested before the ordered sum of the current one, so that their
            // latency (44 % of this kernel's L2 requests miss: 8 MB of fp64 centroids, 4 MB of L2 per XCD) runs under it.
            const int sh = __ffs(hD0) - 1;
            const int j = (tid & (hD0 - 1)) * 2;
            const int ci0 = tid >> sh, cstep = MMIDX_BLOCK >> sh;
            const double2 qj = *(const double2 *)(qv + j);
            double2 cv[PUP];
            auto request = [&](const int base) {
                const int nc_ = (n - base < CH) ? n - base : CH;
#pragma unroll
                for (int u = 0; u < PUP; u++) {
                    const int ci = ci0 + u * cstep;
                    cv[u] = make_double2(0.0, 0.0);
                    if (ci < nc_) cv[u] = *(const double2 *)(A.coarse + (size_t)cidx[base + ci] * (u32)D + j);
                }
            };
            if (n > 0) request(0);
            for (int base = 0; base < n; base += CH) {
                const int nc_ = (n - base < CH) ? n - base : CH;
#pragma unroll
                for (int u = 0; u < PUP; u++) {
                    const int ci = ci0 + u * cstep;
                    if (ci < nc_) {
                        const double d0 = cv[u].x - qj.x, d1 = cv[u].y - qj.y;
                        double *tt = terms + (size_t)ci * (D + MMIDX_TERM_PAD) + j;
                        tt[0] = d0 * d0;
                        tt[1] = d1 * d1;
                    }
                }
                if (base + CH < n) request(base + CH);
                sum_chunk(base, nc_);
            }
        } else
        for (int base = 0; base < n; base += CH) {
            const int nc_ = (n - base < CH) ? n - base : CH;
            const int hD = D >> 1;
            if ((D & 1) == 0 && (hD & (hD - 1)) == 0 && hD <= MMIDX_BLOCK) {
                // D/2 a power of two that divides the block (the usual case, D = 128): a thread owns ONE pair of
                // coordinates (j, j + 1) and walks the candidates -- no division per element, the query pair is
                // loaded once, the row offset is one 32 x 32 -> 64 bit multiply.  Same operations on the same
                // operands as the general branch below: (c_j - q_j)^2 per coordinate.
                constexpr int PU = 6;  // candidate rows in flight per thread (24 = one chunk at D = 128)
                const int sh = __ffs(hD) - 1;
                const int j = (tid & (hD - 1)) * 2;
                const int ci0 = tid >> sh, cstep = MMIDX_BLOCK >> sh;
                const double2 qj = *(const double2 *)(qv + j);
                for (int cb = 0; cb < nc_; cb += PU * cstep) {
                    double2 cv[PU];
#pragma unroll
                    for (int u = 0; u < PU; u++) {
                        const int ci = cb + ci0 + u * cstep;
                        cv[u] = make_double2(0.0, 0.0);
                        if (ci < nc_) cv[u] = *(const double2 *)(A.coarse + (size_t)cidx[base + ci] * (u32)D + j);
                    }
#pragma unroll
                    for (int u = 0; u < PU; u++) {
                        const int ci = cb + ci0 + u * cstep;
                        if (ci < nc_) {
                            const double d0 = cv[u].x - qj.x, d1 = cv[u].y - qj.y;
                            double *tt = terms + (size_t)ci * (D + MMIDX_TERM_PAD) + j;
                            tt[0] = d0 * d0;
                            tt[1] = d1 * d1;
                        }
                    }
                }
            } else if ((D & 1) == 0) {
                // 16-byte loads: element pairs (j, j + 1) of a candidate row; 8 pairs in flight per thread
                for (int e0 = 0; e0 < nc_ * hD; e0 += MMIDX_BLOCK * 8) {
                    double2 cv[8], qq[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u * MMIDX_BLOCK + tid;
                        cv[u] = make_double2(0.0, 0.0);
                        qq[u] = cv[u];
                        if (e < nc_ * hD) {
                            const int ci = e / hD, j = (e - ci * hD) * 2;
                            cv[u] = *(const double2 *)(A.coarse + (size_t)cidx[base + ci] * D + j);
                            qq[u] = *(const double2 *)(qv + j);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u * MMIDX_BLOCK + tid;
                        if (e < nc_ * hD) {
                            const int ci = e / hD, j = (e - ci * hD) * 2;
                            const double d0 = cv[u].x - qq[u].x, d1 = cv[u].y - qq[u].y;
                            double *tt = terms + (size_t)ci * (D + MMIDX_TERM_PAD) + j;
                            tt[0] = d0 * d0;
                            tt[1] = d1 * d1;
                        }
                    }
                }
            } else
                for (int e0 = 0; e0 < nc_ * D; e0 += MMIDX_BLOCK * 8) {  // 8 independent loads in flight per thread
                double cv[8], qq[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u * MMIDX_BLOCK + tid;
                    cv[u] = 0.0;
                    qq[u] = 0.0;
                    if (e < nc_ * D) {
                        const int ci = e / D, j = e - ci * D;
                        cv[u] = A.coarse[(size_t)cidx[base + ci] * D + j];
                        qq[u] = qv[j];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u * MMIDX_BLOCK + tid;
                    if (e < nc_ * D) {
                        const double df = cv[u] - qq[u];
                        const int ci = e / D;
                        terms[e + ci * MMIDX_TERM_PAD] = df * df;
                    }
                }
            }
            sum_chunk(base, nc_);
        }
        if (n <= 64) {
            // the usual case: one wave ranks the candidates by (distance, index) in registers (readlane,
            // no LDS round trips, no barriers) and writes them back in order
            if (tid < 64) {
                const bool have = tid < n;
                const u64 mk = have ? ckey[tid] : MMIDX_KEY_MAX;
                const u32 mv = have ? cidx[tid] : 0xFFFFFFFFu;
                // (equal distances are rare: the first pass compares the keys only and counts the equal ones, itself
                //  included; the index / position tie-breaks are read only if some lane saw a tie)
                int rank = 0, eqc = 0;
                for (int j = 0; j < n; j++) {  // (entries past n would never precede a real one)
                    const u32 olo = wave_read_u32((u32)mk, j), ohi = wave_read_u32((u32)(mk >> 32), j);
                    const u64 ok = ((u64)ohi << 32) | olo;
                    rank += ok < mk;
                    eqc += ok == mk;
                }
                if (__builtin_amdgcn_ballot_w64(have && eqc > 1)) {  // wave-uniform
                    rank = 0;
                    for (int j = 0; j < n; j++) {
                        const u32 olo = wave_read_u32((u32)mk, j), ohi = wave_read_u32((u32)(mk >> 32), j), ov = wave_read_u32(mv, j);
                        const u64 ok = ((u64)ohi << 32) | olo;
                        rank += (ok < mk) || (ok == mk && (ov < mv || (ov == mv && j < tid)));
                    }
                }
                // (every lane has read its entry before any lane writes: one wave, program order)
                ckey[rank] = mk;
                cidx[rank] = mv;
                if (rank < R) {
                    sel_k[rank] = mk;
                    sel_i[rank] = (int)mv;
                }
            }
        } else {
            const int Pn = pow2ceil(n < 2 ? 2 : n);
            for (int i = n + tid; i < Pn; i += MMIDX_BLOCK) {
                ckey[i] = MMIDX_KEY_MAX;
                cidx[i] = 0xFFFFFFFFu;
            }
            block_bitonic_sort<u32>(ckey, cidx, Pn);
            for (int i = tid; i < R; i += MMIDX_BLOCK) {
                sel_k[i] = ckey[i];
                sel_i[i] = (int)cidx[i];
            }
        }
        __syncthreads();
        // no two equal keys among the w+1 best (the usual case): the answer is the sorted prefix
        bool plain = false;
        if (w < 64 && tid < 64) {
            const bool eq = tid < w && sel_k[tid] == sel_k[tid + 1];
            plain = __ballot(eq) == 0;
            if (plain && tid < w) {
                A.cells[(size_t)q * w + tid] = sel_i[tid];
                A.cdsel[(size_t)q * w + tid] = keyd(sel_k[tid]);
            }
        }
        if (tid == 0 && !plain) {
            int32_t *out = A.cells + (size_t)q * w;
            double *dout = A.cdsel + (size_t)q * w;
            if (sel_k[w - 1] == sel_k[w]) {
                // tie straddles the boundary: the bounded queue's closed form over the candidates in
                // arrival (= index) order; non-candidates are > tau and never count
                const u64 tk = sel_k[w - 1];
                int b = 0;
                while (b < w && sel_k[b] < tk) b++;
                // ties (== tk) and better (< tk) candidates are all inside the sorted prefix up to the
                // last entry equal to tk
                int last = w;
                while (last + 1 < n && ckey[last + 1] == tk) last++;
                // p = ties among the first w arrivals with key <= tk: walk indices ascending
                int p = 0, nonjunk = 0, prev = -1;
                while (nonjunk < w) {
                    int best = 0x7fffffff, bpos = -1;
                    for (int t = 0; t <= last; t++) {
                        const int ci = (int)cidx[t];
                        if (ci > prev && ci < best) {
                            best = ci;
                            bpos = t;
                        }
                    }
                    if (bpos < 0) break;
                    prev = best;
                    nonjunk++;
                    if (ckey[bpos] == tk) p++;
                }
                const int e = b - (w - p);
                // kept ties: tie ranks e..p-1 in index order; ties sit at sorted positions b..last with
                // ascending index (sort key = (distance, index))
                for (int r = 0; r < w - b; r++) sel_i[b + r] = (int)cidx[b + e + r];
            }
            int a = 0;
            while (a < w) {
                int bnd = a;
                while (bnd + 1 < w && sel_k[bnd + 1] == sel_k[a]) bnd++;
                for (int t = a; t <= bnd; t++) {
                    out[t] = sel_i[bnd - (t - a)];
                    dout[t] = keyd(sel_k[a]);
                }
                a = bnd + 1;
            }
        }
        return;
    }
    // ---- overflow: exact row + round-based selection (massive ties / degenerate bounds) ----------
    double *row = A.row_scratch + (size_t)q * C;
    u64 key[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = tid + i * MMIDX_BLOCK;
        u64 k = MMIDX_KEY_MAX;
        if (c < C) {
            double acc = 0.0;
            for (int j = 0; j < D; j++) {
                const double df = A.coarseT[(size_t)j * C + c] - qv[j];
                acc += df * df;
            }
            row[c] = acc;
            k = dkey(acc);
        }
        key[i] = k;
    }
    __syncthreads();
    for (int r = 0; r < R; r++) {
        u64 bk = MMIDX_KEY_MAX;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int c = tid + i * MMIDX_BLOCK;
            if (c < C && key[i] < bk) {
                bk = key[i];
                bi = c;
            }
        }
        block_min_pair(bk, bi, s_k, s_i);
#pragma unroll
        for (int i = 0; i < PER; i++)
            if (tid + i * MMIDX_BLOCK == bi) key[i] = MMIDX_KEY_MAX;
        if (tid == 0) {
            sel_k[r] = bk;
            sel_i[r] = bi;
        }
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        coarse_emit(row, C, w, R, sel_k, sel_i, A.cells + (size_t)q * w);
        for (int t = 0; t < w; t++) A.cdsel[(size_t)q * w + t] = row[A.cells[(size_t)q * w + t]];
    }
}

template <int PER>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select_approx(const ApproxSel A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *ckey = (u64 *)smem;                       // [CSEL_CAP] thread minima, then exact candidate keys
    u32 *cidx = (u32 *)(ckey + MMIDX_CSEL_CAP);    // [CSEL_CAP]
    u64 *sel_k = (u64 *)(cidx + MMIDX_CSEL_CAP);   // [w+1]
    int *sel_i = (int *)(sel_k + (A.w + 1));       // [w+1]
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    __shared__ u32 s_n4[4];  // statics total 64 B: keeps the dynamic LDS base 16-byte aligned
    u32 &s_n = s_n4[0];
    const int q = blockIdx.x, tid = threadIdx.x, C = A.C, D = A.D, w = A.w;
    const int R = w + 1;  // host guarantees R <= 256 <= C
    const float *srow = A.S + (size_t)q * C;
    const double qn = A.qn[q];
    // |d~ - d| <= eps for every centroid of this query:
    //   dot product: 2 (D + 3) 2^-24 (1.01) |q| |c|     (input rounding + D fused multiply-adds)
    //   fp64 epilogue and exact-sum rounding: 1e-12 (|c|^2 + |q|^2)
    //   fp32 store of d~: 2^-23 |d~| <= 2^-23 (|c| + |q|)^2
    const double qnorm = sqrt(qn);
    const double sumn = A.cnorm_max + qnorm;
    const double eps = (2.0 * (double)(D + 3) * 0x1p-24 * 1.01 * qnorm * A.cnorm_max + 1e-12 * (A.cn_max + qn) +
                        0x1p-23 * sumn * sumn) * (1.0 + 1e-9);
    float dt[PER];  // d~
    float lmin = __int_as_float(0x7f800000);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = tid + i * MMIDX_BLOCK;
        const float v = (c < C) ? srow[c] : __int_as_float(0x7f800000);
        dt[i] = v;
        lmin = v < lmin ? v : lmin;
    }
    // tau: (R-th smallest per-thread minimum of d~) + eps >= the R-th smallest upper bound d~ + eps
    // R-th smallest of the 256 minima by rank counting (one barrier instead of a 36-stage sort)
    float *fmin = (float *)ckey;
    fmin[tid] = lmin < 0.0f ? 0.0f : lmin;
    if (tid == 0) s_n = 0;
    __syncthreads();
    {
        const float mine = fmin[tid];
        int rank = 0;
#pragma unroll 32
        for (int t = 0; t < MMIDX_BLOCK; t++) {
            const float o = fmin[t];
            rank += (o < mine) || (o == mine && t < tid);
        }
        if (rank == R - 1) sel_k[0] = dkey((double)mine);
    }
    __syncthreads();
    const double tau = keyd(sel_k[0]) + eps;
    __syncthreads();
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
    const double cut = tau + eps;  // candidate iff d~ - eps <= tau
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = tid + i * MMIDX_BLOCK;
        const bool pass = (c < C) && ((double)dt[i] <= cut);
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(&s_n, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < MMIDX_CSEL_CAP) cidx[slot] = (u32)c;
            }
        }
    }
    __syncthreads();
    coarse_select_finish<PER>(A, q, (int)s_n, ckey, cidx, sel_k, sel_i, s_k, s_i);
}


// ------------------------------------------------------------------------------------------------
// K1e + K1f: the same certified selection without materialising the [nq][C] matrix of d~.
//
// K1c writes 4 bytes per (query, centroid) and K1d reads them back -- 1 GiB of HBM traffic per 16384
// queries at C = 8192 -- although only ~w+3 entries per row matter.  Here the dot products run on the
// bf16 matrix cores with a two-term split (x = xh + xl + O(2^-16 x); q.c ~ qh.ch + qh.cl + ql.ch, each
// product exact in fp32), 16x the fp32 MFMA rate, and the epilogue keeps, per group of 8 centroids (the 8
// columns one lane holds for a row: no cross-lane work), the smallest d~, its position and the runner-up
// (K1e: [nq][C/8] float2, a quarter of the traffic).  K1f then
//   * takes tau = (w+1)-th smallest of its 256 per-thread minima + eps16: w+1 distinct centroids have an
//     exact distance <= their d~ + eps16, so tau bounds the (w+1)-th smallest exact distance;
//   * nominates the minimum of every group with d~ <= tau + eps16 (any centroid with exact distance <= tau
//     satisfies that), and all 8 members of a group whose runner-up passes too (rare: 8 (w+1) / C);
//   * hands the nominees to the shared exact second half.
// Error bound of the split dot product (|x - xh - xl| <= 2^-16 |x| elementwise, |xl| <= 2^-8 |x|):
//   |q.c - (qh.ch + qh.cl + ql.ch)| <= 3.1 * 2^-16 |q||c|   (dropped terms, Cauchy-Schwarz)
//   accumulation of 3D exact products in fp32, any order, at most 2^-22 relative per step (twice the IEEE
//   unit roundoff: the MFMA's internal summation order and rounding are not documented)
//                                   <= (3D + 16) 2^-22 |q||c|
//   d~ = |c|^2 + |q|^2 - 2 S evaluated in fp32 from fp32 copies of the norms: 2^-21 (|c| + |q|)^2
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define G16_BQ 128     // queries per block (32 per wave)
#define G16_BC 128     // centroids per tile
#define G16_KC 128     // k per LDS tile
#define G16_STRIDE 272 // bytes per LDS row: 256 + 16 (conflict-free 16-byte fragment reads)

// one wave per row: X (fp64) -> bf16 head / tail (zero padded to Dp), optional fp32 copy and squared norm
__global__ __launch_bounds__(MMIDX_BLOCK) void k_split_bf16(const double *__restrict__ X, __bf16 *__restrict__ H,
                                                            __bf16 *__restrict__ L, float *__restrict__ X32,
                                                            double *__restrict__ nrm2, int D, int Dp, long long n, u32 *__restrict__ zero_word = nullptr) {
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;  // (a counter of a later kernel of the same stream: saves a memset launch)
    const long long r = (long long)blockIdx.x * (MMIDX_BLOCK / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int j = lane; j < Dp; j += 64) {
        const double v = j < D ? X[(size_t)r * D + j] : 0.0;
        const float f = (float)v;
        const __bf16 h = (__bf16)f;
        const __bf16 l = (__bf16)(f - (float)h);  // f - h is exact in fp32
        H[(size_t)r * Dp + j] = h;
        L[(size_t)r * Dp + j] = l;
        if (X32 && j < D) X32[(size_t)r * D + j] = f;
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (nrm2 && lane == 0) nrm2[r] = s * (1.0 + 1e-12);  // only ever used inside error bounds: round up
}

// row-wise minimum over the 16 lanes that share lane >> 4 (all of them receive it); VALU only
__device__ __forceinline__ float row16_min(float v) {
    float o;
    o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v = o < v ? o : v;
    o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v = o < v ? o : v;
    o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v = o < v ? o : v;
    o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    v = o < v ? o : v;
    return v;
}

// K1e.  grid = (ceil(nq / 128), csplit); block = 4 waves, wave w owns query rows [32 w, 32 w + 32).
// Output: gpair[q][G = Cp / 8] = (smallest, second smallest | position of the smallest) of d~ over group
// g = 16 t + fr = centroids { 128 t + fr + 16 ct : ct = 0..7 }.
__global__ __launch_bounds__(MMIDX_BLOCK, 2) void k_coarse_gmin16(const __bf16 *__restrict__ Qh, const __bf16 *__restrict__ Ql,
                                                               const __bf16 *__restrict__ Ch, const __bf16 *__restrict__ Cl,
                                                               const double *__restrict__ cn, const double *__restrict__ qn,
                                                               float2 *__restrict__ gpair, int Cp, int Dp, int nq, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Bh = smem, *Bl = smem + G16_BC * G16_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int q0 = blockIdx.x * G16_BQ + wave * 32;
    const int ntiles = Cp / G16_BC;
    const int t_per = (ntiles + gridDim.y - 1) / gridDim.y;
    const int t_lo = blockIdx.y * t_per, t_hi = (t_lo + t_per < ntiles) ? t_lo + t_per : ntiles;
    const int nkc = (Dp + G16_KC - 1) / G16_KC;
    float qn_r[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = q0 + rt * 16 + 4 * fg + r;
            qn_r[rt][r] = q < nq ? (float)qn[q] : 0.0f;
        }
    bf16x8 ah[2][4], al[2][4];  // A fragments of one k chunk: [row tile][k step of 32]
    auto load_a = [&](int kc) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            int q = q0 + rt * 16 + fr;
            q = q < nq ? q : nq - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int k = kc * G16_KC + ks * 32 + fg * 8;
                if (k < Dp) {
                    ah[rt][ks] = *(const bf16x8 *)(Qh + (size_t)q * Dp + k);
                    al[rt][ks] = *(const bf16x8 *)(Ql + (size_t)q * Dp + k);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        ah[rt][ks][e] = (__bf16)0.0f;
                        al[rt][ks][e] = (__bf16)0.0f;
                    }
                }
            }
        }
    };
    if (nkc == 1) load_a(0);
    for (int t = t_lo; t < t_hi; t++) {
        const int c0 = t * G16_BC;
        f32x4 acc[2][8];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; kc++) {
            const int kbase = kc * G16_KC;
            const int kw = (Dp - kbase < G16_KC) ? Dp - kbase : G16_KC;  // multiple of 32
            const int upr = kw >> 3;                                      // 16-byte units per row
            __syncthreads();  // the previous tile's fragment reads are done
            if (upr == 16) {  // full-width chunk: all 16 loads of the thread in flight, then the LDS stores
                uint4 vh[8], vl[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int u = tid + i * MMIDX_BLOCK, row = u >> 4, cu = u & 15;
                    const size_t src = (size_t)(c0 + row) * Dp + kbase + cu * 8;
                    vh[i] = *(const uint4 *)(Ch + src);
                    vl[i] = *(const uint4 *)(Cl + src);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int u = tid + i * MMIDX_BLOCK, row = u >> 4, cu = u & 15;
                    *(uint4 *)(Bh + row * G16_STRIDE + cu * 16) = vh[i];
                    *(uint4 *)(Bl + row * G16_STRIDE + cu * 16) = vl[i];
                }
            } else {
                for (int u = tid; u < G16_BC * upr; u += MMIDX_BLOCK) {
                    const int row = u / upr, cu = u - row * upr;
                    const size_t src = (size_t)(c0 + row) * Dp + kbase + cu * 8;
                    *(uint4 *)(Bh + row * G16_STRIDE + cu * 16) = *(const uint4 *)(Ch + src);
                    *(uint4 *)(Bl + row * G16_STRIDE + cu * 16) = *(const uint4 *)(Cl + src);
                }
            }
            if (nkc > 1) load_a(kc);
            __syncthreads();
            // touch the next tile (one word per 128-byte line, this k chunk): after the scan kernels the centroids are
            // not in L2 any more, and the staging loads of the next tile would wait on HBM with nothing else to do
            unsigned touch_h = 0, touch_l = 0;
            if (t + 1 < t_hi && kw == G16_KC && Dp == G16_KC) {
                const size_t nxt = (size_t)(c0 + G16_BC) * Dp + (size_t)tid * 64;  // 256 threads x 128 B = the 32 KiB of a tile
                touch_h = *(const volatile unsigned *)(Ch + nxt);
                touch_l = *(const volatile unsigned *)(Cl + nxt);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (ks * 32 < kw) {  // block-uniform
                    // two column tiles at a time: four independent accumulators between dependent MFMAs
#pragma unroll
                    for (int ct = 0; ct < 8; ct += 2) {
                        bf16x8 bh[2], bl[2];
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const int off = ((ct + u) * 16 + fr) * G16_STRIDE + (ks * 32 + fg * 8) * 2;
                            bh[u] = *(const bf16x8 *)(Bh + off);
                            bl[u] = *(const bf16x8 *)(Bl + off);
                        }
#pragma unroll
                        for (int u = 0; u < 2; u++)
#pragma unroll
                            for (int rt = 0; rt < 2; rt++)
                                acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bh[u], acc[rt][ct + u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; u++)
#pragma unroll
                            for (int rt = 0; rt < 2; rt++)
                                acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bl[u], acc[rt][ct + u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; u++)
#pragma unroll
                            for (int rt = 0; rt < 2; rt++)
                                acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][ks], bh[u], acc[rt][ct + u], 0, 0, 0);
                    }
                }
            }
            asm volatile("" ::"v"(touch_h), "v"(touch_l));  // (keeps the two loads; their data is not used)
        }
        // epilogue: d~ = |c|^2 + |q|^2 - 2 S.  A group = the 8 columns one lane holds for a row
        // (c0 + fr + 16 ct, ct = 0..7): its two smallest d~ and the position of the smallest, no cross-lane work.
        // The epilogue's VALU work was half of this kernel's time (timing experiment: 155 us with, 48 us without), so
        // it is four instructions per value: t = fma(-2, S, |c|^2) (|q|^2, constant along the row, is added to the two
        // survivors only), the column index packed into t's three low mantissa bits (v_and_or_b32: the minimum then
        // carries its own position), runner-up = med3(best, runner-up, t), best = min(best, t).
        // Roundings: fp32 copies of the norms, the fma, the final addition (4 x 2^-24) and the three borrowed bits
        // (2^-20, on either survivor) -- inside eps16's last term (2^-21 + 2^-20)(|c| + |q|)^2.  Nothing here can
        // overflow or produce a NaN: K1f sends rows with (|c|max + |q|)^2 >= 1e37 to the exact path before it looks at
        // these values, and the +inf norms of the padding centroids are clamped to a finite "far".
        float cn_c[8];
#pragma unroll
        for (int ct = 0; ct < 8; ct++) {
            const float cf = (float)cn[c0 + ct * 16 + fr];
            cn_c[ct] = cf < 3.0e38f ? cf : 3.0e38f;
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float m1 = __int_as_float(0x7f7ffff8), m2 = m1;  // (largest finite value with the position bits clear)
#pragma unroll
                for (int ct = 0; ct < 8; ct++) {
                    const float t = __builtin_fmaf(-2.0f, acc[rt][ct][r], cn_c[ct]);
                    const float tp = __int_as_float((__float_as_int(t) & ~7) | ct);
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, tp);
                    // min(m1, tp) as the median with a value below everything (|t| < 1e37 here): a minnum of a value
                    // assembled from bits would be preceded by a canonicalising v_max, a fifth instruction per value
                    m1 = __builtin_amdgcn_fmed3f(m1, tp, -3.0e38f);
                }
                const int a1 = __float_as_int(m1) & 7;
                const float d1 = m1 + qn_r[rt][r];
                float d2 = m2 + qn_r[rt][r];
                d2 = d2 < 0.0f ? 0.0f : d2;
                // the position of the minimum rides in the 3 low mantissa bits of the runner-up (masked off by the reader)
                const float m2p = __int_as_float((__float_as_int(d2) & ~7) | a1);
                const int q = q0 + rt * 16 + 4 * fg + r;
                if (q < nq) gpair[(size_t)q * G + (size_t)t * 16 + fr] = make_float2(d1, m2p);
            }
        }
    }
}

// K1e', Dp == 128 (one k chunk: the usual D = 128): the same computation with the centroid tiles brought in by LDS-DMA
// (global_load_lds_dwordx4) into TWO buffers of half a tile (64 centroids x 256 B x {head, tail} = 32 KiB each), so that
// the next half tile lands while the matrix cores work on the current one -- in K1e the staging round trip
// (global -> registers -> LDS, two barriers) cost a third of the kernel (timing experiment: 155 us with, 103 us without).
//   * the DMA writes lane-linear (wave-uniform base + lane x 16 B), so the rows cannot be padded; the 16-byte units of a
//     row are XOR-swizzled instead -- unit u of row r is FETCHED into slot u ^ (r & 15) by choosing the lane's global
//     address, and read back from that slot: the 16 lanes of a fragment read (rows r .. r + 15, same u) hit 16 different
//     slots = all 64 banks;
//   * the two buffers are two distinct __shared__ arrays and the loop is unrolled by two halves, so that the compiler can
//     tell the buffer being filled from the one being read (it waits for ALL outstanding LDS-DMA before a ds_read it
//     cannot disambiguate); the barrier that ends a half drains the DMA issued at its start;
//   * groups, epilogue and output are those of K1e (a group = the 8 columns a lane holds over the two halves of a tile);
//   * no "touch" of the tile after next as in K1e: with the DMA half a tile ahead it measured slightly slower.
__global__ __launch_bounds__(MMIDX_BLOCK, 2) void k_coarse_gmin16_dma(const __bf16 *__restrict__ Qh, const __bf16 *__restrict__ Ql,
                                                                   const __bf16 *__restrict__ Ch, const __bf16 *__restrict__ Cl,
                                                                   const double *__restrict__ cn, const double *__restrict__ qn,
                                                                   float2 *__restrict__ gpair, int Cp, int nq, int G) {
    constexpr int Dp = G16_KC;               // 128 bf16 = 256 B = 16 units per row
    constexpr int HROWS = G16_BC / 2;        // centroids per half tile
    constexpr int HBYTES = HROWS * Dp * 2;   // 16 KiB per part (head / tail)
    __shared__ __attribute__((aligned(1024))) unsigned char buf0[2 * HBYTES];
    __shared__ __attribute__((aligned(1024))) unsigned char buf1[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int q0 = blockIdx.x * G16_BQ + wave * 32;
    const int ntiles = Cp / G16_BC;
    const int t_per = (ntiles + gridDim.y - 1) / gridDim.y;
    const int t_lo = blockIdx.y * t_per, t_hi = (t_lo + t_per < ntiles) ? t_lo + t_per : ntiles;
    if (t_lo >= t_hi) return;  // block-uniform
    float qn_r[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = q0 + rt * 16 + 4 * fg + r;
            qn_r[rt][r] = q < nq ? (float)qn[q] : 0.0f;
        }
    bf16x8 ah[2][4], al[2][4];  // A fragments: [row tile][k step of 32], resident for the whole block
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
        int q = q0 + rt * 16 + fr;
        q = q < nq ? q : nq - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            ah[rt][ks] = *(const bf16x8 *)(Qh + (size_t)q * Dp + ks * 32 + fg * 8);
            al[rt][ks] = *(const bf16x8 *)(Ql + (size_t)q * Dp + ks * 32 + fg * 8);
        }
    }
    // DMA of one half tile (first centroid row0) into a buffer: 4 + 4 wave-instructions of 1 KiB (4 rows) per wave
    auto issue = [&](unsigned char *buf, int row0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pu = (i * 4 + wave) * 64 + lane;  // 16-byte slot of the part this lane fills
            const int row = pu >> 4, slot = pu & 15;
            const size_t src = ((size_t)(row0 + row) * Dp + (size_t)((slot ^ (row & 15)) * 8)) * 2;  // bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)Ch + src),
                                             (__attribute__((address_space(3))) void *)(buf + (i * 4 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)Cl + src),
                                             (__attribute__((address_space(3))) void *)(buf + HBYTES + (i * 4 + wave) * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[2][8];
    // the four column tiles of one half: acc[rt][ct0 .. ct0 + 3] += A x B
    auto half_mma = [&](const unsigned char *buf, const int ct0) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int cp = 0; cp < 4; cp += 2) {
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int row = (cp + u) * 16 + fr;
                    const int off = row * (Dp * 2) + (((ks * 4 + fg) ^ fr) * 16);  // (row & 15 == fr)
                    bh[u] = *(const bf16x8 *)(buf + off);
                    bl[u] = *(const bf16x8 *)(buf + HBYTES + off);
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][ct0 + cp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bh[u], acc[rt][ct0 + cp + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][ct0 + cp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bl[u], acc[rt][ct0 + cp + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][ct0 + cp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][ks], bh[u], acc[rt][ct0 + cp + u], 0, 0, 0);
            }
        }
    };
    issue(buf0, t_lo * G16_BC);
    __syncthreads();
    for (int t = t_lo; t < t_hi; t++) {
        const int c0 = t * G16_BC;
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the tile's squared norms are requested before the DMA: a wait for them then does not wait for the DMA)
        float cn_c[8];
#pragma unroll
        for (int ct = 0; ct < 8; ct++) {
            const float cf = (float)cn[c0 + ct * 16 + fr];
            cn_c[ct] = cf < 3.0e38f ? cf : 3.0e38f;
        }
        // first half (columns c0 + fr + 16 ct, ct = 0..3) from buf0 while the second half lands in buf1
        issue(buf1, c0 + HROWS);
        half_mma(buf0, 0);
        __syncthreads();  // buf1 has landed (every wave waited for its own DMA), buf0 is free
        if (t + 1 < t_hi) issue(buf0, c0 + G16_BC);
        half_mma(buf1, 4);
        // epilogue: as in K1e
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float m1 = __int_as_float(0x7f7ffff8), m2 = m1;
#pragma unroll
                for (int ct = 0; ct < 8; ct++) {
                    const float tv = __builtin_fmaf(-2.0f, acc[rt][ct][r], cn_c[ct]);
                    const float tp = __int_as_float((__float_as_int(tv) & ~7) | ct);
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, tp);
                    m1 = __builtin_amdgcn_fmed3f(m1, tp, -3.0e38f);
                }
                const int a1 = __float_as_int(m1) & 7;
                const float d1 = m1 + qn_r[rt][r];
                float d2 = m2 + qn_r[rt][r];
                d2 = d2 < 0.0f ? 0.0f : d2;
                const float m2p = __int_as_float((__float_as_int(d2) & ~7) | a1);
                const int q = q0 + rt * 16 + 4 * fg + r;
                if (q < nq) gpair[(size_t)q * G + (size_t)t * 16 + fr] = make_float2(d1, m2p);
            }
        }
        __syncthreads();  // the next tile's first half has landed in buf0, buf1 is free
    }
}

// K1e' for long vectors (Dp a multiple of 128, e.g. the 1024 dimensions of YFCC100MExample.java:85): the same LDS-DMA double
// buffering with the k chunks of a tile streamed through the two half-tile buffers one after the other -- (chunk, half) is the
// unit -- the accumulators kept across the chunks of a tile, the query fragments of a chunk reloaded when the previous ones are dead.
// K1e itself (k_coarse_gmin16) stages every chunk through registers behind two barriers.
__global__ __launch_bounds__(MMIDX_BLOCK, 2) void k_coarse_gmin16_dma_kc(const __bf16 *__restrict__ Qh, const __bf16 *__restrict__ Ql,
                                                                   const __bf16 *__restrict__ Ch, const __bf16 *__restrict__ Cl,
                                                                   const double *__restrict__ cn, const double *__restrict__ qn,
                                                                   float2 *__restrict__ gpair, int Cp, int nq, int G, int Dp) {
    constexpr int KC = G16_KC;               // 128 bf16 = 256 B = 16 units per row and k chunk (Dp = nkc x KC)
    constexpr int HROWS = G16_BC / 2;        // centroids per half tile
    constexpr int HBYTES = HROWS * KC * 2;   // 16 KiB per part (head / tail)
    const int nkc = Dp / KC;
    __shared__ __attribute__((aligned(1024))) unsigned char buf0[2 * HBYTES];
    __shared__ __attribute__((aligned(1024))) unsigned char buf1[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int q0 = blockIdx.x * G16_BQ + wave * 32;
    const int ntiles = Cp / G16_BC;
    const int t_per = (ntiles + gridDim.y - 1) / gridDim.y;
    const int t_lo = blockIdx.y * t_per, t_hi = (t_lo + t_per < ntiles) ? t_lo + t_per : ntiles;
    if (t_lo >= t_hi) return;  // block-uniform
    float qn_r[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int q = q0 + rt * 16 + 4 * fg + r;
            qn_r[rt][r] = q < nq ? (float)qn[q] : 0.0f;
        }
    bf16x8 ah[2][4], al[2][4];    // A fragments of the current k chunk: [row tile][k step of 32]
    auto load_a = [&](bf16x8 (&xh)[2][4], bf16x8 (&xl)[2][4], const int kc) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            int q = q0 + rt * 16 + fr;
            q = q < nq ? q : nq - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                xh[rt][ks] = *(const bf16x8 *)(Qh + (size_t)q * Dp + kc * KC + ks * 32 + fg * 8);
                xl[rt][ks] = *(const bf16x8 *)(Ql + (size_t)q * Dp + kc * KC + ks * 32 + fg * 8);
            }
        }
    };
    load_a(ah, al, 0);
    // DMA of one half tile (first centroid row0) into a buffer: 4 + 4 wave-instructions of 1 KiB (4 rows) per wave
    auto issue = [&](unsigned char *buf, int row0, int kbase) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pu = (i * 4 + wave) * 64 + lane;  // 16-byte slot of the part this lane fills
            const int row = pu >> 4, slot = pu & 15;
            const size_t src = ((size_t)(row0 + row) * Dp + (size_t)kbase + (size_t)((slot ^ (row & 15)) * 8)) * 2;  // bytes
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)Ch + src),
                                             (__attribute__((address_space(3))) void *)(buf + (i * 4 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const unsigned char *)Cl + src),
                                             (__attribute__((address_space(3))) void *)(buf + HBYTES + (i * 4 + wave) * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[2][8];
    // the four column tiles of one half: acc[rt][ct0 .. ct0 + 3] += A x B
    auto half_mma = [&](const unsigned char *buf, const int ct0) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int cp = 0; cp < 4; cp += 2) {
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int row = (cp + u) * 16 + fr;
                    const int off = row * (KC * 2) + (((ks * 4 + fg) ^ fr) * 16);  // (row & 15 == fr)
                    bh[u] = *(const bf16x8 *)(buf + off);
                    bl[u] = *(const bf16x8 *)(buf + HBYTES + off);
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][ct0 + cp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bh[u], acc[rt][ct0 + cp + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][ct0 + cp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bl[u], acc[rt][ct0 + cp + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
                        acc[rt][ct0 + cp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][ks], bh[u], acc[rt][ct0 + cp + u], 0, 0, 0);
            }
        }
    };
    issue(buf0, t_lo * G16_BC, 0);
    __syncthreads();
    for (int t = t_lo; t < t_hi; t++) {
        const int c0 = t * G16_BC;
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the tile's squared norms are requested before the DMA: a wait for them then does not wait for the DMA)
        float cn_c[8];
#pragma unroll
        for (int ct = 0; ct < 8; ct++) {
            const float cf = (float)cn[c0 + ct * 16 + fr];
            cn_c[ct] = cf < 3.0e38f ? cf : 3.0e38f;
        }
#pragma unroll 1
        for (int kc = 0; kc < nkc; kc++) {
            const bool last = kc + 1 == nkc;
            // first half of this chunk from buf0 while its second half lands in buf1 and the next chunk's queries are requested
            issue(buf1, c0 + HROWS, kc * KC);
            half_mma(buf0, 0);
            __syncthreads();  // buf1 has landed (every wave waited for its own DMA), buf0 is free
            if (!last) issue(buf0, c0, (kc + 1) * KC);
            else if (t + 1 < t_hi) issue(buf0, c0 + G16_BC, 0);
            half_mma(buf1, 4);
            load_a(ah, al, last ? 0 : kc + 1);  // (the fragments are dead here: 222 registers leave no room for a second set)
            if (!last) __syncthreads();  // the next chunk's first half has landed in buf0, buf1 is free
        }
        // epilogue: as in K1e
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float m1 = __int_as_float(0x7f7ffff8), m2 = m1;
#pragma unroll
                for (int ct = 0; ct < 8; ct++) {
                    const float tv = __builtin_fmaf(-2.0f, acc[rt][ct][r], cn_c[ct]);
                    const float tp = __int_as_float((__float_as_int(tv) & ~7) | ct);
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, tp);
                    m1 = __builtin_amdgcn_fmed3f(m1, tp, -3.0e38f);
                }
                const int a1 = __float_as_int(m1) & 7;
                const float d1 = m1 + qn_r[rt][r];
                float d2 = m2 + qn_r[rt][r];
                d2 = d2 < 0.0f ? 0.0f : d2;
                const float m2p = __int_as_float((__float_as_int(d2) & ~7) | a1);
                const int q = q0 + rt * 16 + 4 * fg + r;
                if (q < nq) gpair[(size_t)q * G + (size_t)t * 16 + fr] = make_float2(d1, m2p);
            }
        }
        __syncthreads();  // the next tile's first half has landed in buf0, buf1 is free
    }
}

// K6a': certified nearest-centroid assignment with the bf16-split dot products of K1e (the encoder's and the learner's
// argmin).  Each block keeps 128 vectors' fragments in registers and walks ALL centroid tiles; every lane tracks the
// smallest and second smallest d~ (and the index of the smallest) over the columns it holds for each of its rows, the 16
// lanes of a row are merged at the end (DPP), and the vector is certified when second - best > 2 eps16 -- otherwise
// (exact ties included) it is flagged and the exact fp64 kernel redoes it, exactly as with K6a.
// FROMX: the vectors come straight from their fp64 rows X[n][D] (D a multiple of 8, one k chunk: D <= 128) -- head / tail split and
// squared norms in registers -- instead of from k_split_bf16's copies: one pass over the input instead of a 1.5 x larger round trip
// through HBM (VLAD assigns ten million 64-d descriptors per call, the encoder two million 128-d vectors per chunk).
template <bool FROMX>
__global__ __launch_bounds__(MMIDX_BLOCK, 2) void k_assign_gmin16_t(const __bf16 *__restrict__ Xh, const __bf16 *__restrict__ Xl,
                                                                 const __bf16 *__restrict__ Ch, const __bf16 *__restrict__ Cl,
                                                                 const double *__restrict__ cn, const double *__restrict__ xn,
                                                                 int32_t *__restrict__ cell_out, unsigned char *__restrict__ amb,
                                                                 double cnorm_max, double cn_max, int Cp, int Dp, long long n,
                                                                 const double *__restrict__ X, int D, int32_t *__restrict__ aidx,
                                                                 int32_t *__restrict__ acount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Bh = smem, *Bl = smem + G16_BC * G16_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const long long q0 = (long long)blockIdx.x * G16_BQ + wave * 32;
    const int ntiles = Cp / G16_BC;
    const int nkc = (Dp + G16_KC - 1) / G16_KC;
    float xn_r[2][4];
    [[maybe_unused]] double xrow[2] = {0.0, 0.0};  // FROMX: ||x||^2 (rounded up) of row rt * 16 + fr, in all four lanes that share fr
    if constexpr (!FROMX) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const long long q = q0 + rt * 16 + 4 * fg + r;
                xn_r[rt][r] = q < n ? (float)xn[q] : 0.0f;
            }
    }
    bf16x8 ah[2][4], al[2][4];
    auto load_a = [&](int kc) {
        if constexpr (FROMX) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                long long q = q0 + rt * 16 + fr;
                q = q < n ? q : n - 1;
                double pn = 0.0;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const int k = ks * 32 + fg * 8;
                    if (k < D) {  // (D a multiple of 8: a lane's eight values are all inside the row or all padding)
                        const double2 *xp = (const double2 *)(X + (size_t)q * D + k);
                        const double2 v0 = xp[0], v1 = xp[1], v2 = xp[2], v3 = xp[3];
                        const double vv[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float f = (float)vv[e];
                            const __bf16 hh = (__bf16)f;
                            ah[rt][ks][e] = hh;
                            al[rt][ks][e] = (__bf16)(f - (float)hh);  // f - head is exact in fp32
                            pn += vv[e] * vv[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            ah[rt][ks][e] = (__bf16)0.0f;
                            al[rt][ks][e] = (__bf16)0.0f;
                        }
                    }
                }
                pn += __shfl_xor(pn, 16);
                pn += __shfl_xor(pn, 32);
                xrow[rt] = pn * (1.0 + 1e-12);  // only ever used inside error bounds: round up (as k_split_bf16)
#pragma unroll
                for (int r = 0; r < 4; r++) xn_r[rt][r] = (float)__shfl(xrow[rt], 4 * fg + r);
            }
            return;
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            long long q = q0 + rt * 16 + fr;
            q = q < n ? q : n - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int k = kc * G16_KC + ks * 32 + fg * 8;
                if (k < Dp) {
                    ah[rt][ks] = *(const bf16x8 *)(Xh + (size_t)q * Dp + k);
                    al[rt][ks] = *(const bf16x8 *)(Xl + (size_t)q * Dp + k);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        ah[rt][ks][e] = (__bf16)0.0f;
                        al[rt][ks][e] = (__bf16)0.0f;
                    }
                }
            }
        }
    };
    if (nkc == 1) load_a(0);
    const float inf = __int_as_float(0x7f800000);
    float b1[2][4], b2[2][4];
    int i1[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            b1[rt][r] = inf;
            b2[rt][r] = inf;
            i1[rt][r] = 0;
        }
    for (int t = 0; t < ntiles; t++) {
        const int c0 = t * G16_BC;
        f32x4 acc[2][8];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; kc++) {
            const int kbase = kc * G16_KC;
            const int kw = (Dp - kbase < G16_KC) ? Dp - kbase : G16_KC;
            const int upr = kw >> 3;
            __syncthreads();
            if (upr == 16) {
                uint4 vh[8], vl[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int u = tid + i * MMIDX_BLOCK, row = u >> 4, cu = u & 15;
                    const size_t src = (size_t)(c0 + row) * Dp + kbase + cu * 8;
                    vh[i] = *(const uint4 *)(Ch + src);
                    vl[i] = *(const uint4 *)(Cl + src);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int u = tid + i * MMIDX_BLOCK, row = u >> 4, cu = u & 15;
                    *(uint4 *)(Bh + row * G16_STRIDE + cu * 16) = vh[i];
                    *(uint4 *)(Bl + row * G16_STRIDE + cu * 16) = vl[i];
                }
            } else {
                for (int u = tid; u < G16_BC * upr; u += MMIDX_BLOCK) {
                    const int row = u / upr, cu = u - row * upr;
                    const size_t src = (size_t)(c0 + row) * Dp + kbase + cu * 8;
                    *(uint4 *)(Bh + row * G16_STRIDE + cu * 16) = *(const uint4 *)(Ch + src);
                    *(uint4 *)(Bl + row * G16_STRIDE + cu * 16) = *(const uint4 *)(Cl + src);
                }
            }
            if (nkc > 1) load_a(kc);
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (ks * 32 < kw) {
#pragma unroll
                    for (int ct = 0; ct < 8; ct += 2) {
                        bf16x8 bh[2], bl[2];
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            const int off = ((ct + u) * 16 + fr) * G16_STRIDE + (ks * 32 + fg * 8) * 2;
                            bh[u] = *(const bf16x8 *)(Bh + off);
                            bl[u] = *(const bf16x8 *)(Bl + off);
                        }
#pragma unroll
                        for (int u = 0; u < 2; u++)
#pragma unroll
                            for (int rt = 0; rt < 2; rt++)
                                acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bh[u], acc[rt][ct + u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; u++)
#pragma unroll
                            for (int rt = 0; rt < 2; rt++)
                                acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bl[u], acc[rt][ct + u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; u++)
#pragma unroll
                            for (int rt = 0; rt < 2; rt++)
                                acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][ks], bh[u], acc[rt][ct + u], 0, 0, 0);
                    }
                }
            }
        }
        float cn_c[8];
#pragma unroll
        for (int ct = 0; ct < 8; ct++) cn_c[ct] = (float)cn[c0 + ct * 16 + fr];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int ct = 0; ct < 8; ct++) {
                    const float dv = (cn_c[ct] + xn_r[rt][r]) - 2.0f * acc[rt][ct][r];
                    const bool lt1 = dv < b1[rt][r];
                    b2[rt][r] = lt1 ? b1[rt][r] : (dv < b2[rt][r] ? dv : b2[rt][r]);
                    i1[rt][r] = lt1 ? c0 + ct * 16 + fr : i1[rt][r];
                    b1[rt][r] = lt1 ? dv : b1[rt][r];
                }
    }
    // merge the 16 lanes of every row: (best, its index, second best)
    auto dppf = [](float v, int ctrl) -> float {
        switch (ctrl) {
            case 0: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));
            case 1: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));
            case 2: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xf, 0xf, true));
            default: return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xf, 0xf, true));
        }
    };
    auto dppi = [](int v, int ctrl) -> int {
        switch (ctrl) {
            case 0: return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);
            case 1: return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);
            case 2: return __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);
            default: return __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);
        }
    };
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float m1 = b1[rt][r], m2 = b2[rt][r];
            int ix = i1[rt][r];
#pragma unroll
            for (int step = 0; step < 4; step++) {
                const float o1 = dppf(m1, step), o2 = dppf(m2, step);
                const int oi = dppi(ix, step);
                const bool take = o1 < m1 || (o1 == m1 && oi < ix);
                const float hi1 = take ? m1 : o1;  // the larger of the two bests
                const float lo2 = o2 < m2 ? o2 : m2;
                m2 = hi1 < lo2 ? hi1 : lo2;
                ix = take ? oi : ix;
                m1 = take ? o1 : m1;
            }
            const long long q = q0 + rt * 16 + 4 * fg + r;
            double xnd_x = 0.0;
            if constexpr (FROMX) xnd_x = __shfl(xrow[rt], 4 * fg + r);
            if (fr == 0 && q < n) {
                const double xnd = FROMX ? xnd_x : xn[q];
                const double xnorm = sqrt(xnd), sumn = cnorm_max + xnorm;
                const double eps = (2.0 * 3.1 * 0x1p-16 * xnorm * cnorm_max + 2.0 * (3.0 * (double)Dp + 16.0) * 0x1p-22 * xnorm * cnorm_max +
                                    1e-12 * (cn_max + xnd) + 0x1p-21 * sumn * sumn) * (1.0 + 1e-9);
                cell_out[q] = ix;
                // (beyond 1e37 the fp32 quantities above may have overflowed -- an infinite dot product would even make a
                //  far centroid look nearest: such vectors go to the exact kernel)
                const bool sure = ((double)m2 - (double)m1) > 2.0 * eps && sumn * sumn < 1e37;  // inf - x = inf > ... when there is one centroid
                amb[q] = sure ? 0 : 1;
                if (!sure && aidx) aidx[atomicAdd(acount, 1)] = (int32_t)q;  // (the list the exact kernel redoes: ~1e-4 of the vectors)
            }
        }
}

// K1f: front end of the selection over group minima (see above); one block per query
template <int PER>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select_grp(const ApproxSel A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *ckey = (u64 *)smem;                       // [CSEL_CAP] exact candidate keys (second half)
    u32 *cidx = (u32 *)(ckey + MMIDX_CSEL_CAP);    // [CSEL_CAP]
    u64 *sel_k = (u64 *)(cidx + MMIDX_CSEL_CAP);   // [w+1]
    int *sel_i = (int *)(sel_k + (A.w + 1));       // [w+1]
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    __shared__ u32 s_n4[4];  // 0: candidates (statics total 64 B + 16 B: the dynamic LDS base stays 16-byte aligned)
    __shared__ float s_tau4[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = A.C, w = A.w, G = A.G;
    const int R = w + 1;  // host guarantees R <= min(G, 256)
    constexpr int PERG = (PER + 7) / 8;  // groups per thread
    // the second half stages its difference terms after sel_i; this front end uses that space first
    float *gsort = (float *)(sel_i + ((A.w + 2) & ~1));  // [256] thread minima
    const double qn = A.qn[q];
    const double qnorm = sqrt(qn);
    const double sumn = A.cnorm_max + qnorm;
    // (2^-21 (|c| + |q|)^2: fp32 copies of |c|^2, |q|^2 and the three fp32 operations of the epilogue, five roundings of at
    //  most 2^-24 each; + 2^-20: the three mantissa bits borrowed from the runner-up)
    const double eps16 = (2.0 * 3.1 * 0x1p-16 * qnorm * A.cnorm_max + 2.0 * (3.0 * (double)A.Dp + 16.0) * 0x1p-22 * qnorm * A.cnorm_max +
                          1e-12 * (A.cn_max + qn) + (0x1p-21 + 0x1p-20) * sumn * sumn) * (1.0 + 1e-9);
    const float inf = __int_as_float(0x7f800000);
    float m1[PERG], m2[PERG];
    int a1[PERG];
    float gm = inf;
#pragma unroll
    for (int i = 0; i < PERG; i++) {
        const int g = tid + i * MMIDX_BLOCK;
        float2 v = make_float2(inf, inf);
        if (g < G) v = A.gpair[(size_t)q * G + g];
        m1[i] = v.x < 0.0f ? 0.0f : v.x;  // exact distances are >= 0
        a1[i] = __float_as_int(v.y) & 7;
        m2[i] = __int_as_float(__float_as_int(v.y) & ~7);  // rounded towards zero: never above the true runner-up
        gm = m1[i] < gm ? m1[i] : gm;
    }
    if (tid == 0) s_n4[0] = 0;
    // ---- tau: the R-th smallest of the 256 thread minima (R distinct centroids lie at or below it) ----
    // The values are >= 0 (or NaN / inf, which order above everything finite), so their bit patterns order them: one
    // wave takes all 256 (four per lane) and bisects on the pattern -- four compares + popcounts per step, ~24 steps,
    // scalar bookkeeping -- instead of every thread ranking its value against the other 255.
    gsort[tid] = gm;
    __syncthreads();
    if (wv == 0) {
        const u32 k0 = (u32)__float_as_int(gsort[lane]) & 0x7fffffffu, k1 = (u32)__float_as_int(gsort[64 + lane]) & 0x7fffffffu,
                  k2 = (u32)__float_as_int(gsort[128 + lane]) & 0x7fffffffu, k3 = (u32)__float_as_int(gsort[192 + lane]) & 0x7fffffffu;
        const u32 mn01 = k0 < k1 ? k0 : k1, mn23 = k2 < k3 ? k2 : k3, mx01 = k0 < k1 ? k1 : k0, mx23 = k2 < k3 ? k3 : k2;
        u32 lo_k = wave_min_u32(mn01 < mn23 ? mn01 : mn23), hi_k = wave_max_u32(mx01 < mx23 ? mx23 : mx01);
        while (lo_k < hi_k) {  // smallest pattern with at least R values at or below it == the R-th smallest value
            const u32 mid = lo_k + ((hi_k - lo_k) >> 1);
            const int c = (int)__popcll(__builtin_amdgcn_ballot_w64(k0 <= mid)) + (int)__popcll(__builtin_amdgcn_ballot_w64(k1 <= mid)) +
                          (int)__popcll(__builtin_amdgcn_ballot_w64(k2 <= mid)) + (int)__popcll(__builtin_amdgcn_ballot_w64(k3 <= mid));
            if (c >= R) hi_k = mid;
            else lo_k = mid + 1;
        }
        if (lane == 0) s_tau4[0] = __int_as_float((int)hi_k);
    }
    __syncthreads();
    // any centroid with exact distance <= tau has d~ <= tau + eps16
    const double cut = ((double)s_tau4[0] + eps16) + eps16;
    if (!(cut < (double)inf) || !(sumn * sumn < 1e37)) {  // (beyond 1e37 K1e's fp32 values may have overflowed: they are not looked at)
        // magnitudes beyond fp32 / bf16 (inf or NaN in d~): nothing can be certified -> the exact row (overflow path)
        __syncthreads();
        coarse_select_finish<PER>(A, q, MMIDX_CSEL_CAP + 1, ckey, cidx, sel_k, sel_i, s_k, s_i);
        return;
    }
    // ---- candidates: the minimum of every group at or under the cut; the whole group when its runner-up is too
    const u64 lane_lt = (1ull << lane) - 1ull;
    auto push = [&](bool pass, int c) {
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) base = atomicAdd(&s_n4[0], (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < MMIDX_CSEL_CAP) cidx[slot] = (u32)c;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < PERG; i++) {
        const int g = tid + i * MMIDX_BLOCK;
        const int cb = (g >> 4) * G16_BC + (g & 15);  // column ct of the group is centroid cb + 16 ct
        const bool hot = (double)m1[i] <= cut;
        push(hot && cb + 16 * a1[i] < C, cb + 16 * a1[i]);
        const bool all = hot && (double)m2[i] <= cut;
        if (__ballot(all)) {  // rare: two of the w+1 nearest in one group of 8
#pragma unroll
            for (int ct = 0; ct < 8; ct++) push(all && ct != a1[i] && cb + 16 * ct < C, cb + 16 * ct);
        }
    }
    __syncthreads();
    coarse_select_finish<PER>(A, q, (int)s_n4[0], ckey, cidx, sel_k, sel_i, s_k, s_i);
}

// ------------------------------------------------------------------------------------------------
// K1f in two kernels.  The front end of k_coarse_select_grp (read the query's group minima, find tau, nominate) is a
// stream over 8 KiB per query between block barriers, held to four blocks per CU by the registers of the selection
// behind it: 0.09 of the kernel's 0.18 ms per 16384 queries at 1.5 TB/s.  k_coarse_front does it with ONE WAVE per
// query and no barrier -- the 256 partial minima of the bisection are four per lane, candidates are compacted with
// ballots -- and leaves the candidate list in global memory; k_coarse_select_list is the selection alone.
// Same tau (the R-th smallest of 256 minima over disjoint sets of groups), same cut, same candidates.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_front(const ApproxSel A) {
    constexpr int GPL = 16;  // groups per lane: G <= 1024 (the host checks)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = (int)blockIdx.x * (MMIDX_BLOCK / 64) + wv;
    if (q >= A.nq) return;  // (wave-uniform; the kernel has no barrier)
    const int C = A.C, w = A.w, G = A.G;
    const int R = w + 1;
    const double qn = A.qn[q];
    const double qnorm = sqrt(qn);
    const double sumn = A.cnorm_max + qnorm;
    // (the error bound of k_coarse_select_grp)
    const double eps16 = (2.0 * 3.1 * 0x1p-16 * qnorm * A.cnorm_max + 2.0 * (3.0 * (double)A.Dp + 16.0) * 0x1p-22 * qnorm * A.cnorm_max +
                          1e-12 * (A.cn_max + qn) + (0x1p-21 + 0x1p-20) * sumn * sumn) * (1.0 + 1e-9);
    const float inf = __int_as_float(0x7f800000);
    float m1[GPL], ry[GPL];
    float km[4] = {inf, inf, inf, inf};
#pragma unroll
    for (int i = 0; i < GPL; i++) {
        const int g = lane + 64 * i;
        const int gc = g < G ? g : G - 1;
        const float2 v = A.gpair[(size_t)q * G + gc];  // (a load on a clamped index whose value every lane uses: see k_coarse_front_sel)
        m1[i] = fmaxf(v.x, g < G ? 0.0f : inf);  // exact distances are >= 0; a group past the end never qualifies
        ry[i] = v.y;
        km[i & 3] = m1[i] < km[i & 3] ? m1[i] : km[i & 3];
    }
    // ---- tau: the R-th smallest of the 256 partial minima (R distinct centroids lie at or below it): bisection on the bit
    //      patterns (values >= 0; NaN / inf order above everything finite)
    float tau;
    {
        const u32 k0 = (u32)__float_as_int(km[0]) & 0x7fffffffu, k1 = (u32)__float_as_int(km[1]) & 0x7fffffffu,
                  k2 = (u32)__float_as_int(km[2]) & 0x7fffffffu, k3 = (u32)__float_as_int(km[3]) & 0x7fffffffu;
        const u32 mn01 = k0 < k1 ? k0 : k1, mn23 = k2 < k3 ? k2 : k3, mx01 = k0 < k1 ? k1 : k0, mx23 = k2 < k3 ? k3 : k2;
        u32 lo_k = wave_min_u32(mn01 < mn23 ? mn01 : mn23), hi_k = wave_max_u32(mx01 < mx23 ? mx23 : mx01);
        while (lo_k < hi_k) {
            const u32 mid = lo_k + ((hi_k - lo_k) >> 1);
            const int c = (int)__popcll(__builtin_amdgcn_ballot_w64(k0 <= mid)) + (int)__popcll(__builtin_amdgcn_ballot_w64(k1 <= mid)) +
                          (int)__popcll(__builtin_amdgcn_ballot_w64(k2 <= mid)) + (int)__popcll(__builtin_amdgcn_ballot_w64(k3 <= mid));
            if (c >= R) hi_k = mid;
            else lo_k = mid + 1;
        }
        tau = __int_as_float((int)hi_k);
    }
    // any centroid with exact distance <= tau has d~ <= tau + eps16
    const double cut = ((double)tau + eps16) + eps16;
    u32 *list = A.clist + (size_t)q * MMIDX_CLIST;
    if (!(cut < (double)inf) || !(sumn * sumn < 1e37)) {  // nothing can be certified: the exact row
        if (lane == 0) list[0] = MMIDX_CSEL_CAP + 1;
        return;
    }
    // ---- candidates: the minimum of every group at or under the cut; the whole group when its runner-up is too
    const u64 lane_lt = (1ull << lane) - 1ull;
    u32 n = 0;  // wave-uniform
    auto push = [&](const bool pass, const int c) {
        const u64 mask = __builtin_amdgcn_ballot_w64(pass);
        if (pass) {
            const u32 slot = n + (u32)__popcll(mask & lane_lt);
            if (slot < MMIDX_CLIST - 1) list[1 + slot] = (u32)c;  // (entry 0 is the count: one row read in the selection kernel)
        }
        n += (u32)__popcll(mask);
    };
#pragma unroll
    for (int i = 0; i < GPL; i++) {
        const int g = lane + 64 * i;
        const int cb = (g >> 4) * G16_BC + (g & 15);  // column ct of the group is centroid cb + 16 ct
        const int a1 = __float_as_int(ry[i]) & 7;
        const float m2 = __int_as_float(__float_as_int(ry[i]) & ~7);  // rounded towards zero: never above the true runner-up
        const bool hot = (double)m1[i] <= cut;
        push(hot && cb + 16 * a1 < C, cb + 16 * a1);
        const bool all = hot && (double)m2 <= cut;
        if (__builtin_amdgcn_ballot_w64(all)) {  // rare: two of the w+1 nearest in one group of 8
#pragma unroll
            for (int ct = 0; ct < 8; ct++) push(all && ct != a1 && cb + 16 * ct < C, cb + 16 * ct);
        }
    }
    if (lane == 0) list[0] = n <= MMIDX_CLIST - 1 ? n : (u32)(MMIDX_CSEL_CAP + 1);
}

// K1f in ONE kernel where the exact stage fits a wave: front end as k_coarse_front, then -- for a query with at most 64 candidates
// (the usual 1.3 (w + 1)) -- the exact distances, the ranking and the bounded-queue rule by the same wave, no block barrier and no
// block-wide staging (k_coarse_select_list spent 117 us per 16384 queries of the headline workload between barriers: one block per
// query, 24 rows of terms through LDS, then one lane per candidate adding them up while 230 lanes wait).  NJB = D / 16 blocks of
// dimensions; the keys are the staged form's bit for bit (same operations in the same order).  The queries it cannot serve keep their
// candidate list and are appended to A.defer: k_coarse_select_defer (k_coarse_select_list over that list) answers them.
// A block is ONE wave: the queries' lifetimes differ by a factor of three (33 to 64 candidates, hits and misses of their rows), and in
// a block of four a finished wave's registers and LDS waited for the slowest one -- 67 % of the wave slots busy, 84 -> 62 us.
template <int NJB>
__global__ __launch_bounds__(64, 4) void k_coarse_front_sel(const ApproxSel A) {
    constexpr int GPL = 16;  // groups per lane: G <= 1024 (the host checks)
    __shared__ __attribute__((aligned(16))) double s_terms[16][65];  // the exact stage's tile; the keys and the selection live in it afterwards
    __shared__ u32 s_cidx[128];  // up to 128 candidates: the exact stage takes them 64 at a time
    const int lane = threadIdx.x;
    const int q = (int)blockIdx.x;
    u64 *ckey = (u64 *)&s_terms[0][0], *sel_k = ckey + 128;
    u32 *cidx = s_cidx;
    int *sel_i = (int *)(sel_k + 64);  // [128] (ranks of up to 128 candidates pass through it)
    const int C = A.C, w = A.w, G = A.G, D = A.D;
    const int R = w + 1;
    const double qn = A.qn[q];
    const double qnorm = sqrt(qn);
    const double sumn = A.cnorm_max + qnorm;
    const double eps16 = (2.0 * 3.1 * 0x1p-16 * qnorm * A.cnorm_max + 2.0 * (3.0 * (double)A.Dp + 16.0) * 0x1p-22 * qnorm * A.cnorm_max +
                          1e-12 * (A.cn_max + qn) + (0x1p-21 + 0x1p-20) * sumn * sumn) * (1.0 + 1e-9);
    const float inf = __int_as_float(0x7f800000);
    u32 *list = A.clist + (size_t)q * MMIDX_CLIST;
    u32 n = 0;  // wave-uniform
#ifdef CFS_TIMING
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, t0 = __builtin_readcyclecounter(), t1;
#define CFS_TICK(i) do { t1 = __builtin_readcyclecounter(); tacc[i] += t1 - t0; t0 = t1; } while (0)
#else
#define CFS_TICK(i) do { } while (0)
#endif
    {
        float m1[GPL], ry[GPL];
        float km[4] = {inf, inf, inf, inf};
#pragma unroll
        for (int i = 0; i < GPL; i++) {
            const int g = lane + 64 * i;
            const int gc = g < G ? g : G - 1;
            const float2 v = A.gpair[(size_t)q * G + gc];
            // (a load on a clamped index ...
            // ... whose value every lane USES: written as selects of v the compiler sank each load into its own `g < G` block with a
            //     full wait behind it -- sixteen memory round trips in a row, a quarter of this kernel's time)
            m1[i] = fmaxf(v.x, g < G ? 0.0f : inf);  // exact distances are >= 0; a group past the end never qualifies
            ry[i] = v.y;                              // (read only for a group that qualifies)
            km[i & 3] = m1[i] < km[i & 3] ? m1[i] : km[i & 3];
        }
        // ---- tau: the R-th smallest of the 256 partial minima, by bisection on the bit patterns (as k_coarse_front)
        float tau;
        {
            const u32 k0 = (u32)__float_as_int(km[0]) & 0x7fffffffu, k1 = (u32)__float_as_int(km[1]) & 0x7fffffffu,
                      k2 = (u32)__float_as_int(km[2]) & 0x7fffffffu, k3 = (u32)__float_as_int(km[3]) & 0x7fffffffu;
            const u32 mn01 = k0 < k1 ? k0 : k1, mn23 = k2 < k3 ? k2 : k3, mx01 = k0 < k1 ? k1 : k0, mx23 = k2 < k3 ? k3 : k2;
            u32 lo_k = wave_min_u32(mn01 < mn23 ? mn01 : mn23), hi_k = wave_max_u32(mx01 < mx23 ? mx23 : mx01);
            CFS_TICK(0);
            // (hi_k always has >= R values at or under it: stopping 12 bits early leaves a tau at most 2^-11 above the R-th smallest --
            //  a few candidates more at most, the exact stage decides either way -- and saves a dozen dependent rounds)
            while (hi_k - lo_k > 0xFFFu) {
                const u32 mid = lo_k + ((hi_k - lo_k) >> 1);
                const int c = (int)__popcll(__builtin_amdgcn_ballot_w64(k0 <= mid)) + (int)__popcll(__builtin_amdgcn_ballot_w64(k1 <= mid)) +
                              (int)__popcll(__builtin_amdgcn_ballot_w64(k2 <= mid)) + (int)__popcll(__builtin_amdgcn_ballot_w64(k3 <= mid));
                if (c >= R) hi_k = mid;
                else lo_k = mid + 1;
            }
            tau = __int_as_float((int)hi_k);
        }
        CFS_TICK(1);
        const double cut = ((double)tau + eps16) + eps16;
        if (!(cut < (double)inf) || !(sumn * sumn < 1e37)) {  // nothing can be certified: the exact row
            if (lane == 0) {
                list[0] = MMIDX_CSEL_CAP + 1;
                A.defer[1 + atomicAdd(A.defer, 1u)] = (u32)q;
            }
            return;
        }
        // ---- candidates (as k_coarse_front; their order is immaterial: the ranking below is by (distance, index)) into the wave's LDS
        //      list.  A group whose runner-up qualifies as well (rare: two of the w + 1 nearest in one group of 8) is only noted here
        //      and expanded afterwards from a reload; the global list is written only for a query that is handed on, by a compact
        //      loop over reloaded pairs (~40 scattered stores less for every other query, and a tenth of the code).
        const u64 lane_lt = (1ull << lane) - 1ull;
        float cutf = (float)cut;  // the largest float at or under the cut: x <= cutf exactly when (double)x <= cut
        if ((double)cutf > cut) cutf = __int_as_float(__float_as_int(cutf) - 1);  // (cut > 0 and finite)
        // (one compare per group and bit masks first, then a loop over the groups that qualified -- one per lane and round, three or
        //  four rounds for ~40 of 1024 groups -- instead of sixteen rounds of ballots and prefix counts)
        u32 allm = 0, hotm = 0, a1lo = 0, a1hi = 0;  // a1 of group i: three bits at 3 i of a1hi:a1lo (ten groups in the low word)
#pragma unroll
        for (int i = 0; i < GPL; i++) {
            const u32 rb = (u32)__float_as_int(ry[i]);
            const float m2 = __int_as_float((int)(rb & ~7u));  // rounded towards zero: never above the true runner-up
            const bool hot = m1[i] <= cutf;
            hotm |= hot ? 1u << i : 0u;
            allm |= (hot && m2 <= cutf) ? 1u << i : 0u;
            if (i < 10) a1lo |= (rb & 7u) << (3 * i);
            else a1hi |= (rb & 7u) << (3 * (i - 10));
        }
        while (__builtin_amdgcn_ballot_w64(hotm != 0)) {  // (wave-uniform)
            const bool act = hotm != 0;
            const int i = act ? __ffs(hotm) - 1 : 0;
            hotm &= hotm - 1;
            const int a1 = (int)((i < 10 ? a1lo >> (3 * i) : a1hi >> (3 * (i - 10))) & 7u);
            const int g = lane + 64 * i;
            const int c = (g >> 4) * G16_BC + (g & 15) + 16 * a1;  // column ct of the group is centroid cb + 16 ct
            const bool pass = act && c < C;
            const u64 mask = __builtin_amdgcn_ballot_w64(pass);
            if (pass) {
                const u32 slot = n + (u32)__popcll(mask & lane_lt);
                if (slot < 128) cidx[slot] = (u32)c;
            }
            n += (u32)__popcll(mask);
        }
        if (__builtin_amdgcn_ballot_w64(allm != 0)) {  // (wave-uniform)
            u32 *cnt = (u32 *)sel_i;                     // (free until the ranking)
            if (lane == 0) *cnt = n;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (u32 mm = allm; mm; mm &= mm - 1) {
                const int g = lane + 64 * (__ffs(mm) - 1);
                const int cb = (g >> 4) * G16_BC + (g & 15);
                const int a1 = __float_as_int(A.gpair[(size_t)q * G + g].y) & 7;
                for (int ct = 0; ct < 8; ct++)
                    if (ct != a1 && cb + 16 * ct < C) {
                        const u32 slot = atomicAdd(cnt, 1u);
                        if (slot < 128) cidx[slot] = (u32)(cb + 16 * ct);
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            n = (u32)__builtin_amdgcn_readfirstlane((int)*(volatile u32 *)cnt);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (cnt sits in the tile's memory)
            __builtin_amdgcn_wave_barrier();
        }
        CFS_TICK(2);
        if (n > 128 || (int)n < R) {  // (fewer than w + 1 candidates cannot happen with a valid cut; k_coarse_select_list's business either way)
            n = 0;
#pragma unroll 1
            for (int i = 0; i < GPL; i++) {
                const int g = lane + 64 * i;
                const float2 v = A.gpair[(size_t)q * G + (g < G ? g : G - 1)];
                const int cb = (g >> 4) * G16_BC + (g & 15);
                const int a1 = __float_as_int(v.y) & 7;
                const float m2 = __int_as_float(__float_as_int(v.y) & ~7);
                const bool hot = g < G && (double)fmaxf(v.x, 0.0f) <= cut;
                const bool all = hot && (double)m2 <= cut;
#pragma unroll 1
                for (int ct = 0; ct < 8; ct++) {
                    const bool pass = hot && (ct == a1 || all) && cb + 16 * ct < C;
                    const u64 mask = __builtin_amdgcn_ballot_w64(pass);
                    if (pass) {
                        const u32 slot = n + (u32)__popcll(mask & lane_lt);
                        if (slot < MMIDX_CLIST - 1) list[1 + slot] = (u32)(cb + 16 * ct);
                    }
                    n += (u32)__popcll(mask);
                }
            }
            if (lane == 0) {
                list[0] = n <= MMIDX_CLIST - 1 ? n : (u32)(MMIDX_CSEL_CAP + 1);
                A.defer[1 + atomicAdd(A.defer, 1u)] = (u32)q;
            }
            return;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- exact distances: a LANE per candidate adds the terms (c_j - q_j)^2 in dimension order (IVFPQ.java:583); the terms of 16
    //      dimensions at a time come through the wave's LDS tile, computed by all lanes from coalesced loads (8 lanes x 16 bytes = one
    //      128-byte line of a candidate's row, 8 candidates per load instruction); the next 16 dimensions' rows are requested before
    //      the current ones are summed
    {
        double(*terms)[65] = s_terms;  // [16][64 + 1]: lane c reads terms[j][c] (consecutive banks), the 8 lanes of a row write two apart
        const int cl = lane >> 3, jp = lane & 7;
        // (Asking for all lines of all candidate rows up front -- 40 % of them miss the L2: 8192 rows are 8 MiB against 4 MiB per XCD -- was
        //  tried with LDS-DMA loads into a sink: 89 -> 101 us per 16384 queries, the extra requests cost more than the misses they hid.)
        const unsigned char *cbase = (const unsigned char *)A.coarse;  // (wave-uniform base + a 32-bit byte offset per lane + a constant:
        const unsigned char *qbase = (const unsigned char *)(A.Q + (size_t)q * D);  //  no address arithmetic inside the steps)
        const u32 qo = 16u * (u32)jp;
        // NG = load instructions per block of 16 dimensions, a compile-time bound on ceil(candidates / 8): every load of an instance is
        // unconditional (clamped candidates), every step of it unrolled -- the waits are counted, not full.  Candidates base .. base + 63
        // (more than 64 -- a few queries in a hundred at w = 32 -- take a second pass instead of the block-per-query kernel, whose
        // single launch for them cost 17 us of every step).
        u64 key0 = MMIDX_KEY_MAX, key1 = MMIDX_KEY_MAX;
        for (int base = 0; base < (int)n; base += 64) {
            const int cn = (int)n - base < 64 ? (int)n - base : 64;
            const int ng = (cn + 7) >> 3;
            double acc = 0.0;
            auto exact = [&](auto ngc) {
                constexpr int NG = decltype(ngc)::value;
                u32 rowb[NG];  // (byte offsets: C * D * 8 < 2^32 for every C this kernel takes)
#pragma unroll
                for (int gq = 0; gq < NG; gq++) {
                    const int ci = gq * 8 + cl;
                    const int cc = ci < cn ? ci : cn - 1;
                    rowb[gq] = (cidx[base + cc] * (u32)D + 2u * (u32)jp) * 8u;
                }
                double2 buf[2][NG], qv[2];
#pragma unroll
                for (int gq = 0; gq < NG; gq++) buf[0][gq] = *(const double2 *)(cbase + rowb[gq]);
                qv[0] = *(const double2 *)(qbase + qo);
#pragma unroll
                for (int jb = 0; jb < NJB; jb++) {
                    const int cu = jb & 1, nx = cu ^ 1;
                    if (jb + 1 < NJB) {  // the next block's rows into the other register set first (two sets, no copies)
#pragma unroll
                        for (int gq = 0; gq < NG; gq++) buf[nx][gq] = *(const double2 *)(cbase + rowb[gq] + (jb + 1) * 128);
                        qv[nx] = *(const double2 *)(qbase + qo + (jb + 1) * 128);
                    }
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) {
                        const double d0 = buf[cu][gq].x - qv[cu].x, d1 = buf[cu][gq].y - qv[cu].y;
                        terms[2 * jp][gq * 8 + cl] = d0 * d0;  // (columns >= cn: copies of the last candidate, never read back)
                        terms[2 * jp + 1][gq * 8 + cl] = d1 * d1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        double tv[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) tv[j] = terms[8 * h + j][lane];
#pragma unroll
                        for (int j = 0; j < 8; j++) acc += tv[j];  // (lane >= cn: junk, never used)
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            };
            if (ng <= 4) exact(std::integral_constant<int, 4>());
            else if (ng <= 6) exact(std::integral_constant<int, 6>());
            else exact(std::integral_constant<int, 8>());
            const u64 kk = lane < cn ? dkey(acc) : MMIDX_KEY_MAX;
            if (base == 0) key0 = kk;
            else key1 = kk;
        }
        ckey[lane] = key0;  // (the tile is free: the keys move in)
        ckey[64 + lane] = key1;
    }
    CFS_TICK(3);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- ranking by (distance, index), bounded-queue rule: coarse_select_finish's n <= 64 branch, a lane for a thread.  Every lane
    //      counts the keys under its own from broadcast LDS reads, eight a round (lanes past n hold the largest key); equal keys show
    //      as two lanes with one rank and send the wave through the slower (key, index, lane) count.
    {
        const bool have0 = lane < (int)n, have1 = lane + 64 < (int)n;
        const u64 mk0 = ckey[lane], mk1 = ckey[64 + lane];
        const u32 mv0 = have0 ? cidx[lane] : 0xFFFFFFFFu, mv1 = have1 ? cidx[64 + lane] : 0xFFFFFFFFu;
        int rank0 = 0, rank1 = 0;
        if (n <= 64) {
            for (int j = 0; j < (int)n; j += 8) {
                u64 ok[8];
#pragma unroll
                for (int t = 0; t < 8; t++) ok[t] = ckey[j + t];
#pragma unroll
                for (int t = 0; t < 8; t++) rank0 += ok[t] < mk0;
            }
        } else {
            for (int j = 0; j < (int)n; j += 8) {
                u64 ok[8];
#pragma unroll
                for (int t = 0; t < 8; t++) ok[t] = ckey[j + t];
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    rank0 += ok[t] < mk0;
                    rank1 += ok[t] < mk1;
                }
            }
        }
        if (have0) sel_i[rank0] = lane;
        if (have1) sel_i[rank1] = lane + 64;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool dup = (have0 && sel_i[rank0] != lane) || (have1 && sel_i[rank1] != lane + 64);
        if (__builtin_amdgcn_ballot_w64(dup)) {  // wave-uniform: equal keys somewhere -- (key, index, position) decides
            rank0 = rank1 = 0;
            for (int j = 0; j < (int)n; j++) {
                const u64 ok = ckey[j];
                const u32 ov = cidx[j];
                rank0 += (ok < mk0) || (ok == mk0 && (ov < mv0 || (ov == mv0 && j < lane)));
                rank1 += (ok < mk1) || (ok == mk1 && (ov < mv1 || (ov == mv1 && j < lane + 64)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (every lane has read the entries before any lane writes)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (have0) {
            ckey[rank0] = mk0;
            cidx[rank0] = mv0;
            if (rank0 < R) sel_k[rank0] = mk0;
        }
        if (have1) {
            ckey[rank1] = mk1;
            cidx[rank1] = mv1;
            if (rank1 < R) sel_k[rank1] = mk1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (sel_i held the ranks' owners until here)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (have0 && rank0 < R) sel_i[rank0] = (int)mv0;
        if (have1 && rank1 < R) sel_i[rank1] = (int)mv1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    CFS_TICK(4);
#ifdef CFS_TIMING
    if (lane == 0 && (q & 4095) == 77)
        printf("[cfs] q %d n %u: cycles load+min %llu bisect %llu push %llu exact %llu rank %llu\n", q, n, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
#endif
    bool plain;
    {
        const bool eq = lane < w && sel_k[lane] == sel_k[lane + 1];
        plain = __builtin_amdgcn_ballot_w64(eq) == 0;
        if (plain && lane < w) {
            A.cells[(size_t)q * w + lane] = sel_i[lane];
            A.cdsel[(size_t)q * w + lane] = keyd(sel_k[lane]);
        }
    }
    if (lane == 0 && !plain) {  // equal keys among the w + 1 best: coarse_select_finish's closed form
        int32_t *out = A.cells + (size_t)q * w;
        double *dout = A.cdsel + (size_t)q * w;
        if (sel_k[w - 1] == sel_k[w]) {
            const u64 tk = sel_k[w - 1];
            int b = 0;
            while (b < w && sel_k[b] < tk) b++;
            int last = w;
            while (last + 1 < (int)n && ckey[last + 1] == tk) last++;
            int p = 0, nonjunk = 0, prev = -1;
            while (nonjunk < w) {
                int best = 0x7fffffff, bpos = -1;
                for (int t = 0; t <= last; t++) {
                    const int ci = (int)cidx[t];
                    if (ci > prev && ci < best) {
                        best = ci;
                        bpos = t;
                    }
                }
                if (bpos < 0) break;
                prev = best;
                nonjunk++;
                if (ckey[bpos] == tk) p++;
            }
            const int e = b - (w - p);
            for (int r = 0; r < w - b; r++) sel_i[b + r] = (int)cidx[b + e + r];
        }
        int a = 0;
        while (a < w) {
            int bnd = a;
            while (bnd + 1 < w && sel_k[bnd + 1] == sel_k[a]) bnd++;
            for (int t = a; t <= bnd; t++) {
                out[t] = sel_i[bnd - (t - a)];
                dout[t] = keyd(sel_k[a]);
            }
            a = bnd + 1;
        }
    }
}

template <int PER>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select_list(const ApproxSel A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *ckey = (u64 *)smem;                       // [CSEL_CAP] exact candidate keys
    u32 *cidx = (u32 *)(ckey + MMIDX_CSEL_CAP);    // [CSEL_CAP]
    u64 *sel_k = (u64 *)(cidx + MMIDX_CSEL_CAP);   // [w+1]
    int *sel_i = (int *)(sel_k + (A.w + 1));       // [w+1]
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    __shared__ u32 s_pad[8];  // (statics total 80 B as in k_coarse_select_grp: the dynamic LDS base stays 16-byte aligned)
    const int q = blockIdx.x, tid = threadIdx.x;
    {   // one coalesced row: the count in entry 0, the candidates behind it (entries past the count are never looked at)
        const u32 c = A.clist[(size_t)q * MMIDX_CLIST + tid];
        if (tid == 0) s_pad[0] = c;
        else cidx[tid - 1] = c;
    }
    __syncthreads();
    const int n = (int)s_pad[0];
    coarse_select_finish<PER>(A, q, n, ckey, cidx, sel_k, sel_i, s_k, s_i);
}

// the queries k_coarse_front_sel left (more than 64 candidates, or nothing certifiable): k_coarse_select_list over its list of them
template <int PER>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_coarse_select_defer(const ApproxSel A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *ckey = (u64 *)smem;
    u32 *cidx = (u32 *)(ckey + MMIDX_CSEL_CAP);
    u64 *sel_k = (u64 *)(cidx + MMIDX_CSEL_CAP);
    int *sel_i = (int *)(sel_k + (A.w + 1));
    __shared__ u64 s_k[MMIDX_BLOCK / 64];
    __shared__ int s_i[MMIDX_BLOCK / 64];
    __shared__ u32 s_pad[8];
    const int tid = threadIdx.x;
    const u32 nd = *A.defer;
    for (u32 i = blockIdx.x; i < nd; i += gridDim.x) {
        __syncthreads();  // (the previous query's LDS is done with)
        const int q = (int)A.defer[1 + i];
        const u32 c = A.clist[(size_t)q * MMIDX_CLIST + tid];
        if (tid == 0) s_pad[0] = c;
        else cidx[tid - 1] = c;
        __syncthreads();
        coarse_select_finish<PER>(A, q, (int)s_pad[0], ckey, cidx, sel_k, sel_i, s_k, s_i);
    }
}

// ------------------------------------------------------------------------------------------------
// K2 + K3: per (query, probe, chunk) work item: residual -> transform -> ADC lookup table in LDS
// -> coalesced scan of the list's PQ codes -> threshold-filtered candidate buffer.
//
//   residual  r = centroid - q                    computeResidualVector IVFPQ.java:642-648
//   transform permute / rotate                    IVFPQ.java:420-424, PQ.java:294-298
//   LUT[s][j] = sum_t (r[s*dsub+t]-pq[s][j][t])^2  computeLookupADC IVFPQ.java:525-538
//   d(code)   = sum_s LUT[s][code[s]], s ascending IVFPQ.java:435-438 / PQ.java:308-311
//
// Candidates: the bounded queue (IVFPQ.java:445) is replaced by an exact selection.  Each query
// owns a global threshold T (u64 bits of a distance that is provably >= its (k+1)-th smallest
// candidate distance); a code survives when d <= T.  Survivors go to a block-local LDS buffer,
// which is pruned to the K1 = k+1 smallest by (distance, position) when it fills; every prune
// publishes a tighter T with atomicMin.  At the end the block appends its <= K1 survivors to the
// query's pool; k_merge sorts the pool.  Stale T only costs work, never correctness.
//
// Work order: items are (query, probe) pairs.  Pass A takes probe rank 0 of every query (the
// nearest cell, which fixes a tight T); pass B takes the remaining pairs sorted by cell
// (k_pair_*), so that the blocks scanning one inverted list run back to back and re-read its
// codes from the XCD's L2 instead of HBM; the block -> item map sends consecutive items to the
// same XCD (blocks are dispatched round-robin over the 8 XCDs).
// ------------------------------------------------------------------------------------------------
struct ScanParams {
    const double *Q;         // [nq][D]
    const double *coarse;    // [C][D] (IVFPQ)
    const double *pqT;       // [m][dsub][ks]
    const int32_t *perm;     // [D] or null
    const double *rot;       // [D][D] or null
    const int32_t *cells;    // [nq][w] (IVFPQ) or null (PQ)
    const int64_t *list_off; // [nlists+1]
    const void *codes;       // [n][m] CodeT
    const double *sdc_tt;    // SDC mode (PQ.computeKnnSDC): [nq][m][ks][dsub] squared-difference terms, else null
    const int32_t *order;    // sorted pair ids (q*w + rank) or null = natural order
    const int32_t *n_order;  // device count of valid entries in order[]
    u64 *T;                  // [nq]
    u32 *pool_cnt;           // [nq]
    u64 *pool_key;           // [nq][poolq]
    u64 *pool_val;           // [nq][poolq]  probe_rank << 32 | position in list
    int D, m, ks, dsub, w, transform, ivf;
    int n_items;             // items of this launch
    int rank_lo, nrank;      // natural order: item -> (q = item / nrank, rank = rank_lo + item % nrank)
    int xcd_remap;           // 1: consecutive items -> same XCD
    int chunk;               // codes per work item
    unsigned vb_base, vgrid; // K3f: first virtual block of this launch / total number of virtual blocks
    const int32_t *order_ch; // with `order`: chunk of each item (fallback launches), else null = blockIdx.y
    u32 *fb_count;           // K3h: items handed back to K3
    int32_t *fb_items, *fb_ch;
    int code_lo, code_hi;    // only list positions [code_lo, code_hi) are scanned by this launch (pass A prefix / remainder)
    double *glut;            // GLUT kernels (table larger than the LDS): scratch of m * ks doubles per block, else null
    const double *lut_pre;   // K3h: the queries' exact tables of their nearest cell [nq][m][ks], built by k_lut_pre (null: every block builds its own)
    int K1;                  // k + 1
    int cap;                 // LDS candidate capacity (>= K1 + SEG, power of two)
    int poolq;
};

#define MMIDX_SEGU 2  // codes per thread per segment
#define MMIDX_SEG (MMIDX_BLOCK * MMIDX_SEGU)

// 8 * (byte b of w) in ONE VALU instruction (SDWA byte select on the shift's operand): the byte offset of a fp64 table
// entry.  (The compiler's own sequence is v_bfe_u32 + v_lshl_add_u32.)  b is a constant after unrolling.
__device__ __forceinline__ u32 byte_x8(u32 w, int b) {
    u32 r;
    const u32 three = 3u;
    switch (b & 3) {
        case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(three), "v"(w)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(three), "v"(w)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(three), "v"(w)); break;
        default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(three), "v"(w)); break;
    }
    return r;
}
typedef __attribute__((address_space(3))) const double lds_cdouble;

template <int M, typename CodeT>
struct CodeVec {
    static constexpr int BYTES = M * (int)sizeof(CodeT);
    static constexpr int WORDS = (BYTES + 3) / 4;
    u32 wd[WORDS];
    __device__ __forceinline__ void load(const CodeT *p) {
        if constexpr (BYTES % 16 == 0) {
            const uint4 *s = (const uint4 *)p;
#pragma unroll
            for (int i = 0; i < BYTES / 16; i++) {
                uint4 v = s[i];
                wd[4 * i] = v.x;
                wd[4 * i + 1] = v.y;
                wd[4 * i + 2] = v.z;
                wd[4 * i + 3] = v.w;
            }
        } else if constexpr (BYTES % 8 == 0) {
            const uint2 *s = (const uint2 *)p;
#pragma unroll
            for (int i = 0; i < BYTES / 8; i++) {
                uint2 v = s[i];
                wd[2 * i] = v.x;
                wd[2 * i + 1] = v.y;
            }
        } else if constexpr (BYTES % 4 == 0) {
            const u32 *s = (const u32 *)p;
#pragma unroll
            for (int i = 0; i < BYTES / 4; i++) wd[i] = s[i];
        } else {
#pragma unroll
            for (int i = 0; i < WORDS; i++) wd[i] = 0;
            const unsigned char *s = (const unsigned char *)p;
#pragma unroll
            for (int i = 0; i < BYTES; i++) wd[i >> 2] |= (u32)s[i] << (8 * (i & 3));
        }
    }
    __device__ __forceinline__ int get(int s) const {
        if constexpr (sizeof(CodeT) == 1) return (wd[s >> 2] >> (8 * (s & 3))) & 0xff;
        else return (wd[s >> 1] >> (16 * (s & 1))) & 0xffff;
    }
};

// prune the block's candidate buffer to the K1 smallest by (key, position); publishes T
__device__ __forceinline__ void scan_prune(u64 *bkey, u32 *bval, u32 *s_cnt, int K1, u64 *Tq) {
    __syncthreads();
    const int n = (int)*s_cnt;
    const int P = pow2ceil(n < 2 ? 2 : n);
    for (int i = n + threadIdx.x; i < P; i += blockDim.x) {
        bkey[i] = MMIDX_KEY_MAX;
        bval[i] = 0xFFFFFFFFu;
    }
    block_bitonic_sort<u32>(bkey, bval, P);
    if (threadIdx.x == 0) {
        if (n >= K1) {
            *s_cnt = (u32)K1;
            atomicMin(Tq, bkey[K1 - 1]);
        }
    }
    __syncthreads();
}

// LUT[s][j] for idx = s*ks + j; the dsub loads of an entry are issued together (they are
// L2-resident: the transposed codebook is 256 KiB) and two entries are in flight per thread.
template <int DSUB>
__device__ __forceinline__ void build_lut(double *lut, const double *tr, const double *__restrict__ pqT, int m,
                                          int ks, int dsub_rt) {
    const int total = m * ks;
    if constexpr (DSUB > 0) {
#if MMIDX_LUT_PAIRS
        if ((ks & 1) == 0) {  // two adjacent entries per thread: 16-byte loads, half as many load instructions
            const int hk = ks >> 1;
#pragma unroll 2
            for (int pi = threadIdx.x; pi < (total >> 1); pi += blockDim.x) {
                const int s = pi / hk, j = (pi - s * hk) * 2;
                const double *pp = pqT + (size_t)s * DSUB * ks + j;
                double2 pv[DSUB];
#pragma unroll
                for (int t = 0; t < DSUB; t++) pv[t] = *(const double2 *)(pp + (size_t)t * ks);
                const double *tv = tr + s * DSUB;
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int t = 0; t < DSUB; t++) {
                    const double d0 = tv[t] - pv[t].x, d1 = tv[t] - pv[t].y;
                    a0 += d0 * d0;
                    a1 += d1 * d1;
                }
                *(double2 *)(lut + s * ks + j) = make_double2(a0, a1);
            }
            return;
        }
#endif
#pragma unroll 2
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int s = idx / ks, j = idx - s * ks;
            const double *pp = pqT + (size_t)s * DSUB * ks + j;
            double pv[DSUB];
#pragma unroll
            for (int t = 0; t < DSUB; t++) pv[t] = pp[(size_t)t * ks];
            const double *tv = tr + s * DSUB;
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < DSUB; t++) {
                const double df = tv[t] - pv[t];
                acc += df * df;
            }
            lut[idx] = acc;
        }
    } else {
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int s = idx / ks, j = idx - s * ks;
            const double *pp = pqT + (size_t)s * dsub_rt * ks + j;
            const double *tv = tr + s * dsub_rt;
            double acc = 0.0;
            int t = 0;
            for (; t + 4 <= dsub_rt; t += 4) {
                const double p0 = pp[(size_t)t * ks], p1 = pp[(size_t)(t + 1) * ks], p2 = pp[(size_t)(t + 2) * ks],
                             p3 = pp[(size_t)(t + 3) * ks];
                double df = tv[t] - p0;
                acc += df * df;
                df = tv[t + 1] - p1;
                acc += df * df;
                df = tv[t + 2] - p2;
                acc += df * df;
                df = tv[t + 3] - p3;
                acc += df * df;
            }
            for (; t < dsub_rt; t++) {
                const double df = tv[t] - pp[(size_t)t * ks];
                acc += df * df;
            }
            lut[idx] = acc;
        }
    }
}

__device__ __forceinline__ void build_lut_any(double *lut, const double *tr, const double *pqT, int m, int ks,
                                              int dsub) {
    switch (dsub) {
        case 4: build_lut<4>(lut, tr, pqT, m, ks, dsub); break;
        case 8: build_lut<8>(lut, tr, pqT, m, ks, dsub); break;
        case 16: build_lut<16>(lut, tr, pqT, m, ks, dsub); break;
        default: build_lut<0>(lut, tr, pqT, m, ks, dsub); break;
    }
}

// residual (centroid - q, IVFPQ.java:645) or the query itself (PQ), then permute / rotate
__device__ __forceinline__ double *query_vector(const ScanParams &P, int q, int cell, double *vec) {
    const int D = P.D, tid = threadIdx.x;
    double *r = vec, *tr = vec + D;
    for (int i = tid; i < D; i += blockDim.x) {
        const double qv = P.Q[(size_t)q * D + i];
        r[i] = P.ivf ? (P.coarse[(size_t)cell * D + i] - qv) : qv;
    }
    __syncthreads();
    if (P.transform == 2) {
        for (int i = tid; i < D; i += blockDim.x) tr[i] = r[P.perm[i]];
        __syncthreads();
        return tr;
    }
    if (P.transform == 1) {
        for (int j = tid; j < D; j += blockDim.x) {
            double total = 0.0;
            for (int i = 0; i < D; i++) total += r[i] * P.rot[(size_t)i * D + j];
            tr[j] = total;
        }
        __syncthreads();
        return tr;
    }
    return r;
}

// SU = codes per thread per segment: 2 by default; 1 for pass A (one cold list per query: the block is a
// long dependent chain, so LDS is traded for a fourth resident block per CU)
// NT = threads per block (256, or 512 for pass A: twice the waves over the same LDS footprint)
// SDC = symmetric distances (PQ.computeKnnSDC PQ.java:334-374): instead of the LUT sum, the distance is ONE chain
// over all D dimensions of (pq[s][code_s][t] - pq[s][querycode_s][t])^2, s outer, t inner (PQ.java:349-363); the
// squared terms come from a per-query table in global memory (L2-resident, 256 KiB at m*ks*dsub = 32768).
// GLUT: the lookup table does not fit the LDS (m * ks * 8 bytes beyond ~140 KiB: short codes with many sub-quantizers) and
// lives in the block's slice of a global scratch buffer instead (L2-resident; the gather becomes global loads).  The slow,
// complete path: same arithmetic, same order.
template <int M, typename CodeT, int SU, int NT, bool SDC, bool GLUT = false>
__global__ __launch_bounds__(NT) void k_scan(const ScanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = (M > 0) ? M : P.m;
    const int ks = P.ks, D = P.D;
    double *lut, *vec;
    if constexpr (GLUT) {
        lut = P.glut + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)m * ks;
        vec = (double *)smem;
    } else {
        lut = (double *)smem;                     // [m*ks]
        vec = lut + (size_t)m * ks;               // [2*D]
    }
    u64 *bkey = (u64 *)(vec + (P.transform ? 2 : 1) * (size_t)D);  // [cap] (the transform needs a second vector)
    u32 *bval = (u32 *)(bkey + P.cap);            // [cap]
    u32 *s_cnt = bval + P.cap;                    // [4]

    // ---- block -> work item ---------------------------------------------------------------------
    int item = blockIdx.x;
    const int n_items = P.order ? *P.n_order : P.n_items;  // pass B: count left by the coarse bound
    if (P.xcd_remap) {
        const int per = (n_items + 7) >> 3;
        if ((int)(blockIdx.x >> 3) >= per) return;
        item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    if (item >= n_items) return;
    int q, pr;
    if (P.order) {
        const int e = P.order[item];
        q = e / P.w;
        pr = e - q * P.w;
    } else {
        q = item / P.nrank;
        pr = P.rank_lo + (item - q * P.nrank);
    }
    const int ch = P.order_ch ? P.order_ch[item] : (P.ivf ? (int)blockIdx.y : pr);  // flat PQ: the item rank is the chunk of the single list
    int cell = 0;
    if (P.ivf) {
        cell = P.cells[(size_t)q * P.w + pr];
        if (cell < 0) return;
    }
    const int64_t beg = P.list_off[cell];
    const int64_t len = P.list_off[cell + 1] - beg;
    int64_t c0 = (int64_t)ch * P.chunk;
    if (c0 >= len) return;
    int64_t c1 = (c0 + P.chunk < len) ? c0 + P.chunk : len;
    c0 = c0 < (int64_t)P.code_lo ? (int64_t)P.code_lo : c0;
    c1 = c1 > (int64_t)P.code_hi ? (int64_t)P.code_hi : c1;
    if (c0 >= c1) return;
    const int tid = threadIdx.x;
    const CodeT *codes = (const CodeT *)P.codes + (size_t)beg * m;

    // first segment's codes: issued before the LUT build so that HBM latency hides under it
    CodeVec<(M > 0 ? M : 4), CodeT> cur[SU], nxt[SU];
    if constexpr (M > 0) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int64_t i = c0 + u * NT + tid;
            cur[u].load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
        }
    }

    if (tid == 0) s_cnt[0] = 0;
    const double *TT = nullptr;
    if constexpr (SDC) {
        TT = P.sdc_tt + (size_t)q * m * ks * P.dsub;
    } else {
        const double *tr = query_vector(P, q, cell, vec);
        build_lut_any(lut, tr, P.pqT, m, ks, P.dsub);
    }
    u64 *Tq = P.T + q;
    u64 T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();

    // ---- scan -----------------------------------------------------------------------------------
    const int limit = P.cap - NT * SU;
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
    for (int64_t seg = c0; seg < c1; seg += (NT * SU)) {
        double d[SU];
        const bool more = seg + NT * SU < c1;
        if constexpr (M > 0) {
            if (more) {
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const int64_t i = seg + NT * SU + u * NT + tid;
                    nxt[u].load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
                }
            }
#pragma unroll
            for (int u = 0; u < SU; u++) d[u] = 0.0;
            if constexpr (SDC) {
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    for (int s = 0; s < M; s++) {
                        const double *tt = TT + ((size_t)s * ks + cur[u].get(s)) * P.dsub;
                        for (int t = 0; t < P.dsub; t++) d[u] += tt[t];
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < M; s++) {
#pragma unroll
                    for (int u = 0; u < SU; u++) d[u] += lut[s * ks + cur[u].get(s)];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int64_t i = seg + u * NT + tid;
                const CodeT *cp = codes + (size_t)(i < c1 ? i : c1 - 1) * m;
                double a = 0.0;
                if constexpr (SDC) {
                    for (int s = 0; s < m; s++) {
                        const double *tt = TT + ((size_t)s * ks + (int)cp[s]) * P.dsub;
                        for (int t = 0; t < P.dsub; t++) a += tt[t];
                    }
                } else {
                    for (int s = 0; s < m; s++) a += lut[s * ks + (int)cp[s]];
                }
                d[u] = a;
            }
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int64_t i = seg + u * NT + tid;
            const u64 key = dkey(d[u]);
            const bool pass = (i < c1) && key <= T;
            const u64 mask = __ballot(pass);
            if (mask) {
                u32 base = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if ((tid & 63) == leader) base = atomicAdd(s_cnt, (u32)__popcll(mask));
                base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                if (pass) {
                    const u32 slot = base + (u32)__popcll(mask & lane_lt);
                    bkey[slot] = key;
                    bval[slot] = (u32)i;
                }
            }
        }
        __syncthreads();
        const bool need = (int)*s_cnt > limit;  // uniform: nobody writes s_cnt until the barrier below
        __syncthreads();
        if (need) {
            scan_prune(bkey, bval, s_cnt, P.K1, Tq);
            T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (M > 0) {
            if (more) {
#pragma unroll
                for (int u = 0; u < SU; u++) cur[u] = nxt[u];
            }
        }
    }
    // ---- hand the survivors to the query's pool ------------------------------------------------
    __syncthreads();
    if ((int)*s_cnt > P.K1) scan_prune(bkey, bval, s_cnt, P.K1, Tq);
    T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = (int)*s_cnt;
    for (int base0 = 0; base0 < n; base0 += NT) {
        const int i = base0 + tid;
        const bool pass = (i < n) && bkey[i] <= T;
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(P.pool_cnt + q, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < (u32)P.poolq) {
                    P.pool_key[(size_t)q * P.poolq + slot] = bkey[i];
                    P.pool_val[(size_t)q * P.poolq + slot] = ((u64)pr << 32) | (u64)bval[i];
                }
            }
        }
    }
}

// K3 for an exact table of twice the LDS (m = 128 byte codes: 128 x 256 x 8 B = 256 KiB -- Example.java:74 names a pq_1024_128x8
// codebook): the sum over the sub-quantizers is taken in TWO sweeps over the item's codes, each with half the table in LDS.  Sweep 1
// leaves every code's partial sum over s = 0 .. M/2 - 1 (from 0.0, s ascending) in the block's slice of the global scratch that
// the table-in-global kernels use; sweep 2 continues each from there over s = M/2 .. M - 1 -- the same additions in the same order
// as one pass over a whole table (IVFPQ.java:435-438), so the same bits -- and feeds K3's candidate buffer.  The table-in-global
// form (k_scan<0, .., GLUT>) read every code byte and every table entry through global loads: 0.76 ms per 1024 queries of the
// 100 k x 1024-d shape; this one 0.1 ms.
template <int M, int NT>
__global__ __launch_bounds__(NT) void k_scan_split(const ScanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int H = M / 2;
    const int ks = P.ks, D = P.D;
    double *lut = (double *)smem;                  // [H * ks]
    double *vec = lut + (size_t)H * ks;            // [D] or [2 D]
    u64 *bkey = (u64 *)(vec + (P.transform ? 2 : 1) * (size_t)D);
    u32 *bval = (u32 *)(bkey + P.cap);
    u32 *s_cnt = bval + P.cap;
    int item = blockIdx.x;
    const int n_items = P.order ? *P.n_order : P.n_items;
    if (P.xcd_remap) {
        const int per = (n_items + 7) >> 3;
        if ((int)(blockIdx.x >> 3) >= per) return;
        item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    if (item >= n_items) return;
    int q, pr;
    if (P.order) {
        const int e = P.order[item];
        q = e / P.w;
        pr = e - q * P.w;
    } else {
        q = item / P.nrank;
        pr = P.rank_lo + (item - q * P.nrank);
    }
    const int ch = P.order_ch ? P.order_ch[item] : (P.ivf ? (int)blockIdx.y : pr);
    int cell = 0;
    if (P.ivf) {
        cell = P.cells[(size_t)q * P.w + pr];
        if (cell < 0) return;
    }
    const int64_t beg = P.list_off[cell];
    const int64_t len = P.list_off[cell + 1] - beg;
    int64_t c0 = (int64_t)ch * P.chunk;
    if (c0 >= len) return;
    int64_t c1 = (c0 + P.chunk < len) ? c0 + P.chunk : len;
    c0 = c0 < (int64_t)P.code_lo ? (int64_t)P.code_lo : c0;
    c1 = c1 > (int64_t)P.code_hi ? (int64_t)P.code_hi : c1;
    if (c0 >= c1) return;
    const int tid = threadIdx.x;
    const unsigned char *codes = (const unsigned char *)P.codes + (size_t)beg * M;
    double *part = P.glut + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)M * ks;  // [<= chunk <= M * ks] (the host checks)
    if (tid == 0) s_cnt[0] = 0;
    const double *tr = query_vector(P, q, cell, vec);
    build_lut_any(lut, tr, P.pqT, H, ks, P.dsub);
    __syncthreads();
    // ---- sweep 1: s = 0 .. H - 1
    for (int64_t i0 = c0; i0 < c1; i0 += 2 * NT) {
        CodeVec<H, unsigned char> cv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int64_t i = i0 + u * NT + tid;
            cv[u].load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
        }
        double d[2] = {0.0, 0.0};
#pragma unroll
        for (int s = 0; s < H; s++) {
#pragma unroll
            for (int u = 0; u < 2; u++) d[u] += lut[s * ks + cv[u].get(s)];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int64_t i = i0 + u * NT + tid;
            if (i < c1) part[i - c0] = d[u];
        }
    }
    __syncthreads();
    build_lut_any(lut, tr + (size_t)H * P.dsub, P.pqT + (size_t)H * P.dsub * ks, H, ks, P.dsub);
    u64 *Tq = P.T + q;
    u64 T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // ---- sweep 2: s = H .. M - 1 on top of the partial sums; K3's candidate buffer from here on
    const int limit = P.cap - NT;
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
    for (int64_t seg = c0; seg < c1; seg += NT) {
        const int64_t i = seg + tid;
        const int64_t ic = i < c1 ? i : c1 - 1;
        CodeVec<H, unsigned char> cv;
        cv.load(codes + (size_t)ic * M + H);
        double d = part[ic - c0];
#pragma unroll
        for (int s = 0; s < H; s++) d += lut[s * ks + cv.get(s)];
        const u64 key = dkey(d);
        const bool pass = (i < c1) && key <= T;
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(s_cnt, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                bkey[slot] = key;
                bval[slot] = (u32)i;
            }
        }
        __syncthreads();
        const bool need = (int)*s_cnt > limit;  // uniform: nobody writes s_cnt until the barrier below
        __syncthreads();
        if (need) {
            scan_prune(bkey, bval, s_cnt, P.K1, Tq);
            T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if ((int)*s_cnt > P.K1) scan_prune(bkey, bval, s_cnt, P.K1, Tq);
    T = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = (int)*s_cnt;
    for (int base0 = 0; base0 < n; base0 += NT) {
        const int i = base0 + tid;
        const bool pass = (i < n) && bkey[i] <= T;
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(P.pool_cnt + q, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < (u32)P.poolq) {
                    P.pool_key[(size_t)q * P.poolq + slot] = bkey[i];
                    P.pool_val[(size_t)q * P.poolq + slot] = ((u64)pr << 32) | (u64)bval[i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3h: pass A without a sorted candidate buffer (byte codes, long lists).
//
// In K3 the bookkeeping around the gather -- two block barriers per 256-code segment to agree on
// "buffer full?", and a 45-stage bitonic sort of the 512-slot buffer every ~150 accepted candidates
// -- costs more than the gather itself (measured on cfg4: 1.97 ms per 16384 lists, of which the
// table build is 0.25 ms and the lookups 0.52 ms).  K3h keeps the same exact fp64 sums but decides
// "could this code still be among the K1 best of the list?" with a histogram instead of a sort:
//
//   * segment 0 fixes a monotone map d -> bucket (MMIDX_HB uniform buckets over
//     [min0 - 0.4 (q40 - min0), q40 + 0.1 (q40 - min0)), clamped; min0 / q40 = minimum and ~40 %
//     quantile of the first 256 distances) -- the region where the K1 smallest of a long list end up;
//   * a code is a candidate iff bucket(d) <= Tb, the wave's current threshold bucket; candidates
//     bump hist[bucket] and append their list position (4 bytes) to the wave's own quarter of a
//     position buffer (count in a register: no atomic with a return value);
//   * every segment at first, then every fourth, each wave recomputes Tb = the first bucket whose
//     cumulative count reaches K1 (a 256-entry prefix sum inside the wave).  Tb only decreases, so
//     every code whose bucket is <= the final Tb was appended, whatever the interleaving of the
//     waves: NO barrier in the loop;
//   * at the end (one barrier) the final Tb is read off the complete histogram, the appended
//     positions are re-evaluated (exact sum again, same order -> same bits) and the entries with
//     bucket <= Tb -- at least K1, at most MMIDX_HKEEP, typically K1 + 2 -- go to the query's pool;
//     the largest of them is the published threshold.  No sort anywhere.
//
// A position buffer overflow costs a second pass over the list inside the block (the histogram is
// complete regardless).  Degenerate lists (more than 256 entries under the final Tb: massive
// ties, or a segment 0 that is not representative) are not handled here: the item is appended to a
// fallback list and a K3 launch over that list redoes it, so results never depend on the heuristics.
// ------------------------------------------------------------------------------------------------
// Pass A's exact tables for a whole batch, ahead of the scan: LUT[q][s][j] = sum_t (r[s dsub + t] - pq[s][j][t])^2 for the query's
// NEAREST cell, t ascending from 0.0 (computeLookupADC, IVFPQ.java:525-538) -- the same operations in the same order as build_lut.
// A block is (sub-quantizer s, QB queries); thread j holds entry j's dsub coordinates in registers and walks the queries, whose
// sub-residuals sit in LDS (broadcast reads).  A block of k_scan_hist that builds its own table re-reads the whole codebook from
// L2 -- 256 KiB per query at 16 x 256 x 8, 2 MiB at 64 x 256 x 16 (the 1024-d shape: 8.6 GB of L2 reads per 4096 queries, the
// whole exposed table build of its one-block-per-CU pass A) -- here a codebook row is read once per QB queries.
#define LUTPRE_QB 64
template <int DSUB>
__global__ __launch_bounds__(256) void k_lut_pre(const double *__restrict__ Q, const double *__restrict__ coarse, const int32_t *__restrict__ cells,
                                                 const int32_t *__restrict__ perm, const double *__restrict__ pqT, double *__restrict__ lut, int D,
                                                 int m, int ks, int w, int ivf, long long nq) {
    __shared__ double s_r[LUTPRE_QB * DSUB];
    const int s = blockIdx.x, j = threadIdx.x;
    const long long q0 = (long long)blockIdx.y * LUTPRE_QB;
    const int nb = (int)(nq - q0 < LUTPRE_QB ? nq - q0 : LUTPRE_QB);
    for (int idx = j; idx < LUTPRE_QB * DSUB; idx += 256) {
        const int qi = idx / DSUB, t = idx - qi * DSUB;
        const long long q = q0 + (qi < nb ? qi : nb - 1);
        const int d = s * DSUB + t, src = perm ? perm[d] : d;
        const double qv = Q[(size_t)q * D + src];
        int cell = ivf ? cells[(size_t)q * w] : 0;
        cell = cell < 0 ? 0 : cell;
        s_r[idx] = ivf ? coarse[(size_t)cell * D + src] - qv : qv;
    }
    double pv[DSUB];
#pragma unroll
    for (int t = 0; t < DSUB; t++) pv[t] = j < ks ? pqT[((size_t)s * DSUB + t) * ks + j] : 0.0;
    __syncthreads();
    if (j >= ks) return;
    for (int qi = 0; qi < nb; qi++) {
        const double *tv = s_r + qi * DSUB;
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < DSUB; t++) {
            const double df = tv[t] - pv[t];
            acc += df * df;
        }
        lut[((size_t)(q0 + qi) * m + s) * ks + j] = acc;
    }
}

#define MMIDX_HB 256
#define MMIDX_HKEEP 256  // most entries one item may emit (>= K1 required: the host checks; the pool has room for them)
#define MMIDX_HPOS 4     // appended positions re-evaluated per thread per round at the end
#ifndef MMIDX_K3H_SDWA
#define MMIDX_K3H_SDWA 1  // K3h: table offsets by SDWA byte select (1 VALU per lookup instead of 2)
#endif
#ifndef MMIDX_K3H_PAIR
#define MMIDX_K3H_PAIR 2  // K3h: 1 + this many segments per round of the scan loop (3 costs a block per CU)
#endif
#ifndef MMIDX_K3H_PAIR512
#define MMIDX_K3H_PAIR512 1  // the same for 512-thread blocks (80 VGPRs at six waves per SIMD)
#endif
#ifndef MMIDX_K3H_WPS512
#define MMIDX_K3H_WPS512 6
#endif
#ifndef MMIDX_HREF_LATE
#define MMIDX_HREF_LATE 12  // ... and from this round (of U segments) on every fourth
#endif
#ifndef MMIDX_HREF_EARLY
#define MMIDX_HREF_EARLY 16  // the threshold bucket is re-derived every round for this many segments, then every second round ...
#endif

// KS = 256: the usual codebook size as a compile-time constant (the table row of sub-quantizer s then sits at an
// immediate offset of the gather's ds_read instead of costing a VALU add per lookup); KS = 0: ks from the parameters.
#define MMIDX_HWV 16   // most waves per block (NT = 1024: m = 64, whose 128 KiB table leaves room for one block per CU)
#define MMIDX_HCNT (MMIDX_HWV + 4)  // counters behind the histogram: [0, HWV) appended per wave, HWV..HWV+1 largest kept key (u64), +2 kept, +3 pool base
template <int M, int KS, int NT>
__global__ __launch_bounds__(NT, NT == 512 ? MMIDX_K3H_WPS512 : 1) void k_scan_hist(const ScanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WV = NT / 64;  // waves: 4, 8 (NT = 512: three blocks per CU, six waves per SIMD over the same tables) or 16 (NT = 1024: one block per CU)
    static_assert(WV <= MMIDX_HWV, "counter layout");
    const int ks = KS > 0 ? KS : P.ks, D = P.D;
    double *lut = (double *)smem;                                        // [M*ks]
    double *vec = lut + (size_t)M * ks;                                  // [D] or [2D]
    double *s_red = vec + (P.transform ? 2 : 1) * (size_t)D;             // [2 HWV] wave minima / r-th values of segment 0
    // [HB], 16-byte aligned (read as uint4).  The offset is computed on the index, not on the pointer value: a cast
    // through uintptr_t loses the LDS address space and every access below becomes a FLAT instruction
    const size_t hist_off = ((((size_t)M * ks + (P.transform ? 2 : 1) * (size_t)D + 2 * MMIDX_HWV) * 8) + 15) & ~(size_t)15;
    u32 *hist = (u32 *)(smem + hist_off);
    u32 *s_cnt = hist + MMIDX_HB;                                        // [HCNT]
    u32 *posbuf = s_cnt + MMIDX_HCNT;                                    // [cap] list positions, an equal share per wave

    int item = blockIdx.x;
    if (P.order) {  // (a shard's pass A: the queries whose nearest list is here, k_passa_items)
        const int lim = *P.n_order < (int)gridDim.x ? *P.n_order : (int)gridDim.x;
        if (item >= lim) return;
        item = P.order[item];
    }
    if (item >= P.n_items) return;
    const int q = item / P.nrank;
    const int pr = P.rank_lo + (item - q * P.nrank);
    const int ch = P.ivf ? (int)blockIdx.y : pr;
    int cell = 0;
    if (P.ivf) {
        cell = P.cells[(size_t)q * P.w + pr];
        if (cell < 0) return;
    }
    const int64_t beg = P.list_off[cell];
    const int64_t len = P.list_off[cell + 1] - beg;
    int64_t c0 = (int64_t)ch * P.chunk;
    if (c0 >= len) return;
    int64_t c1 = (c0 + P.chunk < len) ? c0 + P.chunk : len;
    c0 = c0 < (int64_t)P.code_lo ? (int64_t)P.code_lo : c0;
    c1 = c1 > (int64_t)P.code_hi ? (int64_t)P.code_hi : c1;
    if (c0 >= c1) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned char *codes = (const unsigned char *)P.codes + (size_t)beg * M;

    CodeVec<M, unsigned char> cur, nxt;
    {
        const int64_t i = c0 + tid;
        cur.load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
    }
    for (int i = tid; i < MMIDX_HB + MMIDX_HCNT; i += NT) hist[i] = 0;  // histogram and the counters
    if (P.lut_pre) {  // the table was built ahead of the launch (k_lut_pre): 16-byte loads straight into LDS
        const double2 *src = (const double2 *)(P.lut_pre + (size_t)q * M * ks);
        for (int i = tid; i < (M * ks) >> 1; i += NT) ((double2 *)lut)[i] = src[i];
    } else {
        const double *tr = query_vector(P, q, cell, vec);
        build_lut_any(lut, tr, P.pqT, M, ks, P.dsub);
    }
    __syncthreads();

    // (0.0 + x == x bit for bit: the table entries are sums of squares from +0.0, never -0.0, so the reference's
    //  `d = 0; d += LUT[0][..]` is the first entry itself)
    // table entry of sub-quantizer s for the code's byte s.  KS = 256: the byte offset 8 * byte comes from one SDWA shift and
    // the row from the ds_read's immediate offset (the table starts the block's LDS: the kernel has no static LDS, so
    // the dynamic segment begins at LDS address 0 -- checked by the host against the function's static size)
    auto entry = [&](const CodeVec<M, unsigned char> &cv, const int s) -> double {
#if MMIDX_K3H_SDWA
        if constexpr (KS == 256) return *(lds_cdouble *)(size_t)(byte_x8(cv.wd[s >> 2], s & 3) + (u32)s * 2048u);
#endif
        return lut[s * ks + cv.get(s)];
    };
    auto exact = [&](const CodeVec<M, unsigned char> &cv) -> double {
        double d = entry(cv, 0);
#pragma unroll
        for (int s = 1; s < M; s++) d += entry(cv, s);
        return d;
    };

    // ---- segment 0: the bucket map and the first threshold bucket ---------------------------------
    // Each wave sorts its 64 distances in registers and reports its minimum and its r-th smallest,
    // r = ceil(K1 / 4).  qr = the largest of the four r-th values has at least 4 r >= K1 entries of
    // segment 0 at or below it, so bucket(qr) is a valid first threshold; and (qr - min) is a scale that
    // far outliers in the list (members of other clusters assigned to this cell) cannot stretch.
    double d = exact(cur);
    const int r_sel = (P.K1 + NT / 64 - 1) / (NT / 64);  // <= 64: the host launches K3h for K1 <= 256 only
    {
        // wmin / the r-th smallest to 2^-20 relative: distances are >= 0, so the high words of their bit patterns order them;
        // the wave bisects on that word (one compare + scalar popcount per step) instead of ranking 64 values against each
        // other (64 x two v_readlane + compares: 8 % of the kernel's VALU instructions).  Reported: the smallest pattern with
        // that high word (a lower bound of the minimum) and the largest pattern whose high word has >= r values at or under
        // it (>= r values of the wave are at or below it: all that the first threshold needs).
        const u32 hw = (c0 + tid < c1) ? (u32)(dkey(d) >> 32) : 0x7FF00000u;  // (NaN / inf / padding: above everything finite)
        u32 lo_k = wave_min_u32(hw), hi_k = wave_max_u32(hw);
        const u32 mn_k = lo_k;
        while (lo_k < hi_k) {  // smallest word with at least r_sel values at or below it
            const u32 mid = lo_k + ((hi_k - lo_k) >> 1);
            if ((int)__popcll(__builtin_amdgcn_ballot_w64(hw <= mid)) >= r_sel) hi_k = mid;
            else lo_k = mid + 1;
        }
        const bool enough = (int)__popcll(__builtin_amdgcn_ballot_w64(hw <= hi_k)) >= r_sel && hi_k < 0x7FF00000u;
        if (lane == 0) {
            s_red[wv] = mn_k < 0x7FF00000u ? keyd((u64)mn_k << 32) : __longlong_as_double(0x7FF0000000000000ll);
            s_red[WV + wv] = enough ? keyd(((u64)hi_k << 32) | 0xFFFFFFFFull) : __longlong_as_double(0x7FF0000000000000ll);
        }
    }
    __syncthreads();
    double lo, inv, qr;
    bool qr_valid = true;
    {
        const double inf = __longlong_as_double(0x7FF0000000000000ll);
        double mn = s_red[0];
        qr = s_red[WV];
#pragma unroll
        for (int i = 1; i < NT / 64; i++) {
            mn = s_red[i] < mn ? s_red[i] : mn;
            qr = s_red[WV + i] > qr ? s_red[WV + i] : qr;
        }
        if (!(qr < inf)) {  // a wave with fewer than r entries (short list): no first threshold, scale from what there is
            qr_valid = false;
            qr = -inf;
#pragma unroll
            for (int i = 0; i < NT / 64; i++) {
                const double a = s_red[WV + i];
                qr = (a < inf && a > qr) ? a : qr;
            }
            if (!(qr > -inf)) qr = mn;
        }
        const double span = qr - mn;  // >= 0
        lo = mn - 0.4 * span;
        const double width = 1.5 * span;  // buckets cover [lo, qr + 0.1 span); everything above clamps into the last
        inv = (width > 0.0 && width < 1e300) ? (double)MMIDX_HB / width : 0.0;
        if (!(inv < 1e300)) inv = 0.0;
    }
    // monotone in d: (d - lo) and the product by inv >= 0 are monotone, so is the truncation
    auto bucket = [&](double dd) -> int {
        const double x = (dd - lo) * inv;
        return x >= (double)(MMIDX_HB - 1) ? MMIDX_HB - 1 : (x > 0.0 ? (int)x : 0);
    };
    // first bucket whose cumulative count reaches K1, from this lane's four buckets hv (HB - 1 if none)
    auto threshold_bucket = [&](const uint4 hv, u32 &upto, u32 &total) -> int {
        const u32 own = hv.x + hv.y + hv.z + hv.w;
        const u32 incl = wave_incl_scan_u32(own);
        total = wave_read_u32(incl, 63);
        upto = 0xFFFFFFFFu;  // cumulative count including the returned bucket
        const u64 reached = __ballot(incl >= (u32)P.K1);
        if (!reached) return MMIDX_HB - 1;
        const int L = __ffsll((long long)reached) - 1;  // the lane whose four buckets cross K1 (wave-uniform)
        const u32 hx = wave_read_u32(hv.x, L), hy = wave_read_u32(hv.y, L), hz = wave_read_u32(hv.z, L),
                  hw = wave_read_u32(hv.w, L);
        u32 c = wave_read_u32(incl, L) - (hx + hy + hz + hw) + hx;
        int cand = 4 * L;
        if (c < (u32)P.K1) { c += hy; cand++; }
        if (c < (u32)P.K1) { c += hz; cand++; }
        if (c < (u32)P.K1) { c += hw; cand++; }
        upto = c;
        return cand;
    };

    // ---- scan: no block barrier, no LDS round trip besides the gather itself -------------------------
    // Each wave appends to its own quarter of the position buffer (count in a register, no atomic) and
    // reads its four histogram buckets TOGETHER with the segment's table lookups; the threshold bucket
    // derived from them applies from the next segment on (a stale histogram only errs on the safe side).
    const u64 lane_lt = (1ull << lane) - 1ull;
    const u32 capw = (u32)P.cap / (NT / 64);
    u32 *mybuf = posbuf + (size_t)wv * capw;
    u32 wcnt = 0;  // wave-uniform
    int Tb = qr_valid ? bucket(qr) : MMIDX_HB - 1;
    // The loop is VALU-bound as much as LDS-bound (PMC: VALU ~78 % busy, LDS ~73 %), so its bookkeeping is kept in
    // scalar registers and 32-bit arithmetic: the list bounds are made wave-uniform explicitly (they come from vector
    // loads), positions are relative to c0 (chunk <= 2^24 codes: the host checks), the bucket clamp is two fp64
    // min / max, and the ballot is the compare mask itself.
    const u32 n_seg = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(c1 - c0));
    const unsigned char *codes0;  // (rebuilt from the kernel argument: a pointer made from integers would be FLAT-addressed)
    {
        const u64 e0 = (u64)(beg + c0);
        const u32 lo32 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)e0), hi32 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(e0 >> 32));
        codes0 = (const unsigned char *)P.codes + (size_t)(((u64)hi32 << 32) | lo32) * M;
    }
    // U = MMIDX_K3H_PAIR + 1 segments per round: their table gathers are independent chains (U times the LDS requests in
    // flight per wave) and the round's bookkeeping -- loop control, prefetch addresses, the threshold refresh -- is paid
    // once.  Segment 0 (already summed into d) is candidate-tested first, on its own.
    constexpr int U = (M >= 64 ? 0 : (NT >= 512 ? MMIDX_K3H_PAIR512 : MMIDX_K3H_PAIR)) + 1;  // (m = 64: a code is sixteen registers)
    // bucket(dd) <= Tb  <=>  (dd - lo) * inv < Tb + 1 (the clamp to [0, HB - 1] cannot change the comparison unless
    // Tb = HB - 1, where everything passes: the bound is NaN then and `!(x >= NaN)` holds).  Two fp64 operations and a compare
    // per code; the bucket itself (clamp, convert) is computed by the candidates only.  A NaN distance passes and lands in
    // the last bucket, which is harmless: appended entries above the final bucket are dropped at the end.
    double TbP1 = Tb >= MMIDX_HB - 1 ? __longlong_as_double(0x7FF8000000000000ll) : (double)(Tb + 1);
    auto offer = [&](const double dd, const u32 pos) {
        const double x = (dd - lo) * inv;
        const bool pass = (pos < n_seg) && !(x >= TbP1);
        const u64 mask = __builtin_amdgcn_ballot_w64(pass);
        if (pass) {
            const int b = (int)__builtin_fmax(__builtin_fmin(x, (double)(MMIDX_HB - 1)), 0.0);
            atomicAdd(hist + b, 1u);
            const u32 slot = wcnt + (u32)__popcll(mask & lane_lt);
            if (slot < capw) mybuf[slot] = ((u32)b << 24) | pos;
        }
        wcnt += (u32)__popcll(mask);  // > capw: overflow, seen at the end
    };
    offer(d, (u32)tid);
    CodeVec<M, unsigned char> cu[U], nx[U];
    auto fetch = [&](CodeVec<M, unsigned char> &dst, const u32 i) {
        const u32 ic = i < n_seg ? i : n_seg - 1u;
        dst.load(codes0 + ic * (u32)M);  // (32-bit offset from a uniform base)
    };
    if (NT < n_seg) {
#pragma unroll
        for (int u = 0; u < U; u++) fetch(cu[u], (u32)(1 + u) * NT + (u32)tid);
    }
    u32 g = 1;
    for (u32 seg = NT; seg < n_seg; seg += U * NT, g++) {
        const bool more = seg + U * NT < n_seg;  // scalar
        if (more) {
#pragma unroll
            for (int u = 0; u < U; u++) fetch(nx[u], seg + (u32)(U + u) * NT + (u32)tid);
        }
        const bool refresh = (g < MMIDX_HREF_EARLY / U || (g < MMIDX_HREF_LATE ? (g & 1) == 0 : (g & 3) == 0));
        uint4 hv;  // live only on refresh rounds
        if (refresh) hv = ((const uint4 *)hist)[lane];  // buckets 4*lane .. 4*lane+3
        double dd[U];
#pragma unroll
        for (int u = 0; u < U; u++) dd[u] = entry(cu[u], 0);
#pragma unroll
        for (int sq = 1; sq < M; sq++) {
#pragma unroll
            for (int u = 0; u < U; u++) dd[u] += entry(cu[u], sq);
        }
#pragma unroll
        for (int u = 0; u < U; u++) offer(dd[u], seg + (u32)u * NT + (u32)tid);
        if (refresh) {
            u32 upto, total;
            const int cand = threshold_bucket(hv, upto, total);
            Tb = cand < Tb ? cand : Tb;
            TbP1 = Tb >= MMIDX_HB - 1 ? __longlong_as_double(0x7FF8000000000000ll) : (double)(Tb + 1);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; u++) cu[u] = nx[u];
        }
    }
    if (lane == 0) s_cnt[wv] = wcnt;
    __syncthreads();
    // ---- final threshold bucket from the complete histogram (identical in every wave) ------------
    u32 kept_total;
    {
        u32 upto, total;
        Tb = threshold_bucket(((const uint4 *)hist)[lane], upto, total);  // fewer than K1 candidates in all: every bucket counts
        kept_total = (upto == 0xFFFFFFFFu) ? total : upto;
    }
    u32 woff[NT / 64 + 1];
    bool overflow = false;  // block-uniform (read after the barrier)
    woff[0] = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const u32 c = s_cnt[i];
        overflow |= c > capw;
        woff[i + 1] = woff[i] + (c > capw ? capw : c);
    }
    if (kept_total > (u32)MMIDX_HKEEP) {  // block-uniform: redo this item with K3
        if (tid == 0) {
            const u32 slot = atomicAdd(P.fb_count, 1u);
            P.fb_items[slot] = q * P.w + pr;
            P.fb_ch[slot] = ch;
        }
        return;
    }
    // Every entry under the final bucket goes to the query's pool (unordered; K4 sorts it) and the
    // largest of them is published as the threshold: at least K1 list entries are <= it.  kept_total
    // exceeds K1 by about half a bucket's population (~2 on cfg4), so the bound is as good as the exact
    // K1-th smallest, without any sort.
    u64 *Tq = P.T + q;
    const u64 Tg = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // other chunks / shards may have published one
    u64 kmax = 0;
    auto keep_entry = [&](bool keep, double dd, u32 p) {
        const u64 key = dkey(dd);
        if (keep) kmax = key > kmax ? key : kmax;
        const bool emit = keep && key <= Tg;
        const u64 mask = __ballot(emit);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader) base = atomicAdd(P.pool_cnt + q, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (emit) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < (u32)P.poolq) {
                    P.pool_key[(size_t)q * P.poolq + slot] = key;
                    P.pool_val[(size_t)q * P.poolq + slot] = ((u64)pr << 32) | (u64)p;
                }
            }
        }
    };
    if (!overflow) {
        // ---- the appended entries carry their bucket: pick those under the final one (each wave its own
        //      region), then evaluate only them -- kept_total <= HKEEP = one per thread -------------------
        u32 *keptpos = hist;  // the histogram is dead (every wave has derived the final bucket): barrier first
        __syncthreads();
        // the pool space of all kept_total entries is reserved now, so that the reservation's round trip overlaps the
        // compaction and the code loads (entries above another chunk's threshold are written too: harmless)
        if (tid == 0) s_cnt[MMIDX_HWV + 3] = atomicAdd(P.pool_cnt + q, kept_total);
        const u32 mine = s_cnt[wv];  // <= capw here
        for (u32 e0 = 0; e0 < mine; e0 += 64) {
            const u32 e = e0 + (u32)lane;
            const u32 v = e < mine ? mybuf[e] : 0xFFFFFFFFu;
            const bool keep = e < mine && (int)(v >> 24) <= Tb;
            const u64 mask = __ballot(keep);
            if (mask) {
                u32 base = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if (lane == leader) base = atomicAdd(s_cnt + MMIDX_HWV + 2, (u32)__popcll(mask));
                base = wave_read_u32(base, leader);
                if (keep) keptpos[base + (u32)__popcll(mask & lane_lt)] = v & 0xFFFFFFu;
            }
        }
        __syncthreads();
        const bool have = (u32)tid < kept_total;  // == s_cnt[HWV + 2]
        const u32 p = have ? (u32)c0 + keptpos[tid] : (u32)c0;
        CodeVec<M, unsigned char> cv;
        cv.load(codes + (size_t)p * M);
        const double dd = exact(cv);
        const u64 key = dkey(dd);
        if (have) {
            kmax = key;
            const u32 slot = s_cnt[MMIDX_HWV + 3] + (u32)tid;  // (written before the barrier above)
            if (slot < (u32)P.poolq) {
                P.pool_key[(size_t)q * P.poolq + slot] = key;
                P.pool_val[(size_t)q * P.poolq + slot] = ((u64)pr << 32) | (u64)p;
            }
        }
    } else {
        // ---- the position buffer overflowed (the histogram is still complete): second pass over the
        //      list with the final threshold bucket ------------------------------------------------
        {
            const int64_t i = c0 + tid;
            cur.load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
        }
        for (int64_t seg = c0; seg < c1; seg += NT) {
            const bool more = seg + NT < c1;
            if (more) {
                const int64_t i = seg + NT + tid;
                nxt.load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
            }
            const double dd = exact(cur);
            const int64_t i = seg + tid;
            keep_entry((i < c1) && bucket(dd) <= Tb, dd, (u32)i);
            if (more) cur = nxt;
        }
    }
    // ---- threshold: the largest kept key, if the list had K1 entries at all ----------------------
    if (kept_total >= (u32)P.K1) {  // block-uniform
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 o = __shfl_xor(kmax, off);
            kmax = o > kmax ? o : kmax;
        }
        u64 *s_max = (u64 *)(s_cnt + MMIDX_HWV);  // zeroed with the histogram, 8-byte aligned
        if (lane == 0) atomicMax(s_max, kmax);
        __syncthreads();
        if (tid == 0) atomicMin(Tq, *s_max);
    }
}

// ------------------------------------------------------------------------------------------------
// K3f: filtered scan (pass B, byte codes).  The fp64 gather of K3 is LDS-bank-conflict bound
// (random 8-byte reads over 2 KiB rows: ~3x serialisation, measured).  K3f first runs a
// conflict-free LOWER-BOUND filter and evaluates the exact fp64 sum only for the survivors:
//
//   with T the query's current threshold, min_s = min_j LUT[s][j], Smin = sum_s min_s and
//   delta = (T - Smin) / 254:      q8[s][j] = min(255, floor((LUT[s][j] - min_s) / delta))
//   sum_s q8[s][code[s]] * delta + Smin <= d(code), so a code with sum_s q8 >= 256 has d > T and
//   can be dropped; Smin >= T drops the whole list before a single code is read.
//
// A q8 row is 256 bytes = exactly one pass over the 64 LDS banks, and ds_read_b64 serves 8-byte
// slots: two lanes either hit the same slot (broadcast) or different bank pairs -> no conflicts.
// The wanted byte is picked from the 8-byte slot with v_perm_b32.  Survivors (typically < 1 % of
// the codes) are queued in LDS and verified 256 at a time with the exact sequential fp64 sum from
// the fp64 LUT, so results stay bit-identical to K3.  All roundings of the filter are directed so
// that it can only under-estimate (never drops a candidate with d <= T).
// ------------------------------------------------------------------------------------------------
#define MMIDX_FSEGU 2  // codes per thread per segment in the filtered scan (4 measured no faster: the per-item
                       // fixed cost -- LUT build, minima, q8 table -- dominates, not the lookup loop)
#define MMIDX_FSEG (MMIDX_BLOCK * MMIDX_FSEGU)
#define MMIDX_SURV_CAP 1024  // >= FSEG + VROUND
#define MMIDX_VROUND MMIDX_BLOCK

template <int DSUB>
__device__ __forceinline__ double lut_entry(const double *tr, const double *__restrict__ pqT, int s, int j, int ks,
                                            int dsub_rt) {
    double acc = 0.0;
    if constexpr (DSUB > 0) {
        const double *pp = pqT + (size_t)s * DSUB * ks + j;
        double pv[DSUB];
#pragma unroll
        for (int t = 0; t < DSUB; t++) pv[t] = pp[(size_t)t * ks];
        const double *tv = tr + s * DSUB;
#pragma unroll
        for (int t = 0; t < DSUB; t++) {
            const double df = tv[t] - pv[t];
            acc += df * df;
        }
    } else {
        const double *pp = pqT + (size_t)s * dsub_rt * ks + j;
        const double *tv = tr + s * dsub_rt;
        for (int t = 0; t < dsub_rt; t++) {
            const double df = tv[t] - pp[(size_t)t * ks];
            acc += df * df;
        }
    }
    return acc;
}

// one work item of K3f; vb = virtual block index
template <int M>
__device__ __forceinline__ void scan_filt_body(const ScanParams &P, const unsigned vb, const int n_items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ks = P.ks, D = P.D;
    double *lut = (double *)smem;                        // [M*ks]   exact fp64 table
    double *vec = lut + (size_t)M * ks;                  // [2*D]
    u64 *bkey = (u64 *)(vec + 2 * (size_t)D);            // [cap]    candidates: distance bits
    u64 *s_min = bkey + P.cap;                           // [4][M]   per-wave minima, then [0..M) final
    u64 *s_T = s_min + 4 * M;                            // [2]      block-uniform copy of the threshold
    u32 *bval = (u32 *)(s_T + 2);                        // [cap]    candidates: list position
    u32 *surv = bval + P.cap;                            // [SURV_CAP] filter survivors: list position
    u32 *s_cnt = surv + MMIDX_SURV_CAP;                  // [4]: 0 candidates, 1 survivors
    unsigned char *lut8 = (unsigned char *)(s_cnt + 4);  // [M][256] quantised lower-bound table

    int item = (int)vb;
    if (P.xcd_remap) {
        const int per = (n_items + 7) >> 3;
        if ((int)(vb >> 3) >= per) return;
        item = (int)(vb & 7) * per + (int)(vb >> 3);
    }
    if (item >= n_items) return;
    int q, pr;
    if (P.order) {
        const int e = P.order[item];
        q = e / P.w;
        pr = e - q * P.w;
    } else {
        q = item / P.nrank;
        pr = P.rank_lo + (item - q * P.nrank);
    }
    // flat PQ: the item rank is the chunk of the single list; order_ch: (pair, chunk) items handed back by K3g
    const int ch = (P.order && P.order_ch) ? P.order_ch[item] : (P.ivf ? (int)blockIdx.y : pr);
    int cell = 0;
    if (P.ivf) {
        cell = P.cells[(size_t)q * P.w + pr];
        if (cell < 0) return;
    }
    const int64_t beg = P.list_off[cell];
    const int64_t len = P.list_off[cell + 1] - beg;
    int64_t c0 = (int64_t)ch * P.chunk;
    if (c0 >= len) return;
    int64_t c1 = (c0 + P.chunk < len) ? c0 + P.chunk : len;
    c0 = c0 < (int64_t)P.code_lo ? (int64_t)P.code_lo : c0;
    c1 = c1 > (int64_t)P.code_hi ? (int64_t)P.code_hi : c1;
    if (c0 >= c1) return;
    const int tid = threadIdx.x;
    const unsigned char *codes = (const unsigned char *)P.codes + (size_t)beg * M;
    u64 *Tq = P.T + q;

    CodeVec<M, unsigned char> cur[MMIDX_FSEGU], nxt[MMIDX_FSEGU];
#pragma unroll
    for (int u = 0; u < MMIDX_FSEGU; u++) {
        const int64_t i = c0 + u * MMIDX_BLOCK + tid;
        cur[u].load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
    }
    if (tid < 2) s_cnt[tid] = 0;
    // every decision that steers control flow uses ONE copy of T (another block may publish a new
    // value at any time): thread 0 reads it, the block reads the LDS copy after a barrier
    if (tid == 0) s_T[0] = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double *tr = query_vector(P, q, cell, vec);

    // ---- exact LUT column j = tid, per-sub-quantizer minima ---------------------------------------
    {
        u64 mk[M];
        const bool hasj = tid < ks;
#pragma unroll
        for (int s = 0; s < M; s++) {
            double v = 0.0;
            if (hasj) {
                switch (P.dsub) {
                    case 4: v = lut_entry<4>(tr, P.pqT, s, tid, ks, 4); break;
                    case 8: v = lut_entry<8>(tr, P.pqT, s, tid, ks, 8); break;
                    case 16: v = lut_entry<16>(tr, P.pqT, s, tid, ks, 16); break;
                    default: v = lut_entry<0>(tr, P.pqT, s, tid, ks, P.dsub); break;
                }
                lut[s * ks + tid] = v;
            }
            mk[s] = hasj ? dkey(v) : MMIDX_KEY_MAX;
        }
#pragma unroll
        for (int s = 0; s < M; s++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const u64 o = __shfl_xor(mk[s], off);
                mk[s] = o < mk[s] ? o : mk[s];
            }
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int s = 0; s < M; s++) s_min[(tid >> 6) * M + s] = mk[s];
        }
    }
    __syncthreads();
    if (tid < M) {  // final minimum of sub-quantizer tid -> s_min[tid] (read partials first)
        u64 a = s_min[tid];
        for (int wv = 1; wv < MMIDX_BLOCK / 64; wv++) {
            const u64 b = s_min[wv * M + tid];
            a = b < a ? b : a;
        }
        __builtin_amdgcn_s_waitcnt(0);
        s_min[tid] = a;  // row 0 of the partials; rows 1..3 are dead now
    }
    __syncthreads();
    double smin = 0.0;
#pragma unroll
    for (int s = 0; s < M; s++) smin += keyd(s_min[s]);
    // directed roundings: smin_lo <= the real sum of minima; 1/delta slightly small -> q8 under-estimates
    const double smin_lo = smin * (1.0 - 0x1p-40);
    u64 T = s_T[0];
    u64 T_built = MMIDX_KEY_MAX;  // threshold the current lut8 was built for
    bool exhausted = false;       // no remaining code of this list can reach the threshold

    // (re)build the quantised table for threshold Tn; block-uniform. Returns false when the list is
    // exhausted (Smin >= T).
    auto rebuild = [&](u64 Tn) -> bool {
        const bool finiteT = Tn < 0x7FF0000000000000ull;
        const double Td = keyd(Tn);
        if (finiteT && !(smin_lo < Td)) return false;
        // (a threshold in fp64's denormal range makes 254 / (T - Smin) overflow: no filter for such an item either)
        const bool nofilter = !finiteT || !((Td - smin_lo) > Td * 0x1p-30) || !(254.0 / (Td - smin_lo) < 1e300);
        if (tid < ks) {
            const double inv = nofilter ? 0.0 : (254.0 / (Td - smin_lo)) * (1.0 - 0x1p-40);
#pragma unroll
            for (int s = 0; s < M; s++) {
                const double x = (lut[s * ks + tid] - keyd(s_min[s])) * inv;
                const u32 qv = (x >= 255.0) ? 255u : (u32)x;  // x >= 0, finite
                lut8[s * 256 + tid] = (unsigned char)(nofilter ? 0u : qv);
            }
        }
        __syncthreads();
        return true;
    };
    if (!rebuild(T)) return;
    T_built = T;

    // ---- filter scan ------------------------------------------------------------------------------
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
    for (int64_t seg = c0; seg < c1 && !exhausted; seg += MMIDX_FSEG) {
        const bool more = seg + MMIDX_FSEG < c1;
        if (more) {
#pragma unroll
            for (int u = 0; u < MMIDX_FSEGU; u++) {
                const int64_t i = seg + MMIDX_FSEG + u * MMIDX_BLOCK + tid;
                nxt[u].load(codes + (size_t)(i < c1 ? i : c1 - 1) * M);
            }
        }
        u32 acc[MMIDX_FSEGU];
        uint2 va[MMIDX_FSEGU], vb[MMIDX_FSEGU];
        // software pipeline over the sub-quantizers: the slots of s+1 are requested before the bytes
        // of s are extracted, so every v_perm waits on reads issued a full row earlier (counted
        // lgkmcnt instead of a drain)
#pragma unroll
        for (int u = 0; u < MMIDX_FSEGU; u++) {
            acc[u] = 0;
            va[u] = *(const uint2 *)(lut8 + ((u32)cur[u].get(0) & 0xF8u));
        }
#pragma unroll
        for (int s = 0; s < M; s++) {
            if (s + 1 < M) {
#pragma unroll
                for (int u = 0; u < MMIDX_FSEGU; u++)
                    vb[u] = *(const uint2 *)(lut8 + (s + 1) * 256 + ((u32)cur[u].get(s + 1) & 0xF8u));
            }
#pragma unroll
            for (int u = 0; u < MMIDX_FSEGU; u++)
                acc[u] += __builtin_amdgcn_perm(va[u].y, va[u].x, ((u32)cur[u].get(s) & 7u) | 0x0C0C0C00u);
#pragma unroll
            for (int u = 0; u < MMIDX_FSEGU; u++) va[u] = vb[u];
        }
#pragma unroll
        for (int u = 0; u < MMIDX_FSEGU; u++) {
            const int64_t i = seg + u * MMIDX_BLOCK + tid;
            const bool pass = (i < c1) && acc[u] <= 255u;
            const u64 mask = __ballot(pass);
            if (mask) {
                u32 base = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if ((tid & 63) == leader) base = atomicAdd(s_cnt + 1, (u32)__popcll(mask));
                base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                if (pass) surv[base + (u32)__popcll(mask & lane_lt)] = (u32)i;
            }
        }
        __syncthreads();
        int ns = (int)s_cnt[1];  // uniform: no writer until the barrier below
        __syncthreads();
        // ---- exact verification of the queued survivors, MMIDX_VROUND at a time ------------------
        while (ns >= MMIDX_VROUND || (!more && ns > 0)) {
            if ((int)s_cnt[0] > P.cap - MMIDX_VROUND) {  // uniform (s_cnt[0] stable here)
                scan_prune(bkey, bval, s_cnt, P.K1, Tq);
                if (tid == 0) s_T[0] = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                T = s_T[0];
                if (T < T_built) {  // tighter threshold: sharper filter for the rest of the list
                    if (!rebuild(T)) {
                        exhausted = true;  // queued survivors still need their exact check below
                    }
                    T_built = T;
                }
            }
            const int take = ns < MMIDX_VROUND ? ns : MMIDX_VROUND;
            const int base_s = ns - take;
            bool pass = false;
            u64 key = 0;
            u32 pos = 0;
            if (tid < take) {
                pos = surv[base_s + tid];
                CodeVec<M, unsigned char> cv;
                cv.load(codes + (size_t)pos * M);
                double d = 0.0;
#pragma unroll
                for (int s = 0; s < M; s++) d += lut[s * ks + cv.get(s)];
                key = dkey(d);
                pass = key <= T;
            }
            const u64 mask = __ballot(pass);
            if (mask) {
                u32 base = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if ((tid & 63) == leader) base = atomicAdd(s_cnt, (u32)__popcll(mask));
                base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                if (pass) {
                    const u32 slot = base + (u32)__popcll(mask & lane_lt);
                    bkey[slot] = key;
                    bval[slot] = pos;
                }
            }
            ns = base_s;
            __syncthreads();
            if (tid == 0) s_cnt[1] = (u32)ns;
            __syncthreads();
            if (exhausted && ns > 0) continue;  // drain the queue, then leave the segment loop
        }
        if (exhausted) {
            // survivors below one verification round are still queued: verify them too
            while (ns > 0) {
                const int take = ns < MMIDX_VROUND ? ns : MMIDX_VROUND;
                const int base_s = ns - take;
                if ((int)s_cnt[0] > P.cap - MMIDX_VROUND) scan_prune(bkey, bval, s_cnt, P.K1, Tq);
                bool pass = false;
                u64 key = 0;
                u32 pos = 0;
                if (tid < take) {
                    pos = surv[base_s + tid];
                    CodeVec<M, unsigned char> cv;
                    cv.load(codes + (size_t)pos * M);
                    double d = 0.0;
#pragma unroll
                    for (int s = 0; s < M; s++) d += lut[s * ks + cv.get(s)];
                    key = dkey(d);
                    pass = key <= T;
                }
                const u64 mask = __ballot(pass);
                if (mask) {
                    u32 base = 0;
                    const int leader = __ffsll((long long)mask) - 1;
                    if ((tid & 63) == leader) base = atomicAdd(s_cnt, (u32)__popcll(mask));
                    base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                    if (pass) {
                        const u32 slot = base + (u32)__popcll(mask & lane_lt);
                        bkey[slot] = key;
                        bval[slot] = pos;
                    }
                }
                ns = base_s;
                __syncthreads();
                if (tid == 0) s_cnt[1] = (u32)ns;
                __syncthreads();
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < MMIDX_FSEGU; u++) cur[u] = nxt[u];
        }
    }
    // ---- hand the survivors to the query's pool ------------------------------------------------
    __syncthreads();
    if ((int)*s_cnt > P.K1) scan_prune(bkey, bval, s_cnt, P.K1, Tq);
    const u64 Tfin = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // per-lane use only
    const int n = (int)*s_cnt;
    for (int base0 = 0; base0 < n; base0 += MMIDX_BLOCK) {
        const int i = base0 + tid;
        const bool pass = (i < n) && bkey[i] <= Tfin;
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(P.pool_cnt + q, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < (u32)P.poolq) {
                    P.pool_key[(size_t)q * P.poolq + slot] = bkey[i];
                    P.pool_val[(size_t)q * P.poolq + slot] = ((u64)pr << 32) | (u64)bval[i];
                }
            }
        }
    }
}

// K3f launches.  Pass B's item count is only known on the device (what the coarse bound left), and a grid sized
// for the worst case costs 0.1 ms of pure block dispatch when -- as on the benchmark -- nothing is left.  So the
// host sizes the main launch from the count it saw last time (a hint, read back asynchronously), and a small grid
// of looping blocks covers whatever lies beyond it.  Virtual block vb -> item is the same map in both.
template <int M>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_scan_filt(const ScanParams P) {
    const int n_items = P.order ? *P.n_order : P.n_items;  // pass B: count left by the coarse bound
    scan_filt_body<M>(P, P.vb_base + blockIdx.x, n_items);
}
// (the loop costs registers -- the optimiser keeps far more live across it -- which is why it is not the main kernel)
template <int M>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_scan_filt_tail(const ScanParams P) {
    const int n_items = P.order ? *P.n_order : P.n_items;
    const unsigned vlim = P.xcd_remap ? (unsigned)(((n_items + 7) >> 3) << 3) : (unsigned)n_items;
    const unsigned vend = vlim < P.vgrid ? vlim : P.vgrid;
    for (unsigned vb = P.vb_base + blockIdx.x; vb < vend; vb += gridDim.x) {
        scan_filt_body<M>(P, vb, n_items);
        __syncthreads();  // LDS is reused by the next item
    }
}

// ------------------------------------------------------------------------------------------------
// K3s: seeded scan for pass A (probe rank 0: no threshold exists yet).  Same exact results as K3,
// but only ~0.5 K codes pay the fp64 gather:
//   stage 0  exact distances of the first 512 codes of the chunk -> T0 = their (k+1)-th smallest
//            (a valid threshold, but loose: a ~20 % quantile of the list)
//   stage 1  conflict-free u8 sums (table built for T0) for ALL codes of the chunk, kept in LDS
//            (1 byte per code), plus a 256-bin histogram of the sums <= 254
//   stage 2  the histogram gives b* = smallest sum with at least k+1 codes at or below it; every such
//            code has d <= Smin + (b* + m) * delta (each of the m truncations loses < 1), so
//            T1 = that bound is a valid and far tighter threshold (within m * delta of the true one)
//   stage 3  codes whose stored sum can still reach min(T0, T1) -- a few hundred -- get the exact
//            fp64 sum; the rest of the block is K3's prune / pool hand-off.
// ------------------------------------------------------------------------------------------------
#define MMIDX_SEED_N0 512
#define MMIDX_SEGU_S 8  // codes per thread per streaming segment of stage 1: pass A streams cold lists from HBM,
                       // so two segments (32 KiB per block) are kept in flight
#define MMIDX_SEG_S (MMIDX_BLOCK * MMIDX_SEGU_S)
template <int M>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_scan_seed(const ScanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ks = P.ks, D = P.D;
    double *lut = (double *)smem;                        // [M*ks]
    double *vec = lut + (size_t)M * ks;                  // [2*D]
    u64 *bkey = (u64 *)(vec + 2 * (size_t)D);            // [cap] cap >= K1 + 512
    u64 *s_min = bkey + P.cap;                           // [4][M]
    u64 *s_T = s_min + 4 * M;                            // [2]
    u32 *bval = (u32 *)(s_T + 2);                        // [cap]
    u32 *surv = bval + P.cap;                            // [SURV_CAP]
    u32 *hist = surv + MMIDX_SURV_CAP;                   // [256]
    u32 *s_cnt = hist + 256;                             // [4]: 0 candidates, 1 survivors, 2 b*
    unsigned char *lut8 = (unsigned char *)(s_cnt + 4);  // [M][256]
    unsigned char *sums = lut8 + M * 256;                // [chunk] u8 lower-bound sums

    const int item = blockIdx.x;
    if (item >= P.n_items) return;
    const int q = item / P.nrank;
    const int pr = P.rank_lo + (item - q * P.nrank);
    const int ch = P.ivf ? (int)blockIdx.y : pr;  // flat PQ: the item rank is the chunk of the single list
    int cell = 0;
    if (P.ivf) {
        cell = P.cells[(size_t)q * P.w + pr];
        if (cell < 0) return;
    }
    const int64_t beg = P.list_off[cell];
    const int64_t len = P.list_off[cell + 1] - beg;
    int64_t c0 = (int64_t)ch * P.chunk;
    if (c0 >= len) return;
    int64_t c1 = (c0 + P.chunk < len) ? c0 + P.chunk : len;
    c0 = c0 < (int64_t)P.code_lo ? (int64_t)P.code_lo : c0;
    c1 = c1 > (int64_t)P.code_hi ? (int64_t)P.code_hi : c1;
    if (c0 >= c1) return;
    const int clen = (int)(c1 - c0);
    const int tid = threadIdx.x;
    const unsigned char *codes = (const unsigned char *)P.codes + (size_t)(beg + c0) * M;  // chunk-relative
    u64 *Tq = P.T + q;
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;

    CodeVec<M, unsigned char> cur[MMIDX_SEGU_S], nxt[MMIDX_SEGU_S];
#pragma unroll
    for (int u = 0; u < MMIDX_SEGU_S; u++) {
        const int i = u * MMIDX_BLOCK + tid;
        cur[u].load(codes + (size_t)(i < clen ? i : clen - 1) * M);
    }
    if (tid < 4) s_cnt[tid] = 0;
    hist[tid] = 0;
    const double *tr = query_vector(P, q, cell, vec);
    {
        u64 mk[M];
        const bool hasj = tid < ks;
#pragma unroll
        for (int s = 0; s < M; s++) {
            double v = 0.0;
            if (hasj) {
                switch (P.dsub) {
                    case 4: v = lut_entry<4>(tr, P.pqT, s, tid, ks, 4); break;
                    case 8: v = lut_entry<8>(tr, P.pqT, s, tid, ks, 8); break;
                    case 16: v = lut_entry<16>(tr, P.pqT, s, tid, ks, 16); break;
                    default: v = lut_entry<0>(tr, P.pqT, s, tid, ks, P.dsub); break;
                }
                lut[s * ks + tid] = v;
            }
            mk[s] = hasj ? dkey(v) : MMIDX_KEY_MAX;
        }
#pragma unroll
        for (int s = 0; s < M; s++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const u64 o = __shfl_xor(mk[s], off);
                mk[s] = o < mk[s] ? o : mk[s];
            }
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int s = 0; s < M; s++) s_min[(tid >> 6) * M + s] = mk[s];
        }
    }
    __syncthreads();
    if (tid < M) {
        u64 a = s_min[tid];
        for (int wv = 1; wv < MMIDX_BLOCK / 64; wv++) {
            const u64 b = s_min[wv * M + tid];
            a = b < a ? b : a;
        }
        s_min[tid] = a;
    }
    __syncthreads();
    double smin = 0.0;
#pragma unroll
    for (int s = 0; s < M; s++) smin += keyd(s_min[s]);
    const double smin_lo = smin * (1.0 - 0x1p-40), smin_hi = smin * (1.0 + 0x1p-40);

    // ---- stage 0: exact distances of the first n0 codes -> candidates -> T0 ---------------------
    const int n0 = clen < MMIDX_SEED_N0 ? clen : MMIDX_SEED_N0;
#pragma unroll
    for (int u = 0; u < MMIDX_SEGU; u++) {
        const int i = u * MMIDX_BLOCK + tid;
        if (i < n0) {
            double d = 0.0;
#pragma unroll
            for (int s = 0; s < M; s++) d += lut[s * ks + cur[u].get(s)];
            bkey[i] = dkey(d);
            bval[i] = (u32)(c0 + i);
        }
    }
    if (tid == 0) s_cnt[0] = (u32)n0;
    __syncthreads();
    if (n0 > P.K1) scan_prune(bkey, bval, s_cnt, P.K1, Tq);  // publishes T0 when n0 >= K1
    if (tid == 0) s_T[0] = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    u64 T = s_T[0];
    const bool finiteT = T < 0x7FF0000000000000ull;
    bool run_filter = clen > n0;
    // (if the whole chunk was covered, or no threshold exists and none can be built, skip ahead)
    double inv = 0.0, Td = keyd(T);
    bool usable = finiteT && (smin_lo < Td) && ((Td - smin_lo) > Td * 0x1p-30) && (254.0 / (Td - smin_lo) < 1e300);
    if (run_filter && finiteT && !(smin_lo < Td)) run_filter = false;  // nothing else can qualify
    if (run_filter) {
        inv = usable ? (254.0 / (Td - smin_lo)) * (1.0 - 0x1p-40) : 0.0;
        if (tid < ks) {
#pragma unroll
            for (int s = 0; s < M; s++) {
                const double x = (lut[s * ks + tid] - keyd(s_min[s])) * inv;
                const u32 qv = (x >= 255.0) ? 255u : (u32)x;
                lut8[s * 256 + tid] = (unsigned char)(usable ? qv : 0u);
            }
        }
        __syncthreads();
        // ---- stage 1: u8 sums of every code of the chunk + histogram -------------------------------
        for (int seg = 0; seg < clen; seg += MMIDX_SEG_S) {
            const bool more = seg + MMIDX_SEG_S < clen;
            if (more) {
#pragma unroll
                for (int u = 0; u < MMIDX_SEGU_S; u++) {
                    const int i = seg + MMIDX_SEG_S + u * MMIDX_BLOCK + tid;
                    nxt[u].load(codes + (size_t)(i < clen ? i : clen - 1) * M);
                }
            }
#pragma unroll
            for (int h2 = 0; h2 < MMIDX_SEGU_S; h2 += 2) {  // two codes at a time: ILP without 8 live accumulators
                u32 acc0 = 0, acc1 = 0;
#pragma unroll
                for (int s = 0; s < M; s++) {
                    const u32 b0 = (u32)cur[h2].get(s), b1 = (u32)cur[h2 + 1].get(s);
                    const uint2 v0 = *(const uint2 *)(lut8 + s * 256 + (b0 & 0xF8u));
                    const uint2 v1 = *(const uint2 *)(lut8 + s * 256 + (b1 & 0xF8u));
                    acc0 += __builtin_amdgcn_perm(v0.y, v0.x, (b0 & 7u) | 0x0C0C0C00u);
                    acc1 += __builtin_amdgcn_perm(v1.y, v1.x, (b1 & 7u) | 0x0C0C0C00u);
                }
                const int i0 = seg + h2 * MMIDX_BLOCK + tid, i1 = i0 + MMIDX_BLOCK;
                if (i0 < clen) {
                    sums[i0] = (unsigned char)(acc0 > 255u ? 255u : acc0);
                    if (usable && i0 >= n0 && acc0 <= 254u) atomicAdd(hist + acc0, 1u);
                }
                if (i1 < clen) {
                    sums[i1] = (unsigned char)(acc1 > 255u ? 255u : acc1);
                    if (usable && i1 >= n0 && acc1 <= 254u) atomicAdd(hist + acc1, 1u);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < MMIDX_SEGU_S; u++) cur[u] = nxt[u];
            }
        }
        __syncthreads();
        // ---- stage 2: b* and the tighter threshold ---------------------------------------------------
        u32 thr = 255u;  // codes with sum <= thr survive; 255 = everything (no usable filter)
        if (usable) {
            if (tid == 0) {
                u32 cum = 0;
                int bstar = -1;
                for (int b = 0; b < 255; b++) {
                    cum += hist[b];
                    if ((int)cum >= P.K1) {
                        bstar = b;
                        break;
                    }
                }
                u64 Tn = T;
                if (bstar >= 0) {
                    const double delta_hi = ((Td - smin_lo) / 254.0) * (1.0 + 0x1p-38);  // >= 1 / inv
                    const double t1 = (smin_hi + (double)(bstar + M) * delta_hi) * (1.0 + 0x1p-40);
                    const u64 k1 = dkey(t1);
                    if (k1 < Tn) Tn = k1;
                }
                atomicMin(Tq, Tn);
                s_T[0] = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            T = s_T[0];
            // d <= T needs  sum <= (T - smin_lo) * inv  (the sum times 1/inv plus smin_lo is a lower bound)
            const double lim = (keyd(T) - smin_lo) * inv;
            thr = lim >= 254.0 ? 254u : (lim < 0.0 ? 0u : (u32)lim + 1u);
            if (thr > 254u) thr = 254u;
        }
        // ---- stage 3: exact verification of the codes that can still qualify -------------------------
        for (int seg = n0; seg < clen; seg += MMIDX_SEG) {
            const bool more = seg + MMIDX_SEG < clen;
#pragma unroll
            for (int u = 0; u < MMIDX_SEGU; u++) {
                const int i = seg + u * MMIDX_BLOCK + tid;
                const bool pass = (i < clen) && (u32)sums[i] <= thr;
                const u64 mask = __ballot(pass);
                if (mask) {
                    u32 base = 0;
                    const int leader = __ffsll((long long)mask) - 1;
                    if ((tid & 63) == leader) base = atomicAdd(s_cnt + 1, (u32)__popcll(mask));
                    base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                    if (pass) surv[base + (u32)__popcll(mask & lane_lt)] = (u32)i;
                }
            }
            __syncthreads();
            int ns = (int)s_cnt[1];
            __syncthreads();
            while (ns >= MMIDX_VROUND || (!more && ns > 0)) {
                if ((int)s_cnt[0] > P.cap - MMIDX_VROUND) {
                    scan_prune(bkey, bval, s_cnt, P.K1, Tq);
                    if (tid == 0) s_T[0] = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    T = s_T[0];
                }
                const int take = ns < MMIDX_VROUND ? ns : MMIDX_VROUND;
                const int base_s = ns - take;
                bool pass = false;
                u64 key = 0;
                u32 pos = 0;
                if (tid < take) {
                    pos = surv[base_s + tid];
                    CodeVec<M, unsigned char> cv;
                    cv.load(codes + (size_t)pos * M);
                    double d = 0.0;
#pragma unroll
                    for (int s = 0; s < M; s++) d += lut[s * ks + cv.get(s)];
                    key = dkey(d);
                    pass = key <= T;
                }
                const u64 mask = __ballot(pass);
                if (mask) {
                    u32 base = 0;
                    const int leader = __ffsll((long long)mask) - 1;
                    if ((tid & 63) == leader) base = atomicAdd(s_cnt, (u32)__popcll(mask));
                    base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                    if (pass) {
                        const u32 slot = base + (u32)__popcll(mask & lane_lt);
                        bkey[slot] = key;
                        bval[slot] = (u32)(c0 + pos);
                    }
                }
                ns = base_s;
                __syncthreads();
                if (tid == 0) s_cnt[1] = (u32)ns;
                __syncthreads();
            }
        }
    }
    // ---- hand the survivors to the query's pool ------------------------------------------------
    __syncthreads();
    if ((int)*s_cnt > P.K1) scan_prune(bkey, bval, s_cnt, P.K1, Tq);
    const u64 Tfin = __hip_atomic_load(Tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // per-lane use only
    const int n = (int)*s_cnt;
    for (int base0 = 0; base0 < n; base0 += MMIDX_BLOCK) {
        const int i = base0 + tid;
        const bool pass = (i < n) && bkey[i] <= Tfin;
        const u64 mask = __ballot(pass);
        if (mask) {
            u32 base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if ((tid & 63) == leader) base = atomicAdd(P.pool_cnt + q, (u32)__popcll(mask));
            base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
            if (pass) {
                const u32 slot = base + (u32)__popcll(mask & lane_lt);
                if (slot < (u32)P.poolq) {
                    P.pool_key[(size_t)q * P.poolq + slot] = bkey[i];
                    P.pool_val[(size_t)q * P.poolq + slot] = ((u64)pr << 32) | (u64)bval[i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-batch ordering of the (query, probe rank >= rank_lo) pairs by cell: counting sort with
// atomics.  The order inside a cell is arbitrary -- it only decides which block runs when.
// ------------------------------------------------------------------------------------------------
// Coarse bound (exact pruning of whole probes before any LUT is built).  The ADC distance of a
// code in cell c is ||(c - q) - p||^2 with p the concatenation of the chosen sub-centroids, so by
// the reverse triangle inequality d >= (||c - q|| - Rmax)^2 whenever ||c - q|| > Rmax, where
// Rmax^2 = sum_s max_j ||pq[s][j]||^2 bounds ||p|| (a permutation keeps norms; the bound is not
// used with a rotation matrix, which is an input and need not be exactly orthogonal).  A pair whose
// bound exceeds the query's threshold T (valid after pass A) cannot contribute and is left out of
// pass B.  Margins make the test conservative against fp64 rounding.
struct PairBound {
    const double *Q;       // [nq][D]
    const double *coarse;  // [C][D]
    const u64 *T;          // [nq]
    const double *cdist;   // [nq][C] coarse distances when this process computed them, else null
    const double *cdsel;   // [nq][w] exact distances of the selected cells (approximate coarse path), else null
    const int64_t *list_off;  // [C+1] pairs whose list is empty here (other shards' lists) are dropped
    double rmax;           // sqrt(sum_s max_j ||pq[s][j]||^2) * (1 + 1e-12)
    int D, C;
    int enabled;
    double shrink;         // RandomRotation: lengths shrink by at most this factor under the (nearly orthogonal) matrix; 1 otherwise
};
// Written branch-free on purpose: with early returns hipcc (ROCm 7.2) sank the zero-extension of
// the cell index into a divergent region and the later atomicAdd(cnt + c) used a garbage high
// dword on the lanes that had returned early (memory aperture violation).
__device__ __forceinline__ bool pair_keep(const PairBound &B, long long e, int q, int c) {
    const u64 T = B.T[q];
    const bool nonempty = B.list_off[c + 1] > B.list_off[c];
    double cd = 0.0;
    if (B.cdsel) {  // K1d left the exact distance of every selected cell, in probe order
        cd = B.cdsel[e];
    } else if (B.cdist) {  // K1a computed ||c - q||^2 for every (q, c)
        cd = B.cdist[(size_t)q * B.C + c];
    } else {
        const double *cc = B.coarse + (size_t)c * B.D, *qq = B.Q + (size_t)q * B.D;
        const int Dn = nonempty ? B.D : 0;  // (a trip count, not an early return: see above)
        for (int j = 0; j < Dn; j++) {
            const double df = cc[j] - qq[j];
            cd += df * df;
        }
    }
    const double r = sqrt(cd) * (1.0 - 1e-12) * B.shrink;
    const double gap = r - B.rmax;
    const double lb = gap * gap * (1.0 - 1e-9);
    const bool prune = (B.enabled != 0) & (T < 0x7FF0000000000000ull) & (gap > 0.0) & (lb > keyd(T));
    return nonempty & !prune;
}

// one launch instead of four memsets at the start of a search step: thresholds = +inf (all ones), pool counters,
// per-cell pair counters and the K3h fallback header = 0
__global__ void k_step_init(u64 *__restrict__ T, u32 *__restrict__ pool_cnt, long long nq, int32_t *__restrict__ pcount, int C,
                            int32_t *__restrict__ fb_header) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) {
        T[i] = MMIDX_KEY_MAX;
        pool_cnt[i] = 0;
    }
    if (pcount && i <= C) pcount[i] = 0;  // ([C] = number of surviving pairs of the step)
    if (fb_header && i < 4) fb_header[i] = 0;
}
// Pass A on a shard: most queries' nearest cell lives on another rank, and a block per query that only finds an empty
// list still costs its dispatch (0.1 ms per 114688 of them).  This lists the queries whose nearest list is non-empty
// here -- order[0 .. count) -- so that K3h is launched over about nq / world blocks; the grid is sized for the expected
// count plus a margin, and whatever does not fit is appended to K3h's hand-back list (the K3 launch behind it).
__global__ __launch_bounds__(MMIDX_BLOCK) void k_passa_items(const int32_t *__restrict__ cells, int w, const int64_t *__restrict__ list_off,
                                                             long long nq, int grid_main, int32_t *__restrict__ order,
                                                             int32_t *__restrict__ count, u32 *__restrict__ fb_count,
                                                             int32_t *__restrict__ fb_items, int32_t *__restrict__ fb_ch) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool own = false;
    if (q < nq) {
        const int cell = cells[(size_t)q * w];
        own = cell >= 0 && list_off[cell + 1] > list_off[cell];
    }
    const u64 mask = __builtin_amdgcn_ballot_w64(own);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    u32 base = 0;
    if (lane == leader) base = (u32)atomicAdd(count, (int)__popcll(mask));
    base = wave_read_u32(base, leader);
    if (own) {
        const u32 slot = base + (u32)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot < (u32)grid_main) {
            order[slot] = (int32_t)q;
        } else {
            const u32 f = atomicAdd(fb_count, 1u);
            fb_items[f] = (int32_t)(q * w);  // (query, probe rank 0)
            fb_ch[f] = 0;
        }
    }
}
// cdist_out[q][r] = dist[q][cells[q][r]] (exact coarse path: the selected cells' distances for other ranks)
__global__ void k_gather_cdist(const double *__restrict__ dist, const int32_t *__restrict__ cells, double *__restrict__ out, int C, int w,
                               long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = cells[e];
    out[e] = c >= 0 ? dist[(size_t)(e / w) * C + c] : 0.0;
}
// cand != null: the kept pairs are not counted per cell yet -- they are appended to cand[0 .. *ncand) (any order) for K3s
// (k_pair_smin, mmidx_scan_grp.h), and k_pair_recount counts what that leaves
__global__ void k_pair_hist(const int32_t *__restrict__ cells, int w, int rank_lo, long long npairs,
                            int32_t *__restrict__ cnt, unsigned char *__restrict__ keep, const PairBound B,
                            int32_t *__restrict__ cand, int32_t *__restrict__ ncand) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npairs) return;
    const int q = (int)(e / w);
    if ((int)(e - (long long)q * w) < rank_lo) return;
    const int c = cells[e];
    const unsigned cu = c >= 0 ? (unsigned)c : 0u;
    const bool k = (c >= 0) & pair_keep(B, e, q, (int)cu);
    keep[e] = k ? 1 : 0;
    const u64 mk = __builtin_amdgcn_ballot_w64(k);
    const int lane = (int)(threadIdx.x & 63);
    const int leader = mk ? __ffsll((long long)mk) - 1 : 0;
    if (cand) {
        u32 base = 0;
        if (mk && lane == leader) base = (u32)atomicAdd(ncand, (int)__popcll(mk));
        base = wave_read_u32(base, leader);
        const u32 slot = base + (u32)__popcll(mk & ((1ull << lane) - 1ull));
        if (k) cand[slot] = (int32_t)e;
        return;
    }
    if (k) atomicAdd(cnt + (size_t)cu, 1);
    // total of the step in cnt[C]: one atomic per wave that kept anything (none at all on separable data, where the
    // kernels behind this one then leave at once)
    if (mk && lane == leader) atomicAdd(cnt + B.C, (int)__popcll(mk));
}
// single block: exclusive scan of cnt[C] -> start[C]; start[C] = total; cursor zeroed
// (host_hint: pinned host word that receives the total as well -- the next call sizes pass B's launch from it; written
//  from here it needs no copy node in the stream)
__global__ __launch_bounds__(1024) void k_pair_scan(const int32_t *__restrict__ cnt, int C, int32_t *__restrict__ start,
                                                    int32_t *__restrict__ cursor, int32_t *host_hint) {
    __shared__ u32 s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (cnt[C] == 0) {  // nothing survived the coarse bound: no order to build (k_pair_scatter keeps nothing either)
        if (tid == 0) {
            start[C] = 0;
            if (host_hint) *host_hint = 0;
        }
        return;
    }
    const int per = (C + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < C) ? lo + per : C;
    u32 sum = 0;
    for (int c = lo; c < hi; c++) sum += (u32)cnt[c];
    const u32 incl = wave_incl_scan_u32(sum);  // inside the wave: DPP, no barrier
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) base += (i < wv) ? s_wave[i] : 0u;
    u32 run = base + incl - sum;  // exclusive prefix of this thread's range
    for (int c = lo; c < hi; c++) {
        start[c] = (int32_t)run;
        cursor[c] = 0;
        run += (u32)cnt[c];
    }
    if (tid == 1023) {
        start[C] = (int32_t)(base + incl);
        if (host_hint) *host_hint = (int32_t)(base + incl);
    }
}
__global__ void k_pair_scatter(const int32_t *__restrict__ cells, int w, int rank_lo, long long npairs,
                               const int32_t *__restrict__ start, int32_t *__restrict__ cursor,
                               int32_t *__restrict__ order, const unsigned char *__restrict__ keep) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npairs) return;
    if ((int)(e % w) < rank_lo) return;
    if (!keep[e]) return;
    const int c = cells[e];
    const int pos = start[c] + atomicAdd(cursor + c, 1);
    order[pos] = (int32_t)e;
}

// ------------------------------------------------------------------------------------------------
// K4: per-query merge of the candidate pool: exact k+1 smallest by (distance, offer order), then
// ASS.lookUp order (ASS:345-358): ascending distance, equal distances later-offered first.
// mode 0: final results (iid / dist / count / tie flag); mode 1: sorted partial list for the
// cross-shard merge (pdist, pkey = probe_rank << 32 | iid, pcount).
// ------------------------------------------------------------------------------------------------
#define MMIDX_MCAP 2048  // smallest K5 buffer (entries); the host doubles it until it exceeds 2 (k + 1)
#define MMIDX_K_MAX 4095  // largest k: the candidate buffers (pow2 >= k + 1 + segment) and the merge buffers (pow2 >= 2 (k + 1)) must fit
                          // the 160 KiB LDS next to the lookup table
struct ShardDest {
    double *pd;
    long long *pk;
    int32_t *pc;
};
struct MergeParams {
    const u64 *T;            // [nq] final thresholds (null: no filtering)
    const u32 *pool_cnt;
    const u64 *pool_key;
    const u64 *pool_val;
    int poolq;
    const int32_t *cells;    // [nq][w] or null
    const int64_t *list_off;
    const int32_t *ids;
    int w, k, mode;
    int cap;                 // LDS entries (power of two, > k + 1)
    int32_t *iid_out;        // [nq][k]
    double *dist_out;        // [nq][k]
    int32_t *count_out;      // [nq]
    int32_t *flag_out;       // [nq] tie straddles k (mode 0)
    double *pdist;           // [nq][k+1] (mode 1)
    long long *pkey;         // [nq][k+1]
    // mode 1 inside a sharded handle (mmidx_create_sharded): query q belongs to shard q / dest_per, and its partial list goes
    // straight into THAT shard's receive buffers -- dest[o].pd / .pk laid out [n_shards][dest_per][k+1], dest[o].pc
    // [n_shards][dest_per], row (dest_me, q % dest_per) -- by stores over xGMI (peer access; the same device for virtual
    // shards).  Only the valid entries travel.  dest == null: the dense local arrays above.
    const ShardDest *dest;
    int dest_per, dest_me;
};

// NT = threads per block: 128 when k + 1 <= 128 (the survivors fit one per thread and twice as many queries are in
// flight per CU: the kernel is a chain of dependent memory round trips), else 256
template <int NT>
__global__ __launch_bounds__(NT) void k_merge(const MergeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int CAP = P.cap;           // power of two, > K1
    u64 *key = (u64 *)smem;          // [CAP]
    u64 *val = key + CAP;            // [CAP]
    int64_t *s_off = (int64_t *)(val + CAP);  // [w] start of every probed list (the id lookups at the end)
    const int q = blockIdx.x, tid = threadIdx.x;
    const int K1 = P.k + 1;
    __shared__ u32 s_m;
    int n = (int)P.pool_cnt[q];
    if (n > P.poolq) n = P.poolq;
    const u64 *pk = P.pool_key + (size_t)q * P.poolq;
    const u64 *pv = P.pool_val + (size_t)q * P.poolq;
    // Entries above the query's final threshold cannot be among the K1 best (at least K1 candidates are at or
    // below it): drop them while loading.  What is left is K1 plus a few entries, small enough to be ordered
    // by counting ranks -- one pass of broadcast LDS reads, no 36-stage sorting network.
    // where this query's partial list goes (mode 1): the local dense arrays, or the owner shard's receive buffers
    double *o_pd = P.pdist;
    long long *o_pk = P.pkey;
    int32_t *o_pc = P.count_out + q;
    if (P.mode == 1) {
        size_t row = (size_t)q * K1;
        if (P.dest) {
            const int o = q / P.dest_per, ql = q - o * P.dest_per;
            const ShardDest d = P.dest[o];
            row = ((size_t)P.dest_me * P.dest_per + ql) * K1;
            o_pd = d.pd;
            o_pk = d.pk;
            o_pc = d.pc + (size_t)P.dest_me * P.dest_per + ql;
        }
        o_pd += row;
        o_pk += row;
    }
    if (n == 0 && P.mode == 1) {  // nothing on this shard for this query (the usual case for most queries of a rank):
        if (tid == 0) *o_pc = 0;  // the merge reads count entries, the list itself stays unwritten
        return;
    }
    const u64 T = P.T ? __hip_atomic_load(P.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : MMIDX_KEY_MAX;
    if (tid == 0) s_m = 0;
    const bool offs_in_lds = P.cells && P.w <= 1024;  // (the host sized the LDS accordingly)
    if (offs_in_lds)
        for (int t = tid; t < P.w; t += NT) s_off[t] = P.list_off[P.cells[(size_t)q * P.w + t]];
    auto list_start = [&](int rank) -> int64_t {
        if (offs_in_lds) return s_off[rank];
        return P.list_off[P.cells ? P.cells[(size_t)q * P.w + rank] : 0];
    };
    __syncthreads();
    {
        const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
        for (int base0 = 0; base0 < n; base0 += NT) {
            const int e = base0 + tid;
            u64 kk = MMIDX_KEY_MAX, vv = 0;
            if (e < n) {  // both loads in flight together
                kk = pk[e];
                vv = pv[e];
            }
            const bool pass = e < n && kk <= T;
            const u64 mask = __ballot(pass);
            if (mask) {
                u32 base = 0;
                const int leader = __ffsll((long long)mask) - 1;
                if ((tid & 63) == leader) base = atomicAdd(&s_m, (u32)__popcll(mask));
                base = wave_read_u32(base, leader);  // (v_readlane: no trip through the LDS pipe)
                if (pass) {
                    const u32 slot = base + (u32)__popcll(mask & lane_lt);
                    if (slot < (u32)CAP) {
                        key[slot] = kk;
                        val[slot] = vv;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int m = (int)s_m;
    int kept = 0;
    if (m <= NT) {
        const bool have = tid < m;
        const u64 mk = have ? key[tid] : MMIDX_KEY_MAX, mv = have ? val[tid] : MMIDX_KEY_MAX;
        // rank = entries that precede mine in (key, offer order).  Equal keys are rare, so the first pass reads and
        // compares the keys only (one broadcast LDS read and half the VALU work per entry) and counts the equal ones
        // (itself included); only a wave that saw a tie reads the offer orders as well.
        int rank = 0, eqc = 0;
        for (int j = 0; j < m; j++) {
            const u64 ok = key[j];
            rank += ok < mk;
            eqc += ok == mk;
        }
        if (__builtin_amdgcn_ballot_w64(have && eqc > 1)) {  // wave-uniform
            rank = 0;
            for (int j = 0; j < m; j++) {
                const u64 ok = key[j], ov = val[j];
                rank += (ok < mk) || (ok == mk && ov < mv);
            }
        }
        __syncthreads();  // every entry has been read
        if (have) {
            key[rank] = mk;
            val[rank] = mv;
        }
        __syncthreads();
        kept = m < K1 ? m : K1;
    } else if (m <= CAP) {
        const int Pn = pow2ceil(m);
        for (int i = m + tid; i < Pn; i += NT) {
            key[i] = MMIDX_KEY_MAX;
            val[i] = MMIDX_KEY_MAX;
        }
        block_bitonic_sort<u64>(key, val, Pn);
        kept = m < K1 ? m : K1;
    } else {
        // more survivors than the buffer holds: chunked selection over the whole pool
        int consumed = 0;
        do {
            int take = n - consumed;
            if (take > CAP - kept) take = CAP - kept;
            __syncthreads();
            for (int i = tid; i < take; i += NT) {
                key[kept + i] = pk[consumed + i];
                val[kept + i] = pv[consumed + i];
            }
            const int filled = kept + take;
            const int Pn = pow2ceil(filled < 2 ? 2 : filled);
            for (int i = filled + tid; i < Pn; i += NT) {
                key[i] = MMIDX_KEY_MAX;
                val[i] = MMIDX_KEY_MAX;
            }
            block_bitonic_sort<u64>(key, val, Pn);
            consumed += take;
            kept = filled < K1 ? filled : K1;
        } while (consumed < n);
    }
    const int total = kept;  // min(n, K1), sorted by (key, offer order)
    const int cnt = total < P.k ? total : P.k;
    if (P.mode == 1) {
        const int nwrite = P.dest ? total : K1;  // (over xGMI only what the merge will read)
        for (int i = tid; i < nwrite; i += NT) {
            double dd = __longlong_as_double(0x7FF0000000000000ll);
            long long kk = -1;
            if (i < total) {
                const u64 v = val[i];
                const int rank = (int)(v >> 32);
                const u32 pos = (u32)v;
                const int iid = P.ids[list_start(rank) + pos];
                dd = keyd(key[i]);
                kk = (long long)(((u64)rank << 32) | (u32)iid);
            }
            o_pd[i] = dd;
            o_pk[i] = kk;
        }
        if (tid == 0) *o_pc = total;
        return;
    }
    for (int i = tid; i < P.k; i += NT) {
        int iid = -1;
        double dd = __longlong_as_double(0x7FF0000000000000ll);
        if (i < cnt) {
            // run of equal distances [a, b] inside the first cnt entries: reverse it
            int a = i, b = i;
            const u64 ki = key[i];
            while (a > 0 && key[a - 1] == ki) a--;
            while (b + 1 < cnt && key[b + 1] == ki) b++;
            const int src = a + (b - i);
            const u64 v = val[src];
            const int rank = (int)(v >> 32);
            const u32 pos = (u32)v;
            iid = P.ids[list_start(rank) + pos];
            dd = keyd(ki);
        }
        P.iid_out[(size_t)q * P.k + i] = iid;
        P.dist_out[(size_t)q * P.k + i] = dd;
    }
    if (tid == 0) {
        P.count_out[q] = cnt;
        P.flag_out[q] = (total > P.k && key[P.k - 1] == key[P.k]) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// K5: cross-shard merge of sorted partial lists [nshards][nq][K1] -> final results.  Offer order
// across shards is (probe_rank, iid): inside one inverted list the reference appends in iid order
// (IVFPQ.java:339, :699-700), so this equals the single-queue offer order.
// ------------------------------------------------------------------------------------------------
// NT = threads per block: 128 when everything of a query fits one entry per thread (as in K4), else 256.
// nflag_out (may be null): number of flagged queries of the launch (what the sharded handle reads back to decide
// whether the tie replay has to run).
template <int NT>
__global__ __launch_bounds__(NT) void k_merge_partials(int k, int nq, int nshards, int mcap,
                                                       const double *__restrict__ pdist,
                                                       const long long *__restrict__ pkey,
                                                       const int32_t *__restrict__ pcount,
                                                       const long long *__restrict__ poff,
                                                       int32_t *iid_out, double *dist_out,
                                                       int32_t *count_out, int32_t *flag_out, int32_t *nflag_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *key = (u64 *)smem;
    u64 *val = key + mcap;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int K1 = k + 1;
    int kept = 0;
    for (int s = 0; s < nshards;) {
        // pack as many shards as fit
        int filled = kept;
        while (s < nshards) {
            int c = pcount[(size_t)s * nq + q];
            if (c > K1) c = K1;
            if (filled + c > mcap) break;
            // dense [nshards][nq][K1] or, with poff, ragged: list (s, q) starts at poff[s * nq + q]
            const size_t base = poff ? (size_t)poff[(size_t)s * nq + q] : ((size_t)s * nq + q) * K1;
            for (int i = tid; i < c; i += NT) {
                key[filled + i] = dkey(pdist[base + i]);
                val[filled + i] = (u64)pkey[base + i];
            }
            filled += c;
            s++;
        }
        if (kept == 0 && s == nshards && filled <= NT) {
            // the usual case (phase 2 drops everything above the global threshold: about k + 1 entries over all
            // shards): one entry per thread, ordered by counting ranks as in K4 -- keys only unless a wave sees a tie
            __syncthreads();
            const bool have = tid < filled;
            const u64 mk = have ? key[tid] : MMIDX_KEY_MAX, mv = have ? val[tid] : MMIDX_KEY_MAX;
            int rank = 0, eqc = 0;
            for (int j = 0; j < filled; j++) {
                const u64 ok = key[j];
                rank += ok < mk;
                eqc += ok == mk;
            }
            if (__builtin_amdgcn_ballot_w64(have && eqc > 1)) {  // wave-uniform
                rank = 0;
                for (int j = 0; j < filled; j++) {
                    const u64 ok = key[j], ov = val[j];
                    rank += (ok < mk) || (ok == mk && ov < mv);
                }
            }
            __syncthreads();  // every entry has been read
            if (have) {
                key[rank] = mk;
                val[rank] = mv;
            }
            __syncthreads();
        } else {
            const int Pn = pow2ceil(filled < 2 ? 2 : filled);
            for (int i = filled + tid; i < Pn; i += NT) {
                key[i] = MMIDX_KEY_MAX;
                val[i] = MMIDX_KEY_MAX;
            }
            block_bitonic_sort<u64>(key, val, Pn);
        }
        kept = filled < K1 ? filled : K1;
    }
    const int cnt = kept < k ? kept : k;
    for (int i = tid; i < k; i += NT) {
        int iid = -1;
        double dd = __longlong_as_double(0x7FF0000000000000ll);
        if (i < cnt) {
            int a = i, b = i;
            const u64 ki = key[i];
            while (a > 0 && key[a - 1] == ki) a--;
            while (b + 1 < cnt && key[b + 1] == ki) b++;
            iid = (int)(u32)val[a + (b - i)];
            dd = keyd(ki);
        }
        iid_out[(size_t)q * k + i] = iid;
        dist_out[(size_t)q * k + i] = dd;
    }
    if (tid == 0) {
        count_out[q] = cnt;
        // a tie that straddles position k: the bounded queue's replay decides which of the equal candidates stay
        // (mmidx_shard_tie_phase_device); everything strictly better than the k-th distance is already final
        const int fl = (kept > k && key[k - 1] == key[k]) ? 1 : 0;
        if (flag_out) flag_out[q] = fl;
        if (fl && nflag_out) atomicAdd(nflag_out, 1);
    }
}

// dense partial lists [nq][K1] -> ragged: the pcount[q] valid entries of list q at out[poff[q] ...] (one wave per query)
__global__ __launch_bounds__(MMIDX_BLOCK) void k_compact_partials(const double *__restrict__ pdist, const long long *__restrict__ pkey,
                                                                  const int32_t *__restrict__ pcount, const long long *__restrict__ poff,
                                                                  double *__restrict__ out_d, long long *__restrict__ out_k, int K1,
                                                                  long long nq) {
    const long long q = (long long)blockIdx.x * (MMIDX_BLOCK / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int c = pcount[q] < K1 ? pcount[q] : K1;
    const long long o = poff[q];
    for (int i = threadIdx.x & 63; i < c; i += 64) {
        out_d[o + i] = pdist[(size_t)q * K1 + i];
        out_k[o + i] = pkey[(size_t)q * K1 + i];
    }
}

// ------------------------------------------------------------------------------------------------
// K4b: exact replay for queries whose k-th and (k+1)-th distances are equal (tie straddling the
// queue boundary).  One block per flagged query re-scans the probed lists in the reference's offer
// order and applies the bounded queue's closed form: with tau the k-th distance,
// b = #(d < tau) (already final, sorted, in the first b output slots), p = ties among the first k
// offers with d <= tau, e = b - (k - p): the kept ties are those with tie rank e..p-1 in offer
// order.  Output tail = kept ties, later-offered first.
// ------------------------------------------------------------------------------------------------
struct TieParams {
    ScanParams S;
    const int32_t *flag;   // [nq]
    const int32_t *ids;
    int32_t *iid_out;      // [nq][k]
    double *dist_out;      // [nq][k]
    int k;
    int nq;
};

template <typename CodeT, bool SDC, bool GLUT = false>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_tie_resolve(const TieParams TP) {
    const ScanParams &P = TP.S;
    const int tid = threadIdx.x, k = TP.k;
    // A block reads the flags of MMIDX_BLOCK queries at once and replays the flagged ones (rare): a block per query that only reads a
    // zero flag was 52 us per 131072 queries, and so was a block walking its queries one dependent flag load at a time.
    __shared__ int s_fl[MMIDX_BLOCK];
    __shared__ int s_nfl;
    for (int qb = (int)blockIdx.x * MMIDX_BLOCK; qb < TP.nq; qb += (int)gridDim.x * MMIDX_BLOCK) {
    if (tid == 0) s_nfl = 0;
    __syncthreads();
    if (qb + tid < TP.nq && TP.flag[qb + tid]) s_fl[atomicAdd(&s_nfl, 1)] = qb + tid;
    __syncthreads();
    const int nfl = s_nfl;
    for (int fi = 0; fi < nfl; fi++) {
    const int q = s_fl[fi];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, ks = P.ks;
    double *lut, *vec;
    if constexpr (GLUT) {  // (table in the block's slice of the global scratch, see k_scan)
        lut = P.glut + (size_t)blockIdx.x * (size_t)m * ks;
        vec = (double *)smem;
    } else {
        lut = (double *)smem;
        vec = lut + (size_t)m * ks;
    }
    __shared__ int s_wsum[MMIDX_BLOCK / 64][2];
    __shared__ int s_state[4];  // nonjunk, ties, p (or -1), emitted
    const u64 tau = dkey(TP.dist_out[(size_t)q * k + (k - 1)]);
    int b = 0;
    while (b < k && dkey(TP.dist_out[(size_t)q * k + b]) < tau) b++;  // uniform per thread
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
    const int nprobe = P.ivf ? P.w : 1;
    int e = 0, pfin = 0;
    for (int pass = 0; pass < 2; pass++) {
        if (tid == 0) {
            s_state[0] = 0;
            s_state[1] = 0;
            s_state[2] = -1;
            s_state[3] = 0;
        }
        __syncthreads();
        bool done = false;
        for (int pr = 0; pr < nprobe && !done; pr++) {
            int cell = 0;
            if (P.ivf) {
                cell = P.cells[(size_t)q * P.w + pr];
                if (cell < 0) break;
            }
            const int64_t beg = P.list_off[cell];
            const int64_t len = P.list_off[cell + 1] - beg;
            if (len == 0) continue;
            __syncthreads();
            const double *TT = nullptr;
            if constexpr (SDC) {
                TT = P.sdc_tt + (size_t)q * m * ks * P.dsub;
            } else {
                const double *tr = query_vector(P, q, cell, vec);
                build_lut_any(lut, tr, P.pqT, m, ks, P.dsub);
            }
            __syncthreads();
            const CodeT *codes = (const CodeT *)P.codes + (size_t)beg * m;
            for (int64_t base = 0; base < len && !done; base += MMIDX_BLOCK) {
                const int64_t i = base + tid;
                bool nonjunk = false, tie = false;
                if (i < len) {
                    const CodeT *cp = codes + (size_t)i * m;
                    double a = 0.0;
                    if constexpr (SDC) {
                        for (int s = 0; s < m; s++) {
                            const double *tt = TT + ((size_t)s * ks + (int)cp[s]) * P.dsub;
                            for (int t = 0; t < P.dsub; t++) a += tt[t];
                        }
                    } else {
                        for (int s = 0; s < m; s++) a += lut[s * ks + (int)cp[s]];
                    }
                    const u64 key = dkey(a);
                    nonjunk = key <= tau;
                    tie = key == tau;
                }
                // block-wide exclusive prefix of (nonjunk, tie) in offer order
                const u64 mj = __ballot(nonjunk), mt = __ballot(tie);
                const int wave = tid >> 6;
                if ((tid & 63) == 0) {
                    s_wsum[wave][0] = __popcll(mj);
                    s_wsum[wave][1] = __popcll(mt);
                }
                __syncthreads();
                int pj = s_state[0], pt = s_state[1];
                for (int wv = 0; wv < wave; wv++) {
                    pj += s_wsum[wv][0];
                    pt += s_wsum[wv][1];
                }
                pj += __popcll(mj & lane_lt);
                pt += __popcll(mt & lane_lt);
                if (pass == 0) {
                    // the k-th non-junk offer fixes p = ties offered up to and including it
                    if (nonjunk && pj + 1 == k) s_state[2] = pt + (tie ? 1 : 0);
                } else {
                    if (tie && pt >= e && pt < pfin) {
                        // kept tie with rank pt: later-offered first -> slot k-1-(pt-e)
                        const int slot = k - 1 - (pt - e);
                        TP.iid_out[(size_t)q * k + slot] = TP.ids[beg + i];
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    int tj = 0, tt = 0;
                    for (int wv = 0; wv < MMIDX_BLOCK / 64; wv++) {
                        tj += s_wsum[wv][0];
                        tt += s_wsum[wv][1];
                    }
                    s_state[0] += tj;
                    s_state[1] += tt;
                }
                __syncthreads();
                if (pass == 0 && s_state[2] >= 0) done = true;
                if (pass == 1 && s_state[1] >= pfin) done = true;
            }
        }
        __syncthreads();
        if (pass == 0) {
            pfin = s_state[2];
            e = b - (k - pfin);
        }
        __syncthreads();
    }
    __syncthreads();  // (the static state and the table are reused by the next query)
    }
    __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K6a: coarse assignment for encoding.  cell[v] = argmin_c sum_j (coarse[c][j]-x[j])^2, first
// index wins ties (computeNearestCoarseIndex IVFPQ.java:547-564; the early break at :554 cannot
// change the argmin because partial sums are non-decreasing).
// ------------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_assign_coarse(const double *__restrict__ coarseT,
                                                               const double *__restrict__ X,
                                                               int32_t *__restrict__ cell_out,
                                                               int C, int D, long long n) {
    __shared__ u64 s_k[MMIDX_BLOCK / 64][QT];
    __shared__ int s_i[MMIDX_BLOCK / 64][QT];
    const long long v0 = (long long)blockIdx.x * QT;
    u64 bk[QT];
    int bi[QT];
#pragma unroll
    for (int t = 0; t < QT; t++) {
        bk[t] = MMIDX_KEY_MAX;
        bi[t] = 0x7fffffff;
    }
    for (int c = threadIdx.x; c < C; c += MMIDX_BLOCK) {
        double acc[QT];
#pragma unroll
        for (int t = 0; t < QT; t++) acc[t] = 0.0;
        for (int j = 0; j < D; j++) {
            const double cj = coarseT[(size_t)j * C + c];
#pragma unroll
            for (int t = 0; t < QT; t++) {
                const long long v = (v0 + t < n) ? v0 + t : n - 1;
                const double df = cj - X[(size_t)v * D + j];
                acc[t] += df * df;
            }
        }
#pragma unroll
        for (int t = 0; t < QT; t++) {
            const u64 k = dkey(acc[t]);
            if (k < bk[t]) {  // strict: first index wins inside a thread (c ascending)
                bk[t] = k;
                bi[t] = c;
            }
        }
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < QT; t++) {
        u64 k = bk[t];
        int i = bi[t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 ok = __shfl_xor(k, off);
            const int oi = __shfl_xor(i, off);
            if (ok < k || (ok == k && oi < i)) {
                k = ok;
                i = oi;
            }
        }
        if ((threadIdx.x & 63) == 0) {
            s_k[wave][t] = k;
            s_i[wave][t] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x < QT && v0 + threadIdx.x < n) {
        const int t = threadIdx.x;
        u64 k = s_k[0][t];
        int i = s_i[0][t];
        for (int wv = 1; wv < MMIDX_BLOCK / 64; wv++) {
            if (s_k[wv][t] < k || (s_k[wv][t] == k && s_i[wv][t] < i)) {
                k = s_k[wv][t];
                i = s_i[wv][t];
            }
        }
        // Double.MAX_VALUE start + strict '<' : a vector whose every distance is >= MAX_VALUE
        // (inf / NaN) gets -1 in the reference
        cell_out[v0 + t] = (k < dkey(1.7976931348623157e308)) ? i : -1;
    }
}

// ------------------------------------------------------------------------------------------------
// K6a (certified approximate path): nearest coarse centroid of every vector to encode.
// One block = 64 vectors against ALL centroids: the vectors' fp32 copy stays in LDS, centroid tiles
// of 256 stream through, dot products on the fp32 matrix cores (as K1c), and every lane keeps the
// two smallest d~ = |c|^2 + |x|^2 - 2<x,c> of its rows.  With eps the error bound of K1d:
//   second - best > 2 eps  =>  every other centroid has d >= d~ - eps > best + eps >= d(best):
//   the argmin is certified (and unique, so the reference's first-wins tie rule is moot);
// otherwise the vector is flagged and re-done by the exact kernel k_assign_coarse (a ~1e-4 fraction).
// ------------------------------------------------------------------------------------------------
#define ASG_BM 64
#define ASG_BN 256
#define ASG_BK 32
__global__ __launch_bounds__(MMIDX_BLOCK) void k_assign_approx(const float *__restrict__ CT32, const float *__restrict__ X32,
                                                               const double *__restrict__ cn, const double *__restrict__ xn,
                                                               int32_t *__restrict__ cell_out, unsigned char *__restrict__ amb,
                                                               double cnorm_max, double cn_max, int C, int D, long long n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lda = D + 2;                               // (2*row + k) mod 32 spreads the fragment reads
    float *As = (float *)smem;                           // [64][D+2]
    float *Bs = As + ((ASG_BM * lda + 3) & ~3);          // [32][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long v0 = (long long)blockIdx.x * ASG_BM;
    for (int e = tid; e < ASG_BM * D; e += MMIDX_BLOCK) {
        const int r = e / D, j = e - r * D;
        const long long v = v0 + r;
        As[r * lda + j] = (v < n) ? X32[(size_t)v * D + j] : 0.f;
    }
    float m1[4], m2[4];
    int i1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        m1[r] = __int_as_float(0x7f800000);
        m2[r] = __int_as_float(0x7f800000);
        i1[r] = 0x7fffffff;
    }
    double xnr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const long long v = v0 + wave * 16 + 4 * (lane >> 4) + r;
        xnr[r] = (v < n) ? xn[v] : 0.0;
    }
    const int fr = lane & 15, fk = lane >> 4;
    for (int c0 = 0; c0 < C; c0 += ASG_BN) {
        f32x4 acc[16];
#pragma unroll
        for (int t = 0; t < 16; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < D; k0 += ASG_BK) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; u++) {  // B: 32 k x 256 c = 2048 float4
                const int p = tid + u * 256, r = p >> 6, c4 = (p & 63) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + r < D) {
                    const float *src = CT32 + (size_t)(k0 + r) * C + c0 + c4;
                    if (c0 + c4 + 3 < C) {
                        v = *(const float4 *)src;
                    } else {
                        float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int e = 0; e < 4; e++)
                            if (c0 + c4 + e < C) tmp[e] = src[e];
                        v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                    }
                }
                *(float4 *)(Bs + r * ASG_BN + c4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < ASG_BK / 4; kk++) {
                const int kidx = k0 + kk * 4 + fk;
                const float a = (kidx < D) ? As[(wave * 16 + fr) * lda + kidx] : 0.f;
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float b = Bs[(kk * 4 + fk) * ASG_BN + t * 16 + fr];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
        // epilogue of this centroid tile: running two smallest d~ per row (C/D: col = lane & 15, row = 4*(lane>>4)+reg)
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int c = c0 + t * 16 + (lane & 15);
            if (c < C) {
                const double cnc = cn[c];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float d = (float)((cnc + xnr[r]) - 2.0 * (double)acc[t][r]);
                    if (d < m1[r] || (d == m1[r] && c < i1[r])) {
                        m2[r] = m1[r];
                        m1[r] = d;
                        i1[r] = c;
                    } else if (d < m2[r]) {
                        m2[r] = d;
                    }
                }
            }
        }
    }
    // merge across the 16 lanes that share a row
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const float om1 = __shfl_xor(m1[r], off), om2 = __shfl_xor(m2[r], off);
            const int oi1 = __shfl_xor(i1[r], off);
            if (om1 < m1[r] || (om1 == m1[r] && oi1 < i1[r])) {
                m2[r] = (m1[r] < om2) ? m1[r] : om2;
                m1[r] = om1;
                i1[r] = oi1;
            } else {
                m2[r] = (om1 < m2[r]) ? om1 : m2[r];
            }
        }
        const long long v = v0 + wave * 16 + 4 * (lane >> 4) + r;
        if ((lane & 15) == 0 && v < n) {
            const double xnorm = sqrt(xnr[r]);
            const double sumn = cnorm_max + xnorm;
            const double eps = (2.0 * (double)(D + 3) * 0x1p-24 * 1.01 * xnorm * cnorm_max + 1e-12 * (cn_max + xnr[r]) +
                                0x1p-23 * sumn * sumn) * (1.0 + 1e-9);
            const bool ok = ((double)m2[r] - (double)m1[r]) > 2.0 * eps;  // inf - x = inf > ... when C == 1
            cell_out[v] = i1[r];
            amb[v] = ok ? 0 : 1;
        }
    }
}

// compaction of the flagged vectors and write-back of their exact cells
// indices of the set flags, any order (one atomic per wave)
__global__ void k_compact_flags(const unsigned char *__restrict__ flag, long long n, int32_t *__restrict__ idx, int32_t *__restrict__ count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool f = i < n && flag[i] != 0;
    const u64 mk = __builtin_amdgcn_ballot_w64(f);
    if (!mk) return;
    const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)mk) - 1;
    u32 base = 0;
    if (lane == leader) base = (u32)atomicAdd(count, (int)__popcll(mk));
    base = wave_read_u32(base, leader);
    if (f) idx[base + (u32)__popcll(mk & ((1ull << lane) - 1ull))] = (int32_t)i;
}
__global__ void k_gather_rows(const double *__restrict__ X, const int32_t *__restrict__ idx, double *__restrict__ out, int D,
                              long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * D) return;
    const long long r = e / D;
    out[e] = X[(size_t)idx[r] * D + (e - r * D)];
}
__global__ void k_scatter_cells(const int32_t *__restrict__ idx, const int32_t *__restrict__ cells, int32_t *__restrict__ out,
                                long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[idx[e]] = cells[e];
}

// ------------------------------------------------------------------------------------------------
// K6b: product quantisation.  Per vector: residual (centroid - x) -> transform -> for every
// sub-quantizer the first-minimum centroid (IVFPQ.java:316-335, :613-631; PQ.java:237-252).
// VT vectors per block; thread j owns centroid j of every sub-quantizer.
// ------------------------------------------------------------------------------------------------
template <int VT, typename CodeT>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_encode_pq(const double *__restrict__ X,
                                                           const int32_t *__restrict__ cell,
                                                           const double *__restrict__ coarse,
                                                           const double *__restrict__ pqT,
                                                           const int32_t *__restrict__ perm,
                                                           const double *__restrict__ rot,
                                                           CodeT *__restrict__ code_out, int D, int m,
                                                           int ks, int dsub, int transform, int ivf,
                                                           long long n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *r = (double *)smem;            // [VT][D]
    double *tr = r + (size_t)VT * D;       // [VT][D]
    u64 *red_k = (u64 *)(tr + (size_t)VT * D);       // [m][VT][4]
    int *red_i = (int *)(red_k + (size_t)m * VT * 4);
    const long long v0 = (long long)blockIdx.x * VT;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < VT * D; idx += MMIDX_BLOCK) {
        const int t = idx / D, i = idx - t * D;
        const long long v = (v0 + t < n) ? v0 + t : n - 1;
        const double xv = X[(size_t)v * D + i];
        double val = xv;
        if (ivf) {
            int c = cell[v];
            if (c < 0) c = 0;
            val = coarse[(size_t)c * D + i] - xv;
        }
        r[idx] = val;
    }
    __syncthreads();
    if (transform == 2) {
        for (int idx = tid; idx < VT * D; idx += MMIDX_BLOCK) {
            const int t = idx / D, i = idx - t * D;
            tr[idx] = r[t * D + perm[i]];
        }
        __syncthreads();
    } else if (transform == 1) {
        for (int idx = tid; idx < VT * D; idx += MMIDX_BLOCK) {
            const int t = idx / D, j = idx - t * D;
            double total = 0.0;
            for (int i = 0; i < D; i++) total += r[t * D + i] * rot[(size_t)i * D + j];
            tr[idx] = total;
        }
        __syncthreads();
    } else {
        tr = r;
    }
    const int wave = tid >> 6;
    for (int s = 0; s < m; s++) {
        u64 bk[VT];
        int bi[VT];
#pragma unroll
        for (int t = 0; t < VT; t++) {
            bk[t] = MMIDX_KEY_MAX;
            bi[t] = 0x7fffffff;
        }
        for (int j = tid; j < ks; j += MMIDX_BLOCK) {
            double acc[VT];
#pragma unroll
            for (int t = 0; t < VT; t++) acc[t] = 0.0;
            for (int tt = 0; tt < dsub; tt++) {
                const double pv = pqT[((size_t)s * dsub + tt) * ks + j];
#pragma unroll
                for (int t = 0; t < VT; t++) {
                    const double df = pv - tr[t * D + s * dsub + tt];
                    acc[t] += df * df;
                }
            }
#pragma unroll
            for (int t = 0; t < VT; t++) {
                const u64 k = dkey(acc[t]);
                if (k < bk[t]) {
                    bk[t] = k;
                    bi[t] = j;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < VT; t++) {
            u64 k = bk[t];
            int i = bi[t];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const u64 ok = __shfl_xor(k, off);
                const int oi = __shfl_xor(i, off);
                if (ok < k || (ok == k && oi < i)) {
                    k = ok;
                    i = oi;
                }
            }
            if ((tid & 63) == 0) {
                red_k[((size_t)s * VT + t) * 4 + wave] = k;
                red_i[((size_t)s * VT + t) * 4 + wave] = i;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < m * VT; idx += MMIDX_BLOCK) {
        const int s = idx / VT, t = idx - s * VT;
        if (v0 + t >= n) continue;
        u64 k = red_k[(size_t)idx * 4];
        int i = red_i[(size_t)idx * 4];
        for (int wv = 1; wv < MMIDX_BLOCK / 64; wv++) {
            const u64 ok = red_k[(size_t)idx * 4 + wv];
            const int oi = red_i[(size_t)idx * 4 + wv];
            if (ok < k || (ok == k && oi < i)) {
                k = ok;
                i = oi;
            }
        }
        code_out[(size_t)(v0 + t) * m + s] = (CodeT)i;
    }
}

// K6b': the same arithmetic with one thread per vector (no transform or a permutation; dsub = 4, 8, 16).  K6b reads the
// whole codebook from L2 for every 8 vectors; here a block of 256 vectors stages one sub-quantizer's table (ks x dsub
// doubles) in LDS at a time and every thread walks it with broadcast reads, keeping its sub-vector in registers: the
// first minimum in j order wins, as in computeNearestProductIndex (IVFPQ.java:613-631).  pq is the file-order table
// [m][ks][dsub].
template <int DSUB, typename CodeT>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_encode_pq_rows(const double *__restrict__ X, const int32_t *__restrict__ cell,
                                                                const double *__restrict__ coarse, const double *__restrict__ pq,
                                                                const int32_t *__restrict__ perm, CodeT *__restrict__ code_out, int D,
                                                                int m, int ks, int transform, int ivf, long long n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *tab = (double *)smem;                        // [ks][DSUB] of the current sub-quantizer
    CodeT *cbuf = (CodeT *)(tab + (size_t)ks * DSUB);    // [256][m] the block's codes, written out coalesced at the end
    const int tid = threadIdx.x;
    const long long v0 = (long long)blockIdx.x * MMIDX_BLOCK, v = v0 + tid;
    const long long vv = v < n ? v : n - 1;
    int c = 0;
    if (ivf) {
        c = cell[vv];
        c = c < 0 ? 0 : c;
    }
    const double *xrow = X + (size_t)vv * D, *crow = coarse + (size_t)c * D;
    for (int s = 0; s < m; s++) {
        __syncthreads();  // the previous table has been consumed
        const double *src = pq + (size_t)s * ks * DSUB;
        for (int i = tid; i < ks * DSUB; i += MMIDX_BLOCK) tab[i] = src[i];
        double r[DSUB];
#pragma unroll
        for (int t = 0; t < DSUB; t++) {
            const int idx = s * DSUB + t;
            const int from = transform == 2 ? perm[idx] : idx;
            const double xv = xrow[from];
            r[t] = ivf ? crow[from] - xv : xv;
        }
        __syncthreads();
        double best = __longlong_as_double(0x7FF0000000000000ll);
        int bi = 0;
#pragma unroll 4
        for (int j = 0; j < ks; j++) {
            const double *pv = tab + (size_t)j * DSUB;  // the same address in every lane: LDS broadcast
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < DSUB; t++) {
                const double df = pv[t] - r[t];
                acc += df * df;
            }
            const bool lt = acc < best;  // strict: the first minimum wins (sums are >= 0 and finite)
            best = lt ? acc : best;
            bi = lt ? j : bi;
        }
        cbuf[(size_t)tid * m + s] = (CodeT)bi;
    }
    __syncthreads();
    const long long nvalid = (n - v0 < MMIDX_BLOCK) ? n - v0 : MMIDX_BLOCK;
    for (long long i = tid; i < nvalid * m; i += MMIDX_BLOCK) code_out[(size_t)v0 * m + i] = cbuf[i];
}

// ------------------------------------------------------------------------------------------------
// inverted-list maintenance: move the existing CSR entries / place the new ones
// ------------------------------------------------------------------------------------------------
template <typename CodeT>
__global__ void k_move_old(const int64_t *__restrict__ off_old, const int64_t *__restrict__ off_new,
                           int nlists, const CodeT *__restrict__ codes_old,
                           const int32_t *__restrict__ ids_old, CodeT *__restrict__ codes_new,
                           int32_t *__restrict__ ids_new, int m, long long n_old) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_old) return;
    int lo = 0, hi = nlists;  // largest c with off_old[c] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off_old[mid] <= e) lo = mid;
        else hi = mid;
    }
    const long long dst = off_new[lo] + (e - off_old[lo]);
    ids_new[dst] = ids_old[e];
    for (int s = 0; s < m; s++) codes_new[(size_t)dst * m + s] = codes_old[(size_t)e * m + s];
}

template <typename CodeT>
__global__ void k_place_new(const long long *__restrict__ dest, const CodeT *__restrict__ codes_p,
                            const int32_t *__restrict__ ids_p, CodeT *__restrict__ codes_new,
                            int32_t *__restrict__ ids_new, int m, long long n_new) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_new) return;
    const long long dst = dest[e];
    ids_new[dst] = ids_p[e];
    for (int s = 0; s < m; s++) codes_new[(size_t)dst * m + s] = codes_p[(size_t)e * m + s];
}

__global__ void k_iota(int32_t *out, int32_t start, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = start + (int32_t)e;
}

// stored form <-> centroid index (PQ.transformToByte PQ.java:552-558: stored = idx - 128)
__global__ void k_bias_codes(const unsigned char *in, signed char *out, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = (signed char)((int)in[e] - 128);
}
__global__ void k_unbias_codes(const signed char *in, unsigned char *out, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = (unsigned char)((int)in[e] + 128);
}

// ------------------------------------------------------------------------------------------------
// instrumentation (profiling mode only): algorithmic work of a launch = sum over (query, probe) of
// the probed list lengths; number of queries that needed the tie replay.
// ------------------------------------------------------------------------------------------------
// profiling counters in one launch: total[0] += sum of the probed lists' lengths (cells may be null: flat PQ counts on the
// host), total[1] += number of flagged queries (flag may be null).  Eight independent loads per thread are in flight
// before their dependent list_off loads: the kernel sits inside the timed region of bench.py.
__global__ void k_count_stats(const int32_t *__restrict__ cells, const int64_t *__restrict__ list_off, long long n, int w,
                              const int32_t *__restrict__ flag, long long nq, u64 *__restrict__ total) {
    const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    u64 v = 0, f = 0, v0 = 0;  // v0: the probe-rank-0 lists (what pass A scans)
    if (cells) {
        for (long long e0 = t0; e0 < n; e0 += 8 * stride) {
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const long long e = e0 + u * stride;
                c[u] = e < n ? cells[e] : -1;
            }
            int64_t a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a[u] = c[u] >= 0 ? list_off[c[u]] : 0;
                b[u] = c[u] >= 0 ? list_off[c[u] + 1] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                v += (u64)(b[u] - a[u]);
                if ((e0 + u * stride) % w == 0) v0 += (u64)(b[u] - a[u]);
            }
        }
    }
    if (flag)
        for (long long e = t0; e < nq; e += stride) f += flag[e] ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off);
        f += __shfl_xor(f, off);
        v0 += __shfl_xor(v0, off);
    }
    __shared__ u64 s_v[16], s_f[16], s_v0[16];  // one set of atomics per block, not per wave (they all hit the same words)
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_v[wv] = v;
        s_f[wv] = f;
        s_v0[wv] = v0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; i++) {
            v += s_v[i];
            f += s_f[i];
            v0 += s_v0[i];
        }
        if (v) atomicAdd(total, v);
        if (f) atomicAdd(total + 1, f);
        if (v0) atomicAdd(total + 2, v0);
    }
}

// thresholds cross the process boundary as doubles (+inf = "fewer than k+1 candidates so far") so
// that a MIN all-reduce over ranks is meaningful
__global__ void k_T_export(const u64 *__restrict__ T, double *__restrict__ out, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = (T[e] >= 0x7FF0000000000000ull) ? __longlong_as_double(0x7FF0000000000000ll) : keyd(T[e]);
}
__global__ void k_T_import(const double *__restrict__ in, u64 *__restrict__ T, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double v = in[e];
    if (v >= 0.0 && v < __longlong_as_double(0x7FF0000000000000ll)) {
        const u64 k = dkey(v);
        if (k < T[e]) T[e] = k;
    }
}

// SDC term table of one query code: TT[q][s][a][t] = (pq[s][a][t] - pq[s][b_s][t])^2, b = stored code of the
// vector with internal id iid (PQ.java:337-344; iid == position in the flat PQ index, PQ.java:303,318)
template <typename CodeT>
__global__ void k_sdc_terms(const double *__restrict__ pq, const CodeT *__restrict__ codes, const int32_t *__restrict__ qpos,
                            double *__restrict__ TT, int m, int ks, int dsub, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long long per = (long long)m * ks * dsub;
    const int q = (int)(e / per);
    const long long r = e - (long long)q * per;
    const int s = (int)(r / ((long long)ks * dsub));
    const int t = (int)(r % dsub);
    const int b = (int)codes[(size_t)qpos[q] * m + s];
    const double df = pq[r] - pq[((size_t)s * ks + b) * dsub + t];
    TT[e] = df * df;
}


// ------------------------------------------------------------------------------------------------
// per-id utilities of IVFPQ (J/datastructures/IVFPQ.java:464-497 computeDistanceIVFADC, :801-880 getPQCodeByte /
// getPQCodeShort / getInvertedListId): the reference reads the record {list id, code} of an internal id from BDB; here
// the record lives in the list-major arrays, found through an iid -> position map built on demand.
// ------------------------------------------------------------------------------------------------
__global__ void k_iid_max(const int32_t *__restrict__ ids, long long n, int32_t *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int v = i < n ? ids[i] : -1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v >= 0) atomicMax(out, v);
}
__global__ void k_inv_scatter(const int32_t *__restrict__ ids, long long n, int32_t *__restrict__ inv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids[i] >= 0) inv[ids[i]] = (int32_t)i;
}
// pos_out[i] = position of iids[i] (-1: not indexed); code_out[i][m] in stored form (byte: index - 128, PQ.java:552-558)
template <typename CodeT>
__global__ void k_lookup_codes(const int32_t *__restrict__ iids, long long n, const int32_t *__restrict__ inv, long long inv_size,
                               const CodeT *__restrict__ codes, int m, int32_t *__restrict__ pos_out, CodeT *__restrict__ code_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t id = iids[i];
    const int32_t pos = (id >= 0 && id < inv_size) ? inv[id] : -1;
    pos_out[i] = pos;
    const long long src = pos >= 0 ? pos : 0;
    for (int s = 0; s < m; s++) {
        const CodeT v = pos >= 0 ? codes[src * m + s] : (CodeT)0;
        code_out[i * m + s] = sizeof(CodeT) == 1 ? (CodeT)(v - 128) : v;
    }
}
// computeDistanceIVFADC for pair i = (query Q[i], indexed vector at position pos[i] of list cell[i]): one block per pair;
// residual / transform exactly as the search does (query_vector), then the table entries the code selects, t ascending,
// summed in sub-quantizer order (IVFPQ.java:487-495 over :525-538).  Thread 0 runs the sequential chain.
template <typename CodeT>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_pair_distance(const ScanParams P, const double *__restrict__ pq, const int32_t *__restrict__ pos,
                                                               const int32_t *__restrict__ cell, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *vec = (double *)smem;  // [2 * D]
    const int i = blockIdx.x;
    const int32_t p = pos[i];
    const int c = P.ivf ? cell[i] : 0;
    if (p < 0 || c < 0) {  // (uniform)
        if (threadIdx.x == 0) out[i] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    const double *tr = query_vector(P, i, c, vec);
    if (threadIdx.x == 0) {
        const CodeT *code = (const CodeT *)P.codes + (size_t)p * P.m;
        double d = 0.0;
        for (int s = 0; s < P.m; s++) {
            const double *pp = pq + ((size_t)s * P.ks + code[s]) * P.dsub;
            double acc = 0.0;
            for (int t = 0; t < P.dsub; t++) {
                const double df = tr[s * P.dsub + t] - pp[t];
                acc += df * df;
            }
            d += acc;
        }
        out[i] = d;
    }
}

// validation of appended records before they are committed (a bad list id would break every later CSR build, a code >= ks
// would index past the lookup table): flag |= 1 for a list id outside [0, nlists), |= 2 for a code value >= ks
template <typename CodeT>
__global__ void k_check_records(const int32_t *__restrict__ cells, int nlists, const CodeT *__restrict__ codes, int ks, int m, long long n,
                                int32_t *__restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = 0;
    if (cells && (cells[i] < 0 || cells[i] >= nlists)) bad |= 1;
    for (int s = 0; s < m; s++)
        if ((int)codes[i * m + s] >= ks) bad |= 2;
    if (bad) atomicOr(flag, bad);
}


// ------------------------------------------------------------------------------------------------
// Straddling ties across shards.  The single queue of IVFPQ.java:445 sees every candidate in offer order (probe rank, list
// position); with the lists of a query spread over ranks no rank sees that sequence.  The closed form of the bounded queue
//  needs, for tau = the k-th smallest distance: b = #{d < tau}, p = the ties among the first k offers
// with d <= tau, and then keeps the ties of rank e .. p-1 (e = b - (k - p)) in global offer order.  Every list lives on
// exactly one rank, so three passes with two tiny reductions in between reproduce it:
//   phase 0  per (flagged query, local list): n(d <= tau), n(d == tau)            -> SUM all-reduce of counts[F][w][2]
//   phase 1  the list holding the k-th "d <= tau" offer is known to everyone; its owner counts the ties among that list's
//            first j* such offers                                                 -> SUM all-reduce of pB[F]
//   phase 2  owners write the iids of their ties of global rank e .. p-1 into the slots they occupy in the answer
//            (later-offered first: slot k-1-(rank-e))                            -> MAX all-reduce of tie_iids[F][k]
// One block per flagged query; the lists are streamed in offer order with the exact table of the search.
// ------------------------------------------------------------------------------------------------
struct TieShardParams {
    ScanParams S;          // Q, coarse, pqT, perm, rot, cells, list_off, codes, D, m, ks, dsub, w, transform, ivf
    const int32_t *ids;
    const int32_t *fq;     // [F] query index, -1 = unused slot
    const double *tau;     // [F]
    int32_t *counts;       // [F][w][2]
    int32_t *pB;           // [F]
    int32_t *tie_iids;     // [F][k] (-1 where nothing is written)
    int k, phase;
};

template <typename CodeT, bool GLUT = false>
__global__ __launch_bounds__(MMIDX_BLOCK) void k_shard_tie(const TieShardParams TP) {
    const ScanParams &P = TP.S;
    const int f = blockIdx.x, tid = threadIdx.x, k = TP.k, w = P.w;
    const int q = TP.fq[f];
    if (q < 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, ks = P.ks;
    double *lut, *vec;
    if constexpr (GLUT) {  // (table in the block's slice of the global scratch, see k_scan)
        lut = P.glut + (size_t)blockIdx.x * (size_t)m * ks;
        vec = (double *)smem;
    } else {
        lut = (double *)smem;
        vec = lut + (size_t)m * ks;
    }
    __shared__ int s_wsum[MMIDX_BLOCK / 64][2];
    __shared__ int s_run[2];   // running (d <= tau, d == tau) counts of the list being streamed
    __shared__ int s_plan[6];  // r*, j*, ties before r*, b, e, p
    const u64 tau = dkey(TP.tau[f]);
    const u64 lane_lt = (1ull << (tid & 63)) - 1ull;
    int32_t *cnt = TP.counts + (size_t)f * w * 2;
    if (TP.phase > 0) {
        if (tid == 0) {
            int L = 0, TB = 0, rs = -1, js = 0, tbs = 0, b = 0;
            for (int r = 0; r < w; r++) {
                const int nj = cnt[2 * r], nt = cnt[2 * r + 1];
                if (rs < 0 && L + nj >= k) {
                    rs = r;
                    js = k - L;
                    tbs = TB;
                }
                L += nj;
                TB += nt;
                b += nj - nt;
            }
            const int p = tbs + (TP.phase == 2 ? TP.pB[f] : 0);
            s_plan[0] = rs;
            s_plan[1] = js;
            s_plan[2] = tbs;
            s_plan[3] = b;
            s_plan[4] = b - (k - p);
            s_plan[5] = p;
        }
        __syncthreads();
    }
    const int r_star = TP.phase > 0 ? s_plan[0] : -1, j_star = TP.phase > 0 ? s_plan[1] : 0;
    const int e = TP.phase == 2 ? s_plan[4] : 0, pfin = TP.phase == 2 ? s_plan[5] : 0;
    if (TP.phase > 0 && r_star < 0) return;  // fewer than k candidates with d <= tau: nothing straddles
    int ties_before = 0;  // ties in the lists of lower probe rank (phase 2)
    for (int pr = 0; pr < w; pr++) {
        if (TP.phase == 2 && pr > 0) ties_before += cnt[2 * (pr - 1) + 1];
        if (TP.phase == 1 && pr != r_star) continue;
        const int cell = P.cells[(size_t)q * w + pr];
        if (cell < 0) break;
        const int64_t beg = P.list_off[cell];
        const int64_t len = P.list_off[cell + 1] - beg;
        if (len == 0) continue;  // (a list of another rank, or an empty one)
        if (TP.phase == 2 && (ties_before >= pfin || ties_before + cnt[2 * pr + 1] <= e)) continue;  // no kept tie in this list
        __syncthreads();
        if (tid == 0) s_run[0] = s_run[1] = 0;
        const double *tr = query_vector(P, q, cell, vec);
        build_lut_any(lut, tr, P.pqT, m, ks, P.dsub);
        __syncthreads();
        const CodeT *codes = (const CodeT *)P.codes + (size_t)beg * m;
        bool done = false;
        for (int64_t base = 0; base < len && !done; base += MMIDX_BLOCK) {
            const int64_t i = base + tid;
            bool nonjunk = false, tie = false;
            if (i < len) {
                const CodeT *cp = codes + (size_t)i * m;
                double a = 0.0;
                for (int s = 0; s < m; s++) a += lut[s * ks + (int)cp[s]];
                const u64 key = dkey(a);
                nonjunk = key <= tau;
                tie = key == tau;
            }
            const u64 mj = __ballot(nonjunk), mt = __ballot(tie);
            const int wave = tid >> 6;
            if ((tid & 63) == 0) {
                s_wsum[wave][0] = __popcll(mj);
                s_wsum[wave][1] = __popcll(mt);
            }
            __syncthreads();
            int pj = s_run[0], pt = s_run[1];
            for (int wv = 0; wv < wave; wv++) {
                pj += s_wsum[wv][0];
                pt += s_wsum[wv][1];
            }
            pj += __popcll(mj & lane_lt);  // offers with d <= tau before mine in this list
            pt += __popcll(mt & lane_lt);  // ties before mine in this list
            if (TP.phase == 1) {
                if (nonjunk && pj + 1 == j_star) TP.pB[f] = pt + (tie ? 1 : 0);  // (one writer)
            } else if (TP.phase == 2) {
                const int rank = ties_before + pt;
                if (tie && rank >= e && rank < pfin) TP.tie_iids[(size_t)f * k + (k - 1 - (rank - e))] = TP.ids[beg + i];
            }
            __syncthreads();
            if (tid == 0) {
                int tj = 0, tt = 0;
                for (int wv = 0; wv < MMIDX_BLOCK / 64; wv++) {
                    tj += s_wsum[wv][0];
                    tt += s_wsum[wv][1];
                }
                s_run[0] += tj;
                s_run[1] += tt;
            }
            __syncthreads();
            if (TP.phase == 1 && s_run[0] >= j_star) done = true;
            if (TP.phase == 2 && ties_before + s_run[1] >= pfin) done = true;
        }
        if (TP.phase == 0 && tid == 0) {
            cnt[2 * pr] = s_run[0];
            cnt[2 * pr + 1] = s_run[1];
        }
    }
}
