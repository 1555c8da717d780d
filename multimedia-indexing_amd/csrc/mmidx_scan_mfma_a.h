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

// One wave per pair: the K1-th largest of the (up to NS x 256) accumulator values sweep 1 kept for it (NS chunks of 256: pieces x 1 or 2), to 2^-13 relative (a bisection
// on the key bits with ballots; the low ten bits stay zero, which can only lower the value).  K1 DISTINCT codes of the list have
// acc >= a*, and a code's distance is at most ||r||^2 + err - 2 acc / s^2 (k_scan_mfma's certificate): that bound at a* is a valid
// threshold.  Pairs with fewer than K1 values (short lists) or an unusable scale keep T = +inf and go to the redo.
template <int NS>
__global__ __launch_bounds__(256) void k_a1_select(const MfmaParams P) {
    const int lane = (int)(threadIdx.x & 63);
    const long long npairs = *P.S.n_order;
    // (a loop over the device-side pair count: a shard of a sharded handle holds an eighth of the pairs its grid would have to cover)
    for (long long slot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); slot < npairs; slot += (long long)gridDim.x * 4) {
    const int e = P.S.order[slot];
    const int q = e / P.S.w;
    const int cell = P.S.cells[e];
    const long long len = P.S.list_off[cell + 1] - P.S.list_off[cell];
    const long long pieces_l = (len + P.sub - 1) / P.sub;
    // (chunks of 256 values: a piece is one, or two where the eight-wave sweep instance sets the stride)
    const int pieces = (int)(pieces_l < (long long)P.nsub ? pieces_l : (long long)P.nsub) * (P.a_cstride >> 8);
    const double2 rc = P.a_rowc[slot];
    if (!(rc.x == rc.x)) continue;
    u32 key[4 * NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        if (j < pieces) {
            const float4 v = *(const float4 *)(P.a_cand + (size_t)slot * P.nsub * (size_t)P.a_cstride + (size_t)j * 256 + lane * 4);
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
    if (K <= 0x007FFFFFu) continue;  // fewer than K1 values
    const float a = a1_unkey(K);
    if (!(a == a) || !(fabsf(a) < 3e38f)) continue;
    const double ub = (rc.x + rc.y * (double)a) * (1.0 + 1e-12);
    if (!(ub >= 0.0) || !(ub < 1e300)) continue;
    if (lane == 0) atomicMin(P.S.T + q, dkey(ub));
    }
}

// ---- sweep 2's bits as a flat record list -----------------------------------------------------------------------------------------
// k_a1_item_scan: a_icnt[v] (bits per item, counted by sweep 2) -> exclusive prefix in place, a_icnt[items] = total (one block).
__global__ __launch_bounds__(1024) void k_a1_item_scan(const MfmaParams P) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nv = *P.n_groups * P.nsub;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nv; base += 1024) {
        const int i = base + tid;
        const u32 c = i < nv ? P.a_icnt[i] : 0u;
        const u32 incl = wave_incl_scan_u32(c);
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        // (saturating: duplicate-heavy long lists with many queries per list can set more than 2^32 bits in one call; a wrapped prefix
        //  would let items write overlapping record ranges.  Past the cap every item goes to the redo)
        unsigned long long before = s_carry;
#pragma unroll
        for (int j = 0; j < 16; j++) before += (j < wv) ? s_wave[j] : 0u;
        const unsigned long long sat = 0xFFFFFFFFull;
        const unsigned long long mine = before + incl - c;
        if (i < nv) P.a_icnt[i] = (u32)(mine < sat ? mine : sat);
        __syncthreads();
        if (tid == 1023) s_carry = (u32)(before + incl < sat ? before + incl : sat);
        __syncthreads();
    }
    if (tid == 0) P.a_icnt[nv] = s_carry;
    // the rounds' slot ranges: empty (k_a1_records narrows them down)
    const u32 tot = s_carry < P.a_rec_cap ? s_carry : P.a_rec_cap;
    const u32 nrnd = (tot + (u32)P.a_rnd_size - 1u) / (u32)P.a_rnd_size;
    for (u32 i = (u32)tid; i < nrnd; i += 1024u) P.a_rnd[i] = make_uint2(0xFFFFFFFFu, 0u);
}
// the pairs' exact residuals c - q (IVFPQ.java:645) and what the verification needs of a pair, slot by slot: k_a1_verify reads the rows
// of a round's pairs as ONE contiguous block instead of chasing order[] -> cells[] -> list_off[] -> two rows per pair
__global__ __launch_bounds__(256) void k_a1_rows(const MfmaParams P, int D) {
    const long long npairs = *P.S.n_order;
    const int per = 256 / (D / 2 < 256 ? D / 2 : 256);  // pairs per block: a thread per 16 bytes of a row (D <= 128: 64 threads per pair)
    const int tp = D / 2;
    const int t = (threadIdx.x % tp) * 2;
    if (threadIdx.x >= per * tp) return;
    for (long long slot = (long long)blockIdx.x * per + threadIdx.x / tp; slot < npairs; slot += (long long)gridDim.x * per) {
    const int e = P.S.order[slot];
    const int q = e / P.S.w;
    const int cell = P.S.cells[e];
    if (!P.R) {
        const double2 cc = *(const double2 *)(P.S.coarse + (size_t)cell * D + t), qq = *(const double2 *)(P.S.Q + (size_t)q * D + t);
        *(double2 *)(P.a_rows + (size_t)slot * D + t) = make_double2(cc.x - qq.x, cc.y - qq.y);
    }
    if (t == 0) {
        const long long beg = P.S.list_off[cell];
        P.a_meta[slot] = make_int4(q, e - q * P.S.w, (int)(u32)beg, (int)(beg >> 32));
        P.a_metaT[slot] = P.S.T[q];  // (final: the kernel runs behind k_a1_select, and nothing lowers a threshold before pass A's verification is over)
    }
    }
}
// k_a1_records: a block per item turns its bitmap -- per tile pair and lane the packed compares of a_scan_tiles<MODE 2>, read 16 bytes
// per thread -- into records {pair slot in order[], position in the index} at a_rec[a_icnt[item] ..): per-thread bit counts, a block
// scan, no atomics.  The list is in item order, so the pairs of any window of it form a contiguous range of slots.
#define A1R_NT 256
__global__ __launch_bounds__(A1R_NT) void k_a1_records(const MfmaParams P) {
    __shared__ u32 s_wave[A1R_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nv = *P.n_groups * P.nsub;
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
        const unsigned char *bm = P.a_bm + (size_t)v * P.a_bm_stride;
        const u32 off0 = P.a_icnt[v], cnt = P.a_icnt[v + 1] - off0;
        if (cnt == 0) continue;  // (block-uniform)
        if ((unsigned long long)off0 + cnt > (unsigned long long)P.a_rec_cap) {  // the list is full: the item's queries go to the exact kernels
            for (int r = tid; r < np; r += A1R_NT) P.redo[P.S.order[first + r] / P.S.w] = 1;
            for (u32 i = off0 + (u32)tid; i < off0 + cnt && i < P.a_rec_cap; i += A1R_NT) P.a_rec[i] = make_uint2(0xFFFFFFFFu, 0u);
            continue;
        }
        const u32 posb = (u32)(beg + c0);  // (an index holds fewer than 2^31 codes)
        if (tid == 0) {  // the rounds this item's records fall into take in its slot range (two atomics per round and item)
            const u32 rsh = (u32)__builtin_ctz((unsigned)P.a_rnd_size);
            for (u32 r = off0 >> rsh; r <= (off0 + cnt - 1u) >> rsh; r++) {
                atomicMin(&P.a_rnd[r].x, (u32)first);
                atomicMax(&P.a_rnd[r].y, (u32)(first + np - 1));
            }
        }
        // words of 16 bits (<= 32 rows) or 32 bits; a thread takes 16 bytes per step: its loads' bit counts first, ONE block scan, then
        // the same loads again (L2-resident) with the records written at the thread's own offset
        const int wpl = ntl <= 2 ? 8 : 4;                 // words per 16-byte load
        const int nload = npt * 64 / wpl;                 // (64 words per tile pair)
        u32 mine = 0;
        // Up to 16 loads per thread (a piece of <= 16384 codes with <= 32 rows: the usual item is ONE piece of ~12 k codes, 12 loads per
        // thread): ALL of them in flight at once and kept in registers for the walk below.  (Four at a time and a second read of every
        // word inside the walk, the item was fifteen memory round trips in a row: 110 k cycles, 263 us per 131072 queries.)
        constexpr int KEEP = 16;
        const bool small = nload <= KEEP * A1R_NT;  // (block-uniform)
        const int nu = (nload + A1R_NT - 1) / A1R_NT;
        uint4 keep[KEEP];
        if (small) {
#pragma unroll
            for (int u = 0; u < KEEP; u++) {
                keep[u] = make_uint4(0u, 0u, 0u, 0u);
                if (u < nu) {  // (block-uniform)
                    const int lu = tid + u * A1R_NT;
                    keep[u] = ((const uint4 *)bm)[lu < nload ? lu : 0];  // (clamped inside the item's own slice; the value is zeroed below)
                }
            }
#pragma unroll
            for (int u = 0; u < KEEP; u++) {
                if (tid + u * A1R_NT >= nload) keep[u] = make_uint4(0u, 0u, 0u, 0u);
                mine += (u32)(__popc(keep[u].x) + __popc(keep[u].y) + __popc(keep[u].z) + __popc(keep[u].w));
            }
        } else {
            for (int l = tid; l < nload; l += 4 * A1R_NT) {  // (four loads in flight: one at a time the loop is a chain of L2 round trips)
                uint4 a[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int lu = l + u * A1R_NT;
                    a[u] = ((const uint4 *)bm)[lu < nload ? lu : l];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (l + u * A1R_NT < nload) mine += (u32)(__popc(a[u].x) + __popc(a[u].y) + __popc(a[u].z) + __popc(a[u].w));
            }
        }
        const u32 incl = wave_incl_scan_u32(mine);
        __syncthreads();  // (s_wave of the previous item has been read)
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        u32 o = off0 + incl - mine;
#pragma unroll
        for (int j = 0; j < A1R_NT / 64; j++) o += (j < wv) ? s_wave[j] : 0u;
        if (!mine) continue;  // (no barrier below)
        // the set bits of a 16-byte load, 64 at a time (the masks are sparse -- ~1 % of the bits --: walking the words costs a branch per
        // word and lane, 0.35 ms per 131072 queries; walking the bits costs nothing for an empty word)
        auto walk = [&](const int l, const uint4 a) {
            if (!(a.x | a.y | a.z | a.w)) return;
#pragma unroll
            for (int hq = 0; hq < 2; hq++) {
                u64 w64 = hq ? (((u64)a.w << 32) | (u64)a.z) : (((u64)a.y << 32) | (u64)a.x);
                while (w64) {
                    const int bit = __ffsll((long long)w64) - 1 + 64 * hq;  // bit of the 128-bit load
                    w64 &= w64 - 1ull;
                    int x, b, h;
                    if (ntl <= 2) {  // 16-bit words: word = bit >> 4; inside it tile h = bit 3, compare b = bits 0..2
                        x = 8 * l + (bit >> 4);
                        h = (bit >> 3) & 1;
                        b = bit & 7;
                    } else {         // 32-bit words: tile h = bit 4, compare b = bits 0..3
                        x = 4 * l + (bit >> 5);
                        h = (bit >> 4) & 1;
                        b = bit & 15;
                    }
                    const int ln = x & 63, pr = x >> 6, n = ln & 15, g = ln >> 4;
                    P.a_rec[o++] = make_uint2((u32)(first + (b >> 2) * 16 + 4 * g + (b & 3)), posb + (u32)((2 * pr + h) * 16 + n));
                }
            }
        };
        if (small) {
#pragma unroll
            for (int u = 0; u < KEEP; u++)
                if (u < nu) walk(tid + u * A1R_NT, keep[u]);  // (words past the item's end are zero)
        } else {
            for (int l = tid; l < nload; l += A1R_NT) walk(l, ((const uint4 *)bm)[l]);
        }
    }
}

// ---- exact distances of the records ----------------------------------------------------------------------------------------------
// One block of sixteen waves per CU walks the flat list in rounds of 2048 records (two per thread; eight waves and 1024 records with
// 16-dimensional sub-quantizers), every slot used.  The pairs of a round form a contiguous slot range [a_rnd], handled 64 rows at a
// time: their exact residuals (k_a1_rows) arrive as one contiguous block and stand in LDS for the window.  Sub-quantizer by
// sub-quantizer the codebook slice pq[s] (ks x dsub doubles) stands in LDS too, and every record adds its table entry
// sum_t (r[t] - pq[s][code_s][t])^2, t ascending from 0.0 (IVFPQ.java:531-534), to its running sum in sub-quantizer order (:435-438):
// the bits of the fp64 table lookup.  A thread reads a random 64-byte entry from LDS instead of from L2 (one exact distance is
// m x dsub x 8 bytes of codebook: 1 KiB at 16 x 8, and ~110 of them per query).
// What the structure is for (measured, batch 131072: 14.5 M records): a block per item with eight records per thread left half the
// slots empty and spilled (2.3 ms); rounds over the flat list 1.3 ms, of which 0.6 ms were the rounds' own chains of dependent loads
// (records -> codes; order[] -> cells[] -> list_off[] -> rows) with one block per CU and nothing to overlap them -- hence the rows and
// the pair records precomputed slot by slot, the NEXT round's records and slot range requested a round ahead, and the slices
// double-buffered in LDS AND in registers (slice s + 2 is requested while slice s is used and slice s + 1, requested an iteration ago,
// is stored: one barrier per sub-quantizer, two iterations for the L2 round trip; the loads carry no predicate -- behind a predicated
// load the compiler waits for ALL outstanding loads).  (Two slices per barrier interval -- four LDS buffers, the next interval's two
// slices requested while this one's are used -- measured SLOWER, 0.90 -> 1.23 ms: the store of the prefetched pair then waits for
// loads issued in the same interval.)  Entries are 16 bytes apart from a multiple of 64 and rows 16 bytes apart from a
// multiple of 256, so that the 16 lanes of a ds_read_b128 group spread over all banks (a 64-byte stride leaves them 4 slots).
#define A1V_NT(dsub) ((dsub) >= 16 ? 512 : 1024)    // (16-dimensional entries: 64 registers of operands per record -- eight waves with 256 registers each)
#define A1V_RND(dsub) ((dsub) >= 16 ? 512 : 2048)   // records per round
#define A1V_ROWS 64
struct A1VLds {
    size_t estr, rstr, pq, rows, T, beg, q, rank, cnt, base, total;
    __host__ __device__ A1VLds(int m, int dsub) {
        estr = (size_t)dsub * 8 + 16;
        rstr = (size_t)m * dsub * 8 + 16;
        size_t o = 0;
        pq = o; o += 2 * 256 * estr;
        rows = o; o += A1V_ROWS * rstr;
        T = o; o += A1V_ROWS * 8;
        beg = o; o += A1V_ROWS * 8;
        q = o; o += A1V_ROWS * 4;
        rank = o; o += A1V_ROWS * 4;
        cnt = o; o += A1V_ROWS * 4;
        base = o; o += A1V_ROWS * 4;
        total = (o + 15) & ~(size_t)15;
    }
};
template <int M, int DSUB>
__global__ __launch_bounds__(A1V_NT(DSUB), 1) void k_a1_verify(const MfmaParams P) {
    constexpr int D = M * DSUB;
    constexpr int ESTR = DSUB * 8 + 16;      // bytes between codebook entries in LDS
    constexpr int RSTR = D * 8 + 16;         // bytes between residual rows in LDS
    constexpr int PIECES = DSUB / 2;         // 16-byte pieces per entry
    constexpr int NW = (M + 7) / 8;
    constexpr int NT = A1V_NT(DSUB), RND = A1V_RND(DSUB);
    constexpr int R = RND / NT;              // records per thread and round
    constexpr int RP = (A1V_ROWS * (D / 2) + NT - 1) / NT;  // 16-byte pieces of a window's rows per thread
    static_assert((M & 1) == 0, "the loop over the sub-quantizers is unrolled by two");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const A1VLds L(M, DSUB);
    unsigned char *s_pq = smem + L.pq;       // [2][256][ESTR]
    unsigned char *s_rows = smem + L.rows;   // [64][RSTR]
    u64 *s_T = (u64 *)(smem + L.T);
    long long *s_beg = (long long *)(smem + L.beg);
    int *s_q = (int *)(smem + L.q);
    int *s_rank = (int *)(smem + L.rank);
    u32 *s_cnt = (u32 *)(smem + L.cnt);
    u32 *s_base = (u32 *)(smem + L.base);
    const int tid = threadIdx.x, lane = tid & 63;
    const int nv = *P.n_groups * P.nsub;
    const int ks = P.S.ks;
    const u32 tot_raw = P.a_icnt[nv];
    const u32 total = tot_raw < P.a_rec_cap ? tot_raw : P.a_rec_cap;
    const u32 nrnd = (total + (u32)RND - 1u) / (u32)RND;
    constexpr int PQP = (256 * PIECES + NT - 1) / NT;  // 16-byte pieces of a codebook slice per thread
    u32 nver = 0;
    auto slice_load = [&](const int s, double2 (&pv)[PQP]) {
        const double2 *src = (const double2 *)(P.pq + (size_t)s * ks * DSUB);
#pragma unroll
        for (int i = 0; i < PQP; i++) {
            const int idx = tid + i * NT;
            pv[i] = src[idx < ks * PIECES ? idx : 0];  // (no predicate; entries beyond ks are never referenced)
        }
    };
    auto slice_store = [&](const int buf, const double2 (&pv)[PQP]) {
        unsigned char *dp = s_pq + (size_t)buf * 256 * ESTR;
#pragma unroll
        for (int i = 0; i < PQP; i++) {
            const int idx = tid + i * NT;
            if (idx < 256 * PIECES) *(double2 *)(dp + (size_t)(idx / PIECES) * ESTR + (size_t)(idx % PIECES) * 16) = pv[i];
        }
    };
    auto rec_load = [&](const u32 rnd, uint2 (&rc)[R], uint2 &rr) {
        const u32 rc_ = rnd < nrnd ? rnd : (nrnd ? nrnd - 1u : 0u);  // (past the end: the last round again, not used)
#pragma unroll
        for (int j = 0; j < R; j++) {
            const u32 idx = rc_ * (u32)RND + (u32)(j * NT + tid);
            rc[j] = P.a_rec[idx < total ? idx : (total ? total - 1u : 0u)];
        }
        rr = P.a_rnd[rc_];
    };
    if (nrnd == 0) return;
#ifdef A1V_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_readcyclecounter(), t1;
#define A1V_TICK(i) do { t1 = __builtin_readcyclecounter(); tacc[i] += t1 - t0; t0 = t1; } while (0)
#else
#define A1V_TICK(i) do { } while (0)
#endif
    uint2 rec_n[R], rr_n;
    rec_load(blockIdx.x, rec_n, rr_n);
    for (u32 rnd = blockIdx.x; rnd < nrnd; rnd += gridDim.x) {
        // ---- the round's records (requested a round ago) and their codes (as 64-bit words: byte s is picked by selects and a shift --
        //      an array indexed by the loop counter would live in scratch memory) ----
        uint2 rec[R];
        bool valid[R];
        u64 cw[R][NW];
        const u32 slo = rr_n.x, shi = rr_n.y;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const u32 idx = rnd * (u32)RND + (u32)(j * NT + tid);
            rec[j] = rec_n[j];
            valid[j] = idx < total && rec[j].x != 0xFFFFFFFFu;
            CodeVec<M, unsigned char> cv;
            cv.load((const unsigned char *)P.S.codes + (size_t)(valid[j] ? rec[j].y : 0u) * M);
#pragma unroll
            for (int i = 0; i < NW; i++) cw[j][i] = (u64)cv.wd[2 * i] | (2 * i + 1 < CodeVec<M, unsigned char>::WORDS ? (u64)cv.wd[2 * i + 1] << 32 : 0ull);
        }
        rec_load(rnd + gridDim.x, rec_n, rr_n);  // the next round's (block-uniform address arithmetic; in flight under this round)
        if (slo > shi) continue;                 // (block-uniform: a round of invalid records)
        A1V_TICK(0);
#ifdef A1V_TIMING
        tacc[4]++;
#endif
        for (u32 w0 = slo; w0 <= shi; w0 += A1V_ROWS) {  // 64 pairs at a time (one window unless k is tiny)
            const int nr = (int)(shi - w0 + 1 < (u32)A1V_ROWS ? shi - w0 + 1 : (u32)A1V_ROWS);
            double2 pa[PQP], pb[PQP];
            slice_load(0, pa);
            // the window's rows: one contiguous block of nr x D doubles; the pairs' records
            double2 rw[RP];
            const double *rsrc = (P.R ? P.R : P.a_rows) + (size_t)w0 * D;
#pragma unroll
            for (int i = 0; i < RP; i++) {
                const int idx = tid + i * NT;
                rw[i] = *(const double2 *)(rsrc + (size_t)(idx < nr * (D / 2) ? idx : 0) * 2);
            }
            const u32 mslot = w0 + (u32)(tid < nr ? tid : 0);
            const int4 mt = P.a_meta[mslot];
            const u64 Tq = P.a_metaT[mslot];
            A1V_TICK(5);
            __syncthreads();  // (the previous window / round is done with the LDS)
            A1V_TICK(6);
#pragma unroll
            for (int i = 0; i < RP; i++) {
                const int idx = tid + i * NT;
                if (idx < nr * (D / 2)) *(double2 *)(s_rows + (size_t)(idx / (D / 2)) * RSTR + (size_t)(idx % (D / 2)) * 16) = rw[i];
            }
            if (tid < nr) {
                s_q[tid] = mt.x;
                s_rank[tid] = mt.y;
                s_beg[tid] = (long long)(((u64)(u32)mt.w << 32) | (u64)(u32)mt.z);
                s_T[tid] = Tq;
            }
            A1V_TICK(7);
            slice_store(0, pa);
            slice_load(1, pa);  // (M >= 2)
            bool act[R];
            u32 rowb[R];
            double d[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                act[j] = valid[j] && rec[j].x >= w0 && rec[j].x - w0 < (u32)A1V_ROWS;
                rowb[j] = act[j] ? (rec[j].x - w0) * (u32)RSTR : 0u;
                d[j] = 0.0;
            }
            __syncthreads();
            A1V_TICK(1);
            auto compute = [&](const int s) {
                const unsigned char *bp = s_pq + (size_t)(s & 1) * 256 * ESTR;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    u64 wsel = cw[j][0];
#pragma unroll
                    for (int i = 1; i < NW; i++) wsel = (s >> 3) == i ? cw[j][i] : wsel;
                    const u32 byte = (u32)(wsel >> (8 * (s & 7))) & 0xFFu;
                    // (16-byte LDS reads spelled out: the row offset is a runtime multiple of 16 the compiler cannot see)
                    const double2 *pe = (const double2 *)(bp + (size_t)byte * ESTR);
                    const double2 *re = (const double2 *)(s_rows + rowb[j] + (size_t)s * (DSUB * 8));
                    double2 pv2[PIECES], rv2[PIECES];
#pragma unroll
                    for (int t = 0; t < PIECES; t++) {
                        pv2[t] = pe[t];
                        rv2[t] = re[t];
                    }
                    double e1 = 0.0;
#pragma unroll
                    for (int t = 0; t < PIECES; t++) {
                        const double d0 = rv2[t].x - pv2[t].x, d1 = rv2[t].y - pv2[t].y;
                        e1 += d0 * d0;
                        e1 += d1 * d1;
                    }
                    d[j] = s == 0 ? e1 : d[j] + e1;  // (0.0 + e_0 = e_0: IVFPQ.java:435-438)
                }
            };
            // iteration s: request slice s + 2, use slice s, store slice s + 1 (requested an iteration ago) into the buffer slice s - 1
            // was read from (everyone is past that iteration's barrier); past the last slice the loads repeat it, into a buffer nobody reads
#pragma unroll 1
            for (int s = 0; s < M; s += 2) {
                slice_load(s + 2 < M ? s + 2 : M - 1, pb);
                compute(s);
                slice_store(1, pa);  // slice s + 1
                __syncthreads();
                slice_load(s + 3 < M ? s + 3 : M - 1, pa);
                compute(s + 1);
                slice_store(0, pb);  // slice s + 2
                __syncthreads();
            }
            A1V_TICK(2);
            // the pool entries: counted per row in LDS, ONE global reservation per row (2048 single atomics on ~19 addresses took 8 us)
            u32 li[R];
            bool keep[R];
            if (tid < A1V_ROWS) s_cnt[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < R; j++) {
                const u32 rwi = act[j] ? rec[j].x - w0 : 0u;
                keep[j] = act[j] && dkey(d[j]) <= s_T[rwi];
                li[j] = keep[j] ? atomicAdd(s_cnt + rwi, 1u) : 0u;
                nver += act[j] ? 1u : 0u;
            }
            __syncthreads();
            if (tid < nr) {
                const u32 c = s_cnt[tid];
                s_base[tid] = c ? atomicAdd(P.S.pool_cnt + s_q[tid], c) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (!keep[j]) continue;
                const u32 rwi = rec[j].x - w0;
                const int q = s_q[rwi];
                const u32 slotp = s_base[rwi] + li[j];
                if (slotp < (u32)P.S.poolq) {
                    P.S.pool_key[(size_t)q * P.S.poolq + slotp] = dkey(d[j]);
                    P.S.pool_val[(size_t)q * P.S.poolq + slotp] = ((u64)(u32)s_rank[rwi] << 32) | (u64)(u32)((long long)rec[j].y - s_beg[rwi]);
                } else {
                    P.redo[q] = 1;
                }
            }
            A1V_TICK(3);
        }
    }
#ifdef A1V_TIMING
    if (tid == 0 && blockIdx.x < 4)
        printf("[a1v] block %d: %llu rounds; cycles per round: records+codes %llu, window: issue %llu barrier %llu rows->LDS %llu rest %llu, s loop %llu, emit %llu\n", (int)blockIdx.x, tacc[4],
               tacc[0] / (tacc[4] ? tacc[4] : 1), tacc[5] / (tacc[4] ? tacc[4] : 1), tacc[6] / (tacc[4] ? tacc[4] : 1), tacc[7] / (tacc[4] ? tacc[4] : 1), tacc[1] / (tacc[4] ? tacc[4] : 1), tacc[2] / (tacc[4] ? tacc[4] : 1), tacc[3] / (tacc[4] ? tacc[4] : 1));
#endif
    if (P.stat) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nver += __shfl_xor(nver, off);
        if (lane == 0 && nver) {
            atomicAdd(P.stat, (unsigned long long)nver);       // (mmidx_stats::verified_codes)
            atomicAdd(P.stat + 9, (unsigned long long)nver);   // (mmidx_stats::mfma_survivors)
        }
    }
}
