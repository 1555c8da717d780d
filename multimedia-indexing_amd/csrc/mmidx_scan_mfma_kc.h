// mmidx_scan_mfma_kc.h -- K3mk: the matrix-core lower bound of pass B (mmidx_scan_mfma.h) for LONG vectors, D a multiple of 128
// beyond 128 -- the reference's flagship shape is 1024 dimensions in 64 x 256 (YFCC100MExample.java:85-90), probed at up to 64
// lists (Example.java:96-97; every probed code is offered, IVFPQ.java:429-446).
//
// K3m keeps the fp16 codebook of ALL dimensions in LDS (64 KiB at D = 128); at D = 1024 it is 512 KiB.  Here the dimensions go
// through in CHUNKS of 128: the accumulators of a wave's code tiles stay in registers across the chunks, the codebook chunk in LDS
// is replaced between them, and the queries' fp16 residual fragments of the chunk are re-read from a staging matrix --
//   * k_pair_resid16: once per call, for every pair pass B kept: the exact residual r = c - q (fp64, IVFPQ.java:645, rows in
//     transformed order), scaled by ONE power of two for the whole launch (from max|centroid element| + max|query element|, so the
//     per-code accumulator start values -||x||^2 s^2 / 2 are shared by all rows) and rounded to fp16 -> R16[slot][D]; ||r||^2 in fp64;
//   * k_scan_mfma_kc<DSUB, TPW>: an item is (group of <= 32 pairs that probe a list, piece of 64 TPW codes of it).  Each wave owns
//     TPW tiles of 16 codes: acc[TPW][2 row tiles] (8 TPW registers).  For every 128-dimension chunk: the block loads the chunk's
//     codebook rows (64 KiB) into LDS, every wave its A fragments (2 row tiles x 4 k steps), then tile by tile the code bytes of the
//     chunk's sub-quantizers, four 16-byte gathers and eight MFMAs.  After the last chunk: K3m's compares, survivor records, upper-bound
//     histogram, threshold updates -- same certificate (D + 4 accumulation steps), same records, same verification and redo kernels.
// Price of the chunking: the codebook travels L2 -> LDS once per (piece, chunk): 1 KiB per code and group at TPW = 16.
#pragma once
#include "mmidx_scan_mfma.h"

#define MFK_G 32  // pairs per group: two 16-row tiles (a batch of 4096 queries probing 64 of 8192 lists gives ~32 pairs per list)

struct MfmaKcParams {
    MfmaParams M;
    const unsigned short *R16;  // [slots][D] fp16 residuals, scaled by 2^er
    const double *nrow;         // [slots] ||r||^2
    const int32_t *scale;       // [0] er (exponent of the residuals' scale), [1] 1: usable
    int D;                      // a multiple of 128
};

struct MfmaKcLds {
    size_t cb, buf, row, misc, total;
    __host__ __device__ MfmaKcLds() {
        size_t o = 0;
        cb = o; o += 65536;                       // one 128-dimension chunk of the codebook: [16][256] rows of 8 halfs
        buf = o; o += 4 * (size_t)MF_BUF * 16;    // the waves' survivor buffers
        row = o; o += MFK_G * sizeof(MfmaRow);
        misc = o; o += 64;
        total = (o + 15) & ~(size_t)15;
    }
};

// largest |element| of X[n] (non-negative floats order like their bit patterns)
__global__ void k_maxabs_f64(const double *__restrict__ X, long long n, u32 *__restrict__ out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double v = fabs(X[i]);
        float f = (float)v;
        if ((double)f < v) f = __int_as_float(__float_as_int(f) + 1);  // (round up; inf / nan propagate as large bit patterns)
        m = fmaxf(m, f);
        if (!(v == v)) m = __int_as_float(0x7F800000);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (u32)__float_as_int(m));
}

// the launch's residual scale from the two bounds: |r_i| <= cmax + qmax
__global__ void k_resid_scale(const u32 *__restrict__ qmax_bits, float cmax, int ep, int32_t *__restrict__ scale) {
    const float b = __int_as_float((int)*qmax_bits) + cmax;
    int er = 0, ok = b < 1e30f;
    if (ok && b > 0.f) {
        int ex;
        (void)frexpf(b, &ex);
        er = 13 - ex;
    }
    ok = ok && er + ep > -100 && er + ep < 100 && er - ep > -60 && er - ep < 60;
    scale[0] = er;
    scale[1] = ok;
}

// one wave per kept pair (slot in order[]): fp16 residual row + ||r||^2
__global__ __launch_bounds__(256) void k_pair_resid16(const double *__restrict__ Q, const double *__restrict__ coarse, const int32_t *__restrict__ cells,
                                                      const int32_t *__restrict__ order, const int32_t *__restrict__ n_order, long long n_flat,
                                                      int w, int D, int ivf, const int32_t *__restrict__ scale, unsigned short *__restrict__ R16,
                                                      double *__restrict__ nrow) {
    const long long n = n_order ? (long long)*n_order : n_flat;
    const long long slot = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n) return;
    const int lane = threadIdx.x & 63;
    const int e = order[slot];
    const int q = e / w;
    int cell = ivf ? cells[e] : 0;
    cell = cell < 0 ? 0 : cell;
    const float s_r = scale[1] ? ldexpf(1.f, scale[0]) : 1.f;
    const double *cc = coarse + (size_t)cell * D, *qv = Q + (size_t)q * D;
    double nr = 0.0;
    for (int d = 2 * lane; d < D; d += 128) {
        const double2 c2 = *(const double2 *)(cc + d), q2 = *(const double2 *)(qv + d);
        const double r0 = ivf ? c2.x - q2.x : q2.x - c2.x, r1 = ivf ? c2.y - q2.y : q2.y - c2.y;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 p2;
        p2[0] = (_Float16)((float)r0 * s_r);
        p2[1] = (_Float16)((float)r1 * s_r);
        *(h2 *)(R16 + (size_t)slot * D + d) = p2;
        nr += r0 * r0;
        nr += r1 * r1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nr += __shfl_xor(nr, off);
    if (lane == 0) nrow[slot] = nr;
}

// survivors of one tile (the registers of its NTL row tiles): K3m's compact lane-level path
template <int NTL, int CAP>
__device__ __forceinline__ void mfk_tile_survivors(const MfmaParams &P, const mf_f4 (&acc)[NTL], const float (&thr)[NTL][4], const long long pos,
                                                   const long long c1, const float kd, const MfmaRow *s_row, uint4 *s_buf, u32 &bufn, const int first,
                                                   const int g) {
    u32 bits = 0;
#pragma unroll
    for (int b = NTL * 4 - 1; b >= 0; b--)
        asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(acc[b >> 2][b & 3]), "v"(thr[b >> 2][b & 3]) : "vcc");
    if (pos >= c1) bits = 0;
    u64 act = __builtin_amdgcn_ballot_w64(bits != 0);
    while (act) {
        if (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const u32 m0 = 0u - ((u32)b & 1u), m1 = 0u - (((u32)b >> 1) & 1u), m2 = 0u - (((u32)b >> 2) & 1u);
            u32 v4[4], v2[2];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 lo = 2 * j < NTL * 4 ? (u32)__float_as_int(acc[(2 * j) >> 2][(2 * j) & 3]) : 0u,
                          hi = 2 * j + 1 < NTL * 4 ? (u32)__float_as_int(acc[(2 * j + 1) >> 2][(2 * j + 1) & 3]) : 0u;
                v4[j] = (hi & m0) | (lo & ~m0);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) v2[j] = (v4[2 * j + 1] & m1) | (v4[2 * j] & ~m1);
            const float a = __int_as_float((int)((v2[1] & m2) | (v2[0] & ~m2)));
            const int qs = (b >> 2) * 16 + 4 * g + (b & 3);
            const MfmaRow rw = s_row[qs];
            const float lbf = fmaf(a, kd, rw.cd);
            const float xb = fmaf(a, rw.kq, rw.cq);
            const u32 bk = (rw.kq != 0.f && xb < 255.f) ? (xb > 0.f ? (u32)(int)xb : 0u) : 0xFFFFFFFFu;
            const u32 o = bufn + mf_mbcnt(act);
            if (o < (u32)CAP) s_buf[o] = make_uint4((u32)(first + qs), (u32)pos, (u32)__float_as_int(lbf), bk);
            else P.redo[rw.q] = 1;
        }
        bufn += (u32)__popcll(act);
        act = __builtin_amdgcn_ballot_w64(bits != 0);
    }
}

template <int DSUB, int TPW>
__global__ __launch_bounds__(MF_NT, TPW > 8 ? 1 : 2) void k_scan_mfma_kc(const MfmaKcParams K) {
    static_assert(DSUB == 8 || DSUB == 16, "sub-quantizers of 8 or 16 dimensions");
    constexpr int NTL = 2, NB = 32 / DSUB;  // row tiles; code bytes per lane and 128-dimension chunk
    const MfmaParams &P = K.M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const MfmaKcLds L;
    MfmaRow *s_row = (MfmaRow *)(smem + L.row);
    u32 *s_misc = (u32 *)(smem + L.misc);
    MfmaChunk ck{0u, 0u, 0u};
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const u32 lds_cb = (u32)(size_t)(__attribute__((address_space(3))) unsigned char *)(smem + L.cb);
    const u32 lane_base = lds_cb + (u32)(4 * g) * 4096u;
    uint4 *s_buf = (uint4 *)(smem + L.buf) + (size_t)wv * MF_BUF;
    const int D = K.D, NK = D / 128, M = D / DSUB;
    const int nv = *P.n_groups * P.nsub;
    if (nv == 0) return;
    const int per = (nv + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const double xmax = P.xmax;
    const double gam = (double)(D + 4) * 0x1p-23 * 1.01;
    const int er = K.scale[0];
    const bool scale_ok = K.scale[1] != 0;
    const double s2 = scale_ok ? ldexp(1.0, er + P.ep) : 1.0, inv_s2 = 1.0 / s2, inv_sr = scale_ok ? ldexp(1.0, -er) : 1.0, inv_sp = ldexp(1.0, -P.ep);
    const double sqrtD = sqrt((double)D) * 1.001;
    const float kinit = (float)(-0.5 * s2), kd = (float)(-2.0 * inv_s2);

    for (;;) {
        __syncthreads();  // (the previous item's LDS reads are done)
        if (tid == 0) {
            s_misc[0] = atomicAdd(P.work + xcd, 1u);
            s_misc[1] = 0;
            s_misc[2] = 0;
        }
        __syncthreads();
        const int it = (int)s_misc[0];
        if (it >= per) break;
        const int v = xcd * per + it;
        if (v >= nv) break;
        const int gi = v / P.nsub, isub = v - gi * P.nsub;
        const int4 gd = P.gdesc[gi];
        const int cell = gd.x, first = gd.y, np = gd.z;
        const long long beg = P.S.list_off[cell];
        const long long len = P.S.list_off[cell + 1] - beg;
        const long long c0 = (long long)isub * P.sub;
        if (c0 >= len) continue;
        const long long c1 = (c0 + P.sub < len) ? c0 + P.sub : len;
        const unsigned char *codes = (const unsigned char *)P.S.codes + (size_t)beg * M;
        const float *xn = P.xn + beg + c0;

        // ---- (a) the rows' constants (K3m's phase (a) without the residuals: those were staged by k_pair_resid16) ----
        if (tid < MFK_G) {
            const int row = tid;
            const int tl = row < np ? row : np - 1;
            const int e = P.S.order[first + tl];
            const int q = e / P.S.w;
            const double nr = K.nrow[first + tl], nrm = sqrt(nr) * (1.0 + 1e-12);
            const double err = nrm * xmax * (4.02 * 0x1p-11 + 2.004 * gam) + xmax * xmax * (1.001 * gam + 0x1p-24) +
                               2.02 * sqrtD * 0x1p-14 * (xmax * inv_sr + nrm * inv_sp) + 2.0 * D * 0x1p-28 * inv_s2 +
                               0x1p-19 * (nr + xmax * xmax + 2.0 * nrm * xmax) + 1e-300;
            const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float th = __int_as_float(0x7F800000);
            double inv0 = 0.0;
            if (row < np) {
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
            rw.slot = first + row;
            rw.inv0 = 0.f;
            rw.pad = 0;
            s_row[row] = rw;
        }
        __syncthreads();
        float thr[NTL][4];
#pragma unroll
        for (int rt = 0; rt < NTL; rt++)
#pragma unroll
            for (int i = 0; i < 4; i++) thr[rt][i] = s_row[rt * 16 + 4 * g + i].thr;
        float thrmin = thr[0][0];
#pragma unroll
        for (int b = 1; b < NTL * 4; b++) thrmin = __builtin_fminf(thrmin, thr[b >> 2][b & 3]);

        // ---- (b) the wave's TPW tiles through all chunks of 128 dimensions ----
        // tile ti of wave wv = tile (ti * 4 + wv) of the piece: the four waves read neighbouring codes
        const int ntiles = (int)((c1 - c0 + 15) >> 4);
        const u32 last = (u32)(c1 - c0 - 1);
        const unsigned char *cbase = codes + (size_t)c0 * M + (u32)(NB * g);
        mf_f4 acc[TPW][NTL];
        const unsigned short *arow[NTL];
#pragma unroll
        for (int rt = 0; rt < NTL; rt++) {
            int r = rt * 16 + n;
            r = r < np ? r : np - 1;
            arow[rt] = K.R16 + (size_t)(first + r) * D + (size_t)(4 * g) * 8;
        }
        for (int kc = 0; kc < NK; kc++) {
            __syncthreads();  // (everybody is done with the previous chunk's rows)
            {
                const uint4 *src = (const uint4 *)(P.pq16 + (size_t)kc * 16 * 256 * 8);
#pragma unroll 1
                for (int i0 = 0; i0 < 16; i0 += 4) {  // (four rows in flight: the accumulators own the registers)
                    uint4 tmp[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) tmp[i] = src[tid + (i0 + i) * MF_NT];
#pragma unroll
                    for (int i = 0; i < 4; i++) ((uint4 *)(smem + L.cb))[tid + (i0 + i) * MF_NT] = tmp[i];
                }
            }
            mf_h8 A[NTL][4];
#pragma unroll
            for (int rt = 0; rt < NTL; rt++)
#pragma unroll
                for (int j = 0; j < 4; j++) A[rt][j] = *(const mf_h8 *)(arow[rt] + (size_t)kc * 128 + (size_t)j * 8);
            __syncthreads();
#pragma unroll
            for (int ti = 0; ti < TPW; ti++) {
                const int tt = ti * 4 + wv;
                if (tt >= ntiles) continue;  // (wave-uniform)
                u32 p = (u32)tt * 16u + (u32)n;
                p = p < last ? p : last;
                const unsigned char *cp = cbase + (size_t)p * M + (size_t)kc * (128 / DSUB);
                u32 c;
                if constexpr (NB == 4) c = *(const u32 *)cp;
                else c = (u32) * (const unsigned short *)cp;
                mf_h8 B[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32 byte = (c >> (8 * (j / (DSUB / 8)))) & 0xFFu;
                    B[j] = *(const __attribute__((address_space(3))) mf_h8 *)(size_t)(lane_base + (byte << 4) + (u32)j * 4096u);
                }
                if (kc == 0) {
                    const float ci = xn[p] * kinit;
                    const mf_f4 c4 = {ci, ci, ci, ci};
#pragma unroll
                    for (int rt = 0; rt < NTL; rt++) acc[ti][rt] = c4;
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int rt = 0; rt < NTL; rt++) acc[ti][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[rt][j], B[j], acc[ti][rt], 0, 0, 0);
            }
        }
        // ---- (c) compares, survivors ----
        u32 bufn = 0;
#pragma unroll
        for (int ti = 0; ti < TPW; ti++) {
            const int tt = ti * 4 + wv;
            if (tt >= ntiles) continue;
            float mxa = acc[ti][0][0];
#pragma unroll
            for (int b = 1; b + 1 < NTL * 4; b += 2) mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, acc[ti][b >> 2][b & 3]), acc[ti][(b + 1) >> 2][(b + 1) & 3]);
            mxa = __builtin_fmaxf(mxa, acc[ti][NTL - 1][3]);
            if (__builtin_amdgcn_ballot_w64(mxa >= thrmin)) {
                const long long pos = c0 + (long long)tt * 16 + n;
                mfk_tile_survivors<NTL, MF_BUF>(P, acc[ti], thr, pos, c1, kd, s_row, s_buf, bufn, first, g);
                if (bufn >= MF_BUF / 2) {
                    mf_flush(P, ck, s_buf, bufn < MF_BUF ? bufn : MF_BUF, s_row, s_misc + 1, first, lane);
                    bufn = 0;
                }
            }
        }
        if (bufn) mf_flush(P, ck, s_buf, bufn < MF_BUF ? bufn : MF_BUF, s_row, s_misc + 1, first, lane);
        __syncthreads();
        // ---- (d) thresholds from the union of the survivors' upper bounds (as K3m's phase (c)) ----
        {
            const u32 tall = s_misc[1];
            for (int qs = wv; qs < MFK_G; qs += MF_NT / 64) {
                if (!((tall >> qs) & 1u)) continue;
                const int q = s_row[qs].q;
                if (!(s_row[qs].kq != 0.f)) continue;
                const double inv0q = 256.0 / keyd(P.T0[q]);
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
                        atomicMin(P.S.T + q, dkey((double)(b + 1) / inv0q * (1.0 + 1e-12)));
                    }
                }
            }
        }
    }
    for (u32 i = ck.used + (u32)lane; i < ck.cap; i += 64)
        if (ck.base + i < P.surv_cap) P.surv[ck.base + i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
}

// ---- K3mk for D a multiple of 256: eight waves per block, LDS-DMA double buffering, contiguous code bytes -------------------------
// What k_scan_mfma_kc measured on the reference's flagship shape (95 M x 1024-d in 64 x 256, 4096 queries between the cells, w = 64:
// 28 ms per batch): removing the codebook copy (global -> registers -> LDS, 64 KiB per chunk and block) saved 8 ms, removing the
// code loads (2 bytes per lane out of 64-byte rows, the rows touched again in every chunk) 7 ms, removing the matrix instructions
// nothing.  Here
//   * the dimensions are dealt out in QUARTERS: lane group g owns dimensions [g D/4, (g+1) D/4) (any bijection of the k index works
//     as long as A and B agree), so that the code bytes a lane needs over all chunks are CONTIGUOUS: one 8-byte load per lane and
//     8 code bytes (CG; 4 where m / 4 is not a multiple of 8; 16 compiles but its 32 registers of code words make the chunk loop
//     spill) -- a tile of 16 codes x 64 bytes comes in whole lines, the second half of a line from L1 / L2;
//   * chunk kc = the 8-dimension units {g NK 4 + 4 kc + j}; its codebook rows (64 KiB) and the 32 residual rows' units (8 KiB,
//     XOR-swizzled: the 16 lanes of a fragment read 16 different slots) arrive by LDS-DMA (global_load_lds_dwordx4) in the buffer
//     the matrix cores are NOT working from: two static buffer pairs, the chunk loop unrolled by two, one barrier per chunk;
//   * 512 threads: 64 code tiles (1024 codes) per item and codebook copy, one block per CU (156 KiB of LDS);
//   * groups of up to 16 pairs skip the second row tile's matrix instructions.
// Certificate, records, verification and redo as K3m.
#define MFK2_NT 512
#define MFK2_TPW 8
#define MFK2_CAP 96  // survivor records a wave stages between flushes

template <int NTL, int CAP>
__device__ __forceinline__ void mfk2_tile_survivors(const MfmaParams &P, MfmaChunk &ck, const mf_f4 (&acc)[NTL], const float (&thr)[NTL][4], const long long pos,
                                                    const long long c1, const float kd, const MfmaRow *s_row, uint4 *s_buf, u32 *s_touch, u32 &bufn,
                                                    const int first, const int g, const int lane) {
    u32 bits = 0;
#pragma unroll
    for (int b = NTL * 4 - 1; b >= 0; b--)
        asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(acc[b >> 2][b & 3]), "v"(thr[b >> 2][b & 3]) : "vcc");
    if (pos >= c1) bits = 0;
    u64 act = __builtin_amdgcn_ballot_w64(bits != 0);
    while (act) {
        if (bufn > (u32)(CAP - 64)) {  // (wave-uniform) room for one record per lane
            mf_flush(P, ck, s_buf, bufn, s_row, s_touch, first, lane);
            bufn = 0;
        }
        if (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            const u32 m0 = 0u - ((u32)b & 1u), m1 = 0u - (((u32)b >> 1) & 1u), m2 = 0u - (((u32)b >> 2) & 1u);
            u32 v4[4], v2[2];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 lo = (u32)__float_as_int(acc[(2 * j) >> 2][(2 * j) & 3]), hi = (u32)__float_as_int(acc[(2 * j + 1) >> 2][(2 * j + 1) & 3]);
                v4[j] = (hi & m0) | (lo & ~m0);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) v2[j] = (v4[2 * j + 1] & m1) | (v4[2 * j] & ~m1);
            const float a = __int_as_float((int)((v2[1] & m2) | (v2[0] & ~m2)));
            const int qs = (b >> 2) * 16 + 4 * g + (b & 3);
            const MfmaRow rw = s_row[qs];
            const float lbf = fmaf(a, kd, rw.cd);
            const float xb = fmaf(a, rw.kq, rw.cq);
            const u32 bk = (rw.kq != 0.f && xb < 255.f) ? (xb > 0.f ? (u32)(int)xb : 0u) : 0xFFFFFFFFu;
            s_buf[bufn + mf_mbcnt(act)] = make_uint4((u32)(first + qs), (u32)pos, (u32)__float_as_int(lbf), bk);
        }
        bufn += (u32)__popcll(act);
        act = __builtin_amdgcn_ballot_w64(bits != 0);
    }
}

template <int DSUB, int CG>
__global__ __launch_bounds__(MFK2_NT, 1) void k_scan_mfma_kc2(const MfmaKcParams K) {
    static_assert(DSUB == 8 || DSUB == 16, "sub-quantizers of 8 or 16 dimensions");
    static_assert(CG == 4 || CG == 8 || CG == 16, "code bytes a lane loads at a time");
    constexpr int NTL = 2, NB = 32 / DSUB, CPG = CG / NB, TPW = MFK2_TPW, NW = MFK2_NT / 64;
    static_assert(CPG >= 2 && CPG % 2 == 0, "the chunk loop is unrolled by two");
    static_assert(MFK_G == 4 * NW, "one DMA instruction per wave brings four residual rows");
    __shared__ __attribute__((aligned(1024))) unsigned char cb0[65536];
    __shared__ __attribute__((aligned(1024))) unsigned char cb1[65536];
    __shared__ __attribute__((aligned(1024))) unsigned char ab0[8192];
    __shared__ __attribute__((aligned(1024))) unsigned char ab1[8192];
    __shared__ uint4 s_bufs[NW * MFK2_CAP];
    __shared__ MfmaRow s_row[MFK_G];
    __shared__ u32 s_misc[8];
    const MfmaParams &P = K.M;
    MfmaChunk ck{0u, 0u, 0u};
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: addresses and branches by wave)
    const int n = lane & 15, g = lane >> 4;
    uint4 *s_buf = s_bufs + (size_t)wv * MFK2_CAP;
    const int D = K.D, NK = D / 128, M = D / DSUB;
    const int nv = *P.n_groups * P.nsub;
    if (nv == 0) return;
    const int per = (nv + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const double xmax = P.xmax;
    const double gam = (double)(D + 4) * 0x1p-23 * 1.01;
    const int er = K.scale[0];
    const bool scale_ok = K.scale[1] != 0;
    const double s2 = scale_ok ? ldexp(1.0, er + P.ep) : 1.0, inv_s2 = 1.0 / s2, inv_sr = scale_ok ? ldexp(1.0, -er) : 1.0, inv_sp = ldexp(1.0, -P.ep);
    const double sqrtD = sqrt((double)D) * 1.001;
    const float kinit = (float)(-0.5 * s2), kd = (float)(-2.0 * inv_s2);
    const unsigned char *pqb = (const unsigned char *)P.pq16;

    for (;;) {
        __syncthreads();  // (the previous item's LDS reads are done)
        if (tid == 0) {
            s_misc[0] = atomicAdd(P.work + xcd, 1u);
            s_misc[1] = 0;
            s_misc[2] = 0;
        }
        __syncthreads();
        const int it = (int)s_misc[0];
        if (it >= per) break;
        const int v = xcd * per + it;
        if (v >= nv) break;
        const int gi = v / P.nsub, isub = v - gi * P.nsub;
        const int4 gd = P.gdesc[gi];
        const int cell = gd.x, first = gd.y, np = gd.z;
        const long long beg = P.S.list_off[cell];
        const long long len = P.S.list_off[cell + 1] - beg;
        const long long c0 = (long long)isub * P.sub;
        if (c0 >= len) continue;
        const long long c1 = (c0 + P.sub < len) ? c0 + P.sub : len;
        const unsigned char *codes = (const unsigned char *)P.S.codes + (size_t)beg * M;
        const float *xn = P.xn + beg + c0;

        // DMA of chunk kc: the codebook rows of its 16 units (slot 4 g + j <- unit g NK 4 + 4 kc + j; 4 KiB = four wave
        // instructions each) and the units of the item's 32 residual rows (rows 4 wv .. 4 wv + 3 from this wave, unit u of row r in
        // slot u ^ (r & 15))
        const int arow_l = 4 * wv + (lane >> 4);
        const unsigned char *asrc = (const unsigned char *)(K.R16 + (size_t)(first + (arow_l < np ? arow_l : np - 1)) * D);
        const int au = (lane & 15) ^ (arow_l & 15);
        const u32 aoff = (u32)((au >> 2) * (NK * 4) + (au & 3)) * 16u;
        auto stage = [&](unsigned char *cb, unsigned char *ab, const int kc) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int ii = i * NW + wv, r = ii >> 2, part = ii & 3;
                const size_t d8 = (size_t)((r >> 2) * (NK * 4) + 4 * kc + (r & 3));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pqb + d8 * 4096 + (size_t)part * 1024 + (size_t)lane * 16),
                                                 (__attribute__((address_space(3))) void *)(cb + ii * 1024), 16, 0, 0);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(asrc + aoff + (size_t)kc * 64),
                                             (__attribute__((address_space(3))) void *)(ab + wv * 1024), 16, 0, 0);
        };
        stage(cb0, ab0, 0);  // (buffer 0 is free: every wave is past the barrier that followed chunk NK - 2 of the previous item)

        // ---- (a) the rows' constants ----
        if (tid < MFK_G) {
            const int row = tid;
            const int tl = row < np ? row : np - 1;
            const int e = P.S.order[first + tl];
            const int q = e / P.S.w;
            const double nr = K.nrow[first + tl], nrm = sqrt(nr) * (1.0 + 1e-12);
            const double err = nrm * xmax * (4.02 * 0x1p-11 + 2.004 * gam) + xmax * xmax * (1.001 * gam + 0x1p-24) +
                               2.02 * sqrtD * 0x1p-14 * (xmax * inv_sr + nrm * inv_sp) + 2.0 * D * 0x1p-28 * inv_s2 +
                               0x1p-19 * (nr + xmax * xmax + 2.0 * nrm * xmax) + 1e-300;
            const u64 T = __hip_atomic_load(P.S.T + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float th = __int_as_float(0x7F800000);
            double inv0 = 0.0;
            if (row < np) {
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
            rw.slot = first + row;
            rw.inv0 = 0.f;
            rw.pad = 0;
            s_row[row] = rw;
        }
        __syncthreads();
        // ---- (b) the item's codes in PASSES of TPW NW tiles (tile ti * NW + wv of a pass is this wave's), each through all chunks ----
        // The rows' constants, the residual rows and the chunk rotation carry over from pass to pass: chunk 0 of the next pass is
        // fetched while the last chunk of this one is worked on, the next pass's code words and accumulator start values are loaded
        // while this pass's accumulators are compared.
        constexpr int PASS = TPW * NW * 16;
        const bool two_rt = np > 16;
        const unsigned char *cbase = codes + (size_t)c0 * M + (size_t)g * (size_t)(M >> 2);  // the lane's quarter of a code
        mf_f4 acc[TPW][NTL];
        u32 cw[TPW][CG / 4];
        int ntiles;
        // one chunk from a buffer pair: the chunk's code bytes are the LOW bytes of the lane's code words, which then move down
        auto chunk = [&](const unsigned char *cb, const unsigned char *ab) {
            mf_h8 A[NTL][4];
#pragma unroll
            for (int rt = 0; rt < NTL; rt++)
#pragma unroll
                for (int j = 0; j < 4; j++) A[rt][j] = *(const mf_h8 *)(ab + (rt * 16 + n) * 256 + (((4 * g + j) ^ n) << 4));
#pragma unroll
            for (int t2 = 0; t2 < TPW; t2 += 2) {  // two tiles per step: eight gathers in flight, then their matrix instructions
                if (t2 * NW + wv >= ntiles) continue;  // (wave-uniform)
                const bool second = (t2 + 1) * NW + wv < ntiles;
                mf_h8 B[2][4];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int ti = t2 + u;
                    const u32 c = cw[ti][0];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u32 byte = (c >> (8 * (j / (DSUB / 8)))) & 0xFFu;
                        B[u][j] = *(const mf_h8 *)(cb + (4 * g + j) * 4096 + (byte << 4));
                    }
                    if constexpr (CG == 4) {
                        cw[ti][0] = c >> (8 * NB);
                    } else if constexpr (NB == 4) {
#pragma unroll
                        for (int x = 0; x + 1 < CG / 4; x++) cw[ti][x] = cw[ti][x + 1];
                    } else {
#pragma unroll
                        for (int x = 0; x + 1 < CG / 4; x++) cw[ti][x] = __builtin_amdgcn_alignbyte(cw[ti][x + 1], cw[ti][x], NB);
                        cw[ti][CG / 4 - 1] >>= 8 * NB;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[t2][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][j], B[0][j], acc[t2][0], 0, 0, 0);
                    if (second) acc[t2 + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][j], B[1][j], acc[t2 + 1][0], 0, 0, 0);
                    if (two_rt) {
                        acc[t2][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1][j], B[0][j], acc[t2][1], 0, 0, 0);
                        if (second) acc[t2 + 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1][j], B[1][j], acc[t2 + 1][1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // the lane's CG code bytes from byte kb of its quarter, for the tiles of the pass at code offset po (tiles beyond the item read
        // its last code: never compared)
        auto load_codes = [&](const u32 po, const u32 lastp, const int kb) {
#pragma unroll
            for (int ti = 0; ti < TPW; ti++) {
                u32 pp = po + (u32)(ti * NW + wv) * 16u + (u32)n;
                pp = pp < lastp ? pp : lastp;
                const unsigned char *cp = cbase + (size_t)pp * M + (size_t)kb;
                if constexpr (CG == 16) {
                    const uint4 t4 = *(const uint4 *)cp;
                    cw[ti][0] = t4.x; cw[ti][1] = t4.y; cw[ti][2] = t4.z; cw[ti][3] = t4.w;
                } else if constexpr (CG == 8) {
                    const uint2 t2 = *(const uint2 *)cp;
                    cw[ti][0] = t2.x; cw[ti][1] = t2.y;
                } else {
                    cw[ti][0] = *(const u32 *)cp;
                }
            }
        };
        const u32 lastp = (u32)(c1 - c0 - 1);  // last code of the item
        float ci[TPW];
        load_codes(0u, lastp, 0);
#pragma unroll
        for (int ti = 0; ti < TPW; ti++) {  // the accumulators start at -||x||^2 s^2 / 2
            u32 pp = (u32)(ti * NW + wv) * 16u + (u32)n;
            pp = pp < lastp ? pp : lastp;
            ci[ti] = xn[pp];
        }
        u32 bufn = 0;
        for (u32 po = 0;; po += PASS) {  // code offset of the pass within the item
#pragma unroll
            for (int ti = 0; ti < TPW; ti++) {
                const float c1v = ci[ti] * kinit;
                const mf_f4 c4 = {c1v, c1v, c1v, c1v};
#pragma unroll
                for (int rt = 0; rt < NTL; rt++) acc[ti][rt] = c4;
            }
            const u32 left = (u32)(c1 - c0) - po;
            const bool more = left > (u32)PASS;
            ntiles = (int)(((more ? (u32)PASS : left) + 15u) >> 4);
            for (int kc = 0; kc < NK; kc += 2) {
                if (kc % CPG == 0 && kc > 0) load_codes(po, lastp, kc * NB);  // (where a quarter has more than CG bytes)
                // every wave waits for ITS part of chunk kc's DMA, then the barrier: the chunk has landed in buffer 0 and nobody reads
                // buffer 1 any more.  (The wait is explicit: a workgroup barrier does not drain the vector-memory counter by itself, and
                // a wait the compiler places later -- in front of the first LDS read -- would also wait for the DMA issued in between.
                // For the same reason no global load may be waited for between a DMA issue and the barrier behind it: the counter is
                // in order.)
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                stage(cb1, ab1, kc + 1);
                chunk(cb0, ab0);
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (kc + 2 < NK) stage(cb0, ab0, kc + 2);
                else if (more) stage(cb0, ab0, 0);
                chunk(cb1, ab1);
            }
            if (more) {  // the next pass's code words and start values: in flight while this pass's accumulators are compared
                load_codes(po + PASS, lastp, 0);
#pragma unroll
                for (int ti = 0; ti < TPW; ti++) {
                    u32 pp = po + (u32)PASS + (u32)(ti * NW + wv) * 16u + (u32)n;
                    pp = pp < lastp ? pp : lastp;
                    ci[ti] = xn[pp];  // (used -- and waited for -- when the next pass starts)
                }
            }
            // ---- (c) compares, survivors (the rows' thresholds come back from LDS: no registers held across the chunks) ----
            float thr[NTL][4];
#pragma unroll
            for (int rt = 0; rt < NTL; rt++)
#pragma unroll
                for (int i = 0; i < 4; i++) thr[rt][i] = s_row[rt * 16 + 4 * g + i].thr;
            float thrmin = thr[0][0];
#pragma unroll
            for (int b = 1; b < NTL * 4; b++) thrmin = __builtin_fminf(thrmin, thr[b >> 2][b & 3]);
#pragma unroll
            for (int ti = 0; ti < TPW; ti++) {
                const int tt = ti * NW + wv;
                if (tt >= ntiles) continue;
                float mxa = acc[ti][0][0];
#pragma unroll
                for (int b = 1; b + 1 < NTL * 4; b += 2) mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, acc[ti][b >> 2][b & 3]), acc[ti][(b + 1) >> 2][(b + 1) & 3]);
                mxa = __builtin_fmaxf(mxa, acc[ti][NTL - 1][3]);
                if (__builtin_amdgcn_ballot_w64(mxa >= thrmin)) {
                    const long long pos = c0 + (long long)po + (long long)tt * 16 + n;
                    mfk2_tile_survivors<NTL, MFK2_CAP>(P, ck, acc[ti], thr, pos, c1, kd, s_row, s_buf, s_misc + 1, bufn, first, g, lane);
                }
            }
            if (!more) break;
        }
        if (bufn) mf_flush(P, ck, s_buf, bufn, s_row, s_misc + 1, first, lane);
        __syncthreads();
        // ---- (d) thresholds from the union of the survivors' upper bounds (as K3m's phase (c)) ----
        {
            const u32 tall = s_misc[1];
            for (int qs = wv; qs < MFK_G; qs += NW) {
                if (!((tall >> qs) & 1u)) continue;
                const int q = s_row[qs].q;
                if (!(s_row[qs].kq != 0.f)) continue;
                const double inv0q = 256.0 / keyd(P.T0[q]);
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
                        atomicMin(P.S.T + q, dkey((double)(b + 1) / inv0q * (1.0 + 1e-12)));
                    }
                }
            }
        }
    }
    for (u32 i = ck.used + (u32)lane; i < ck.cap; i += 64)
        if (ck.base + i < P.surv_cap) P.surv[ck.base + i] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
}
