// mmidx_frontend.h -- kernels of the two steps that feed the index (BASELINE config 5):
//   K7  batched PCA projection  (J/dimreduction/PCA.java:188-208)   -> f64 MFMA GEMM
//   K8  batched VLAD aggregation (J/aggregation/VladAggregator.java:56-70,
//       AbstractFeatureAggregator.java:136-155, VladAggregatorMultipleVocabularies.java:84-101)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) double f64x4;

// ------------------------------------------------------------------------------------------------
// K7: Y[n][nc] = (X[n][ss] - mu[ss]) * Vt[nc][ss]^T          (sampleToEigenSpace PCA.java:196-201)
//
// The one dense contraction of the path: v_mfma_f64_16x16x4_f64.  Block = 4 waves, tile 64 rows x
// 128 components x PCA_BK k; wave w owns rows 16w..16w+15 and all 8 column tiles (8 accumulators of 4
// f64 per lane).  The mean is subtracted while the X tile is staged (sample - means, :199).  Tiles
// go through LDS with a padded row stride (PCA_LD doubles), which spreads the fragment reads
// (row = lane & 15, k = lane >> 4) over the bank pairs; the next tile's global loads are in
// flight while the current one is multiplied, and four waves per SIMD (blocks of the same CU at
// different points of their k loops) keep the matrix cores fed across the barriers.  A/B operand: one f64 per lane,
// A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]; C/D: col = lane & 15,
// row = (lane >> 4) + 4 * reg (the f64 layout, which differs from the f32 maps).
// EJML's matrix-vector product accumulates each output sequentially with separate multiply and add
// (assumption A2); the MFMA chain is fused and k-blocked, so parity for this kernel is a
// tolerance (1e-12 relative to the row norm), never bit-equality.
// ------------------------------------------------------------------------------------------------
#ifndef PCA_BM
#define PCA_BM 64   // rows per block: 16 per wave
#endif
#define PCA_NT (PCA_BM * 4)  // threads per block
#define PCA_BN 128
#ifndef PCA_BK
#define PCA_BK 8    // k per staged tile.  Round 3, same box: at TWO waves per SIMD (196 registers: 64 accumulators + the staging of a
#endif              // wide tile) 16 / 24 / 32 gave 37 / 38.8 / 40.3 TF of the 67.9 the f64 matrix cores deliver back to back; with the
                    // register budget held to FOUR waves per SIMD (PCA_WPS) a narrow tile needs no spill and 8 gives 47.1-47.9 TF
                    // (0.70), 16 at three waves 44.9, 8 at three 45.3, 8 at five (spills) 37.5: the barriers between short tiles
                    // are hidden by the other blocks of the CU, which the wide tile's registers did not leave room for
#ifndef PCA_PAD
#define PCA_PAD 1
#endif
#define PCA_LD (PCA_BK + PCA_PAD)  // padded row stride in doubles
#define PCA_PPR (PCA_BK / 2)  // pairs of doubles per tile row
#define PCA_NA (PCA_BM * PCA_PPR / PCA_NT)  // pairs per thread: X tile
#define PCA_NB (PCA_BN * PCA_PPR / PCA_NT)  // ... Vt tile
#define PCA_LDS_BYTES ((PCA_BM + PCA_BN) * PCA_LD * 8)

#ifndef PCA_WPS
#define PCA_WPS 4  // waves per SIMD the register budget is held to (128 VGPRs: the accumulators + one narrow tile in flight)
#endif
__global__ __launch_bounds__(PCA_NT, PCA_WPS) void k_pca_project(const double *__restrict__ X, const double *__restrict__ mu,
                                                        const double *__restrict__ Vt, double *__restrict__ Y,
                                                        long long n, int nc, int ss) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pca_smem[];
    double *As = (double *)pca_smem;         // [PCA_BM][PCA_LD]
    double *Bs = As + PCA_BM * PCA_LD;       // [PCA_BN][PCA_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = (long long)blockIdx.x * PCA_BM;
    const int col0 = blockIdx.y * PCA_BN;
    f64x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    // staging: A PCA_BM rows x PCA_BK k in pairs of doubles, PCA_NA per thread; B 128 x PCA_BK, PCA_NB per thread
    double2 ra[PCA_NA], rb[PCA_NB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int u = 0; u < PCA_NA; u++) {
            const int p = tid + u * PCA_NT, r = p / PCA_PPR, c = (p % PCA_PPR) * 2;
            const long long gr = row0 + r;
            double2 v = make_double2(0.0, 0.0);
            if (gr < n) {
                const double *src = X + (size_t)gr * ss + k0 + c;
                if (k0 + c + 1 < ss) {
                    v.x = src[0] - mu[k0 + c];
                    v.y = src[1] - mu[k0 + c + 1];
                } else if (k0 + c < ss) {
                    v.x = src[0] - mu[k0 + c];
                }
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < PCA_NB; u++) {
            const int p = tid + u * PCA_NT, r = p / PCA_PPR, c = (p % PCA_PPR) * 2;
            const int gc = col0 + r;
            double2 v = make_double2(0.0, 0.0);
            if (gc < nc) {
                const double *src = Vt + (size_t)gc * ss + k0 + c;
                if (k0 + c + 1 < ss) {
                    v.x = src[0];
                    v.y = src[1];
                } else if (k0 + c < ss) {
                    v.x = src[0];
                }
            }
            rb[u] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int u = 0; u < PCA_NA; u++) {
            const int p = tid + u * PCA_NT, r = p / PCA_PPR, c = (p % PCA_PPR) * 2;
            if constexpr ((PCA_LD & 1) == 0) {
                *(double2 *)(As + r * PCA_LD + c) = ra[u];
            } else {
                As[r * PCA_LD + c] = ra[u].x;
                As[r * PCA_LD + c + 1] = ra[u].y;
            }
        }
#pragma unroll
        for (int u = 0; u < PCA_NB; u++) {
            const int p = tid + u * PCA_NT, r = p / PCA_PPR, c = (p % PCA_PPR) * 2;
            if constexpr ((PCA_LD & 1) == 0) {
                *(double2 *)(Bs + r * PCA_LD + c) = rb[u];
            } else {
                Bs[r * PCA_LD + c] = rb[u].x;
                Bs[r * PCA_LD + c + 1] = rb[u].y;
            }
        }
    };
    load_tiles(0);
    for (int k0 = 0; k0 < ss; k0 += PCA_BK) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (k0 + PCA_BK < ss) load_tiles(k0 + PCA_BK);
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < PCA_BK / 4; kk++) {
            const double a = As[(wave * 16 + fr) * PCA_LD + kk * 4 + fk];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const double b = Bs[(t * 16 + fr) * PCA_LD + kk * 4 + fk];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D (f64): col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int gc = col0 + t * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long gr = row0 + wave * 16 + (lane >> 4) + 4 * r;
            if (gr < n && gc < nc) Y[(size_t)gr * nc + gc] = acc[t][r];
        }
    }
}

// whitening: y <- y / ||y||_2, zero vector -> all ones (Normalization.normalizeL2, Normalization.java:21-37).
// One thread per row, sequential sum in index order: the same arithmetic as the reference given y.
__global__ void k_rows_normalize_l2(double *__restrict__ Y, long long n, int nc) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double *y = Y + (size_t)r * nc;
    double norm2 = 0.0;
    for (int i = 0; i < nc; i++) norm2 += y[i] * y[i];
    norm2 = sqrt(norm2);
    if (norm2 == 0.0) {
        for (int i = 0; i < nc; i++) y[i] = 1.0;
    } else {
        for (int i = 0; i < nc; i++) y[i] = y[i] / norm2;
    }
}

// ------------------------------------------------------------------------------------------------
// K8: VLAD.  One block per image, one vocabulary per call segment.
//   phase 1  thread <-> descriptor: nearest centroid, first minimum wins, sequential j
//            (computeNearestCentroid AFA:136-155; the early break cannot change the argmin).
//            The codebook sits in LDS ([c][dl]); every lane reads the same address (broadcast).
//   phase 2  per centroid, the ordered list of its descriptors (stable: descriptor order).
//   phase 3  thread <-> (centroid, dim): vlad[c*dl+i] += desc[i] - cb[c][i] in DESCRIPTOR ORDER
//            (VladAggregator.java:63-68) -- the accumulation order is the reference's, so the raw
//            VLAD vector is bit-exact.
//   phase 4  optional power (signed sqrt) + L2 normalisation of the sub-vector
//            (VladAggregatorMultipleVocabularies.java:90-91): the squared norm is a block tree
//            reduction, so normalised outputs carry a 1e-12 tolerance (Math.pow itself is only
//            specified to 1 ulp).
// ------------------------------------------------------------------------------------------------
template <int DL>
__global__ __launch_bounds__(256) void k_vlad(const double *__restrict__ codebook, int nc, int dl_rt, int maxnd,
                                              const long long *__restrict__ desc_off, const double *__restrict__ descs,
                                              double *__restrict__ out, int out_stride, int out_shift, int norms_on) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dl = DL > 0 ? DL : dl_rt;
    double *cb = (double *)smem;                 // [nc][dl]
    int *nn = (int *)(cb + (size_t)nc * dl);     // [maxnd] (maxnd even)
    int *lst = nn + maxnd;                       // [maxnd] descriptors grouped by centroid
    int *cstart = lst + maxnd;                   // [nc + 1]
    double *red = (double *)(cstart + ((nc + 2) & ~1));  // [4]
    const int img = blockIdx.x, tid = threadIdx.x;
    const long long d0 = desc_off[img];
    const int nd = (int)(desc_off[img + 1] - d0);
    const double *D = descs + (size_t)d0 * dl;
    double *vout = out + (size_t)img * out_stride + out_shift;
    const int veclen = nc * dl;
    for (int i = tid; i < veclen; i += 256) cb[i] = codebook[i];
    __syncthreads();
    // phase 1 (the descriptor stays in registers when its length is a template constant)
    for (int d = tid; d < nd; d += 256) {
        const double *x = D + (size_t)d * dl;
        int best = -1;
        double mind = 1.7976931348623157e308;
        if constexpr (DL > 0) {
            double xr[DL];
#pragma unroll
            for (int j = 0; j < DL; j++) xr[j] = x[j];
            for (int c = 0; c < nc; c++) {
                const double *cc = cb + (size_t)c * DL;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < DL; j++) {
                    const double df = cc[j] - xr[j];
                    acc += df * df;
                }
                if (acc < mind) {
                    mind = acc;
                    best = c;
                }
            }
        } else {
            for (int c = 0; c < nc; c++) {
                const double *cc = cb + (size_t)c * dl;
                double acc = 0.0;
                for (int j = 0; j < dl; j++) {
                    const double df = cc[j] - x[j];
                    acc += df * df;
                }
                if (acc < mind) {
                    mind = acc;
                    best = c;
                }
            }
        }
        nn[d] = best < 0 ? 0 : best;
    }
    __syncthreads();
    // phase 2: counts -> starts -> stable fill
    for (int c = tid; c < nc; c += 256) {
        int cnt = 0;
        for (int d = 0; d < nd; d++) cnt += (nn[d] == c);
        cstart[c + 1] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        cstart[0] = 0;
        for (int c = 0; c < nc; c++) cstart[c + 1] += cstart[c];
    }
    __syncthreads();
    for (int c = tid; c < nc; c += 256) {
        int p = cstart[c];
        for (int d = 0; d < nd; d++)
            if (nn[d] == c) lst[p++] = d;
    }
    __syncthreads();
    // phase 3
    double ss = 0.0;
    for (int e = tid; e < veclen; e += 256) {
        const int c = e / dl, i = e - c * dl;
        const double cv = cb[e];
        double v = 0.0;
        for (int p = cstart[c]; p < cstart[c + 1]; p++) v += D[(size_t)lst[p] * dl + i] - cv;
        if (norms_on) {
            // normalizePower(0.5): signum(v) * pow(|v|, 0.5)   (Normalization.java:74-79)
            const double a = sqrt(fabs(v));
            v = (v > 0.0) ? a : ((v < 0.0) ? -a : v);
            ss += v * v;
        }
        vout[e] = v;
    }
    if (!norms_on) return;
    // phase 4: L2 over the sub-vector (zero norm -> ones)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const double norm = sqrt(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    for (int e = tid; e < veclen; e += 256) vout[e] = (norm == 0.0) ? 1.0 : vout[e] / norm;
}

// K8': VLAD with the nearest centroids already known (nn[descriptor], from the encoder's certified bf16-MFMA assignment over ALL
// descriptors of the launch: k_split_bf16 + k_assign_gmin16 + exact redo of the flagged few, first index wins as AFA:136-155).
// One block per image: phases 2-4 of k_vlad -- the stable per-centroid descriptor lists, the accumulation in DESCRIPTOR ORDER
// (VladAggregator.java:63-68: the raw VLAD vector is bit-exact), power + L2.  The codebook is read from global memory once per
// element (coalesced), so the block's LDS is three small integer arrays and a CU holds many images.
template <int DL>
__global__ __launch_bounds__(256) void k_vlad_accum(const double *__restrict__ codebook, int nc, int dl_rt, int maxnd, const int32_t *__restrict__ nn_g,
                                                    const long long *__restrict__ desc_off, const double *__restrict__ descs,
                                                    double *__restrict__ out, int out_stride, int out_shift, int norms_on) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dl = DL > 0 ? DL : dl_rt;
    int *nn = (int *)smem;                       // [maxnd]
    int *lst = nn + maxnd;                       // [maxnd] descriptors grouped by centroid
    int *cstart = lst + maxnd;                   // [nc + 1]
    double *red = (double *)(cstart + ((nc + 2) & ~1));  // [4]
    const int img = blockIdx.x, tid = threadIdx.x;
    const long long d0 = desc_off[img];
    const int nd = (int)(desc_off[img + 1] - d0);
    const double *D = descs + (size_t)d0 * dl;
    double *vout = out + (size_t)img * out_stride + out_shift;
    const int veclen = nc * dl;
    for (int d = tid; d < nd; d += 256) {
        const int c = nn_g[d0 + d];
        nn[d] = c < 0 ? 0 : (c < nc ? c : nc - 1);
    }
    for (int c = tid; c <= nc; c += 256) cstart[c] = 0;
    __syncthreads();
    // phase 2: counts (LDS atomics: order does not matter) -> starts -> stable fill (a thread per centroid walks the image's
    // assignments in descriptor order: broadcast LDS reads)
    for (int d = tid; d < nd; d += 256) atomicAdd(cstart + nn[d] + 1, 1);
    __syncthreads();
    if (tid < 64) {  // starts: a wave-wide scan of the counts (64 centroids per step), not 128 dependent LDS updates by one thread
        u32 carry = 0;
        for (int base = 0; base < nc; base += 64) {
            const int c = base + tid;
            const u32 cnt = c < nc ? (u32)cstart[c + 1] : 0u;
            const u32 incl = wave_incl_scan_u32(cnt);
            if (c < nc) cstart[c + 1] = (int)(carry + incl);
            carry += wave_read_u32(incl, 63);
        }
        if (tid == 0) cstart[0] = 0;
    }
    __syncthreads();
    for (int c = tid; c < nc; c += 256) {
        int p = cstart[c];
        for (int d = 0; d < nd; d++)
            if (nn[d] == c) lst[p++] = d;
    }
    __syncthreads();
    // phase 3: thread <-> (centroid, dim): a wave reads whole descriptor rows (coalesced), in descriptor order
    double ss = 0.0;
    if constexpr (DL == 64) {
        // 64-dimensional descriptors: a lane per dimension, a wave per quarter of the centroids.  The rows of a wave's centroids are
        // one contiguous stretch of the grouped list: eight rows are requested together ACROSS centroid boundaries (an image has ~4
        // descriptors per centroid: one centroid at a time left one load in flight per thread and 170 us per image block), then added
        // in list order -- per element the same additions in the same order as before; the next centroid's coordinate is fetched
        // while the current one's rows are added.
        const int wv = tid >> 6, lane = tid & 63;
        const int cpw = (nc + 3) >> 2;
        const int c_lo = wv * cpw < nc ? wv * cpw : nc, c_hi = c_lo + cpw < nc ? c_lo + cpw : nc;
        if (c_lo < c_hi) {
            int c = c_lo, p = cstart[c_lo];
            const int pe = cstart[c_hi];
            int cend = cstart[c + 1];
            double v = 0.0, cv = codebook[(size_t)c * 64 + lane];
            double cvn = c + 1 < c_hi ? codebook[(size_t)(c + 1) * 64 + lane] : 0.0;
            auto emit = [&]() {  // element (c, lane) is complete: power normalisation, next centroid
                if (norms_on) {
                    const double a = sqrt(fabs(v));  // normalizePower(0.5): signum(v) * pow(|v|, 0.5)   (Normalization.java:74-79)
                    v = (v > 0.0) ? a : ((v < 0.0) ? -a : v);
                    ss += v * v;
                }
                vout[(size_t)c * 64 + lane] = v;
                c++;
                v = 0.0;
                cv = cvn;
                if (c < c_hi) cend = cstart[c + 1];
                if (c + 1 < c_hi) cvn = codebook[(size_t)(c + 1) * 64 + lane];
            };
            while (p < pe) {
                constexpr int RU = 8;  // rows in flight per wave (16: more registers, fewer blocks per CU, measured slower)
                const int n8 = pe - p < RU ? pe - p : RU;
                double a[RU];
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const int pu = u < n8 ? p + u : pe - 1;  // (loads run on a clamped index)
                    a[u] = D[(size_t)lst[pu] * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    if (u < n8) {
                        while (p + u >= cend) emit();  // (wave-uniform; centroids without descriptors come out as zeros)
                        v += a[u] - cv;
                    }
                }
                p += n8;
            }
            while (c < c_hi) emit();
        }
    } else
    for (int e = tid; e < veclen; e += 256) {
        const int c = e / dl, i = e - c * dl;
        const double cv = codebook[e];
        double v = 0.0;
        // (four descriptor rows requested together, added in descriptor order: the same additions, more loads in flight)
        int p = cstart[c];
        const int pe = cstart[c + 1];
        for (; p + 4 <= pe; p += 4) {
            const double a0 = D[(size_t)lst[p] * dl + i], a1 = D[(size_t)lst[p + 1] * dl + i], a2 = D[(size_t)lst[p + 2] * dl + i],
                         a3 = D[(size_t)lst[p + 3] * dl + i];
            v += a0 - cv;
            v += a1 - cv;
            v += a2 - cv;
            v += a3 - cv;
        }
        for (; p < pe; p++) v += D[(size_t)lst[p] * dl + i] - cv;
        if (norms_on) {
            const double a = sqrt(fabs(v));  // normalizePower(0.5): signum(v) * pow(|v|, 0.5)   (Normalization.java:74-79)
            v = (v > 0.0) ? a : ((v < 0.0) ? -a : v);
            ss += v * v;
        }
        vout[e] = v;
    }
    if (!norms_on) return;
    // phase 4: L2 over the sub-vector (zero norm -> ones)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const double norm = sqrt(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    for (int e = tid; e < veclen; e += 256) vout[e] = (norm == 0.0) ? 1.0 : vout[e] / norm;
}

// K8'' (round 5): VLAD of an image in ONE kernel and one pass over its descriptors in HBM (VladAggregator.java:56-70).  K8' streamed
// every descriptor twice -- the assignment kernel over all descriptors of the call, then a block per image for the ordered
// accumulation -- with a host synchronisation between them (the number of flagged descriptors).  Here a block takes an image:
//   (1) the vocabulary's bf16 head / tail tiles go to LDS once (<= 128 centroids: one tile of K6a');
//   (2) 128 descriptors at a time: fp64 rows split to bf16 head / tail in registers, d~ = |c|^2 + |x|^2 - 2 x.c on
//       v_mfma_f32_16x16x32_bf16 (three products), best / second best per row, certified as in k_assign_gmin16_t
//       (second - best > 2 eps16); the descriptors the bound cannot certify (exact ties included, ~1e-4) are listed in LDS;
//   (3) those are redone by the block itself in fp64, every sum in dimension order (computeNearestCentroid, AFA:136-155; strict
//       '<': the first minimum wins) -- identical to the exact argmin, no host round trip;
//   (4) phases 2-4 of k_vlad_accum: stable per-centroid lists, accumulation in DESCRIPTOR ORDER (the raw VLAD vector is bit-exact),
//       power + L2.  The rows are read a second time here -- from L2 / the Infinity Cache: the block touched them microseconds ago
//       (all resident blocks together hold ~130 MB of descriptors).  The row loads are double-buffered (eight rows requested while
//       the previous eight are added).
// 64-dimensional descriptors (SURF), vocabularies of <= 128 centroids; other shapes keep K8'.
#define VF_FLAG_CAP 64
__global__ __launch_bounds__(256, 3) void k_vlad_fused(const double *__restrict__ codebook, const double *__restrict__ codebookT, int nc, int maxnd, const __bf16 *__restrict__ Ch,
                                                       const __bf16 *__restrict__ Cl, const double *__restrict__ cn, double cnorm_max, double cn_max, int Dp,
                                                       const long long *__restrict__ desc_off, const double *__restrict__ descs, double *__restrict__ out,
                                                       int out_stride, int out_shift, int norms_on) {
    constexpr int DL = 64;
    constexpr int BSTR = DL * 2 + 16;  // bytes per LDS row of the tiles: 64 bf16 + 16 (conflict-free 16-byte fragment reads); 36 KiB for both
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Bh = smem, *Bl = smem + G16_BC * BSTR;
    int *nn = (int *)(smem + 2 * G16_BC * BSTR);  // [maxnd]
    int *lst = nn + maxnd;                              // [maxnd] descriptors grouped by centroid
    int *cstart = lst + maxnd;                          // [nc + 1]
    int *flg = cstart + ((nc + 2) & ~1);                // [0] count, [1 ..] flagged descriptors
    double *red = (double *)(flg + VF_FLAG_CAP + 2);    // [4]
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const long long d0 = desc_off[img];
    const int nd = (int)(desc_off[img + 1] - d0);
    const double *Dg = descs + (size_t)d0 * DL;
    double *vout = out + (size_t)img * out_stride + out_shift;
    // ---- (1) the vocabulary's tiles (rows = centroids, DL / 8 units of 16 bytes each), counters ----
    {
        const int upr = DL >> 3;  // (Dp >= 64: the first 64 columns of a row)
        for (int u = tid; u < G16_BC * upr; u += 256) {
            const int row = u / upr, cu = u - row * upr;
            const size_t src = (size_t)row * Dp + cu * 8;
            *(uint4 *)(Bh + row * BSTR + cu * 16) = *(const uint4 *)(Ch + src);
            *(uint4 *)(Bl + row * BSTR + cu * 16) = *(const uint4 *)(Cl + src);
        }
        for (int c = tid; c <= nc; c += 256) cstart[c] = 0;
        if (tid == 0) flg[0] = 0;
    }
    __syncthreads();
    float cn_c[8];
#pragma unroll
    for (int ct = 0; ct < 8; ct++) cn_c[ct] = (float)cn[ct * 16 + fr];
    const float inf = __int_as_float(0x7f800000);
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
    // ---- (2) certified assignment, 128 descriptors (32 per wave) at a time ----
#ifdef VF_SKIP_ASSIGN
    for (int d = tid; d < nd; d += 256) nn[d] = d & 127;
#else
    for (int s0 = 0; s0 < nd; s0 += G16_BQ) {
        const int q0 = s0 + wave * 32;
        if (q0 >= nd) continue;  // (wave-uniform; no barrier in this loop)
        bf16x8 ah[2][2], al[2][2];
        float xn_r[2][4];
        double xrow[2];
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            int q = q0 + rt * 16 + fr;
            q = q < nd ? q : nd - 1;
            double pn = 0.0;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const double2 *xp = (const double2 *)(Dg + (size_t)q * DL + ks * 32 + fg * 8);
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
            }
            pn += __shfl_xor(pn, 16);
            pn += __shfl_xor(pn, 32);
            xrow[rt] = pn * (1.0 + 1e-12);  // only ever used inside error bounds: rounded up
#pragma unroll
            for (int r = 0; r < 4; r++) xn_r[rt][r] = (float)__shfl(xrow[rt], 4 * fg + r);
        }
        f32x4 acc[2][8];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int ct = 0; ct < 8; ct += 2) {
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int off = ((ct + u) * 16 + fr) * BSTR + (ks * 32 + fg * 8) * 2;
                    bh[u] = *(const bf16x8 *)(Bh + off);
                    bl[u] = *(const bf16x8 *)(Bl + off);
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++) {
                        acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bh[u], acc[rt][ct + u], 0, 0, 0);
                        acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt][ks], bl[u], acc[rt][ct + u], 0, 0, 0);
                        acc[rt][ct + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][ks], bh[u], acc[rt][ct + u], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float m1 = inf, m2 = inf;
                int ix = 0;
#pragma unroll
                for (int ct = 0; ct < 8; ct++) {
                    const float dv = (cn_c[ct] + xn_r[rt][r]) - 2.0f * acc[rt][ct][r];
                    const bool lt1 = dv < m1;
                    m2 = lt1 ? m1 : (dv < m2 ? dv : m2);
                    ix = lt1 ? ct * 16 + fr : ix;
                    m1 = lt1 ? dv : m1;
                }
#pragma unroll
                for (int step = 0; step < 4; step++) {  // the 16 lanes of a row: (best, its index, second best)
                    const float o1 = dppf(m1, step), o2 = dppf(m2, step);
                    const int oi = dppi(ix, step);
                    const bool take = o1 < m1 || (o1 == m1 && oi < ix);
                    const float hi1 = take ? m1 : o1;
                    const float lo2 = o2 < m2 ? o2 : m2;
                    m2 = hi1 < lo2 ? hi1 : lo2;
                    ix = take ? oi : ix;
                    m1 = take ? o1 : m1;
                }
                const int q = q0 + rt * 16 + 4 * fg + r;
                const double xnd = __shfl(xrow[rt], 4 * fg + r);
                if (fr == 0 && q < nd) {
                    const double xnorm = sqrt(xnd), sumn = cnorm_max + xnorm;
                    const double eps = (2.0 * 3.1 * 0x1p-16 * xnorm * cnorm_max + 2.0 * (3.0 * (double)Dp + 16.0) * 0x1p-22 * xnorm * cnorm_max +
                                        1e-12 * (cn_max + xnd) + 0x1p-21 * sumn * sumn) * (1.0 + 1e-9);
                    const bool sure = ((double)m2 - (double)m1) > 2.0 * eps && sumn * sumn < 1e37;
                    nn[q] = ix < nc ? ix : nc - 1;
                    if (!sure) {
                        const int f = atomicAdd(flg, 1);
                        if (f < VF_FLAG_CAP) flg[1 + f] = q;
                    }
                }
            }
    }
#endif
#ifdef VF_SKIP_ACCUM
    return;
#endif
    __syncthreads();
    // ---- (3) the flagged descriptors in fp64 (normally none or one).  More than the list holds: every descriptor is redone. ----
    {
        const int nf_raw = flg[0];
        const bool all = nf_raw > VF_FLAG_CAP;
        const int nf = all ? nd : nf_raw;
        for (int f = wave; f < nf; f += 4) {
            const int q = all ? f : flg[1 + f];
            const double *x = Dg + (size_t)q * DL;  // (wave-uniform: scalar loads)
            // a lane a centroid (two of them), the TRANSPOSED codebook: a dimension of all centroids is one coalesced load, eight in
            // flight (row-major it was a 512-byte stride between lanes and a round trip per dimension: 40 k cycles per flagged descriptor,
            // with the whole block waiting behind the barrier)
            const int c0 = lane < nc ? lane : nc - 1, c1 = lane + 64 < nc ? lane + 64 : nc - 1;
            double a0 = 0.0, a1 = 0.0;
            for (int j = 0; j < DL; j += 8) {
                double t0v[8], t1v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    t0v[u] = codebookT[(size_t)(j + u) * nc + c0];
                    t1v[u] = codebookT[(size_t)(j + u) * nc + c1];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const double xv = x[j + u];
                    const double e0 = t0v[u] - xv, e1 = t1v[u] - xv;
                    a0 += e0 * e0;
                    a1 += e1 * e1;
                }
            }
            u64 bk = lane < nc ? (u64)__double_as_longlong(a0) : 0xFFFFFFFFFFFFFFFFull;
            int bi = lane < nc ? lane : 0x7fffffff;
            const u64 k1 = (u64)__double_as_longlong(a1);
            if (lane + 64 < nc && k1 < bk) {
                bk = k1;
                bi = lane + 64;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const u64 ok = __shfl_xor(bk, off);
                const int oi = __shfl_xor(bi, off);
                if (ok < bk || (ok == bk && oi < bi)) {
                    bk = ok;
                    bi = oi;
                }
            }
            if (lane == 0) nn[q] = bi < nc ? bi : 0;
        }
    }
    __syncthreads();
    // ---- (4) ordered accumulation, power + L2.  A wave owns the centroids c = wave (mod 4) -- interleaved: popular centroids spread
    //      over the waves -- and keeps their 64-element sums in REGISTERS (a lane a dimension); it lists its own descriptors in
    //      descriptor order (two ballot passes over nn[]) and walks that list: the sum of a centroid receives its descriptors in
    //      ascending order whatever the other centroids do in between (vlad[nn*dl+j] += desc[j] - centroid[nn][j],
    //      VladAggregator.java:63-67) -- no per-centroid lists (a 500-step LDS chain per thread), no centroid boundaries in the loop,
    //      and the vector is written ONCE, normalised (it had been written raw, read back and rewritten).
    constexpr int CPW = 32;  // centroids per wave (nc <= 128): centroid 4 k + wave is the wave's k-th
    {
        int mine = 0;
        for (int d0 = 0; d0 < nd; d0 += 64) {
            const int d = d0 + lane;
            mine += (int)__popcll(__builtin_amdgcn_ballot_w64(d < nd && (nn[d < nd ? d : nd - 1] & 3) == wave));
        }
        if (lane == 0) cstart[wave] = mine;
    }
    __syncthreads();
    int p = 0;
    for (int w2 = 0; w2 < wave; w2++) p += cstart[w2];
    const int pe = p + cstart[wave];
    {
        int run = p;
        const u64 lane_lt = (1ull << lane) - 1ull;
        for (int d0 = 0; d0 < nd; d0 += 64) {
            const int d = d0 + lane;
            const int c = nn[d < nd ? d : nd - 1];
            const bool own = d < nd && (c & 3) == wave;
            const u64 mk = __builtin_amdgcn_ballot_w64(own);
            if (own) lst[run + (int)__popcll(mk & lane_lt)] = d | ((c >> 2) << 24);  // (nd < 2^24)
            run += (int)__popcll(mk);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (a wave reads its own segment of lst[] only)
    double acc[CPW];  // (the centroids' own rows would be another 64 registers: they are requested with the descriptor rows -- L1 / L2 hits)
#pragma unroll
    for (int c = 0; c < CPW; c++) acc[c] = 0.0;
    const double *cbw = codebook + (size_t)wave * 64 + lane;  // centroid 4 k + wave, element lane: cbw[256 k]
    {
        constexpr int RU = 8;
        double ra[RU], rb[RU], ca[RU], cb[RU];
        int ea[RU], eb[RU];
        auto fetch = [&](double (&dst)[RU], double (&cen)[RU], int (&ent)[RU], const int pp) {  // rows of entries pp .. pp + 7 (clamped: always valid addresses)
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const int pu = pp + u < pe ? pp + u : pe - 1;
                ent[u] = __builtin_amdgcn_readfirstlane(lst[pu < 0 ? 0 : pu]);
                dst[u] = Dg[(size_t)(ent[u] & 0xFFFFFF) * 64 + lane];
                cen[u] = cbw[(size_t)(ent[u] >> 24) * 256];
            }
        };
        auto add8 = [&](const double (&src)[RU], const double (&cen)[RU], const int (&ent)[RU], const int pp) {
            const int n8 = pe - pp < RU ? pe - pp : RU;
#pragma unroll
            for (int u = 0; u < RU; u++) {
                if (u < n8) {  // (wave-uniform)
                    switch (ent[u] >> 24) {  // (wave-uniform: the entry came through readfirstlane)
#define VF_CASE(k) case k: acc[k] += src[u] - cen[u]; break;
                        VF_CASE(0) VF_CASE(1) VF_CASE(2) VF_CASE(3) VF_CASE(4) VF_CASE(5) VF_CASE(6) VF_CASE(7)
                        VF_CASE(8) VF_CASE(9) VF_CASE(10) VF_CASE(11) VF_CASE(12) VF_CASE(13) VF_CASE(14) VF_CASE(15)
                        VF_CASE(16) VF_CASE(17) VF_CASE(18) VF_CASE(19) VF_CASE(20) VF_CASE(21) VF_CASE(22) VF_CASE(23)
                        VF_CASE(24) VF_CASE(25) VF_CASE(26) VF_CASE(27) VF_CASE(28) VF_CASE(29) VF_CASE(30) VF_CASE(31)
#undef VF_CASE
                        default: break;
                    }
                }
            }
        };
        if (p < pe) fetch(ra, ca, ea, p);
        while (p < pe) {  // two batches per round: one in flight while the other is added
            if (p + RU < pe) fetch(rb, cb, eb, p + RU);
            add8(ra, ca, ea, p);
            p += RU;
            if (p >= pe) break;
            if (p + RU < pe) fetch(ra, ca, ea, p + RU);
            add8(rb, cb, eb, p);
            p += RU;
        }
    }
    double ss = 0.0;
    if (norms_on) {
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            const double a = sqrt(fabs(acc[c]));  // normalizePower(0.5): signum(v) * pow(|v|, 0.5)   (Normalization.java:74-79)
            acc[c] = (acc[c] > 0.0) ? a : ((acc[c] < 0.0) ? -a : acc[c]);
            if (4 * c + wave < nc) ss += acc[c] * acc[c];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
    }
    const double norm = norms_on ? sqrt(red[0] + red[1] + red[2] + red[3]) : 1.0;
#pragma unroll
    for (int c = 0; c < CPW; c++) {
        const int cg = 4 * c + wave;
        if (cg < nc) vout[(size_t)cg * 64 + lane] = !norms_on ? acc[c] : ((norm == 0.0) ? 1.0 : acc[c] / norm);
    }
}

// L2 over the concatenation when more than one vocabulary (VladAggregatorMultipleVocabularies.java:97-99)
__global__ __launch_bounds__(256) void k_rows_normalize_l2_block(double *__restrict__ Y, int len) {
    __shared__ double red[4];
    double *y = Y + (size_t)blockIdx.x * len;
    double ss = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) ss += y[i] * y[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const double norm = sqrt(red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < len; i += 256) y[i] = (norm == 0.0) ? 1.0 : y[i] / norm;
}
