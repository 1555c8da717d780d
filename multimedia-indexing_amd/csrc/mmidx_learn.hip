// mmidx_learn.hip -- k-means codebook learning on the GPU (coarse and product quantizers).
//
// The reference learns its quantizers offline with Weka's SimpleKMeans
// (J/visual/quantization/AbstractQuantizerLearning.java:39-81, CoarseQuantizerLearning.java:39-72,
// ProductQuantizationLearning.java:247-305).  Weka 3.7.6 is a third-party dependency that is not in the
// reference tree (pom.xml), so this file restates the published algorithm -- Lloyd iterations with random or
// k-means++ seeding, optional min-max attribute normalisation (the default of Weka's EuclideanDistance), empty
// clusters dropped, stop when no assignment changes or at maxIterations -- and NOT Weka's exact random stream or
// floating-point order: learned codebooks are equivalent in kind, not bit-identical ("parity unpinned" for this
// row; tests pin the GPU path against a numpy restatement of the same algorithm from identical seeds instead).
//
// Arithmetic is deterministic: the assignment is the library's exact fp64 argmin (first index wins ties), and a
// centroid is the sum of its members in ascending index order divided by their number (stable radix sort by
// cluster, one block per cluster), so a CPU restatement reproduces the iterations bit for bit.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "mmidx.h"

namespace {

typedef unsigned long long u64;

#define LCK(call)                                                     \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) {                                       \
            fprintf(stderr, "[mmidx] k-means: %s failed: %s\n", #call, hipGetErrorString(e_)); \
            return MMIDX_ERR_HIP;                                     \
        }                                                             \
    } while (0)

// java.util.Random (the generator Weka seeds with setSeed): 48-bit LCG, JDK javadoc
struct JavaRandom {
    u64 s;
    explicit JavaRandom(long long seed) : s(((u64)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int next(int bits) {
        s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int)((long long)s >> (48 - bits));
    }
    int nextInt(int bound) {
        int r = next(31);
        const int m = bound - 1;
        if ((bound & m) == 0) return (int)(((long long)bound * (long long)r) >> 31);
        for (int u = r; u - (r = u % bound) + m < 0; u = next(31)) {
        }
        return r;
    }
    double nextDouble() { return (double)(((long long)next(26) << 27) + next(27)) * 0x1.0p-53; }
};

// per column minimum / maximum: one block per column
__global__ void k_col_minmax(const double *__restrict__ X, long long n, int d, double *__restrict__ mn, double *__restrict__ mx) {
    const int c = blockIdx.x;
    double lo = __longlong_as_double(0x7FF0000000000000ll), hi = -lo;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = X[(size_t)i * d + c];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    __shared__ double s_lo[256], s_hi[256];
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_lo[threadIdx.x] = s_lo[threadIdx.x + off] < s_lo[threadIdx.x] ? s_lo[threadIdx.x + off] : s_lo[threadIdx.x];
            s_hi[threadIdx.x] = s_hi[threadIdx.x + off] > s_hi[threadIdx.x] ? s_hi[threadIdx.x + off] : s_hi[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mn[c] = s_lo[0];
        mx[c] = s_hi[0];
    }
}

// (x - min) / (max - min), 0 for a constant attribute (weka.core.NormalizableDistance.norm)
__global__ void k_normalize(const double *__restrict__ X, double *__restrict__ Y, const double *__restrict__ mn,
                            const double *__restrict__ mx, long long total, int d) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % d);
    const double r = mx[c] - mn[c];
    Y[e] = r > 0.0 ? (X[e] - mn[c]) / r : 0.0;
}

// k-means++: mind2[i] = min(mind2[i], |x_i - c|^2), sequential sum over the dimensions (one thread per row)
__global__ void k_pp_update(const double *__restrict__ X, const double *__restrict__ c, double *__restrict__ mind2, long long n, int d,
                            int first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int j = 0; j < d; j++) {
        const double df = X[(size_t)i * d + j] - c[j];
        acc += df * df;
    }
    mind2[i] = (first || acc < mind2[i]) ? acc : mind2[i];
}

// first index whose inclusive cumulative sum exceeds r (cum is non-decreasing); n - 1 if none
__global__ void k_pp_pick(const double *__restrict__ cum, long long n, double r, long long *__restrict__ out) {
    long long lo = 0, hi = n - 1;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cum[mid] > r) hi = mid;
        else lo = mid + 1;
    }
    *out = lo;
}

__global__ void k_iota(int32_t *__restrict__ v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// number of changed assignments and members per cluster
__global__ void k_changed_counts(const int32_t *__restrict__ a_old, const int32_t *__restrict__ a_new, long long n, u64 *__restrict__ changed,
                                 int32_t *__restrict__ counts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = a_new[i];
    atomicAdd(counts + c, 1);
    if (a_old[i] != c) atomicAdd(changed, 1ULL);
}

// centroid c = (sum of its members in ascending index order) / count; one block per cluster, thread <-> dimension
__global__ __launch_bounds__(256) void k_centroid_mean(const double *__restrict__ X, const int32_t *__restrict__ sorted_idx,
                                                        const long long *__restrict__ off, double *__restrict__ C, int d) {
    const int c = blockIdx.x;
    const long long a = off[c], b = off[c + 1];
    if (b <= a) return;  // empty: the host drops the cluster
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        double acc = 0.0;
        for (long long m = a; m < b; m++) acc += X[(size_t)sorted_idx[m] * d + j];
        C[(size_t)c * d + j] = acc / (double)(b - a);
    }
}

// squared error: |x_i - C[a_i]|^2, sequential over dimensions; per-point values (summed in index order on the host)
__global__ void k_point_sqerr(const double *__restrict__ X, const double *__restrict__ C, const int32_t *__restrict__ a, double *__restrict__ out,
                              long long n, int d) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *c = C + (size_t)a[i] * d;
    double acc = 0.0;
    for (int j = 0; j < d; j++) {
        const double df = X[(size_t)i * d + j] - c[j];
        acc += df * df;
    }
    out[i] = acc;
}

template <typename T>
struct Buf {
    T *p = nullptr;
    ~Buf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

}  // namespace

extern "C" {

int mmidx_kmeans_device(int device, int64_t n, int d, int k, int max_iter, int64_t seed, int flags, const double *dX,
                        const double *init_centroids, double *centroids_out, int32_t *d_assign_out, double *sse_out, int32_t *iters_out,
                        int32_t *k_out, void *stream) {
    if (n < 1 || d < 1 || k < 1 || max_iter < 1 || !dX || !centroids_out) return MMIDX_ERR_INVALID_ARG;
    if ((int64_t)k > n) return MMIDX_ERR_INVALID_ARG;  // Weka would stop seeding at n centroids; a codebook needs k <= n
    if (mmidx_device_count() < 1) return MMIDX_ERR_NO_DEVICE;
    LCK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const bool plus_plus = (flags & MMIDX_KMEANS_PLUS_PLUS) != 0, normalize = (flags & MMIDX_KMEANS_NORMALIZE) != 0;
    const size_t nd = (size_t)n * d;

    // ---- working copy of the data (normalised attributes: Weka's default distance) --------------------
    Buf<double> Xn, mn, mx;
    const double *W = dX;
    std::vector<double> h_mn((size_t)d, 0.0), h_mx((size_t)d, 1.0);
    if (normalize) {
        LCK(Xn.alloc(nd));
        LCK(mn.alloc((size_t)d));
        LCK(mx.alloc((size_t)d));
        hipLaunchKernelGGL(k_col_minmax, dim3((unsigned)d), dim3(256), 0, st, dX, (long long)n, d, mn.p, mx.p);
        hipLaunchKernelGGL(k_normalize, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, dX, Xn.p, mn.p, mx.p, (long long)nd, d);
        LCK(hipGetLastError());
        LCK(hipMemcpyAsync(h_mn.data(), mn.p, (size_t)d * 8, hipMemcpyDeviceToHost, st));
        LCK(hipMemcpyAsync(h_mx.data(), mx.p, (size_t)d * 8, hipMemcpyDeviceToHost, st));
        LCK(hipStreamSynchronize(st));
        W = Xn.p;
    }

    // ---- seeding ----------------------------------------------------------------------------------------
    std::vector<double> Ch((size_t)k * d);
    Buf<double> dC;
    LCK(dC.alloc((size_t)k * d));
    if (init_centroids) {
        for (int c = 0; c < k; c++)
            for (int j = 0; j < d; j++) {
                const double v = init_centroids[(size_t)c * d + j], r = h_mx[(size_t)j] - h_mn[(size_t)j];
                Ch[(size_t)c * d + j] = normalize ? (r > 0.0 ? (v - h_mn[(size_t)j]) / r : 0.0) : v;
            }
    } else {
        JavaRandom rnd(seed);
        std::vector<long long> pick;
        pick.reserve((size_t)k);
        if (!plus_plus) {
            // SimpleKMeans' default seeding: walk j = n-1 .. 0, draw nextInt(j + 1), take that instance, swap it out of range
            std::vector<int32_t> perm((size_t)n);
            for (int64_t i = 0; i < n; i++) perm[(size_t)i] = (int32_t)i;
            for (int64_t j = n - 1; j >= 0 && (int)pick.size() < k; j--) {
                const int r = rnd.nextInt((int)(j + 1));
                pick.push_back(perm[(size_t)r]);
                std::swap(perm[(size_t)j], perm[(size_t)r]);
            }
        } else {
            // k-means++ (Arthur & Vassilvitskii): first centre uniform, the others with probability ~ D^2
            Buf<double> mind2, cum;
            Buf<long long> dpick;
            Buf<unsigned char> tmp;
            LCK(mind2.alloc((size_t)n));
            LCK(cum.alloc((size_t)n));
            LCK(dpick.alloc(1));
            size_t tb = 0;
            LCK(hipcub::DeviceScan::InclusiveSum(nullptr, tb, mind2.p, cum.p, (int)n, st));
            LCK(tmp.alloc(tb));
            pick.push_back(rnd.nextInt((int)n));
            for (int c = 1; c < k; c++) {
                hipLaunchKernelGGL(k_pp_update, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, W + (size_t)pick.back() * d, mind2.p,
                                   (long long)n, d, c == 1 ? 1 : 0);
                LCK(hipcub::DeviceScan::InclusiveSum(tmp.p, tb, mind2.p, cum.p, (int)n, st));
                double total = 0.0;
                LCK(hipMemcpyAsync(&total, cum.p + (n - 1), 8, hipMemcpyDeviceToHost, st));
                LCK(hipStreamSynchronize(st));
                const double prob = rnd.nextDouble();
                long long idx = 0;
                hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(1), 0, st, cum.p, (long long)n, prob * total, dpick.p);
                LCK(hipMemcpyAsync(&idx, dpick.p, 8, hipMemcpyDeviceToHost, st));
                LCK(hipStreamSynchronize(st));
                pick.push_back(idx);
            }
        }
        // gather the picked rows
        for (size_t c = 0; c < pick.size(); c++)
            LCK(hipMemcpyAsync(Ch.data() + c * d, W + (size_t)pick[c] * d, (size_t)d * 8, hipMemcpyDeviceToHost, st));
        LCK(hipStreamSynchronize(st));
    }

    // ---- Lloyd iterations -------------------------------------------------------------------------------
    Buf<int32_t> a_old, a_new, iota, sorted_idx, keys_out, counts;
    Buf<long long> off;
    Buf<u64> changed;
    Buf<unsigned char> stmp;
    LCK(a_old.alloc((size_t)n));
    LCK(a_new.alloc((size_t)n));
    LCK(iota.alloc((size_t)n));
    LCK(sorted_idx.alloc((size_t)n));
    LCK(keys_out.alloc((size_t)n));
    LCK(counts.alloc((size_t)k));
    LCK(off.alloc((size_t)k + 1));
    LCK(changed.alloc(1));
    LCK(hipMemsetAsync(a_old.p, 0xFF, (size_t)n * 4, st));
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, iota.p, (long long)n);
    int kbits = 1;
    while ((1 << kbits) < k) kbits++;
    size_t sb = 0;
    LCK(hipcub::DeviceRadixSort::SortPairs(nullptr, sb, a_new.p, keys_out.p, iota.p, sorted_idx.p, (int)n, 0, kbits, st));
    LCK(stmp.alloc(sb));

    int k_eff = k, iters = 0;
    std::vector<int32_t> h_counts((size_t)k);
    std::vector<long long> h_off((size_t)k + 1);
    mmidx_index *h = nullptr;
    int rc = MMIDX_OK;
    auto sort_and_mean = [&](const double *data, double *d_out) -> int {
        LCK(hipcub::DeviceRadixSort::SortPairs(stmp.p, sb, a_new.p, keys_out.p, iota.p, sorted_idx.p, (int)n, 0, kbits, st));
        hipLaunchKernelGGL(k_centroid_mean, dim3((unsigned)k_eff), dim3(256), 0, st, data, sorted_idx.p, off.p, d_out, d);
        LCK(hipGetLastError());
        return MMIDX_OK;
    };
    for (;;) {
        iters++;
        if (h) mmidx_destroy(h);
        h = nullptr;
        rc = mmidx_create(MMIDX_KIND_IVFPQ, d, 1, 2, k_eff, MMIDX_TR_NONE, nullptr, nullptr, device, &h);
        if (rc) break;
        rc = mmidx_set_coarse(h, Ch.data());
        if (rc) break;
        rc = mmidx_assign_device(h, n, W, a_new.p, st);
        if (rc) break;
        LCK(hipMemsetAsync(changed.p, 0, 8, st));
        LCK(hipMemsetAsync(counts.p, 0, (size_t)k_eff * 4, st));
        hipLaunchKernelGGL(k_changed_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a_old.p, a_new.p, (long long)n, changed.p, counts.p);
        u64 nchanged = 0;
        LCK(hipMemcpyAsync(&nchanged, changed.p, 8, hipMemcpyDeviceToHost, st));
        LCK(hipMemcpyAsync(h_counts.data(), counts.p, (size_t)k_eff * 4, hipMemcpyDeviceToHost, st));
        LCK(hipStreamSynchronize(st));
        h_off[0] = 0;
        for (int c = 0; c < k_eff; c++) h_off[(size_t)c + 1] = h_off[(size_t)c] + h_counts[(size_t)c];
        LCK(hipMemcpyAsync(off.p, h_off.data(), ((size_t)k_eff + 1) * 8, hipMemcpyHostToDevice, st));
        rc = sort_and_mean(W, dC.p);
        if (rc) break;
        LCK(hipMemcpyAsync(Ch.data(), dC.p, (size_t)k_eff * d * 8, hipMemcpyDeviceToHost, st));
        LCK(hipStreamSynchronize(st));
        const bool done = nchanged == 0 || iters >= max_iter;
        // empty clusters are dropped (SimpleKMeans: m_NumClusters -= emptyClusterCount); the survivors keep their order
        int kept = 0;
        for (int c = 0; c < k_eff; c++) {
            if (h_counts[(size_t)c] > 0) {
                if (kept != c) memcpy(Ch.data() + (size_t)kept * d, Ch.data() + (size_t)c * d, (size_t)d * 8);
                kept++;
            }
        }
        const bool dropped = kept != k_eff;
        if (done) {
            if (dropped) {
                // renumber the final assignment to the compacted codebook
                std::vector<int32_t> remap((size_t)k_eff, -1);
                int t = 0;
                for (int c = 0; c < k_eff; c++)
                    if (h_counts[(size_t)c] > 0) remap[(size_t)c] = t++;
                std::vector<int32_t> ha((size_t)n);
                LCK(hipMemcpyAsync(ha.data(), a_new.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
                LCK(hipStreamSynchronize(st));
                for (int64_t i = 0; i < n; i++) ha[(size_t)i] = remap[(size_t)ha[(size_t)i]];
                LCK(hipMemcpyAsync(a_new.p, ha.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
                // offsets of the compacted numbering
                int t2 = 0;
                h_off[0] = 0;
                for (int c = 0; c < k_eff; c++)
                    if (h_counts[(size_t)c] > 0) {
                        h_off[(size_t)t2 + 1] = h_off[(size_t)t2] + h_counts[(size_t)c];
                        t2++;
                    }
                LCK(hipMemcpyAsync(off.p, h_off.data(), ((size_t)kept + 1) * 8, hipMemcpyHostToDevice, st));
            }
            k_eff = kept;
            break;
        }
        k_eff = kept;
        std::swap(a_old.p, a_new.p);
        if (dropped) LCK(hipMemsetAsync(a_old.p, 0xFF, (size_t)n * 4, st));  // numbering changed: everything counts as moved
    }
    if (h) mmidx_destroy(h);
    if (rc) return rc;

    // ---- outputs ------------------------------------------------------------------------------------------
    // squared error in the space the clustering ran in (what SimpleKMeans.getSquaredError reports)
    if (sse_out) {
        Buf<double> perr;
        LCK(perr.alloc((size_t)n));
        LCK(hipMemcpyAsync(dC.p, Ch.data(), (size_t)k_eff * d * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_point_sqerr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, dC.p, a_new.p, perr.p, (long long)n, d);
        std::vector<double> he((size_t)n);
        LCK(hipMemcpyAsync(he.data(), perr.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        LCK(hipStreamSynchronize(st));
        double s = 0.0;
        for (int64_t i = 0; i < n; i++) s += he[(size_t)i];
        *sse_out = s;
    }
    if (normalize) {
        // centroids in the original space = means of the original vectors over the final assignment
        // (SimpleKMeans.moveCentroid averages the un-normalised instances)
        int rc2 = sort_and_mean(dX, dC.p);
        if (rc2) return rc2;
        LCK(hipMemcpyAsync(Ch.data(), dC.p, (size_t)k_eff * d * 8, hipMemcpyDeviceToHost, st));
        LCK(hipStreamSynchronize(st));
    }
    memcpy(centroids_out, Ch.data(), (size_t)k_eff * d * 8);
    if (d_assign_out) LCK(hipMemcpyAsync(d_assign_out, a_new.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    LCK(hipStreamSynchronize(st));
    if (iters_out) *iters_out = iters;
    if (k_out) *k_out = k_eff;
    return MMIDX_OK;
}

int mmidx_kmeans(int device, int64_t n, int d, int k, int max_iter, int64_t seed, int flags, const double *X, const double *init_centroids,
                 double *centroids_out, int32_t *assign_out, double *sse_out, int32_t *iters_out, int32_t *k_out) {
    if (n < 1 || d < 1 || !X) return MMIDX_ERR_INVALID_ARG;
    if (mmidx_device_count() < 1) return MMIDX_ERR_NO_DEVICE;
    LCK(hipSetDevice(device));
    Buf<double> dX;
    Buf<int32_t> dA;
    LCK(dX.alloc((size_t)n * d));
    LCK(dA.alloc((size_t)n));
    LCK(hipMemcpy(dX.p, X, (size_t)n * d * 8, hipMemcpyHostToDevice));
    int rc = mmidx_kmeans_device(device, n, d, k, max_iter, seed, flags, dX.p, init_centroids, centroids_out, dA.p, sse_out, iters_out, k_out,
                                 nullptr);
    if (rc) return rc;
    if (assign_out) LCK(hipMemcpy(assign_out, dA.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return MMIDX_OK;
}

}  // extern "C"
