// mmidx_probe.hip -- two micro-benchmarks that put a measured ceiling next to the rooflines bench.py quotes.
// They are instrumentation entry points of libmmidx_hip.so (include/mmidx.h), not part of the search path.
//
//   mmidx_probe_lds_gather   the inner loop of pass A (k_scan_hist) with everything but the gather removed: per code m random
//                            8-byte LDS reads (row s at s * 2048, slot 8 * byte -- bytes from a register hash, no memory
//                            traffic) added up in fp64, `chains` independent codes in flight per lane, at the kernel's own
//                            occupancy (256-thread blocks, m x 2 KiB of table each).  What the LDS pipe delivers for THIS
//                            access pattern -- random 64-lane ds_read_b64 over a 2 KiB row collide on banks -- is the
//                            ceiling of the exact scan, whatever HBM could stream.
//   mmidx_probe_f64_mfma     v_mfma_f64_16x16x4_f64 issued back to back from every SIMD: the f64 matrix peak the PCA
//                            projection (k_pca_project) is priced against.
#include "../../include/mmidx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef __attribute__((address_space(3))) const double lds_cdouble;

__device__ __forceinline__ u32 byte_x8(u32 w, int b) {  // 8 * (byte b of w) in one VALU instruction (SDWA byte select)
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

// M sub-quantizers, U chains.  The table starts the block's LDS (dynamic segment at address 0: no static LDS here).
template <int M, int U>
__global__ __launch_bounds__(256) void k_probe_lds_gather(int iters, double *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lut = (double *)smem;
    for (int i = threadIdx.x; i < M * 256; i += 256) lut[i] = 1e-3 * (double)((i * 37) & 255);
    __syncthreads();
    u32 st[U];
#pragma unroll
    for (int u = 0; u < U; u++) st[u] = (u32)(blockIdx.x * 256 + threadIdx.x) * 2654435761u + (u32)u * 0x9E3779B9u + 1u;
    double acc = 0.0;
    for (int it = 0; it < iters; it++) {
        u32 wd[U][M / 4];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < M / 4; j++) {  // the "code": M pseudo-random bytes (xorshift per word)
                u32 x = st[u];
                x ^= x << 13;
                x ^= x >> 17;
                x ^= x << 5;
                st[u] = x;
                wd[u][j] = x;
            }
        }
        double dd[U];
#pragma unroll
        for (int u = 0; u < U; u++) dd[u] = *(lds_cdouble *)(size_t)(byte_x8(wd[u][0], 0));
#pragma unroll
        for (int s = 1; s < M; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) dd[u] += *(lds_cdouble *)(size_t)(byte_x8(wd[u][s >> 2], s & 3) + (u32)s * 2048u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += dd[u];
    }
    if (acc == 12345.678) sink[0] = acc;  // (keeps the loop alive)
}

// The same loop with the LAST KG rows of the table served by the vector-memory path instead of the LDS: each block writes its own
// KG x 2 KiB copy to global memory and gathers from it (L1-resident after the first touch: 4 blocks x KG x 2 KiB of the CU's
// 32 KiB) -- does sharing the gather between the two pipes beat the LDS alone?  (pass A.)
template <int M, int U, int KG>
__global__ __launch_bounds__(256) void k_probe_split_gather(int iters, double *__restrict__ gtab, double *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *lut = (double *)smem;
    double *mine = gtab + (size_t)blockIdx.x * KG * 256;
    for (int i = threadIdx.x; i < (M - KG) * 256; i += 256) lut[i] = 1e-3 * (double)((i * 37) & 255);
    for (int i = threadIdx.x; i < KG * 256; i += 256) mine[i] = 1e-3 * (double)((i * 41) & 255);
    __threadfence();
    __syncthreads();
    u32 st[U];
#pragma unroll
    for (int u = 0; u < U; u++) st[u] = (u32)(blockIdx.x * 256 + threadIdx.x) * 2654435761u + (u32)u * 0x9E3779B9u + 1u;
    double acc = 0.0;
    for (int it = 0; it < iters; it++) {
        u32 wd[U][M / 4];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < M / 4; j++) {
                u32 x = st[u];
                x ^= x << 13;
                x ^= x >> 17;
                x ^= x << 5;
                st[u] = x;
                wd[u][j] = x;
            }
        }
        double gg[U][KG];  // the global rows first: their latency runs under the LDS rows
#pragma unroll
        for (int s = M - KG; s < M; s++) {
#pragma unroll
            for (int u = 0; u < U; u++)
                gg[u][s - (M - KG)] = *(const double *)((const char *)(mine + (s - (M - KG)) * 256) + byte_x8(wd[u][s >> 2], s & 3));
        }
        double dd[U];
#pragma unroll
        for (int u = 0; u < U; u++) dd[u] = *(lds_cdouble *)(size_t)(byte_x8(wd[u][0], 0));
#pragma unroll
        for (int s = 1; s < M - KG; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) dd[u] += *(lds_cdouble *)(size_t)(byte_x8(wd[u][s >> 2], s & 3) + (u32)s * 2048u);
        }
#pragma unroll
        for (int s = 0; s < KG; s++) {
#pragma unroll
            for (int u = 0; u < U; u++) dd[u] += gg[u][s];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += dd[u];
    }
    if (acc == 12345.678) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_probe_f64_mfma(int iters, double *__restrict__ sink) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * (double)threadIdx.x, b = 1.0 - 1e-9 * (double)threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 4; t++) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    if (s == 12345.678) sink[0] = s;
}

thread_local char g_perr[256];

template <int M, int U>
hipError_t run_gather(int blocks, int iters, double *sink, hipStream_t st, hipEvent_t e0, hipEvent_t e1, float *ms) {
    const size_t lds = (size_t)M * 2048;
    hipError_t e = hipFuncSetAttribute((const void *)k_probe_lds_gather<M, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_probe_lds_gather<M, U>), dim3((unsigned)blocks), dim3(256), lds, st, iters / 8 + 1, sink);  // warm-up
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL((k_probe_lds_gather<M, U>), dim3((unsigned)blocks), dim3(256), lds, st, iters, sink);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    if (e != hipSuccess) return e;
    return hipEventElapsedTime(ms, e0, e1);
}

}  // namespace

extern "C" {

// out[0] = wave-level ds_read_b64 gathers per second (whole device), out[1] = the same as "codes x m bytes" per second
// (GB/s: what pass A's roofline object calls algorithmic bytes), out[2] = blocks per CU that were resident, out[3] = seconds
int mmidx_probe_lds_gather(int device, int m, int chains, double *out) {
    if (!out) return MMIDX_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MMIDX_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MMIDX_ERR_NO_DEVICE;
    if ((m != 8 && m != 16 && m != 32) || (chains != 1 && chains != 3)) return MMIDX_ERR_INVALID_ARG;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (cus < 1) cus = 256;
    const int per_cu = std::max(1, std::min(8, (160 * 1024) / (m * 2048 + 2048)));  // (K3h: four 34 KiB blocks per CU at m = 16)
    const int blocks = cus * std::min(per_cu, 4) * 4;                                 // four rounds of resident blocks
    const int iters = 4096 / (m / 8);
    double *sink = nullptr;
    hipEvent_t e0, e1;
    if (hipMalloc((void **)&sink, 64) != hipSuccess) return MMIDX_ERR_HIP;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0.f;
    hipError_t e;
    if (m == 8) e = chains == 3 ? run_gather<8, 3>(blocks, iters, sink, nullptr, e0, e1, &ms) : run_gather<8, 1>(blocks, iters, sink, nullptr, e0, e1, &ms);
    else if (m == 16) e = chains == 3 ? run_gather<16, 3>(blocks, iters, sink, nullptr, e0, e1, &ms) : run_gather<16, 1>(blocks, iters, sink, nullptr, e0, e1, &ms);
    else e = chains == 3 ? run_gather<32, 3>(blocks, iters, sink, nullptr, e0, e1, &ms) : run_gather<32, 1>(blocks, iters, sink, nullptr, e0, e1, &ms);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return MMIDX_ERR_HIP;
    const double sec = (double)ms * 1e-3;
    const double codes = (double)blocks * 256.0 * (double)iters * (double)chains;
    out[0] = codes * m / 64.0 / sec;
    out[1] = codes * m / sec / 1e9;
    out[2] = (double)std::min(per_cu, 4);
    out[3] = sec;
    return MMIDX_OK;
}

// pass A's gather with kg of the m = 16 rows on the vector-memory path (kg = 0: the plain LDS loop through the same harness).
// out as mmidx_probe_lds_gather.
int mmidx_probe_split_gather(int device, int kg, double *out) {
    if (!out) return MMIDX_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MMIDX_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MMIDX_ERR_NO_DEVICE;
    if (kg != 1 && kg != 2 && kg != 4) return MMIDX_ERR_INVALID_ARG;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (cus < 1) cus = 256;
    constexpr int M = 16;
    const int blocks = cus * 4 * 4, iters = 2048;
    double *sink = nullptr, *gtab = nullptr;
    if (hipMalloc((void **)&sink, 64) != hipSuccess) return MMIDX_ERR_HIP;
    if (hipMalloc((void **)&gtab, (size_t)blocks * 4 * 256 * 8) != hipSuccess) { (void)hipFree(sink); return MMIDX_ERR_HIP; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0.f;
    hipError_t e = hipSuccess;
    const size_t lds = (size_t)M * 2048;  // (the LDS footprint of the real kernel: four blocks per CU)
    auto run = [&](auto kern) {
        e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, nullptr, iters / 8 + 1, gtab, sink);
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, nullptr, iters, gtab, sink);
        (void)hipEventRecord(e1, nullptr);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    };
    if (kg == 1) run(k_probe_split_gather<M, 3, 1>);
    else if (kg == 2) run(k_probe_split_gather<M, 3, 2>);
    else run(k_probe_split_gather<M, 3, 4>);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    (void)hipFree(gtab);
    if (e != hipSuccess) return MMIDX_ERR_HIP;
    const double sec = (double)ms * 1e-3;
    const double codes = (double)blocks * 256.0 * (double)iters * 3.0;
    out[0] = codes * M / 64.0 / sec;
    out[1] = codes * M / sec / 1e9;
    out[2] = 4.0;
    out[3] = sec;
    return MMIDX_OK;
}

// out[0] = TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 over the whole device, out[1] = seconds
int mmidx_probe_f64_mfma(int device, double *out) {
    if (!out) return MMIDX_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MMIDX_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MMIDX_ERR_NO_DEVICE;
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (cus < 1) cus = 256;
    const int blocks = cus * 8, iters = 20000;
    double *sink = nullptr;
    hipEvent_t e0, e1;
    if (hipMalloc((void **)&sink, 64) != hipSuccess) return MMIDX_ERR_HIP;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe_f64_mfma, dim3((unsigned)blocks), dim3(256), 0, nullptr, iters / 10, sink);
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_probe_f64_mfma, dim3((unsigned)blocks), dim3(256), 0, nullptr, iters, sink);
    (void)hipEventRecord(e1, nullptr);
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return MMIDX_ERR_HIP;
    const double flops = (double)blocks * 4.0 /*waves*/ * (double)iters * 16.0 /*mfma per iteration*/ * 2048.0;
    out[0] = flops / ((double)ms * 1e-3) / 1e12;
    out[1] = (double)ms * 1e-3;
    return MMIDX_OK;
}

}  // extern "C"
