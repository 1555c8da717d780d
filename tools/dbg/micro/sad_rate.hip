// sad_rate.hip -- issue rate of the byte-sum instructions K3g's scan could use to take a table read's eight bytes apart
// (gfx950): v_mqsad_pk_u16_u8 (four masked byte differences into four 16-bit sums in ONE instruction), v_msad_u8,
// v_dot4_u32_u8, against v_add3_u32 / v_perm_b32 / v_and_b32 (the current spread-and-add sequence), plus a check of the
// semantics the scan would rely on.  Build: hipcc --offload-arch=gfx950 -O2 -o sad_rate sad_rate.hip; run on the box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t u32;

#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k_rate(u64 *out, int iters, u32 seed) {
    u64 a0 = threadIdx.x, a1 = threadIdx.x + 1, a2 = threadIdx.x + 2, a3 = threadIdx.x + 3;
    u64 x0 = (u64)seed * 0x9E3779B97F4A7C15ull + threadIdx.x, x1 = x0 * 3 + 1;
    u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, y0 = (u32)x0, y1 = (u32)x1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if constexpr (OP == 0) {  // four independent chains of v_mqsad_pk_u16_u8
                asm volatile("v_mqsad_pk_u16_u8 %0, %1, 1, %0" : "+v"(a0) : "v"(x0));
                asm volatile("v_mqsad_pk_u16_u8 %0, %1, 1, %0" : "+v"(a1) : "v"(x1));
                asm volatile("v_mqsad_pk_u16_u8 %0, %1, 1, %0" : "+v"(a2) : "v"(x0));
                asm volatile("v_mqsad_pk_u16_u8 %0, %1, 1, %0" : "+v"(a3) : "v"(x1));
            } else if constexpr (OP == 1) {  // v_add3_u32
                asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(b0) : "v"(y0), "v"(y1));
                asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(b1) : "v"(y0), "v"(y1));
                asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(b2) : "v"(y0), "v"(y1));
                asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(b3) : "v"(y0), "v"(y1));
            } else if constexpr (OP == 2) {  // v_perm_b32
                asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(b0) : "v"(y0), "v"(y1));
                asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(b1) : "v"(y0), "v"(y1));
                asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(b2) : "v"(y0), "v"(y1));
                asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(b3) : "v"(y0), "v"(y1));
            } else if constexpr (OP == 3) {  // v_msad_u8
                asm volatile("v_msad_u8 %0, %1, 1, %0" : "+v"(b0) : "v"(y0));
                asm volatile("v_msad_u8 %0, %1, 1, %0" : "+v"(b1) : "v"(y1));
                asm volatile("v_msad_u8 %0, %1, 1, %0" : "+v"(b2) : "v"(y0));
                asm volatile("v_msad_u8 %0, %1, 1, %0" : "+v"(b3) : "v"(y1));
            } else if constexpr (OP == 4) {  // v_dot4_u32_u8
                asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(b0) : "v"(y0), "v"(y1));
                asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(b1) : "v"(y0), "v"(y1));
                asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(b2) : "v"(y0), "v"(y1));
                asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(b3) : "v"(y0), "v"(y1));
            } else if constexpr (OP == 5) {  // v_qsad_pk_u16_u8 (unmasked)
                asm volatile("v_qsad_pk_u16_u8 %0, %1, 0, %0" : "+v"(a0) : "v"(x0));
                asm volatile("v_qsad_pk_u16_u8 %0, %1, 0, %0" : "+v"(a1) : "v"(x1));
                asm volatile("v_qsad_pk_u16_u8 %0, %1, 0, %0" : "+v"(a2) : "v"(x0));
                asm volatile("v_qsad_pk_u16_u8 %0, %1, 0, %0" : "+v"(a3) : "v"(x1));
            } else if constexpr (OP == 6) {  // v_pk_add_u16
                asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(b0) : "v"(y0));
                asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(b1) : "v"(y1));
                asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(b2) : "v"(y0));
                asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(b3) : "v"(y1));
            } else if constexpr (OP == 7) {  // v_sad_u8
                asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(b0) : "v"(y0));
                asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(b1) : "v"(y1));
                asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(b2) : "v"(y0));
                asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(b3) : "v"(y1));
            }
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}

// semantics: D = mqsad_pk(S0 = 8 bytes, S1 = ref, S2 = 4 x u16)
__global__ void k_sem(const u64 *s0, const u32 *ref, const u64 *acc, u64 *out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    u64 d = acc[i];
    const u64 a = s0[i];
    const u32 r = ref[i];
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(r));
    out[i] = d;
}

static u64 model(u64 s0, u32 ref, u64 acc) {
    u64 d = 0;
    for (int f = 0; f < 4; f++) {
        const u32 win = (u32)(s0 >> (8 * f));
        u32 sum = (u32)((acc >> (16 * f)) & 0xFFFF);
        for (int b = 0; b < 4; b++) {
            const int rb = (ref >> (8 * b)) & 0xFF, sb = (win >> (8 * b)) & 0xFF;
            if (rb != 0) sum += (u32)(sb > rb ? sb - rb : rb - sb);
        }
        d |= (u64)(sum & 0xFFFF) << (16 * f);
    }
    return d;
}

template <int OP>
static double run(const char *name, u64 *d_out, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<OP><<<blocks, 256>>>(d_out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<OP><<<blocks, 256>>>(d_out, iters, 7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave instructions per SIMD: blocks * 4 waves / (CUs * 4 SIMDs) * iters * REP * 4
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const double waves_per_simd = (double)blocks * 4 / (pr.multiProcessorCount * 4.0);
    const double inst = waves_per_simd * iters * REP * 4.0;
    const double cyc = ms * 1e-3 * pr.clockRate * 1e3;  // clockRate in kHz
    printf("%-22s %8.3f ms  %6.2f cycles per wave instruction (at %d MHz nominal, %.1f waves per SIMD)\n", name, ms, cyc / inst,
           pr.clockRate / 1000, waves_per_simd);
    return cyc / inst;
}

int main() {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) {
        printf("no device\n");
        return 1;
    }
    const int blocks = pr.multiProcessorCount * 4;  // 16 waves per CU = 4 per SIMD
    u64 *d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 8);
    run<1>("v_add3_u32", d_out, blocks);
    run<2>("v_perm_b32", d_out, blocks);
    run<6>("v_pk_add_u16", d_out, blocks);
    run<4>("v_dot4_u32_u8", d_out, blocks);
    run<7>("v_sad_u8", d_out, blocks);
    run<3>("v_msad_u8", d_out, blocks);
    run<5>("v_qsad_pk_u16_u8", d_out, blocks);
    run<0>("v_mqsad_pk_u16_u8", d_out, blocks);
    // semantics
    const int n = 64;
    std::vector<u64> s0(n), acc(n), out(n);
    std::vector<u32> ref(n);
    u64 x = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        s0[i] = x;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        acc[i] = x & 0x3FFF3FFF3FFF3FFFull;
        ref[i] = i < 32 ? 1u : (i < 48 ? 0xFFu : (u32)(x >> 40) & 0xFF00FFFFu);
    }
    s0[0] = 0; s0[1] = 0x0101010101010101ull; s0[2] = 0xFFFFFFFFFFFFFFFFull;
    u64 *d_s0, *d_acc, *d_o;
    u32 *d_ref;
    hipMalloc(&d_s0, n * 8); hipMalloc(&d_acc, n * 8); hipMalloc(&d_o, n * 8); hipMalloc(&d_ref, n * 4);
    hipMemcpy(d_s0, s0.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_acc, acc.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_ref, ref.data(), n * 4, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(d_s0, d_ref, d_acc, d_o, n);
    hipMemcpy(out.data(), d_o, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const u64 m = model(s0[i], ref[i], acc[i]);
        if (m != out[i]) {
            if (bad < 8) printf("semantics differ at %d: s0 %016llx ref %08x acc %016llx -> %016llx, model %016llx\n", i, s0[i], ref[i], acc[i], out[i], m);
            bad++;
        }
    }
    printf("v_mqsad_pk_u16_u8 semantics: %d of %d differ from the model (masked where the REFERENCE byte is 0; windows at bits 0/8/16/24)\n", bad, n);
    return 0;
}
