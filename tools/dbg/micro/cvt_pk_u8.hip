// build: hipcc --offload-arch=gfx950 -O2 -o cvt_pk_u8 cvt_pk_u8.hip
// rounding / saturation of v_cvt_pk_u8_f32 on gfx950 (used by K3g's table build): prints the byte for a few inputs
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float *in, unsigned *out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main() {
    const float h[] = {0.0f, 0.4f, 0.5f, 0.6f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.7f, 255.0f, 255.5f, 300.f, 1e30f, -0.3f, -5.f, INFINITY, -INFINITY, NAN};
    const int n = sizeof(h) / sizeof(h[0]);
    float *d;
    unsigned *o, ho[32];
    hipMalloc(&d, sizeof(h));
    hipMalloc(&o, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%g -> %u\n", h[i], ho[i]);
    return 0;
}
