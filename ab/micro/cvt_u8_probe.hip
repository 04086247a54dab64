// what does v_cvt_pk_u8_f32 do with fractions, negatives and values above 255?  hipcc --offload-arch=gfx950 -O2 ab/micro/cvt_u8_probe.hip -o /tmp/cvt && /tmp/cvt
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xAABBCCDDu);
}
int main() {
    const float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 254.6f, 255.4f, 255.5f, 256.f, 300.f, -0.4f, -0.6f, -3.f, 100.49f, 100.51f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned r[32];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(r, o, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%8.2f -> byte1 = %3u  (dword %08x)\n", h[i], (r[i] >> 8) & 255u, r[i]);
    return 0;
}
