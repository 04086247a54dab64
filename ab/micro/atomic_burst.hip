// microbenchmark: what does a no-return global_atomic_add_f32 instruction cost the issuing wave when 1 in D waves of the chip issue
// them at the same time (D = 1: all 1024 waves at once, like the synchronised grid-gradient flush of the fused kernel)?
// 48 atomic instructions per "flush", lanes contiguous (one 256-byte segment per instruction), segments spread over 32 MB.
// (diagnostic only; build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics ab/micro/atomic_burst.hip -o ab/micro/atomic_burst)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float* buf, long long* cyc, int nflush, int nfloats, int duty, int segs, int by_block) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned off = (unsigned)wave * 6151u;
    float f = threadIdx.x;
    long long tsum = 0;
    for (int it = 0; it < nflush * duty; ++it) {
        if ((it + (by_block ? (int)blockIdx.x : wave)) % duty == 0) {
            long long t0 = clock64();
            for (int i = 0; i < 48; ++i) {
                off = (off * 1664525u + 1013904223u);
                // segs = 1: 64 lanes on one 256 B segment; segs = 4: four 64 B pieces (two grid rows x two lane halves, like the flush)
                float* p = buf + ((off >> 6) % (unsigned)(nfloats / 1024)) * 1024 + (lane / (64 / segs)) * 256 + lane % (64 / segs);
                __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            tsum += clock64() - t0;
        } else {
            for (int i = 0; i < 3000; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));     // ~ one round of compute
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = tsum;
    if (f == 12345.f) buf[0] = f;
}
int main() {
    const int nf = 8 << 20;
    float* buf; long long* cyc;
    (void)hipMalloc(&buf, nf * 4); (void)hipMalloc(&cyc, 256 * 8);
    for (int by_block = 0; by_block < 2; ++by_block)
        for (int duty = 1; duty <= 16; duty *= 2) {
            const int segs = 4;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemset(buf, 0, nf * 4);
                hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, buf, cyc, 16, nf, duty, segs, by_block);
                (void)hipDeviceSynchronize();
            }
            long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s: 1 in %2d flushing at a time: %7.1f ticks per atomic instruction\n", by_block ? "whole workgroups (4 waves of a CU together)" : "single waves", duty, (double)c / 16 / 48);
        }
    return 0;
}
