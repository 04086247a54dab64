// microbenchmark: do vector instructions of a wave run beside its LDS reads when all four waves of a workgroup saturate the LDS pipe?
// Every wave: 16 x ds_read_b64 (or the transposing ds_read_b64_tr_b16), K filler v_fma_f32 after each read, one wait per group of 16.
// (diagnostic only; build: hipcc --offload-arch=gfx950 -O3 ab/micro/lds_valu.hip -o ab/micro/lds_valu)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int TR, int K, int NW>
__global__ void __launch_bounds__(64 * NW) bench(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) short lds[32768];
    for (int i = threadIdx.x; i < 32768; i += 64 * NW) lds[i] = (short)i;
    __syncthreads();
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i;
    const int lane = threadIdx.x & 63;
    // conflict-free: lane reads 8 consecutive bytes
    const __attribute__((address_space(3))) short* base = (const __attribute__((address_space(3))) short*)lds + 4 * lane + 256 * (threadIdx.x >> 6);
    const __attribute__((address_space(3))) short* base2 = (const __attribute__((address_space(3))) short*)lds + 8 * lane + 512 * (threadIdx.x >> 6);
    s16x4 r[16];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (TR == 2) { r[m] = s16x4{0, 0, 0, 0}; }
            else if (TR == 3) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(*(int*)&r[m]) : "v"(base), "n"(m * 2048));
            else if (TR == 4) { if (m & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(__int128*)&r[m - 1]) : "v"(base2), "n"((m >> 1) * 4096)); }
            else if (TR) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[m]) : "v"(base), "n"(m * 2048));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[m]) : "v"(base), "n"(m * 2048));
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i & 7]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 16; ++m) asm volatile("" :: "v"(r[m]));
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int m = 0; m < 16; ++m) s += (float)r[m][0];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int TR, int K, int NW>
void run(float* out, long long* cyc) {
    const int iters = 20000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<TR, K, NW>), dim3(256), dim3(64 * NW), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s  waves/CU %d  K=%2d fillers per read: %7.1f ticks per 16 reads   (fillers alone: %d issue cycles)\n", TR == 2 ? "no read           " : TR == 3 ? "ds_read_b32 x16   " : TR == 4 ? "ds_read_b128 x8   " : TR ? "ds_read_b64_tr_b16" : "ds_read_b64       ", NW, K,
           (double)c / iters, 16 * K * 4);
}
#define SWEEP(TR, NW) run<TR, 0, NW>(out, cyc); run<TR, 8, NW>(out, cyc); run<TR, 16, NW>(out, cyc); run<TR, 24, NW>(out, cyc); run<TR, 32, NW>(out, cyc);
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    SWEEP(2, 4) SWEEP(3, 4) SWEEP(0, 4) SWEEP(4, 4) SWEEP(1, 4) SWEEP(3, 1) SWEEP(0, 1) SWEEP(4, 1)
    return 0;
}
