// microbenchmark for the two-waves-per-SIMD design: does vector / LDS work of ONE wave run beside the MFMAs of ANOTHER wave on the
// same SIMD?  512 threads per workgroup: waves 0-3 (one per SIMD) run role A, waves 4-7 (their SIMD partners) role B, both for the
// same number of iterations; ticks per iteration of each role, alone and together.
// (diagnostic only; build: hipcc --offload-arch=gfx950 -O3 ab/micro/two_waves.hip -o ab/micro/two_waves)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void work(f32x16& acc, float (&f)[8], const __attribute__((address_space(3))) short* lp, s16x4 (&r)[4]) {
    const bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    if (KIND == 1) {                       // 4 MFMAs (one dependent chain)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
    } else if (KIND == 2) {                // 24 v_fma_f32
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i & 7]));
    } else if (KIND == 3) {                // 24 v_pk_fma_f32
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&f[(2 * i) & 6])));
    } else if (KIND == 4) {                // 8 ds_read_b64 + wait
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i & 3]) : "v"(lp), "n"(i * 2048));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 5) {                // 8 v_exp_f32 + 16 v_fma
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i & 7]));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i & 7]));
    }
}

template <int KA, int KB>
__global__ void __launch_bounds__(512) bench(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (short)i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const __attribute__((address_space(3))) short* lp = (const __attribute__((address_space(3))) short*)lds + 4 * lane;
    f32x16 acc = f32x16(0.f);
    float f[8];
    s16x4 r[4] = {};
    for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i;
    long long t0 = clock64();
    if (wave < 4) { for (int it = 0; it < iters; ++it) work<KA>(acc, f, lp, r); }
    else          { for (int it = 0; it < iters; ++it) work<KB>(acc, f, lp, r); }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += (float)r[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0 && (wave == 0 || wave == 4)) cyc[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
}
static const char* NAME[] = {"idle", "4 MFMA 32x32x16 bf16", "24 v_fma_f32", "24 v_pk_fma_f32", "8 ds_read_b64 + wait", "8 v_exp_f32 + 16 v_fma_f32"};
template <int KA, int KB>
void run(float* out, long long* cyc) {
    const int iters = 4000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<KA, KB>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long c[2];
    (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("A: %-28s %7.1f ticks / iteration   |   B (same SIMD): %-28s %7.1f ticks / iteration\n", NAME[KA], (double)c[0] / iters, NAME[KB], (double)c[1] / iters);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 512 * 8);
    run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<3, 0>(out, cyc); run<4, 0>(out, cyc); run<5, 0>(out, cyc);
    run<1, 1>(out, cyc); run<1, 2>(out, cyc); run<1, 3>(out, cyc); run<1, 4>(out, cyc); run<1, 5>(out, cyc);
    run<2, 2>(out, cyc); run<3, 3>(out, cyc); run<2, 3>(out, cyc); run<2, 4>(out, cyc); run<3, 4>(out, cyc); run<4, 4>(out, cyc);
    return 0;
}
