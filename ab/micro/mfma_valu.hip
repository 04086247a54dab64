// microbenchmark: how many cycles does one wave need for a chain of fp32 MFMAs with K filler instructions in every gap?
// (diagnostic only; build: hipcc --offload-arch=gfx950 -O3 ab/micro/mfma_valu.hip -o ab/micro/mfma_valu)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int K>
__device__ __forceinline__ void filler(float (&f)[8], unsigned (&u)[8]) {
    unsigned u_s = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i & 7]));
        if (KIND == 1) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i & 7]));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i & 7]));
        if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&f[(2 * i) & 6])));
        if (KIND == 4) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(u[i & 7]));
        if (KIND == 5) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(u[i & 7]));
        if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(f[i & 7]));
        if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[i & 7]));
        if (KIND == 10) asm volatile("s_nop 0");
        if (KIND == 11) asm volatile("s_waitcnt lgkmcnt(0)");
        if (KIND == 12) asm volatile("s_add_u32 %0, %0, 1" : "+s"(u_s));
        if (KIND == 13) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(u[i & 7]) : : "a0");
        if (KIND == 14) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
        if (KIND == 8) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(f[i & 7]) : "v"(u[(i + 1) & 7]));
        if (KIND == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[i & 7]) : "v"(f[(i + 1) & 7]));
    }
}

template <int MF, int KIND, int K>
__global__ void __launch_bounds__(256) bench(float* out, long long* cyc, int iters) {
    f32x16 acc[2] = {f32x16(0.f), f32x16(0.f)};
    f32x4 acc4[2] = {f32x4(0.f), f32x4(0.f)};
    float f[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x + i; }
    float a = threadIdx.x * 0.5f, b = 1.0f;
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    long long t0 = __builtin_readcyclecounter();
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MF == 1) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 1], 0, 0, 0);
            if (MF == 2) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[m & 1], 0, 0, 0);
            if (MF == 4) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[0], 0, 0, 0);     // ONE dependent chain
            if (MF == 3) acc4[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m & 1], 0, 0, 0);
            filler<KIND, K>(f, u);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    for (int i = 0; i < 4; ++i) s += acc4[0][i] + acc4[1][i];
    for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MF, int KIND, int K>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((bench<MF, KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((bench<MF, KIND, K>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s K=%2d: %7.1f clock64-ticks per MFMA gap\n", name, K, (double)c / iters / 8);
}
#define SWEEP(MF, KIND, name) \
    run<MF, KIND, 0>(name, out, cyc); run<MF, KIND, 4>(name, out, cyc); run<MF, KIND, 8>(name, out, cyc); run<MF, KIND, 12>(name, out, cyc); run<MF, KIND, 16>(name, out, cyc);
int main(int argc, char**) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    SWEEP(0, 0, "no mfma + v_fma_f32")
    SWEEP(0, 1, "no mfma + v_add_u32")
    SWEEP(0, 3, "no mfma + v_pk_fma_f32")
    SWEEP(0, 2, "no mfma + v_exp_f32")
    SWEEP(1, 0, "mfma32x32x2f32 + v_fma_f32")
    SWEEP(1, 6, "mfma32x32x2f32 + v_mul_f32")
    SWEEP(1, 1, "mfma32x32x2f32 + v_add_u32")
    SWEEP(1, 4, "mfma32x32x2f32 + v_alignbit")
    SWEEP(1, 5, "mfma32x32x2f32 + v_xor")
    SWEEP(1, 7, "mfma32x32x2f32 + v_cndmask")
    SWEEP(1, 2, "mfma32x32x2f32 + v_exp_f32")
    SWEEP(1, 3, "mfma32x32x2f32 + v_pk_fma_f32")
    SWEEP(3, 0, "mfma16x16x4f32 + v_fma_f32")
    SWEEP(2, 0, "mfma32x32x16bf16 + v_fma_f32")
    SWEEP(2, 3, "mfma32x32x16bf16 + v_pk_fma_f32")
    SWEEP(4, 0, "mfma32x32x16bf16 one chain + v_fma")
    SWEEP(0, 10, "no mfma + s_nop 0")
    SWEEP(0, 11, "no mfma + s_waitcnt lgkmcnt(0)")
    SWEEP(0, 14, "no mfma + s_waitcnt vmcnt(0) lgkmcnt(0)")
    SWEEP(0, 12, "no mfma + s_add_u32")
    SWEEP(0, 13, "no mfma + accvgpr write+read pair")
    SWEEP(0, 8, "no mfma + v_dot2c_f32_bf16")
    SWEEP(2, 8, "mfma32x32x16bf16 + v_dot2c_f32_bf16")
    SWEEP(0, 9, "no mfma + v_cvt_pk_bf16_f32")
    SWEEP(2, 9, "mfma32x32x16bf16 + v_cvt_pk_bf16_f32")
    if (argc > 1) return 0;
    return 0;
}
