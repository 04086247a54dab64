// microbenchmark: what does one GELU + derivative evaluation cost a SIMD that runs two waves, in the forms the plain-16-bit kernels could use?
// 512 threads per workgroup: waves 0-3 run role A, their SIMD partners 4-7 role B.  Each iteration evaluates 16 values per lane (the four
// row tiles of one hidden layer of fused_q16_kernel) and packs activations + derivatives into 16-bit fragments like the kernel does.
//   V0: the kernels' exact-erf form (nic_device.hpp::gelu_and_grad4: Abramowitz-Stegun 7.1.26, packed fp32)
//   V1: sigmoid form Phi(z) = 1 / (1 + 2^(z w(z^2))), w a quadratic in z^2 (|GELU error| 3.7e-5, derivative 9.3e-5), plain fp32
//   V2: the same, written as packed fp32 pairs
//   V3: the same in packed fp16 (v_pk_fma_f16; results are the fp16 fragments themselves)
//   V4: V1 with the two-coefficient w (3.3e-4 / 6.7e-4); V5: the same as packed fp32 pairs
// (diagnostic only; build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize ab/micro/gelu_probe.hip -o ab/micro/gelu_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../neural_image_compression_v2_amd/csrc/nic_device.hpp"
using namespace nic;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

constexpr float C0 = -2.30087589f, C1 = -1.06770560e-01f, C2 = 1.00279102e-03f, LN2 = 0.69314718056f;
constexpr float D0 = -2.3080629f, D1 = -0.10091118f;

__device__ __forceinline__ uint32_t pk_bf(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <int V>
__device__ __forceinline__ void gelu16(const f32x4 (&z)[4], u32x4 (&a)[2], u32x4 (&d)[2]) {
    if constexpr (V == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 a4[2], d4[2];
            gelu_and_grad4(z[2 * s], a4[0], d4[0]);
            gelu_and_grad4(z[2 * s + 1], a4[1], d4[1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[s][2 * h] = pk_bf(a4[h][0], a4[h][1]); a[s][2 * h + 1] = pk_bf(a4[h][2], a4[h][3]);
                d[s][2 * h] = pk_bf(d4[h][0], d4[h][1]); d[s][2 * h + 1] = pk_bf(d4[h][2], d4[h][3]);
            }
        }
    } else if constexpr (V == 1 || V == 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float av[4], dv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x = z[2 * s + h][i];
                    float w, q;
                    if (V == 1) {
                        const float x2 = fminf(x * x, 64.f);
                        w = fmaf(fmaf(C2, x2, C1), x2, C0);
                        q = fmaf(fmaf(-5.f * LN2 * C2, x2, -3.f * LN2 * C1), x2, -LN2 * C0);
                    } else {
                        const float x2 = x * x;
                        w = fmaf(D1, x2, D0);
                        q = fmaf(-3.f * LN2 * D1, x2, -LN2 * D0);
                    }
                    const float e = __builtin_amdgcn_exp2f(x * w);
                    const float P = __builtin_amdgcn_rcpf(1.0f + e);
                    av[i] = x * P;
                    dv[i] = fmaf(av[i] * (1.0f - P), q, P);
                }
                a[s][2 * h] = pk_bf(av[0], av[1]); a[s][2 * h + 1] = pk_bf(av[2], av[3]);
                d[s][2 * h] = pk_bf(dv[0], dv[1]); d[s][2 * h + 1] = pk_bf(dv[2], dv[3]);
            }
    } else if constexpr (V == 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x2 x = {z[2 * s + h][2 * i], z[2 * s + h][2 * i + 1]};
                    f32x2 x2 = x * x;
                    x2[0] = fminf(x2[0], 64.f); x2[1] = fminf(x2[1], 64.f);
                    const f32x2 w = pk_fma(pk_fma(f32x2(C2), x2, f32x2(C1)), x2, f32x2(C0));
                    const f32x2 q = pk_fma(pk_fma(f32x2(-5.f * LN2 * C2), x2, f32x2(-3.f * LN2 * C1)), x2, f32x2(-LN2 * C0));
                    const f32x2 u = x * w;
                    f32x2 P;
                    P[0] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[0]));
                    P[1] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[1]));
                    const f32x2 av = x * P;
                    const f32x2 dv = pk_fma(av * (f32x2(1.0f) - P), q, P);
                    a[s][2 * h + i] = pk_bf(av[0], av[1]);
                    d[s][2 * h + i] = pk_bf(dv[0], dv[1]);
                }
    } else if constexpr (V == 5) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x2 x = {z[2 * s + h][2 * i], z[2 * s + h][2 * i + 1]};
                    const f32x2 x2 = x * x;
                    const f32x2 w = pk_fma(f32x2(D1), x2, f32x2(D0));
                    const f32x2 q = pk_fma(f32x2(-3.f * LN2 * D1), x2, f32x2(-LN2 * D0));
                    const f32x2 u = x * w;
                    f32x2 P;
                    P[0] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[0]));
                    P[1] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u[1]));
                    const f32x2 av = x * P;
                    const f32x2 dv = pk_fma(av * (f32x2(1.0f) - P), q, P);
                    a[s][2 * h + i] = pk_bf(av[0], av[1]);
                    d[s][2 * h + i] = pk_bf(dv[0], dv[1]);
                }
    } else if constexpr (V == 3) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x2 xf = {z[2 * s + h][2 * i], z[2 * s + h][2 * i + 1]};
                    const h16x2 x = __builtin_convertvector(xf, h16x2);
                    h16x2 x2 = x * x;
                    x2 = __builtin_elementwise_min(x2, h16x2((_Float16)64.f));
                    const h16x2 w = __builtin_elementwise_fma(__builtin_elementwise_fma(h16x2((_Float16)C2), x2, h16x2((_Float16)C1)), x2, h16x2((_Float16)C0));
                    const h16x2 q = __builtin_elementwise_fma(__builtin_elementwise_fma(h16x2((_Float16)(-5.f * LN2 * C2)), x2, h16x2((_Float16)(-3.f * LN2 * C1))), x2, h16x2((_Float16)(-LN2 * C0)));
                    const h16x2 u = x * w;
                    h16x2 P;
                    P[0] = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(u[0]));
                    P[1] = __builtin_amdgcn_rcph((_Float16)1.0f + __builtin_elementwise_exp2(u[1]));
                    const h16x2 av = x * P;
                    const h16x2 dv = __builtin_elementwise_fma(av * (h16x2((_Float16)1.0f) - P), q, P);
                    a[s][2 * h + i] = __builtin_bit_cast(uint32_t, av);
                    d[s][2 * h + i] = __builtin_bit_cast(uint32_t, dv);
                }
    }
}

template <int KIND>
__device__ __forceinline__ void work(f32x16& acc, f32x4 (&z)[4], uint32_t& sink) {
    const bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    if constexpr (KIND == 100) {            // 8 MFMAs 16x16x32 + 4 32x32x16: roughly a layer's matrix work
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc, 0, 0, 0);
    } else if constexpr (KIND == 101) {
    } else {
        u32x4 a[2], d[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(z[t]));
        gelu16<KIND>(z, a, d);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            asm volatile("" : "+v"(a[s]), "+v"(d[s]));
            sink ^= a[s][0] ^ a[s][3] ^ d[s][1] ^ d[s][2];
        }
    }
}

template <int KA, int KB>
__global__ void __launch_bounds__(512) bench(float* out, long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc = f32x16(0.f);
    f32x4 z[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 4; ++i) z[t][i] = (float)((threadIdx.x * 7 + t * 4 + i) % 97) * 0.08f - 3.9f;
    uint32_t sink = 0;
    long long t0 = clock64();
    if (wave < 4) { for (int it = 0; it < iters; ++it) work<KA>(acc, z, sink); }
    else          { for (int it = 0; it < iters; ++it) work<KB>(acc, z, sink); }
    long long t1 = clock64();
    float s = (float)sink;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0 && (wave == 0 || wave == 4)) cyc[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
}
static const char* name(int k) {
    switch (k) {
        case 0: return "V0 erf A&S packed fp32";
        case 1: return "V1 sigmoid3 plain fp32";
        case 2: return "V2 sigmoid3 packed fp32";
        case 3: return "V3 sigmoid3 packed fp16";
        case 4: return "V4 sigmoid2 plain fp32";
        case 5: return "V5 sigmoid2 packed fp32";
        case 100: return "4 MFMA 32x32x16 bf16";
        default: return "idle";
    }
}
template <int KA, int KB>
void run(float* out, long long* cyc) {
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<KA, KB>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long c[2];
    (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("A: %-26s %7.1f ticks / 16 values   |   B (same SIMD): %-26s %7.1f\n", name(KA), (double)c[0] / iters, name(KB), (double)c[1] / iters);
}
// accuracy of the forms against the fp64 definition
template <int V>
__global__ void acc_kernel(float* out) {
    f32x4 z[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 4; ++i) z[t][i] = -9.f + 18.f * (float)((blockIdx.x * 64 + threadIdx.x) * 16 + t * 4 + i) / (float)(256 * 64 * 16);
    u32x4 a[2], d[2];
    gelu16<V>(z, a, d);
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 4; ++j) {
            const int base = ((blockIdx.x * 64 + threadIdx.x) * 16 + s * 8 + 2 * j) * 2;
            uint32_t wa = a[s][j], wd = d[s][j];
            if (V == 3) {
                const h16x2 ha = __builtin_bit_cast(h16x2, wa), hd = __builtin_bit_cast(h16x2, wd);
                out[base] = (float)ha[0]; out[base + 2] = (float)ha[1]; out[base + 1] = (float)hd[0]; out[base + 3] = (float)hd[1];
            } else {
                out[base] = __builtin_bit_cast(float, wa << 16); out[base + 2] = __builtin_bit_cast(float, wa & 0xFFFF0000u);
                out[base + 1] = __builtin_bit_cast(float, wd << 16); out[base + 3] = __builtin_bit_cast(float, wd & 0xFFFF0000u);
            }
        }
}
#include <cmath>
#include <vector>
template <int V>
void accuracy(float* dev) {
    const int n = 256 * 64 * 16;
    hipLaunchKernelGGL((acc_kernel<V>), dim3(256), dim3(64), 0, 0, dev);
    std::vector<float> h(2 * n);
    (void)hipMemcpy(h.data(), dev, 2 * n * 4, hipMemcpyDeviceToHost);
    double ea = 0, ed = 0;
    for (int i = 0; i < n; ++i) {
        const double z = -9.0 + 18.0 * (double)i / (double)n;       // the kernel's fp32 z differs by an ulp: irrelevant at these error levels
        const double P = 0.5 * erfc(-z / sqrt(2.0)), p = exp(-z * z / 2) / sqrt(2 * M_PI);
        ea = fmax(ea, fabs(h[2 * i] - z * P) / fmax(1.0, fabs(z)));
        ed = fmax(ed, fabs(h[2 * i + 1] - (P + z * p)));
    }
    printf("%-26s max |a - gelu| / max(1, |z|) = %.2e   max |d - gelu'| = %.2e   (16-bit rounded results)\n", name(V), ea, ed);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 64 * 16 * 2 * 4 + 256 * 512 * 4); (void)hipMalloc(&cyc, 512 * 8);
    accuracy<0>(out); accuracy<1>(out); accuracy<2>(out); accuracy<3>(out); accuracy<4>(out); accuracy<5>(out);
    run<0, 101>(out, cyc); run<1, 101>(out, cyc); run<2, 101>(out, cyc); run<3, 101>(out, cyc); run<4, 101>(out, cyc); run<5, 101>(out, cyc); run<100, 101>(out, cyc);
    run<0, 0>(out, cyc); run<1, 1>(out, cyc); run<2, 2>(out, cyc); run<3, 3>(out, cyc); run<4, 4>(out, cyc); run<5, 5>(out, cyc);
    run<100, 0>(out, cyc); run<100, 1>(out, cyc); run<100, 2>(out, cyc); run<100, 3>(out, cyc); run<100, 4>(out, cyc); run<100, 5>(out, cyc);
    return 0;
}
