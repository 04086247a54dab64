// layout check for the split-bf16 weight-gradient products (diagnostic): images are [sample][feature] bf16; all operands come from
// ds_read_b64_tr_b16.  (1) 32x32x16: G[o][k] = sum_s DZ[s][o] A1[s][k] (32 samples = 2 k-steps); (2) 16x16x32: P[o][r] = sum_s
// DZ[s][o] X[s][r] for 16 columns r; (3) 4x4x4: W3g[c][k] = sum_s d3[c][s] A1[s][k] for 64 columns k, c < 4; exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int LD = 72;
__device__ s16x4 tr(const __bf16* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p); }
__device__ bf16x8 join(s16x4 a, s16x4 b) { const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; return __builtin_bit_cast(bf16x8, v); }
__global__ void __launch_bounds__(64) k(const float* DZg, const float* A1g, const float* d3g, float* G, float* P, float* W3g) {
    __shared__ __attribute__((aligned(16))) __bf16 DZ[32 * LD], A1[32 * LD], D3[4 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 32 * 64; i += 64) { DZ[(i / 64) * LD + i % 64] = (__bf16)DZg[i]; A1[(i / 64) * LD + i % 64] = (__bf16)A1g[i]; }
    for (int i = lane; i < 128; i += 64) D3[i] = (__bf16)d3g[i];
    __syncthreads();
    const int q = (lane & 15) >> 2, pp = lane & 3, cg = (lane >> 4) & 1, g = lane >> 5, n = lane & 31;
    // (1) 32x32x16, tile (to = 1, tk = 0): rows o = 32 + n, cols k = n
    {
        f32x16 acc = f32x16(0.f);
        for (int hs = 0; hs < 2; ++hs) {
            const int s0 = 16 * hs + 4 * g + q;
            const bf16x8 a = join(tr(DZ + s0 * LD + 32 + 16 * cg + 4 * pp), tr(DZ + (s0 + 8) * LD + 32 + 16 * cg + 4 * pp));
            const bf16x8 b = join(tr(A1 + s0 * LD + 0 + 16 * cg + 4 * pp), tr(A1 + (s0 + 8) * LD + 0 + 16 * cg + 4 * pp));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) G[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + n] = acc[r];
    }
    // (2) 16x16x32: rows o = 16 ot + (lane & 15), cols r = 48 + (lane & 15) of A1 (stands for the X image), all 32 samples in one MFMA
    for (int ot = 0; ot < 4; ++ot) {
        const int kg = lane >> 4;                       // samples 8 kg .. 8 kg + 7
        const bf16x8 a = join(tr(DZ + (8 * kg + q) * LD + 16 * ot + 4 * pp), tr(DZ + (8 * kg + 4 + q) * LD + 16 * ot + 4 * pp));
        const bf16x8 b = join(tr(A1 + (8 * kg + q) * LD + 48 + 4 * pp), tr(A1 + (8 * kg + 4 + q) * LD + 48 + 4 * pp));
        f32x4 acc = f32x4(0.f);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) P[(16 * ot + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
    }
    // (3) 4x4x4 (16 blocks): lane l <-> column k = l; A = d3[c = l & 3][4 samples], the same for every block
    {
        f32x4 acc = f32x4(0.f);
        for (int s0 = 0; s0 < 32; s0 += 4) {
            const s16x4 b = tr(A1 + (s0 + q) * LD + 16 * (lane >> 4) + 4 * pp);             // column l, samples s0 .. s0+3
            const s16x4 a = *reinterpret_cast<const s16x4*>(D3 + (lane & 3) * 32 + s0);
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) W3g[r * 64 + lane] = acc[r];
    }
}
int main() {
    std::vector<float> DZ(32 * 64), A1(32 * 64), d3(4 * 32), G(32 * 32), P(64 * 16), W3(4 * 64);
    srand(2);
    for (auto& v : DZ) v = (float)(rand() % 9 - 4);
    for (auto& v : A1) v = (float)(rand() % 9 - 4);
    for (auto& v : d3) v = (float)(rand() % 9 - 4);
    float *a, *b, *c, *g, *p, *w;
    hipMalloc(&a, DZ.size() * 4); hipMalloc(&b, A1.size() * 4); hipMalloc(&c, d3.size() * 4); hipMalloc(&g, G.size() * 4); hipMalloc(&p, P.size() * 4); hipMalloc(&w, W3.size() * 4);
    hipMemcpy(a, DZ.data(), DZ.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b, A1.data(), A1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, d3.data(), d3.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c, g, p, w);
    hipMemcpy(G.data(), g, G.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(P.data(), p, P.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(W3.data(), w, W3.size() * 4, hipMemcpyDeviceToHost);
    int b1 = 0, b2 = 0, b3 = 0;
    for (int o = 0; o < 32; ++o) for (int kk = 0; kk < 32; ++kk) { float r = 0; for (int s = 0; s < 32; ++s) r += DZ[s * 64 + 32 + o] * A1[s * 64 + kk]; b1 += r != G[o * 32 + kk]; }
    for (int o = 0; o < 64; ++o) for (int r_ = 0; r_ < 16; ++r_) { float r = 0; for (int s = 0; s < 32; ++s) r += DZ[s * 64 + o] * A1[s * 64 + 48 + r_]; b2 += r != P[o * 16 + r_]; }
    for (int cc = 0; cc < 4; ++cc) for (int kk = 0; kk < 64; ++kk) { float r = 0; for (int s = 0; s < 32; ++s) r += d3[cc * 32 + s] * A1[s * 64 + kk]; b3 += r != W3[cc * 64 + kk]; }
    printf("32x32x16 dW: %d mismatches; 16x16x32 partial: %d; 4x4x4 dW3: %d\n", b1, b2, b3);
    return b1 + b2 + b3 ? 1 : 0;
}
