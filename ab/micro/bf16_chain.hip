// layout check for the split-bf16 plan (diagnostic): (1) accumulator-as-B-operand chaining of v_mfma_f32_32x32x16_bf16 with the A
// operand read row-wise from a natural [o][k] bf16 LDS image (two 8-byte reads per fragment), (2) the transposed product with
// the A operand read by ds_read_b64_tr_b16 from the SAME image.  Exact small-integer data: any layout slip shows as a mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__host__ __device__ constexpr int ROWC(int r) { return (r & 3) + 8 * (r >> 2); }
constexpr int LDW = 64 + 8;      // bf16 elements per image row (144 B)

__device__ bf16x8 frag_from_acc(const f32x16& x, int s) {      // registers 8s .. 8s+7 -> 8 bf16 (k-step s)
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 p = {x[8 * s + j], x[8 * s + j + 1]};
        const bf16x2 q = __builtin_convertvector(p, bf16x2);
        f[j] = q[0]; f[j + 1] = q[1];
    }
    return f;
}

// W: [64 o][64 k] natural;  X: [64 k][32 samples];  Z = W X  (rows o);  Y = W^T Z' with Z' := X (rows k) -> Y[k][s] = sum_o W[o][k] X[o][s]
__global__ void __launch_bounds__(64) k(const float* Wg, const float* Xg, float* Zg, float* Yg) {
    __shared__ __attribute__((aligned(16))) __bf16 Ws[64 * LDW];
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    for (int i = lane; i < 64 * 64; i += 64) Ws[(i / 64) * LDW + (i % 64)] = (__bf16)Wg[i];
    __syncthreads();
    f32x16 xt[2];
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 16; ++r) xt[t][r] = Xg[(32 * t + ROWC(r) + 4 * h) * 32 + n];
    // (1) Z = W X: output row tiles to = 0, 1
    for (int to = 0; to < 2; ++to) {
        f32x16 acc = f32x16(0.f);
        for (int t = 0; t < 2; ++t)
            for (int s = 0; s < 2; ++s) {
                const __bf16* row = Ws + (32 * to + n) * LDW + 32 * t + 16 * s + 4 * h;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row), hi = *reinterpret_cast<const bf16x4*>(row + 8);
                bf16x8 a;
                for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_from_acc(xt[t], s), acc, 0, 0, 0);
            }
        for (int r = 0; r < 16; ++r) Zg[(32 * to + ROWC(r) + 4 * h) * 32 + n] = acc[r];
    }
    // (2) Y = W^T X: output row tiles tk = 0, 1 (rows k), contraction over o = the rows of xt
    for (int tk = 0; tk < 2; ++tk) {
        f32x16 acc = f32x16(0.f);
        const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3, cg = (lane >> 4) & 1;
        for (int t = 0; t < 2; ++t)
            for (int s = 0; s < 2; ++s) {
                const int o0 = 32 * t + 16 * s + 4 * h;
                const __bf16* a0 = Ws + (o0 + q) * LDW + 32 * tk + 16 * cg + 4 * p;
                const __bf16* a1 = Ws + (o0 + 8 + q) * LDW + 32 * tk + 16 * cg + 4 * p;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
                bf16x8 a;
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                a = __builtin_bit_cast(bf16x8, av);
                if (Xg[2048] != 0.f)      // plain element loads instead (debug switch)
                    for (int j = 0; j < 8; ++j) a[j] = Ws[(o0 + (j & 3) + 8 * (j >> 2)) * LDW + 32 * tk + n];
                int bad = 0;
                for (int j = 0; j < 8; ++j) bad += (float)a[j] != (float)Ws[(o0 + (j & 3) + 8 * (j >> 2)) * LDW + 32 * tk + n];
                if (t == 0 && s == 0 && tk == 0 && (lane < 6 || (lane >= 16 && lane < 19) || (lane >= 32 && lane < 35)))
                    printf("lane %2d got %4.0f %4.0f %4.0f %4.0f | %4.0f %4.0f %4.0f %4.0f   want %4.0f %4.0f %4.0f %4.0f | %4.0f %4.0f %4.0f %4.0f\n", lane,
                           (float)a[0], (float)a[1], (float)a[2], (float)a[3], (float)a[4], (float)a[5], (float)a[6], (float)a[7],
                           (float)Ws[(o0 + 0) * LDW + n], (float)Ws[(o0 + 1) * LDW + n], (float)Ws[(o0 + 2) * LDW + n], (float)Ws[(o0 + 3) * LDW + n],
                           (float)Ws[(o0 + 8) * LDW + n], (float)Ws[(o0 + 9) * LDW + n], (float)Ws[(o0 + 10) * LDW + n], (float)Ws[(o0 + 11) * LDW + n]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag_from_acc(xt[t], s), acc, 0, 0, 0);
            }
        for (int r = 0; r < 16; ++r) Yg[(32 * tk + ROWC(r) + 4 * h) * 32 + n] = acc[r];
    }
}

int main() {
    std::vector<float> W(64 * 64), X(64 * 32 + 1), Z(64 * 32), Y(64 * 32);
    srand(1);
    for (int i = 0; i < 64 * 64; ++i) W[i] = (float)(((i / 64) * 3 + (i % 64)) % 200);   // W[o][k] = (3 o + k) mod 200: exact in bf16
    for (auto& v : X) v = (float)(rand() % 15 - 7);
    X[2048] = getenv("PLAIN") ? 1.f : 0.f;
    float *dW, *dX, *dZ, *dY;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dZ, Z.size() * 4); hipMalloc(&dY, Y.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dW, dX, dZ, dY);
    hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    int badz = 0, bady = 0;
    for (int o = 0; o < 64; ++o)
        for (int s = 0; s < 32; ++s) {
            float z = 0, y = 0;
            for (int kk = 0; kk < 64; ++kk) { z += W[o * 64 + kk] * X[kk * 32 + s]; y += W[kk * 64 + o] * X[kk * 32 + s]; }
            badz += z != Z[o * 32 + s];
            bady += y != Y[o * 32 + s];
        }
    printf("row-read chain: %d mismatches; transposed-read chain: %d mismatches (of %d)\n", badz, bady, 64 * 32);
    return badz + bady ? 1 : 0;
}
