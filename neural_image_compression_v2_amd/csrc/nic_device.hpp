// Shared device code for the gfx950 kernels: per-axis sample coordinates, positional encodings, GELU,
// Philox noise and the slot layouts that map decoder-input channels onto MFMA operand registers.
// Reference citations are relative to /root/reference/Projects.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nicv2_hip.h"

namespace nic {

constexpr int kC = 12;   // FEATURE_PYRAMID_CHANNELS the MFMA kernels are built for (var2.py:68)
constexpr int kP = 6;    // PE_CHANNELS (var2.py:69)
constexpr int kH = 64;   // HIDDEN_LAYER_CHANNELS (var2.py:72)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// Sample coordinates along one axis (fp_def.py:116-123, 137-138).  q = range + origin is an integer;
// step = 2^e, so t0 = q*2^e, i0 = floor(t0), t1 = t0/2, i1 = floor(t1), k1 = t1 - i1 are all exact.
// ---------------------------------------------------------------------------------------------------
struct Axis {
    int i0, i1;
    float t1, k1;
};

__device__ __forceinline__ Axis axis_coords(int q, int e) {
    Axis a;
    a.i0 = e < 0 ? (q >> (-e)) : (q << e);
    const int e1 = e - 1;
    a.i1 = e1 < 0 ? (q >> (-e1)) : (q << e1);
    a.t1 = ldexpf((float)q, e1);
    a.k1 = a.t1 - (float)a.i1;
    return a;
}

// products / sums that must not be contracted into FMAs: the reference evaluates
// ((g * wx) * wy) (* wz) and adds the corners left to right in fp32 (fp_def.py:141-144, 176-183;
// image_compression.py:95), and the encode is held to bit-exactness against it.
// (HIP's __fmul_rn / __fadd_rn are plain operators and DO get contracted under the default -ffp-contract=fast;
//  the pragma clears the 'contract' flag on these instructions, which fast-honor-pragmas respects.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// ---------------------------------------------------------------------------------------------------
// Positional encodings
// ---------------------------------------------------------------------------------------------------
// tri(x, o) = 2*|((x - o) mod 2) - 1| - 1 with floor-mod (utils.py:226-227)
__device__ __forceinline__ float tri_wave(float x, float o) {
    const float v = x - o;
    const float m = v - 2.0f * floorf(v * 0.5f);
    return 2.0f * fabsf(m - 1.0f) - 1.0f;
}

// row r (0..P-1) of the triangular PE block of one dimension (utils.py:211-223):
// row P-1 is zero, row P-2 = tri(c, 0), then pairs (tri(c/2^o, 0), tri(c/2^o, .5)) going up.
__device__ __forceinline__ float tri_pe_row(float c, int r, int P) {
    const int u = P - 1 - r;            // u = 2*octave + i, offsets (0.5, 0.0) for i = (0, 1)
    const int octave = u >> 1;
    const float off = (u & 1) ? 0.0f : 0.5f;
    const float v = tri_wave(ldexpf(c, -octave), off);
    return u <= 0 ? 0.0f : v;           // octave 0 / offset 0.5 is skipped -> stays zero
}

// sin and cos by 3-term Cody-Waite reduction to [-pi/4, pi/4] + minimax polynomials (branch-free apart
// from the quadrant select; abs error < 2e-7 for |x| < 2^15, far inside the arguments PE sees: c <= a few
// thousand).  The library sinf/cosf carry a Payne-Hanek slow path that bloats the fused kernel.
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
    const float kf = rintf(x * 0.63661977236758134308f);            // x * 2/pi
    float r = fmaf(kf, -1.57079601287841796875f, x);                 // pi/2 split in 3 parts
    r = fmaf(kf, -3.1391647326017846353e-07f, r);
    r = fmaf(kf, -5.3903025299577647655e-15f, r);
    const float r2 = r * r;
    float sp = fmaf(r2, 2.6083159809786593541503e-06f, -0.0001981069071916863322258f);
    sp = fmaf(sp, r2, 0.00833307858556509017944336f);
    sp = fmaf(sp, r2, -0.166666597127914428710938f);
    const float sr = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, -2.7181184236542321741581e-07f, 2.4799044695100747048855e-05f);
    cp = fmaf(cp, r2, -0.00138887774664908647537231f);
    cp = fmaf(cp, r2, 0.0416666641831398010253906f);
    cp = fmaf(cp, r2, -0.5f);
    const float cr = fmaf(cp, r2, 1.0f);
    const int k = (int)kf;
    const float ss = (k & 1) ? cr : sr;
    const float cc = (k & 1) ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}
// row r of the sinusoidal block: even -> sin(c * div[r/2]), odd -> cos (utils.py:198-208)
__device__ __forceinline__ float sin_pe_row(float c, int r, const float* div) {
    const float a = mul_rn(c, div[r >> 1]);
    float sv, cv;
    sincos_cw(a, sv, cv);
    return (r & 1) ? cv : sv;
}

// ---------------------------------------------------------------------------------------------------
// GELU (exact erf form, nn.GELU() default, image_compression.py:59) and its derivative
// ---------------------------------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26: erf(|u|) = 1 - (a1 t + ... + a5 t^5) exp(-u^2), t = 1 / (1 + p |u|), |error| <= 1.5e-7 (fp32 eps
// is 1.2e-7; measured against fp64 over [-8, 8]: GELU within 4.2e-7 absolute, its derivative within 3.2e-7).  With u = z / sqrt(2)
// the exponential is exp(-z^2 / 2) - the very factor the derivative's pdf term needs - so activation + derivative cost one v_exp,
// one v_rcp and ~14 FMAs, branch-free (the library erff is ~3x that, with a branch).
__device__ __forceinline__ void gelu_and_grad(float z, float& a, float& d) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, az, 1.0f));
    const float ex = __expf(-0.5f * z * z);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * (poly * t) * ex;            // (1 - erf(|u|)) / 2
    const float cdf = z >= 0.f ? 1.0f - half_erfc : half_erfc;    // Phi(z)
    a = z * cdf;
    d = fmaf(z * 0.39894228040143267794f, ex, cdf);
}
// Four evaluations at once, written as two interleaved packed-fp32 chains: a dependent v_pk_fma_f32 -> v_pk_fma_f32 pair costs
// a wait state (s_nop) when issued back to back, the second chain fills those slots.  Same arithmetic as gelu_and_grad except
// that exp(-z^2/2) is taken as exp2(z * (z * (-log2(e)/2))): one multiply fewer.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void gelu_and_grad4(const f32x4_t z, f32x4_t& a, f32x4_t& d) {
    const f32x2 za = {z[0], z[1]}, zb = {z[2], z[3]};
    constexpr float kP = 0.3275911f * 0.70710678118654752440f, kE = -0.5f * 1.44269504088896340736f;
    const f32x2 xa = za * (za * f32x2(kE)), xb = zb * (zb * f32x2(kE));      // log2 of exp(-z^2 / 2)
    f32x2 ta, tb, ea, eb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ta[i] = __builtin_amdgcn_rcpf(fmaf(kP, fabsf(za[i]), 1.0f));
        tb[i] = __builtin_amdgcn_rcpf(fmaf(kP, fabsf(zb[i]), 1.0f));
        ea[i] = __builtin_amdgcn_exp2f(xa[i]);
        eb[i] = __builtin_amdgcn_exp2f(xb[i]);
    }
    // Horner with the coefficients halved: (poly * t) * e = (1 - erf(|u|)) / 2 directly
    f32x2 pa = pk_fma(f32x2(0.5f * 1.061405429f), ta, f32x2(0.5f * -1.453152027f));
    f32x2 pb = pk_fma(f32x2(0.5f * 1.061405429f), tb, f32x2(0.5f * -1.453152027f));
    pa = pk_fma(pa, ta, f32x2(0.5f * 1.421413741f));
    pb = pk_fma(pb, tb, f32x2(0.5f * 1.421413741f));
    pa = pk_fma(pa, ta, f32x2(0.5f * -0.284496736f));
    pb = pk_fma(pb, tb, f32x2(0.5f * -0.284496736f));
    pa = pk_fma(pa, ta, f32x2(0.5f * 0.254829592f));
    pb = pk_fma(pb, tb, f32x2(0.5f * 0.254829592f));
    pa = pa * ta;
    pb = pb * tb;
    // Phi(z) = 1/2 + copysign(1/2 - h, z), h = (1 - erf(|u|)) / 2: no compare / select pair
    const f32x2 ga = pk_fma(-pa, ea, f32x2(0.5f)), gb = pk_fma(-pb, eb, f32x2(0.5f));
    f32x2 ca, cb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ca[i] = 0.5f + __builtin_copysignf(ga[i], za[i]);
        cb[i] = 0.5f + __builtin_copysignf(gb[i], zb[i]);
    }
    const f32x2 aa = za * ca, ab = zb * cb;
    const f32x2 da = pk_fma(za * f32x2(0.39894228040143267794f), ea, ca), db = pk_fma(zb * f32x2(0.39894228040143267794f), eb, cb);
    a = f32x4_t{aa[0], aa[1], ab[0], ab[1]};
    d = f32x4_t{da[0], da[1], db[0], db[1]};
}
// The plain 16-bit product modes (NIC_FLAG_BF16 / NIC_FLAG_FP16: fused_q16.hpp) round every activation and derivative to 8 / 11 significant
// bits, so their GELU does not have to be the 1.5e-7 form above.  Phi(z) ~ 1 / (1 + 2^(z w(z^2))) - the "tanh" form of GELU with the
// coefficients of w refitted against the erf definition (ab/micro/gelu_probe.hip measures both forms against fp64):
//   NC = 3: w = c0 + c1 z^2 + c2 z^4 (z^2 clamped at 64: the quartic term would turn the tail around):  |GELU error| <= 3.7e-5, |derivative error| <= 9.3e-5
//   NC = 2: w = c0 + c1 z^2:                                                                           |GELU error| <= 3.4e-4, |derivative error| <= 6.7e-4
// (half an ulp of bfloat16 at 1 is 2e-3).  The derivative is the exact derivative of the approximant, Phi + z Phi' with
// Phi' = -ln2 w'(z) Phi (1 - Phi) - no second exponential.  A value costs ~ 9 (11) multiply-adds + v_exp + v_rcp instead of ~ 18 + v_exp + v_rcp:
// measured on two waves of one SIMD 43 (53) cycles per value against 61 (ab/micro/gelu_probe.hip; a transcendental is ~ 9, a packed pair 4).
// Restated by oracle/nic_oracle.py::gelu_sigmoid (the precision-emulating oracle evaluates the same form).
template <int NC>
__device__ __forceinline__ void gelu_sig4(const f32x4_t z, f32x4_t& a, f32x4_t& d) {
    constexpr float LN2 = 0.69314718055994530942f;
    constexpr float C0 = NC == 3 ? -2.30087589f : -2.3080629f, C1 = NC == 3 ? -1.06770560e-01f : -0.10091118f, C2 = 1.00279102e-03f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x2 x = {z[2 * i], z[2 * i + 1]};
        f32x2 x2 = x * x, w, q;
        if constexpr (NC == 3) {
            x2[0] = fminf(x2[0], 64.f); x2[1] = fminf(x2[1], 64.f);
            w = pk_fma(pk_fma(f32x2(C2), x2, f32x2(C1)), x2, f32x2(C0));
            q = pk_fma(pk_fma(f32x2(-5.f * LN2 * C2), x2, f32x2(-3.f * LN2 * C1)), x2, f32x2(-LN2 * C0));
        } else {
            w = pk_fma(f32x2(C1), x2, f32x2(C0));
            q = pk_fma(f32x2(-3.f * LN2 * C1), x2, f32x2(-LN2 * C0));
        }
        const f32x2 u = x * w;
        f32x2 e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};   // 2^u = inf -> Phi = 0; 2^u = 0 -> Phi = 1
        e = e + f32x2(1.0f);                                                       // one v_pk_add_f32 (the scalar form compiles to two v_add_f32)
        f32x2 P;
        P[0] = __builtin_amdgcn_rcpf(e[0]);
        P[1] = __builtin_amdgcn_rcpf(e[1]);
        const f32x2 av = x * P;
        const f32x2 dv = pk_fma(av * (f32x2(1.0f) - P), q, P);
        a[2 * i] = av[0]; a[2 * i + 1] = av[1];
        d[2 * i] = dv[0]; d[2 * i + 1] = dv[1];
    }
}
__device__ __forceinline__ float gelu_only(float z) {
    float a, d;
    gelu_and_grad(z, a, d);
    return a;
}
__device__ __forceinline__ float sigmoid_f(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }   // v_rcp_f32: 1 ulp

// ---------------------------------------------------------------------------------------------------
// In-kernel noise: Threefry-4x32-12 (add / rotate / xor only: full-rate VALU; Philox's 32-bit multiplies are quarter-rate on
// CDNA and cost ~4.7K of a 58K-cycle round).  Definition restated in oracle/nic_oracle.py::kernel_noise.
// ---------------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ __forceinline__ U4 threefry4x32_12(U4 c, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
    const uint32_t ks[5] = {k0, k1, k2, k3, 0x1BD11BDAu ^ k0 ^ k1 ^ k2 ^ k3};
    uint32_t x0 = c.x + ks[0], x1 = c.y + ks[1], x2 = c.z + ks[2], x3 = c.w + ks[3];
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        if ((r & 1) == 0) {
            x0 += x1; x1 = rotl32(x1, R[r & 7][0]) ^ x0;
            x2 += x3; x3 = rotl32(x3, R[r & 7][1]) ^ x2;
        } else {
            x0 += x3; x3 = rotl32(x3, R[r & 7][0]) ^ x0;
            x2 += x1; x1 = rotl32(x1, R[r & 7][1]) ^ x2;
        }
        if (((r + 1) & 3) == 0) {
            const int q = (r + 1) >> 2;
            x0 += ks[q % 5]; x1 += ks[(q + 1) % 5]; x2 += ks[(q + 2) % 5]; x3 += ks[(q + 3) % 5] + (uint32_t)q;
        }
    }
    return U4{x0, x1, x2, x3};
}

// ---------------------------------------------------------------------------------------------------
// Counter-based sampler (replaces the host RNG calls of random_crop_dataset, image_compression.py:26-50, when the
// caller opts in).  One generator block per (seed, step, crop): block = threefry4x32_12(ctr = (step_lo, step_hi, crop, "SAMP"),
// key = (seed_lo, seed_hi, "NIC2", 1)).  Crop origin along axis a = (word_a * range) >> 32 (multiply-shift: uniform on [0, range)
// up to range / 2^32).  The LOD of a step comes from the block of crop 0xFFFFFFFF: uniform = (word_1 * (max_mip + 1)) >> 32;
// otherwise floor(-log2(U) / 2) with U = (word_0 + 1/2) / 2^32, which is exactly clz(word_0) >> 1 - the reference's
// P(lod = k) = 3/4 * 4^-k (image_compression.py:32) in integer arithmetic, identical on host and device.
// Restated in oracle/nic_oracle.py::sampler_block / sampler_lod / sampler_origins.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ U4 sampler_block(uint64_t seed, uint64_t step, uint32_t crop) {
    const U4 c{(uint32_t)step, (uint32_t)(step >> 32), crop, 0x53414D50u /* "SAMP" */};
    return threefry4x32_12(c, (uint32_t)seed, (uint32_t)(seed >> 32), 0x4E494332u /* "NIC2" */, 1u);
}
__host__ __device__ __forceinline__ int sampler_origin(const U4& b, int axis, uint32_t range) {
    const uint32_t w = axis == 0 ? b.x : (axis == 1 ? b.y : b.z);
    return (int)(((uint64_t)w * (uint64_t)range) >> 32);
}
__host__ __device__ __forceinline__ int clz32(uint32_t x) {
    int n = 0;
    if (x == 0) return 32;
    while (!(x & 0x80000000u)) { x <<= 1; ++n; }
    return n;
}
__host__ __device__ __forceinline__ int sampler_lod(uint64_t seed, uint64_t step, int uniform, int max_mip) {
    const U4 b = sampler_block(seed, step, 0xFFFFFFFFu);
    const int lod = uniform ? (int)(((uint64_t)b.y * (uint64_t)(max_mip + 1)) >> 32) : (clz32(b.x) >> 1);
    return lod > max_mip ? max_mip : lod;
}

struct NoiseSrc {
    int mode;               // NIC_NOISE_*
    const float* tensor;    // [N, Cin] when mode == TENSOR
    uint32_t k0, k1, off_lo, off_hi;
    float scale;            // 2^-num_bits
};

// One generator block = 16 noise values (4 words x 4 bytes) of one sample: value i of a block is byte i & 3 of word i >> 2,
// u8 -> ((u8 + 1/2) / 256 - 1/2) * 2^-bits (uniform on a 2^-8 lattice of the quantisation step, zero mean).
// Block numbering of a sample (restated in oracle/nic_oracle.py::kernel_noise): the G0 channels come in two groups of
// NG0 = K0/2 * C channels (the corners with dx = 0, then dx = 1 - the two lane halves of the fused kernel), each group
// starting a new block; the remaining channels (G1, PE, LOD) follow in blocks of their own.
__device__ __forceinline__ U4 noise_block(const NoiseSrc& ns, uint64_t sample, int blk) {
    U4 c{(uint32_t)sample, (uint32_t)blk + ((uint32_t)(sample >> 32) << 8), ns.off_lo, ns.off_hi};
    return threefry4x32_12(c, ns.k0, ns.k1, 0x4E494332u /* "NIC2" */, 0u);
}
__device__ __forceinline__ uint32_t block_word(const U4& b, int w) { return w == 0 ? b.x : (w == 1 ? b.y : (w == 2 ? b.z : b.w)); }
__device__ __forceinline__ float noise_from_byte(const NoiseSrc& ns, uint32_t word, int byte) {
    const float u8 = (float)((word >> (8 * byte)) & 0xFFu);                     // v_cvt_f32_ubyteN
    return ((u8 + 0.5f) * (1.0f / 256.0f) - 0.5f) * ns.scale;
}
__device__ __forceinline__ float noise_from_block(const NoiseSrc& ns, const U4& b, int i) {
    return noise_from_byte(ns, block_word(b, (i >> 2) & 3), i & 3);
}
// 2D layouts (Cin = 73): 20 six-bit fields per generator block, five per word (bits 6j .. 6j+5 of word f / 5, j = f % 5):
// u6 -> ((u6 + 1/2) / 64 - 1/2) * 2^-bits (uniform on a 2^-6 lattice of the quantisation step, zero mean).  A sample's 73 channels
// take FOUR blocks, one per G0 corner q: block q holds corner q's 12 channels (fields 0..11), G1 channels 3q..3q+2 (12..14), PE rows
// 3q..3q+2 (15..17) and, for q = 0, the LOD channel (18) - a quarter of the 16-sample kernel's lanes, or one of the two corners of
// a lane half of the 32-sample kernels, runs exactly the blocks it consumes (the byte-per-value numbering needed 6 block
// evaluations per sample, 8 on the 16-sample kernel; the generator is ~ 170 vector instructions per block).
__device__ __forceinline__ float noise_field(const NoiseSrc& ns, const U4& b, int f) {
    const uint32_t w = block_word(b, f / 5);
    // ((u + 0.5) / 64 - 0.5) for the 6-bit field u, without an integer-to-float convert: the field becomes the top six mantissa bits
    // of a float in [1, 2) - m = 1 + u / 64 exactly - and m - 191 / 128 is the same (exactly representable) value: shift, and-or, add
    const int sh = 6 * (f % 5);
    const uint32_t bits = ((sh <= 17 ? w << (17 - sh) : w >> (sh - 17)) & 0x007E0000u) | 0x3F800000u;
    return (__builtin_bit_cast(float, bits) - 1.4921875f) * ns.scale;
}

// ---------------------------------------------------------------------------------------------------
// Grid addressing.  Tensor [C, (Z,) Y, X] (fp_def.py:54,76), per-axis node counts in (x, y, z) order.
// ---------------------------------------------------------------------------------------------------
struct GridView {
    const float* p;
    int nx, ny, nz;      // nodes per axis
    int64_t plane;       // elements per channel = nx*ny*nz
    __device__ __forceinline__ int64_t at(int x, int y, int z) const { return ((int64_t)z * ny + y) * nx + x; }
};

// ---------------------------------------------------------------------------------------------------
// Slot layouts for the MFMA kernels.
//
// A wave works on 32 samples; lane l = (p = l & 31, h = l >> 5) belongs to sample p and owns "slots"
// sigma = 0 .. NSLOT-1 of that sample's padded input vector.  Slot sigma of half h is row
//     rho(sigma, h) = 32*(sigma >> 4) + ROW(sigma & 15, h),   ROW(r, h) = (r & 3) + 8*(r >> 2) + 4*h
// of the internal [KPAD, 32] input matrix - exactly the row a 32x32 MFMA accumulator register r of
// lane-half h holds - so (a) the slot values are the B operand of the first layer with no data
// movement (k-step (sigma, {h=0,1})) and (b) the input gradient comes out of the last backward
// product in the registers of the lane that knows the slot's grid address.  The first NGRID slots are
// the gathered grid features (they need gradients), the rest are PE / LOD / the constant 1 that
// carries the first bias / zero padding.  slot_channel() gives the reference's decoder-input channel
// of a slot (image_compression.py:94-96 channel order), -1 for zero padding, -2 for the constant 1.
// ---------------------------------------------------------------------------------------------------
constexpr int kSlotZero = -1;
constexpr int kSlotOne = -2;

__host__ __device__ constexpr int ROW(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__host__ __device__ constexpr int rho_of(int sigma, int h) { return 32 * (sigma >> 4) + ROW(sigma & 15, h); }

template <int METHOD>
struct Layout;

// 2D (COMPRESSION_METHOD 1): Cin = 4*12 + 12 + 2*6 + 1 = 73
struct Layout2D {
    static constexpr int DIM = 2, K0 = 4, K1 = 4, CIN = 73;
    static constexpr int NSLOT = 40;   // 16 + 16 + 8  -> KPAD = 80
    static constexpr int NGRID = 30;   // 24 G0 (2 corners x 12 ch) + 6 G1 channels per half
    static constexpr int TX = 16, TY = 2, TZ = 1;   // wave block of 32 G0 cells, lanes x-fastest (x = the grids' contiguous axis: 2 rows per gather / atomic)
    __host__ __device__ static constexpr int slot_channel(int s, int h) {
        if (s < 24) return 24 * h + s;                  // G0 corner 2h + s/12, channel s%12
        if (s < 30) return 48 + 6 * h + (s - 24);       // G1 channel 6h + ..
        if (s < 32) return kSlotZero;
        if (s < 38) return 60 + 6 * h + (s - 32);       // PE of dimension h
        if (s == 38) return h == 0 ? 72 : kSlotOne;     // LOD | bias carrier
        return kSlotZero;
    }
};
template <>
struct Layout<1> : Layout2D {                           // TF_USE_TRI_PE = True (var2.py:81)
    static constexpr int PE = NIC_PE_TRIANGULAR;
};
template <>
struct Layout<2> : Layout2D {                           // TF_USE_TRI_PE = False
    static constexpr int PE = NIC_PE_SINUSOIDAL;
};
// 3D method 3: Cin = 8*12 + 12 + 3*6 + 1 = 127
template <>
struct Layout<3> {
    static constexpr int DIM = 3, K0 = 8, K1 = 8, CIN = 127;
    static constexpr int PE = NIC_PE_TRIANGULAR;        // fp_def.py:169
    static constexpr int NSLOT = 64;   // KPAD = 128
    static constexpr int NGRID = 54;   // 48 G0 (4 corners) + 6 G1
    static constexpr int TX = 16, TY = 2, TZ = 1;   // x-fastest like 2D (the 2 x 4 x 4 block touched 16 grid rows per gather: -10 %)
    __host__ __device__ static constexpr int slot_channel(int s, int h) {
        if (s < 48) return 48 * h + s;                  // G0 corner 4h + s/12
        if (s < 54) return 96 + 6 * h + (s - 48);
        if (s < 63) return 108 + 9 * h + (s - 54);      // PE rows 9h .. 9h+8
        return h == 0 ? 126 : kSlotOne;
    }
};
// 3D method 4 (tetrahedral G0): Cin = 4*12 + 12 + 18 + 1 = 79
template <>
struct Layout<4> {
    static constexpr int DIM = 3, K0 = 4, K1 = 8, CIN = 79;
    static constexpr int PE = NIC_PE_SINUSOIDAL;        // fp_def.py:208
    static constexpr int NSLOT = 48;   // 16 + 16 + 16 -> KPAD = 96 (42 real slots; a multiple of 8 so that split-bf16 k-steps cover whole slots)
    static constexpr int NGRID = 30;
    static constexpr int TX = 16, TY = 2, TZ = 1;   // x-fastest like 2D (the 2 x 4 x 4 block touched 16 grid rows per gather: -10 %)
    __host__ __device__ static constexpr int slot_channel(int s, int h) {
        if (s < 24) return 24 * h + s;                  // G0 corner 2h + s/12
        if (s < 30) return 48 + 6 * h + (s - 24);
        if (s < 32) return kSlotZero;
        if (s < 41) return 60 + 9 * h + (s - 32);
        if (s == 41) return h == 0 ? 78 : kSlotOne;
        return kSlotZero;
    }
};

// inverse map: internal row rho -> decoder-input channel (or kSlotZero / kSlotOne)
template <class L>
__host__ __device__ constexpr int channel_of_rho(int rho) {
    const int t = rho >> 5, rr = rho & 31;
    const int h = (rr >> 2) & 1;
    const int reg = (rr & 3) + 4 * (rr >> 3);
    const int sigma = 16 * t + reg;
    return sigma < L::NSLOT ? L::slot_channel(sigma, h) : kSlotZero;
}

// corner offsets (dx, dy, dz) of the reference's corner numbering
// 2D  (fp_def.py:82-85):   0:(0,0) 1:(0,1) 2:(1,0) 3:(1,1)            -> dx = q>>1, dy = q&1
// 3D  (fp_def.py:96-103):  q bit0 = dz, bit1 = dy, bit2 = dx
// tetra (fp_def.py:108-111): 0:(0,0,0) 1:(0,1,1) 2:(1,0,1) 3:(1,1,0)  -> dx = q>>1, dy = q&1, dz = dx^dy
// Q1 (fp_def.py:176-183): factor bits (bx,by,bz) the reference multiplies G1 corner q with (1 -> k, 0 -> 1-k):
//   q: 0:(0,0,0) 1:(0,0,1) 2:(0,1,0) 3:(1,0,0) 4:(1,1,0) 5:(1,0,1) 6:(0,1,1) 7:(1,1,1)
__device__ __forceinline__ int g1_ref_weight_bits(int q) {
    // packed as bx | by<<1 | bz<<2, indexed by q
    constexpr uint32_t tbl = (0u << 0) | (4u << 3) | (2u << 6) | (1u << 9) | (3u << 12) | (5u << 15) | (6u << 18) | (7u << 21);
    return (tbl >> (3 * q)) & 7;
}

}  // namespace nic
