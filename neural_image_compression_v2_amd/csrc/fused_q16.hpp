// fused_q16_kernel<Q, MODE, NL>: the fused training step in PLAIN bf16 products (NIC_FLAG_BF16) - one bf16 MFMA per product, fp32
// accumulation - for every layout (2D, 3D method 3, 3D method 4) and decoder depth (NL = 3: the reference's ColorDecoder,
// image_compression.py:57-64; NL = 5: the "4 x 64" decoder of BASELINE.json's north star), 8 waves x 16 samples = two waves per SIMD.
//
// What "plain bf16" rounds (restated in oracle/nic_oracle.py::mlp_forward_backward_bf16, the precision-emulating oracle):
//   * every matrix-product operand is ONE bf16 value: the weights, the (noisy) input slots, the GELU outputs, the dZ of every layer;
//     accumulation, biases (added in fp32, not carried on the constant-one slot), GELU, sigmoid, loss, db_out and the grid-gradient sums are fp32;
//   * the GELU derivatives wait for the backward pass as bf16 (packed pairs: 8 registers per hidden layer instead of 16);
//   * the encode (gathers, G1 blend, PE) and its backward are the fp32 arithmetic of every other kernel.
// Against the split-bf16 kernels (fused_train16.hpp, fused_mlpn.hpp): a third of the matrix instructions, no hi / lo operand split
// (~ 500 vector slots per round), half the LDS image traffic, and registers to spare - which is what lets the 5-layer decoder and
// the 3D layouts run two waves per SIMD.
//
// Quarter layouts (QL<method>): lane l = 16 g + n is quarter g of sample n (a 16x16 accumulator tile gives it rows 4g .. 4g+3).  A quarter
// owns NS input slots: G0 corner(s) - 12 channels each, ONE gather / atomic address per corner -, G1 channels 3g .. 3g+2 (blended over
// the 4 / 8 corners), and a share of the positional-encoding rows, the LOD and the constant one.  Slots 8s .. 8s+7 of the four quarters
// are the B operand of k-step s as they stand; a last half k-step (2D, method 4: slots 16 .. 19) is stored compactly.
//   2D (Cin 73)        20 slots: G0 corner g | G1 3 | zero | PE rows 3g .. 3g+2 | LOD / one / zero            (fused_train16.hpp's layout)
//   3D method 4 (79)   20 slots: tetrahedral corner g | G1 3 | PE rows 4g .. 4g+3 | PE row 16 / 17 / LOD / one
//   3D method 3 (127)  32 slots: corners 2g, 2g+1 | G1 3 | PE rows 4g .. 4g+3 | PE row 16 / 17 / LOD / one
// Hidden activations, weight images, [sample][position] wave images, transposed reads: the rules of fused_train16.hpp.
//
// LDS: bf16 weight images (W1 in slot order with a ZERO column at the constant-one slot - db1 still rides on it through the dW1
// product), fp32 biases, and per wave: X | A_0 .. A_{NL-3} (the hidden activations, stored by the forward pass as the B fragments they
// are, read again as the weight-gradient images) | DZ (two buffers where the LDS allows) | dZ_out.
// Weight gradients: the backward pass is a sequence of PHASES - hidden layer NL-3, .., hidden layer 0, layer 1 - and the four 32x32 tiles
// of phase j belong to the four waves of HALF j & 1 of the workgroup, each contracted over the samples of all eight waves (8 MFMAs
// 32x32x16).  A wave therefore carries (NL - 1) / 2 accumulator tiles instead of NL - 1 (the 5-layer decoder fits two waves per SIMD),
// and while the owners of a phase read the images the other half - their SIMD partners: waves w and w + 4 share a SIMD - is already in
// the vector work of the next layer, storing its dZ into the OTHER DZ buffer: one barrier per phase plus one per round.
#pragma once
#include "fused_train16.hpp"

namespace nic {

// =====================================================================================================
// quarter layouts
// =====================================================================================================
// C = FEATURE_PYRAMID_CHANNELS (var2.py:68; a multiple of 4: a quarter blends C / 4 G1 channels), P = PE_CHANNELS (var2.py:69; even).  The
// reference's defaults are 12 and 6; the other widths exist for the plain-bf16 kernels only.
template <int METHOD, int C = 12, int P = 6>
struct QL;

__host__ __device__ constexpr int align4(int v) { return (v + 3) & ~3; }

// 2D (Cin = 5 C + 2 P + 1): G0 corner g [C] | G1 channels GQ g .. [GQ = C / 4] | zero padding to a multiple of 4 | PE rows (P / 2) g .. of
// dimension g >> 1 [P / 2] | LOD (g = 0) / the constant one (g = 1) | zero padding.  C = 12, P = 6: the 20 slots of fused_train16.hpp.
template <int C_, int P_, int PE_>
struct QL2D {
    static constexpr int DIM = 2, C = C_, P = P_, PE = PE_, GQ = C / 4, PH = P / 2, LEVELS = 1;
    static constexpr int CIN = 5 * C + 2 * P + 1, K0 = 4, K1 = 4, NG0 = 1;
    static constexpr bool TETRA = false;
    static constexpr int NGS = C + GQ, T0 = align4(NGS), NS = align4(T0 + PH + 1);
    static_assert(C % 4 == 0 && P % 2 == 0 && C >= 4 && P >= 2, "channel counts");
    __host__ __device__ static constexpr int slot_channel(int s, int g) {
        if (s < C) return C * g + s;
        if (s < NGS) return 4 * C + GQ * g + (s - C);
        if (s < T0) return kSlotZero;
        if (s < T0 + PH) return 5 * C + PH * g + (s - T0);
        if (s == T0 + PH) return g == 0 ? CIN - 1 : (g == 1 ? kSlotOne : kSlotZero);
        return kSlotZero;
    }
};
template <int C, int P>
struct QL<1, C, P> : QL2D<C, P, NIC_PE_TRIANGULAR> {};
template <int C, int P>
struct QL<2, C, P> : QL2D<C, P, NIC_PE_SINUSOIDAL> {};
// 3D: the 20 tail values (18 PE rows, LOD, the constant one) are dealt out five per quarter: PE rows 4g .. 4g+3, then row 16 / row 17 / LOD / one
__host__ __device__ constexpr int tail3d_channel(int j, int g, int pe0) {      // pe0: first PE channel; LOD = pe0 + 18
    if (j < 4) return pe0 + 4 * g + j;
    return g == 0 ? pe0 + 16 : (g == 1 ? pe0 + 17 : (g == 2 ? pe0 + 18 : kSlotOne));
}
// tetrahedral G0 (fp_def.py:107-112, 187-223): Cin = 5 C + 19: corner g [C] | G1 [GQ] | tail [5] | zero padding
template <int C_, int P_>
struct QL<4, C_, P_> {
    static constexpr int DIM = 3, C = C_, P = P_, GQ = C / 4, LEVELS = 1;
    static constexpr int CIN = 5 * C + 19, K0 = 4, K1 = 8, NG0 = 1;
    static constexpr bool TETRA = true;
    static constexpr int PE = NIC_PE_SINUSOIDAL;                                // fp_def.py:208
    static constexpr int NGS = C + GQ, T0 = NGS, NS = align4(NGS + 5);
    static_assert(C % 4 == 0 && P == 6, "3D: PE_CHANNELS 6");
    __host__ __device__ static constexpr int slot_channel(int s, int g) {
        if (s < C) return C * g + s;
        if (s < NGS) return 4 * C + GQ * g + (s - C);
        if (s < NGS + 5) return tail3d_channel(s - NGS, g, 5 * C);
        return kSlotZero;
    }
};
// 8 raw G0 corners (fp_def.py:89-104, 148-184): Cin = 9 C + 19: corners 2g, 2g + 1 [2 C] (dx = g >> 1, dy = g & 1, dz = s / C) | G1 [GQ] | tail [5] | zero padding
template <int C_, int P_>
struct QL<3, C_, P_> {
    static constexpr int DIM = 3, C = C_, P = P_, GQ = C / 4, LEVELS = 1;
    static constexpr int CIN = 9 * C + 19, K0 = 8, K1 = 8, NG0 = 2;
    static constexpr bool TETRA = false;
    static constexpr int PE = NIC_PE_TRIANGULAR;                                // fp_def.py:169
    static constexpr int NGS = 2 * C + GQ, T0 = NGS, NS = align4(NGS + 5);
    static_assert(C % 4 == 0 && P == 6, "3D: PE_CHANNELS 6");
    __host__ __device__ static constexpr int slot_channel(int s, int g) {
        if (s < 2 * C) return 2 * C * g + s;
        if (s < NGS) return 8 * C + GQ * g + (s - 2 * C);
        if (s < NGS + 5) return tail3d_channel(s - NGS, g, 9 * C);
        return kSlotZero;
    }
};

// MULTI-LEVEL 2D (multilevel.MultiLevelField; BASELINE config 2's "16-level grid" as the extension it is: the reference reads ONE level pair per sample,
// fp_def.py:24-34): a sample reads the first LV level pairs at once and the decoder sees their encodings concatenated,
//     x = [enc_0 | .. | enc_{LV-1} | lod],  enc_l = [G0_l corners (4 C) | blended G1_l (C) | PE_l (2 P)]  at step_number 4^-(l+1) (image_compression.py:79-100)
// i.e. Cin = LV (5 C + 2 P) + 1.  A quarter's slots: its corner g of EVERY pair [LV x C] | its G1 channels of every pair [LV x GQ] | zero padding to a multiple
// of 4 | its PE rows of every pair [LV x PH] | LOD (g = 0) / the constant one (g = 1) | zero padding: the grid slots still come first (the dX tiles), and LV = 1
// is QL2D's layout.  All samples of a lane's macro-tile share their cell in EVERY pair (cells nest: 4^(l+1) pixels, absolute alignment).
template <int LV, int C_, int P_, int PE_>
struct QML {
    static constexpr int DIM = 2, C = C_, P = P_, PE = PE_, GQ = C / 4, PH = P / 2, LEVELS = LV;
    static constexpr int PER = 5 * C + 2 * P;                                   // channels of one pair's encoding
    static constexpr int CIN = LV * PER + 1, K0 = 4, K1 = 4, NG0 = LV;          // NG0: gather / atomic addresses of G0 per lane (one per pair)
    static constexpr bool TETRA = false;
    static constexpr int NGS = LV * (C + GQ), T0 = align4(NGS), NS = align4(T0 + LV * PH + 1);
    static_assert(C % 4 == 0 && P % 2 == 0 && C >= 4 && P >= 2 && LV >= 2 && LV <= NIC_ML_MAX_LEVELS, "multi-level layout");
    __host__ __device__ static constexpr int slot_channel(int s, int g) {
        if (s < LV * C) return (s / C) * PER + C * g + s % C;
        if (s < NGS) return ((s - LV * C) / GQ) * PER + 4 * C + GQ * g + (s - LV * C) % GQ;
        if (s < T0) return kSlotZero;
        if (s < T0 + LV * PH) return ((s - T0) / PH) * PER + 5 * C + PH * g + (s - T0) % PH;
        if (s == T0 + LV * PH) return g == 0 ? CIN - 1 : (g == 1 ? kSlotOne : kSlotZero);
        return kSlotZero;
    }
};

template <class Q>
struct QInfo {
    static constexpr int NG0V = Q::C * Q::NG0;                 // raw G0 values / G0 gradient sums per lane
    static constexpr int NG1V = Q::GQ * Q::K1 * Q::LEVELS;     // raw G1 values / G1 gradient sums per lane
    static constexpr int NGS = Q::NGS;                         // grid slots of a quarter (they come first)
    static constexpr int KF = Q::NS / 8;                       // full k-steps of layer 1 (32 columns each)
    static constexpr bool HALF = (Q::NS % 8) != 0;             // + a compact half k-step (4 slots per quarter, 16 columns)
    static constexpr int KP = 4 * Q::NS;                       // columns of the X / W1 images
    static constexpr int NDX = (NGS + 3) / 4;                  // dX tiles (4 slots each) that hold grid slots
    static constexpr int NT1 = 2 * KF;                         // 32x32 tiles of dW1's full k-steps: tile to * KF + tk (row half to, column block tk)
    static constexpr int LD1 = ((KP - 16 + 63) / 64) * 64 + 16;   // W1 image row stride = 16 mod 64 elements (80, 144, 208, 272): 160- / 288- / .. byte rows spread b128 row reads over the banks (ab/w16/banks.py)
    static constexpr int LDX = KP + 8;
    static constexpr int T1W = (NT1 + 7) / 8;                  // dW1 tiles per wave when every wave owns tiles (NT1 > 4)
    static_assert(Q::NS % 4 == 0 && NT1 >= 2 && NT1 <= 16 && KP <= 256 && (KP <= 128 ? LD1 == (KP <= 80 ? 80 : 144) : true), "layout");
    // in-kernel noise (oracle/nic_oracle.py::kernel_noise_quarter): the real slots of a quarter, in slot order, take six-bit fields 0, 1, 2, ..;
    // 20 fields per generator block, quarter g consumes blocks NBLK g ..
    __host__ __device__ static constexpr int noise_index(int s) { return s < NGS ? s : (s >= Q::T0 && Q::slot_channel(s, 0) != kSlotZero ? NGS + (s - Q::T0) : -1); }
    __host__ __device__ static constexpr int n_real() {
        int n = 0;
        for (int s = 0; s < Q::NS; ++s)
            if (noise_index(s) >= 0) n = noise_index(s) + 1;
        return n;
    }
    static constexpr int NBLK = (n_real() + 19) / 20;
    // compact input column of (slot, quarter) and its inverse
    __host__ __device__ static constexpr int rho(int s, int g) { return s < 8 * KF ? 32 * (s >> 3) + 8 * g + (s & 7) : 32 * KF + 4 * g + (s - 8 * KF); }
    __host__ __device__ static constexpr int channel_of_rho(int r) {
        if (r < 32 * KF) return Q::slot_channel(8 * (r >> 5) + (r & 7), (r >> 3) & 3);
        return Q::slot_channel(8 * KF + (r & 3), (r - 32 * KF) >> 2);
    }
    __host__ __device__ static constexpr int rho_of_channel(int ch) {
        for (int r = 0; r < KP; ++r)
            if (channel_of_rho(r) == ch) return r;
        return -1;
    }
};

template <class Q, int NL>
struct LdsQ {
    using I = QInfo<Q>;
    static constexpr int NH = NL - 2;                          // hidden 64 x 64 layers
    static constexpr int LD1 = I::LD1, LDH = 80, LDZ = 72, LDX = I::LDX;
    static constexpr int OFF_W1 = 0;                           // [64][LD1]
    static constexpr int OFF_WH = OFF_W1 + kH * LD1;           // NH x [64][LDH], columns in position order
    static constexpr int WSZ = kH * LDH;
    static constexpr int OFF_WO = OFF_WH + NH * WSZ;           // [4][LDH], row 3 = zeros
    static constexpr int OFF_B = OFF_WO + 4 * LDH;             // fp32: biases of layer 1 and the hidden layers [NH + 1][64] (natural order), output bias [4]
    static constexpr int NBIAS = (NH + 1) * kH + 4;
    static constexpr int OFF_IMG = (OFF_B + 2 * NBIAS + 7) & ~7;
    // per wave (bf16 elements): X [16][LDX] | A_k [16][LDZ] x NH | DZ [16][LDZ] x DZB | dZ_out [4][16]
    static constexpr int OFF_A = 16 * LDX, ASZ = 16 * LDZ, OFF_DZ = OFF_A + NH * ASZ;
    static constexpr int SCRATCH = Q::LEVELS > 1 ? 0 : 2 * 64 * (I::NG0V + I::NG1V + 1);     // the head of the wave region doubles as fp32 scratch of the flush (group sums, pre-add; not multi-level)
    static constexpr int spw_of(int dzb) { return (OFF_DZ + dzb * ASZ > SCRATCH ? OFF_DZ + dzb * ASZ : SCRATCH) + 64; }
    static constexpr int DZB = (OFF_IMG + 8 * spw_of(2)) * 2 <= 163840 ? 2 : 1;      // two DZ buffers where they fit (not: method 3 with 5 layers)
    static constexpr int OFF_D3 = spw_of(DZB) - 64;                                  // behind the scratch: its fourth row stays zero
    static constexpr int SPW = OFF_D3 + 64;
    static constexpr int TOTAL = OFF_IMG + 8 * SPW;
    // phases of the backward pass: j = 0 .. NH - 1 = hidden layer NH - 1 - j, j = NH = layer 1.  Phase j belongs to half j & 1, accumulator slot j >> 1;
    // with more than 4 dW1 tiles (6 or 8: wide input layouts) wave w owns tile w: the half that does not own phase NH keeps it in an extra slot
    // multi-level layouts (up to 16 dW1 tiles): wave w owns tiles w, w + 8 in slots of their own behind the hidden layers' (D1SLOT ..)
    static constexpr bool ML = Q::LEVELS > 1;
    static constexpr int D1SLOT = (NH + 1) / 2;
    static constexpr int NACC = ML ? D1SLOT + I::T1W : (NH + 2) / 2 + (I::NT1 > 4 ? 1 : 0), XSLOT = (NH + 2) / 2;
    static constexpr int TAIL_HALF = (NH + 1) & 1;                                   // HALF: the 16-column tail of dW1 goes to the half that idles in phase NH
    __host__ __device__ static constexpr int dz_buf(int store) { return DZB == 2 ? (store & 1) : 0; }   // store 0: a_NH (forward), store 1 + j: the dZ of phase j
    static_assert(TOTAL * 2 <= 163840, "LDS");
    static_assert(OFF_WH % 8 == 0 && OFF_WO % 8 == 0 && OFF_B % 8 == 0 && OFF_A % 8 == 0 && ASZ % 8 == 0 && OFF_D3 % 8 == 0 && SPW % 8 == 0, "16-byte alignment");
    // per-workgroup record of decoder-gradient sums (floats)
    static constexpr int REC_W = 0;                            // 8 waves x NACC slots x 1024
    static constexpr int REC_TAIL = 8 * NACC * 1024;           // 8 x 256 (HALF; the waves of TAIL_HALF write theirs): rows 16 (w & 3)..
    static constexpr int REC_WAVE = REC_TAIL + (I::HALF ? 8 * 256 : 0);
    static constexpr int WTAIL = NH * 64 + 192 + 4;            // db_hidden [NH][64] (position order), dW_out [3][64], db_out [3], loss
    static constexpr int REC = REC_WAVE + 8 * WTAIL;
};

// ---- the 16-bit operand type of the products: bfloat16 (NIC_FLAG_BF16) or IEEE half (NIC_FLAG_FP16: the reference's own 16-bit type, utils.py:301-313, and
// BASELINE config 3's "fp16"; v_mfma_f32_*_f16 run at the bf16 rate with 3 more mantissa bits).  Fragments are carried as bf16x8 bit containers either way.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// two fp32 values -> a 16-bit pair, round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).  (Measured and dropped: the bf16 rounding in integer
// arithmetic - bits + 0x8000 per value, one v_perm_b32 per pair - is 3 instructions for 1 and cost the 5-layer 4K launch +5.8 %: this kernel is bound by
// instruction ISSUE, ~ 4 cycles of the SIMD per vector instruction of any kind.)
template <bool F16>
__device__ __forceinline__ uint32_t pk16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    if constexpr (F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <bool F16>
__device__ __forceinline__ __bf16 to16(float v) {                       // one value (bit container)
    if constexpr (F16) return __builtin_bit_cast(__bf16, (_Float16)v);
    else return (__bf16)v;
}
// 8 fp32 values -> one fragment
template <bool F16>
__device__ __forceinline__ bf16x8 cvt8(const float (&x)[8]) {
    u32x4 hp;
#pragma unroll
    for (int i = 0; i < 4; ++i) hp[i] = pk16<F16>(x[2 * i], x[2 * i + 1]);
    return __builtin_bit_cast(bf16x8, hp);
}
template <bool F16>
__device__ __forceinline__ bf16x8 cvt_pair(const f32x4& a, const f32x4& b) {        // registers of row tiles 2s, 2s + 1 = k-step s
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return cvt8<F16>(x);
}
// the matrix instructions of either operand type
template <bool F16>
__device__ __forceinline__ f32x4 mm16(bf16x8 a, bf16x8 b, f32x4 c) {                // 16x16x32
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else return mfma16_bf(a, b, c);
}
template <bool F16>
__device__ __forceinline__ f32x16 mm32(bf16x8 a, bf16x8 b, f32x16 c) {              // 32x32x16
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else return mfma_bf(a, b, c);
}
template <bool F16>
__device__ __forceinline__ f32x4 mm4(s16x4 a, s16x4 b, f32x4 c) {                   // 4x4x4
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
    else return mfma4_bf(a, b, c);
}
template <bool F16>
__device__ __forceinline__ f32x4 mm16h(s16x4 a, s16x4 b, f32x4 c) {                 // 16x16x16
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
    else return mfma16h_bf(a, b, c);
}
// Materialise a packed fragment where it is written.  Left alone, the compiler SINKS the tail of a GELU derivative (z phi(z) + Phi(z), the rounding, the
// packing) from the forward pass down to its use in the backward pass and keeps three fp32 intermediates per value alive across the whole round instead
// of half a register: that - not the derivatives themselves - was what spilled 70 - 150 registers whenever all four layers' derivatives of the 5-layer
// decoder were kept (found in the ISA: v_cvt_pk after the backward pass's MFMAs, fed by "Folded Reload"s).
__device__ __forceinline__ void pin(bf16x8& f) {
    u32x4 t = __builtin_bit_cast(u32x4, f);
    asm volatile("" : "+v"(t));
    f = __builtin_bit_cast(bf16x8, t);
}
// the fp32 values of a packed fragment: elements 0..3 (hi4 = false) or 4..7 (hi4 = true)
template <bool F16>
__device__ __forceinline__ f32x4 unpack4(const bf16x8& f, bool hi4) {
    const u32x4 w = __builtin_bit_cast(u32x4, f);
    const uint32_t a = hi4 ? w[2] : w[0], b = hi4 ? w[3] : w[1];
    if constexpr (F16) {
        const h16x2 ha = __builtin_bit_cast(h16x2, a), hb = __builtin_bit_cast(h16x2, b);
        return f32x4{(float)ha[0], (float)ha[1], (float)hb[0], (float)hb[1]};
    } else {
        return f32x4{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xFFFF0000u), __builtin_bit_cast(float, b << 16),
                     __builtin_bit_cast(float, b & 0xFFFF0000u)};
    }
}
// GELU + derivative of one row tile (four values per lane) in the form the plain 16-bit modes use (nic_device.hpp::gelu_sig4); 0 = the exact-erf form
#ifndef NIC_Q16_GELU
#define NIC_Q16_GELU 2
#endif
__device__ __forceinline__ void gelu_q(const f32x4& z, f32x4& a, f32x4& d) {
    if constexpr (NIC_Q16_GELU == 0) gelu_and_grad4(z, a, d);
    else gelu_sig4<NIC_Q16_GELU>(z, a, d);
}
// (Measured and dropped: the GELU derivatives kept as the fp32 values they are instead of packed bf16 pairs - 16 registers per hidden layer, no conversion,
// no unpacking: 5 layers 2.128 -> 2.125 ms, 3 layers 1.258 -> 1.288: the instructions it saves are paid back in register moves.)
// acc[t] += A_t x B for the NT row tiles of one k-step, A fragments fetched PF tiles ahead
template <int NT, bool ZERO = false, int PFQ = NIC_T16_PF, bool F16 = false, class LoadA>
__device__ __forceinline__ void kstep_b(f32x4 (&acc)[NT], const bf16x8& bf, LoadA&& load_a) {
    constexpr int PF = PFQ < NT ? PFQ : NT;
    bf16x8 af[PF];
#pragma unroll
    for (int t = 0; t < PF; ++t) af[t] = load_a(t);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = mm16<F16>(af[t % PF], bf, ZERO ? f32x4(0.f) : acc[t]);
        if (t + PF < NT) af[t % PF] = load_a(t + PF);
    }
}

template <class Q>
struct CellRawQ {
    float g0[QInfo<Q>::NG0V];      // [corner e of the quarter][channel]
    float g1[QInfo<Q>::NG1V];      // [corner q][cc]
};

// corner offsets: G0 corner e of quarter g, G1 corner q
template <class Q>
__device__ __forceinline__ void q_g0_corner(int g, int e, int& dx, int& dy, int& dz) {
    dx = g >> 1; dy = g & 1;
    dz = Q::DIM == 2 ? 0 : (Q::TETRA ? (dx ^ dy) : e);         // fp_def.py:96-103 (bit 0 of the corner number = dz), :108-111
}
template <class Q>
__device__ __forceinline__ void q_g1_corner(int q, int& dx, int& dy, int& dz) {
    if (Q::DIM == 2) { dx = q >> 1; dy = q & 1; dz = 0; }
    else { dx = (q >> 2) & 1; dy = (q >> 1) & 1; dz = q & 1; }
}

// 16-bit storage widened to fp32: both conversions, one select (kind is launch-uniform; no branch per element)
__device__ __forceinline__ float widen16(uint32_t hbits, bool is_bf16) {
    const float a = __builtin_bit_cast(float, hbits << 16);
    const float b = (float)__builtin_bit_cast(_Float16, (uint16_t)hbits);
    return is_bf16 ? a : b;
}
template <class Q, bool DO_G0, bool DO_G1>
__device__ __forceinline__ void gather_cell_q(const FusedParams& p, uint32_t off0, uint32_t off1, int g, CellRawQ<Q>& raw) {
    // ONE launch-uniform branch around the whole gather (a test per element serialised the loads of the in-round gathers: +30 % kernel time
    // with 16-bit grids); 16-bit values are loaded as raw bits first and widened afterwards
    const int kind = p.grid_kind;
    uint32_t o0[Q::NG0], o1[Q::K1];                                       // element offsets of the corners (channel 0 / channel 3 g)
#pragma unroll
    for (int e = 0; e < (DO_G0 ? Q::NG0 : 0); ++e) {
        int dx, dy, dz;
        q_g0_corner<Q>(g, e, dx, dy, dz);
        o0[e] = off0 + (uint32_t)p.g0.at(dx, dy, dz);
    }
#pragma unroll
    for (int q = 0; q < (DO_G1 ? Q::K1 : 0); ++q) {
        int dx, dy, dz;
        q_g1_corner<Q>(q, dx, dy, dz);
        o1[q] = off1 + (uint32_t)p.g1.at(dx, dy, dz) + (uint32_t)(Q::GQ * g) * (uint32_t)p.g1.plane;
    }
    const char* b0 = reinterpret_cast<const char*>(p.g0.p);
    const char* b1 = reinterpret_cast<const char*>(p.g1.p);
    if (kind == 0) {
        const uint32_t pb0 = (uint32_t)p.g0.plane * 4u, pb1 = (uint32_t)p.g1.plane * 4u;
#pragma unroll
        for (int e = 0; e < (DO_G0 ? Q::NG0 : 0); ++e) {
            uint32_t ob = o0[e] * 4u;
#pragma unroll
            for (int c = 0; c < Q::C; ++c, ob += pb0) raw.g0[e * Q::C + c] = *reinterpret_cast<const float*>(b0 + ob);
        }
#pragma unroll
        for (int q = 0; q < (DO_G1 ? Q::K1 : 0); ++q) {
            uint32_t ob = o1[q] * 4u;
#pragma unroll
            for (int cc = 0; cc < Q::GQ; ++cc, ob += pb1) raw.g1[q * Q::GQ + cc] = *reinterpret_cast<const float*>(b1 + ob);
        }
    } else {
        const uint32_t pb0 = (uint32_t)p.g0.plane * 2u, pb1 = (uint32_t)p.g1.plane * 2u;
        uint32_t h0[DO_G0 ? QInfo<Q>::NG0V : 1], h1[DO_G1 ? QInfo<Q>::NG1V : 1];
#pragma unroll
        for (int e = 0; e < (DO_G0 ? Q::NG0 : 0); ++e) {
            uint32_t ob = o0[e] * 2u;
#pragma unroll
            for (int c = 0; c < Q::C; ++c, ob += pb0) h0[e * Q::C + c] = *reinterpret_cast<const uint16_t*>(b0 + ob);
        }
#pragma unroll
        for (int q = 0; q < (DO_G1 ? Q::K1 : 0); ++q) {
            uint32_t ob = o1[q] * 2u;
#pragma unroll
            for (int cc = 0; cc < Q::GQ; ++cc, ob += pb1) h1[q * Q::GQ + cc] = *reinterpret_cast<const uint16_t*>(b1 + ob);
        }
        const bool is_bf = kind == 1;
#pragma unroll
        for (int i = 0; i < (DO_G0 ? QInfo<Q>::NG0V : 0); ++i) raw.g0[i] = widen16(h0[i], is_bf);
#pragma unroll
        for (int i = 0; i < (DO_G1 ? QInfo<Q>::NG1V : 0); ++i) raw.g1[i] = widen16(h1[i], is_bf);
    }
}

// ---- multi-level layouts (QML): cell offsets, gathers and input slots of every pair; fp32 grids
template <class Q>
__device__ __forceinline__ void cell_offsets_ml(const FusedParams& p, const int (&q)[3], uint32_t (&off0)[Q::LEVELS], uint32_t (&off1)[Q::LEVELS]) {
#pragma unroll
    for (int l = 0; l < Q::LEVELS; ++l) {
        const int e = p.d.log2_step - 2 * l;
        const Axis ax = axis_coords(q[0], e), ay = axis_coords(q[1], e);
        const GridView& a = p.ml[l].g0;
        const GridView& b = p.ml[l].g1;
        off0[l] = (uint32_t)a.at(clampi(ax.i0, 0, a.nx - 2), clampi(ay.i0, 0, a.ny - 2), 0);
        off1[l] = (uint32_t)b.at(clampi(ax.i1, 0, b.nx - 2), clampi(ay.i1, 0, b.ny - 2), 0);
    }
}
template <class Q, bool DO_G0, bool DO_G1>
__device__ __forceinline__ void gather_cell_ml(const FusedParams& p, const uint32_t (&off0)[Q::LEVELS], const uint32_t (&off1)[Q::LEVELS], int g, CellRawQ<Q>& raw) {
    const int dx = g >> 1, dy = g & 1;
#pragma unroll
    for (int l = 0; l < Q::LEVELS; ++l) {
        if constexpr (DO_G0) {
            const GridView& a = p.ml[l].g0;
            uint32_t ob = (off0[l] + (uint32_t)a.at(dx, dy, 0)) * 4u;
            const uint32_t pb = (uint32_t)a.plane * 4u;
            const char* base = reinterpret_cast<const char*>(a.p);
#pragma unroll
            for (int c = 0; c < Q::C; ++c, ob += pb) raw.g0[l * Q::C + c] = *reinterpret_cast<const float*>(base + ob);
        }
        if constexpr (DO_G1) {
            const GridView& b = p.ml[l].g1;
            const uint32_t pb = (uint32_t)b.plane * 4u;
            const char* base = reinterpret_cast<const char*>(b.p);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t ob = (off1[l] + (uint32_t)b.at(q >> 1, q & 1, 0) + (uint32_t)(Q::GQ * g) * (uint32_t)b.plane) * 4u;
#pragma unroll
                for (int cc = 0; cc < Q::GQ; ++cc, ob += pb) raw.g1[(l * 4 + q) * Q::GQ + cc] = *reinterpret_cast<const float*>(base + ob);
            }
        }
    }
}
template <class Q>
__device__ __forceinline__ void encode_ml(const FusedParams& p, const int (&q)[3], int g, float (&xs)[Q::NS], const CellRawQ<Q>& raw) {
    constexpr int L = Q::LEVELS, C = Q::C, GQ = Q::GQ, PH = Q::PH, T0 = Q::T0;
    const nic_path_desc& d = p.d;
#pragma unroll
    for (int c = 0; c < L * C; ++c) xs[c] = raw.g0[c];
#pragma unroll
    for (int s = Q::NGS; s < T0; ++s) xs[s] = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int e = d.log2_step - 2 * l;                                   // this pair's step_number (image_compression.py:79)
        const Axis ax = axis_coords(q[0], e), ay = axis_coords(q[1], e);
        const float c = (g >> 1) ? ay.t1 : ax.t1;                            // PE rows PH (g & 1) + i of dimension g >> 1 on THIS pair's G1-cell coordinate
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int r = PH * (g & 1) + i;
            float v;
            if (Q::PE == NIC_PE_TRIANGULAR) {
                v = tri_pe_row(c, r, Q::P);
            } else {
                const int k = r >> 1;
                const float dv = pick_loaded(k == 0, d.pe_div[0], pick_loaded(k == 1, d.pe_div[1], pick_loaded(k == 2, d.pe_div[2], d.pe_div[3])));
                float sv, cv;
                sincos_cw(mul_rn(c, dv), sv, cv);
                v = (r & 1) ? cv : sv;
            }
            xs[T0 + l * PH + i] = v;
        }
        const G1FactorsT<2> gf = g1_factors<2>(d.g1_weight_mode, ax.k1, ay.k1, 0.f);      // fp_def.py:141-144
#pragma unroll
        for (int cc = 0; cc < GQ; ++cc) {
            float sum = 0.f;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                const uint32_t b = (gf.bits >> (3 * c8)) & 7u;
                float v = raw.g1[(l * 4 + c8) * GQ + cc];
                v = mul_rn(v, (b & 1u) ? gf.fx[1] : gf.fx[0]);
                v = mul_rn(v, (b & 2u) ? gf.fy[1] : gf.fy[0]);
                sum = c8 == 0 ? v : add_rn(sum, v);
            }
            xs[L * C + l * GQ + cc] = sum;
        }
    }
    xs[T0 + L * PH] = g == 0 ? d.lod_value : (g == 1 ? 1.0f : 0.f);
#pragma unroll
    for (int s = T0 + L * PH + 1; s < Q::NS; ++s) xs[s] = 0.f;
}
// Flush of a multi-level macro-tile.  The 16 lanes of a quarter are 16 consecutive cells of pair 0 along x; in pair l they fall into runs of 4^l lanes
// with the SAME cell (one run from pair 2 on): the sums of a run are added across its lanes (segmented suffix sums: 4 shuffle steps, keyed by the cell
// offset - runs are contiguous, so "lane + d has my key" means everything in between has it too) and the run's first lane issues the atomics.
__device__ __forceinline__ float seg_sum16(float v, const bool (&same)[4], int ln) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float pv = __shfl(v, ln + (1 << k));
        v += same[k] ? pv : 0.f;
    }
    return v;
}
template <class Q, int NDX>
__device__ __forceinline__ void flush_ml(const FusedParams& p, const uint32_t (&off0)[Q::LEVELS], const uint32_t (&off1)[Q::LEVELS], const f32x4 (&dxacc)[NDX],
                                         const float (&g1s)[QInfo<Q>::NG1V], int ln, float us = 1.0f) {
    const int n16 = ln & 15, g = ln >> 4;
#pragma unroll
    for (int l = 0; l < Q::LEVELS; ++l) {
#pragma unroll
        for (int grid = 0; grid < 2; ++grid) {
            const uint32_t key = grid == 0 ? off0[l] : off1[l];
            bool same[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = 1 << k;
                const uint32_t pk = (uint32_t)__shfl((int)key, ln + d);
                same[k] = n16 + d < 16 && pk == key;
            }
            const uint32_t prev_key = (uint32_t)__shfl((int)key, ln - 1);        // unconditionally: a shuffle inside "n16 == 0 || .." reads lanes the branch has switched off (they return 0)
            const bool head = n16 == 0 || prev_key != key;
            if (grid == 0) {
                const GridView& a = p.ml[l].g0;
                float v[Q::C];
                uint32_t nz = 0u;
#pragma unroll
                for (int c = 0; c < Q::C; ++c) {
                    v[c] = seg_sum16(dxacc[(l * Q::C + c) >> 2][(l * Q::C + c) & 3], same, ln);
                    nz |= __builtin_bit_cast(uint32_t, v[c]);
                }
                if (head && (nz << 1) != 0u) {
                    uint32_t ob = (key + (uint32_t)a.at(g >> 1, g & 1, 0)) * 4u;
                    const uint32_t pb = (uint32_t)a.plane * 4u;
                    char* gbase = reinterpret_cast<char*>(p.ml[l].g0_grad);
#pragma unroll
                    for (int c = 0; c < Q::C; ++c, ob += pb) atomicAdd(reinterpret_cast<float*>(gbase + ob), v[c] * us);
                }
            } else {
                const GridView& b = p.ml[l].g1;
                float v[4 * Q::GQ];
                uint32_t nz = 0u;
#pragma unroll
                for (int i = 0; i < 4 * Q::GQ; ++i) {
                    v[i] = seg_sum16(g1s[l * 4 * Q::GQ + i], same, ln);
                    nz |= __builtin_bit_cast(uint32_t, v[i]);
                }
                if (head && (nz << 1) != 0u) {
                    const uint32_t pb = (uint32_t)b.plane * 4u;
                    char* gbase = reinterpret_cast<char*>(p.ml[l].g1_grad);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t ob = (key + (uint32_t)b.at(q >> 1, q & 1, 0) + (uint32_t)(Q::GQ * g) * (uint32_t)b.plane) * 4u;
#pragma unroll
                        for (int cc = 0; cc < Q::GQ; ++cc, ob += pb) atomicAdd(reinterpret_cast<float*>(gbase + ob), v[q * Q::GQ + cc] * us);
                    }
                }
            }
        }
    }
}

// the quarter's input slots for the sample at absolute coordinates q (xs indexed by slot; the fp32 arithmetic of every other kernel)
template <class Q>
__device__ __forceinline__ void encode_q(const FusedParams& p, const int (&q)[3], int g, float (&xs)[Q::NS], float (&kf)[3], const CellRawQ<Q>& raw) {
    using I = QInfo<Q>;
    constexpr int D = Q::DIM;
    const nic_path_desc& d = p.d;
    const int e = d.log2_step;
    const Axis ax = axis_coords(q[0], e), ay = axis_coords(q[1], e);
    Axis az;
    if (D == 3) az = axis_coords(q[2], e);
    else { az.i0 = az.i1 = 0; az.t1 = az.k1 = 0.f; }
    kf[0] = ax.k1; kf[1] = ay.k1; kf[2] = az.k1;
#pragma unroll
    for (int c = 0; c < I::NG0V; ++c) xs[c] = raw.g0[c];
    // ---- tail: PE rows, LOD, the constant one, zero padding
    if constexpr (D == 2) {
        constexpr int T0 = Q::T0, PH = Q::PH;
#pragma unroll
        for (int s = I::NGS; s < T0; ++s) xs[s] = 0.f;
        const float c = (g >> 1) ? ay.t1 : ax.t1;                       // rows PH (g & 1) + i of dimension g >> 1 (utils.py:198-227)
#pragma unroll
        for (int i = 0; i < PH; ++i) {
            const int r = PH * (g & 1) + i;
            float v;
            if (Q::PE == NIC_PE_TRIANGULAR) {
                v = tri_pe_row(c, r, Q::P);
            } else {
                const int k = r >> 1;
                const float dv = pick_loaded(k == 0, d.pe_div[0], pick_loaded(k == 1, d.pe_div[1], pick_loaded(k == 2, d.pe_div[2], d.pe_div[3])));
                float sv, cv;
                sincos_cw(mul_rn(c, dv), sv, cv);
                v = (r & 1) ? cv : sv;
            }
            xs[T0 + i] = v;
        }
        xs[T0 + PH] = g == 0 ? d.lod_value : (g == 1 ? 1.0f : 0.f);
#pragma unroll
        for (int s = T0 + PH + 1; s < Q::NS; ++s) xs[s] = 0.f;
    } else {
        constexpr int T0 = Q::T0;                                       // first tail slot
#pragma unroll
        for (int s = T0 + 5; s < Q::NS; ++s) xs[s] = 0.f;
        if constexpr (Q::PE == NIC_PE_TRIANGULAR) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int R = j < 4 ? 4 * g + j : 16 + (g & 1);          // PE row 0 .. 17: dimension R / 6, row R % 6
                const int dm = R >= 12 ? 2 : (R >= 6 ? 1 : 0);
                const float c = dm == 0 ? ax.t1 : (dm == 1 ? ay.t1 : az.t1);
                const float v = tri_pe_row(c, R - 6 * dm, kP);
                xs[T0 + j] = j < 4 ? v : (g < 2 ? v : (g == 2 ? d.lod_value : 1.0f));
            }
        } else {
            // rows 2 U, 2 U + 1 = (sin, cos) of pair U: dimension U / 3, frequency U % 3.  Quarter g: pairs 2g, 2g + 1, and pair 8 (rows 16, 17)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int U = u < 2 ? 2 * g + u : 8;
                const int dm = U >= 6 ? 2 : (U >= 3 ? 1 : 0), k = U - 3 * dm;
                const float c = dm == 0 ? ax.t1 : (dm == 1 ? ay.t1 : az.t1);
                const float dv = pick_loaded(k == 0, d.pe_div[0], pick_loaded(k == 1, d.pe_div[1], d.pe_div[2]));
                float sv, cv;
                sincos_cw(mul_rn(c, dv), sv, cv);
                if (u < 2) { xs[T0 + 2 * u] = sv; xs[T0 + 2 * u + 1] = cv; }
                else xs[T0 + 4] = g == 0 ? sv : (g == 1 ? cv : (g == 2 ? d.lod_value : 1.0f));
            }
        }
    }
    // ---- G1 blend with the reference's factor order (fp_def.py:141-144, 176-183, 215-222)
    const G1FactorsT<D> gf = g1_factors<D>(d.g1_weight_mode, kf[0], kf[1], kf[2]);
#pragma unroll
    for (int cc = 0; cc < Q::GQ; ++cc) {
        float sum = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < Q::K1; ++c8) {
            const uint32_t b = (gf.bits >> (3 * c8)) & 7u;
            float v = raw.g1[c8 * Q::GQ + cc];
            if constexpr (D == 2) {
                v = mul_rn(v, (b & 1u) ? gf.fx[1] : gf.fx[0]);
                v = mul_rn(v, (b & 2u) ? gf.fy[1] : gf.fy[0]);
            } else {
                v = mul_rn(v, gf.x(b));
                v = mul_rn(v, gf.y(b));
                v = mul_rn(v, gf.z(b));
            }
            sum = c8 == 0 ? v : add_rn(sum, v);
        }
        xs[I::NG0V + cc] = sum;
    }
}

// noise on every real channel (image_compression.py:250): caller-supplied tensor, or the in-kernel generator numbered by quarter
template <class Q>
__device__ __forceinline__ void add_noise_q(const NoiseSrc& ns, uint64_t sample_global, int64_t n_local, int g, float (&xs)[Q::NS]) {
    if (ns.mode == NIC_NOISE_NONE) return;
    if (ns.mode == NIC_NOISE_TENSOR) {
        const float* row = ns.tensor + n_local * Q::CIN;
#pragma unroll
        for (int s = 0; s < Q::NS; ++s) {
            const int c0 = Q::slot_channel(s, 0), c1 = Q::slot_channel(s, 1), c2 = Q::slot_channel(s, 2), c3 = Q::slot_channel(s, 3);
            if (c0 < 0 && c1 < 0 && c2 < 0 && c3 < 0) continue;
            const int ch = g == 0 ? c0 : (g == 1 ? c1 : (g == 2 ? c2 : c3));
            xs[s] += ch >= 0 ? row[ch >= 0 ? ch : 0] : 0.f;
        }
        return;
    }
    using I = QInfo<Q>;
    U4 b[I::NBLK];
#pragma unroll
    for (int i = 0; i < I::NBLK; ++i) b[i] = noise_block(ns, sample_global, I::NBLK * g + i);
#pragma unroll
    for (int s = 0; s < Q::NS; ++s) {
        const int f = I::noise_index(s);
        if (f < 0) continue;
        const bool r0 = Q::slot_channel(s, 0) >= 0, r1 = Q::slot_channel(s, 1) >= 0, r2 = Q::slot_channel(s, 2) >= 0, r3 = Q::slot_channel(s, 3) >= 0;
        if (!(r0 || r1 || r2 || r3)) continue;
        const float v = noise_field(ns, b[f / 20], f % 20);
        if (r0 && r1 && r2 && r3) xs[s] += v;
        else xs[s] += (g == 0 ? r0 : (g == 1 ? r1 : (g == 2 ? r2 : r3))) ? v : 0.f;
    }
}

// G1 sums of the lanes whose G0 cells share a G1 cell are added across lanes before the flush (one atomic per node and instruction):
// partner = the lane whose block coordinate differs in the last bit along one axis, when it sits in the same 16-lane block list and
// really has the same (clamped) G1 cell.  Regular tiles: 2^lw x 16 / 2^lw x 1 blocks; packed tiling: 16 consecutive blocks of the
// crop's pk[0] x pk[1] x pk[2] list.
template <class Q>
__device__ __forceinline__ void combine_g1_lanes_q(float (&g1s)[QInfo<Q>::NG1V], uint32_t off1, const int (&blk)[3], int lane, int lw, bool packed,
                                                   const int (&pk)[3], const int (&pk_lc)[3]) {
    constexpr int D = Q::DIM;
    const int T[3] = {packed ? pk[0] : 1 << lw, packed ? pk[1] : 16 >> lw, packed ? pk[2] : 1};
    const int STR[3] = {1, packed ? pk[0] : 1 << lw, packed ? pk[0] * pk[1] : 16};
    const int pl = lane & 15;
    const int lc[3] = {packed ? pk_lc[0] : pl & (T[0] - 1), packed ? pk_lc[1] : pl >> lw, packed ? pk_lc[2] : 0};
#pragma unroll
    for (int a = 0; a < D; ++a) {
        const bool odd = blk[a] & 1;
        const int pc = lc[a] + (odd ? -1 : 1);
        const int ppl = pl + (odd ? -STR[a] : STR[a]);
        const bool inb = pc >= 0 && pc < T[a] && ppl >= 0 && ppl < 16;
        const int partner = inb ? lane + (odd ? -STR[a] : STR[a]) : lane;
        const bool pair = inb && (uint32_t)__shfl((int)off1, partner) == off1;
#pragma unroll
        for (int i = 0; i < QInfo<Q>::NG1V; ++i) {
            const float pv = __shfl(g1s[i], partner);
            g1s[i] = pair ? (odd ? 0.f : g1s[i] + pv) : g1s[i];
        }
    }
}

// Neighbour pre-add of the G0 sums along x inside a wave (2D and method 3: quarter (dx = 1, dy) of cell n - 1 addresses the nodes of
// quarter (dx = 0, dy) of cell n whenever their cell offsets differ by one element) - see preadd_x16; NT tiles of four sums per lane
template <int NT, int NA>
__device__ __forceinline__ void preadd_x_q(f32x4 (&dxacc)[NA], uint32_t off0, int ln) {
    const int n16 = ln & 15, g = ln >> 4;
    const bool recv = g < 2;
    const bool inb = recv ? n16 >= 1 : n16 <= 14;
    const int partner = inb ? (recv ? ln + 31 : ln - 31) : ln;
    const uint32_t poff = (uint32_t)__shfl((int)off0, partner);
    const bool pair = inb && (recv ? off0 == poff + 1u : poff == off0 + 1u);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = __shfl(dxacc[t][r], partner);
            dxacc[t][r] = pair ? (recv ? dxacc[t][r] + pv : 0.f) : dxacc[t][r];
        }
}
// .. and along y across the 8 waves (consecutive waves hold consecutive rows y of the same 16 columns): see preadd_y16
template <int NT, int NA, class Barrier>
__device__ __forceinline__ void preadd_y_q(f32x4 (&dxacc)[NA], uint32_t off0, uint32_t row, int ln, int wave, lds_f* reg0, int region, Barrier&& barrier) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    const int g = ln >> 4;
    lds_f* const mine = opaque(reg0 + wave * region + ln);
    const bool up = (g & 1) != 0;
    ((lds_u32_t*)mine)[4 * NT * 64] = off0;
    if (up) {
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) mine[i * 64] = dxacc[i >> 2][i & 3];
    }
    barrier();
    const int pw = up ? wave + 1 : wave - 1;
    const bool inw = pw >= 0 && pw < 8;
    lds_cf* const theirs = opaque(reg0 + (inw ? pw : wave) * region + (up ? ln - 16 : ln + 16));
    const uint32_t poff = ((const lds_u32_t*)theirs)[4 * NT * 64];
    const bool pair = inw && (up ? poff == off0 + row : off0 == poff + row);
    if (!up) {
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) {
            const float pv = theirs[i * 64];
            dxacc[i >> 2][i & 3] += pair ? pv : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4 * NT; ++i) dxacc[i >> 2][i & 3] = pair ? 0.f : dxacc[i >> 2][i & 3];
    }
    barrier();
}

#ifndef NIC_RQ_SLICES
#define NIC_RQ_SLICES 32       // record slices summed in parallel by reduce_q16_kernel (256 / slices outputs x slices threads per block): with 32 a thread
                               // walks 8 of the <= 256 records - one batch of loads in flight instead of four dependent ones (8 -> 32: the small steps -2 .. -3 %,
                               // 4K launches +- 0.3 %; 16: the 33^3 step +2.5 %)
#endif
#ifndef NIC_Q16_HALF16
#define NIC_Q16_HALF16 2        // 1: always, 2: with 5 layers
#endif
#ifndef NIC_Q16_PIN
#define NIC_Q16_PIN 2        // pin the derivative fragments: 1 everywhere, 2 with 5 layers and for method 3 (measured, interleaved A/B: pinning costs the 3-layer kernels 0.8 % in 2D, 2 % with method 4)
#endif
#ifndef NIC_Q16_PREADD
#define NIC_Q16_PREADD 2
#endif
#ifndef NIC_Q16_LOFF
#define NIC_Q16_LOFF 1        // lane offsets of the round's LDS addresses computed once per launch (see the kernel)
#endif


// =====================================================================================================
#ifdef NIC_Q16_NVGPR
#define NIC_Q16_ATTR __attribute__((amdgpu_waves_per_eu(NIC_Q16_NVGPR, NIC_Q16_NVGPR)))      // diagnostic: how many registers does the kernel really need?
#else
#define NIC_Q16_ATTR
#endif
template <class Q, int MODE, int NL, bool F16 = false>
__global__ void __launch_bounds__(512) NIC_Q16_ATTR fused_q16_kernel(FusedParams p) {
    using S = LdsQ<Q, NL>;
    using I = QInfo<Q>;
    constexpr int NH = S::NH, D = Q::DIM, NS = Q::NS, KF = I::KF, NDX = I::NDX, NG0T = I::NG0V / 4;
    constexpr bool HALF = I::HALF;
    constexpr int LD1 = S::LD1, LDH = S::LDH, LDZ = S::LDZ, LDX = S::LDX;
    static_assert(NL == 3 || NL == 5, "3 or 5 Linear layers");
    // A fragments of a k-step fetched this many row tiles ahead: all four with 3 layers (1.460 -> 1.435 ms at 4K), two with 5 (four: 2.40 -> 2.45)
#ifdef NIC_Q16_PF
    constexpr int KPF = NIC_Q16_PF;
#else
    constexpr int KPF = NL == 3 ? 4 : 2;
#endif
    constexpr bool TRAIN = MODE != MODE_INFER;                  // MODE_INFER: the forward pass alone (decode_image for the layouts only these kernels serve)
    // raw grid values gathered once per macro-tile (every sample a lane handles there lies in the same G0 / G1 cell) and kept in registers where
    // the budget of two waves per SIMD allows, otherwise re-fetched at the end of every round for the next one (L1 / L2 hits; dead through
    // the forward and backward passes): NIC_Q16_HOIST bit 0: G0 kept, bit 1: G1 kept
#ifdef NIC_Q16_HOIST
    constexpr int HOIST = NIC_Q16_HOIST;
#else
    // measured (4K / 128^3 launches, interleaved A/B on one box, hoist 0 / 1 / 3): 2D NL 3: 1.51 / 1.49 / 1.45 ms, NL 5: 3.65 / 3.77 / 4.25 (spills);
    // method 4: 0.616 / 0.558 / 0.543 ms; method 3: 0.525 / 0.539 / 0.601 (spills)
    // (second half of round 3: with the derivatives pinned method 3 keeps both grids' raw values too - 5 spilled registers instead of the 48 that made
    //  hoisting lose: 0.552 -> 0.517 ms at 128^3, 0.585 -> 0.52 with 16-bit grids; its 5-layer form still spills with them: not hoisted there)
    // multi-level layouts re-fetch (their raw values are L x (C + 4 GQ) registers beside as many gradient sums)
    constexpr int HOIST = !TRAIN ? 3 : (Q::LEVELS > 1 ? 0 : (NL == 3 ? 3 : (Q::NG0 == 1 ? 3 : 0)));
#endif
    constexpr bool ML = Q::LEVELS > 1;
    constexpr bool HG0 = (HOIST & 1) != 0, HG1 = (HOIST & 2) != 0;
    // GELU derivatives: kept from the forward pass as packed bf16 (8 registers per hidden layer), pinned where they are written (pin())
    constexpr bool PIN = NIC_Q16_PIN == 1 || (NIC_Q16_PIN == 2 && (NL == 5 || Q::NG0 >= 2)) || (NIC_Q16_PIN == 3 && (NL == 5 || D == 3));     // NG0 >= 2: method 3 and the multi-level layouts (22 -> 4 spilled registers at L 5 / C 4)
    __shared__ __attribute__((aligned(16))) __bf16 smemq[TRAIN ? S::TOTAL : S::OFF_IMG];
    lds_bf* const sm = (lds_bf*)smemq;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The lane-dependent parts of the round's LDS addresses, computed ONCE per launch and kept in 9 registers (element offsets; the wave's image region
    // included where the address is wave-private).  The phases of a round re-derive their lane ids from an opaque copy of the lane number (opaque_i: left
    // alone, the optimiser hoists everything computed from them, ~ 70 registers), which made every pointer of a round cost its own mask / shift / multiply /
    // add chain - ~ 185 of the ~ 2 400 instructions of a 5-layer round: with the offsets at hand a pointer is one add.
    constexpr bool LOFF = NIC_Q16_LOFF != 0 && TRAIN && (Q::LEVELS == 1 || NIC_Q16_LOFF == 2);
    int lo_rH = 0, lo_r1 = 0, lo_rX = 0, lo_rZ = 0, lo_tH = 0, lo_t1 = 0, lo_b44 = 0, lo_tZ = 0, lo_tX = 0;
    if constexpr (LOFF) {
        const int n16 = lane & 15, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h32 = lane >> 5, cg = (lane >> 4) & 1;
        lo_rH = opaque_i(n16 * LDH + 8 * g);                                       // row n16, columns 8 g ..: forward A fragments of the 64-column weight images
        lo_r1 = LD1 == LDH ? lo_rH : opaque_i(n16 * LD1 + 8 * g);                  // .. of W1
        lo_rX = opaque_i(wave * S::SPW + n16 * LDX + 8 * g);                       // this wave's X image: fragment stores
        lo_rZ = opaque_i(wave * S::SPW + n16 * LDZ + 8 * g);                       // .. its A_k / DZ images
        lo_tH = opaque_i((4 * g + q4) * LDH + 8 * p4);                             // transposed reads of the hidden weight images
        lo_t1 = LD1 == LDH ? lo_tH : opaque_i((4 * g + q4) * LD1 + 8 * p4);        // .. of W1
        lo_b44 = opaque_i(wave * S::SPW + 4 * q4 * LDZ + 16 * g + 4 * p4);         // 4x4x4 B operands out of this wave's DZ image
        lo_tZ = opaque_i((4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4);             // 32x32x16 operands out of the DZ / A_k images of a source wave
        lo_tX = opaque_i((4 * q4 + 2 * h32) * LDX + 16 * cg + 4 * p4);             // .. out of its X image
    }
#ifdef NIC_STAMPS
    unsigned long long stamp_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t0)::"memory");
#endif

    // ---------------- prologue: bf16 weight images, fp32 biases, wave regions zeroed
    stage_all<kH * LD1, 512>(tid,
        [&](int idx) {
            const int o = idx / LD1, r = idx - o * LD1;
            const int ch = r < I::KP ? I::channel_of_rho(r) : kSlotZero;
            const float v = p.W[0][o * Q::CIN + (ch >= 0 ? ch : 0)];
            return ch >= 0 ? v : 0.f;                                   // the constant-one column stays zero: b1 is added in fp32
        },
        [&](int idx, float v) { sm[S::OFF_W1 + idx] = to16<F16>(v); });
#ifdef NIC_STAMPS
    unsigned long long stamp_t1, stamp_t2;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t1)::"memory");
#endif
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const float* Wk = p.W[1 + k];
        stage_all<kH * LDH, 512>(tid,
            [&](int idx) {
                const int o = idx / LDH, ps = idx - o * LDH;
                const float v = Wk[o * kH + hid16(ps < kH ? ps : 0)];
                return ps < kH ? v : 0.f;
            },
            [&](int idx, float v) { sm[S::OFF_WH + k * S::WSZ + idx] = to16<F16>(v); });
    }
    stage_all<4 * LDH, 512>(tid,
        [&](int idx) {
            const int c = idx / LDH, ps = idx - c * LDH;
            return (c < 3 && ps < kH) ? p.W[NL - 1][c * kH + hid16(ps)] : 0.f;
        },
        [&](int idx, float v) { sm[S::OFF_WO + idx] = to16<F16>(v); });
#ifdef NIC_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t2)::"memory");
#endif
    lds_f* const Bs = (lds_f*)(sm + S::OFF_B);
    for (int idx = tid; idx < (NH + 1) * kH; idx += 512) Bs[idx] = p.b[idx / kH][idx % kH];
    if (tid < 4) Bs[(NH + 1) * kH + tid] = tid < 3 ? p.b[NL - 1][tid] : 0.f;
#ifdef NIC_STAMPS
    unsigned long long stamp_t3, stamp_t4;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t3)::"memory");
#endif
    if (TRAIN)
        for (int idx = tid; idx < 8 * S::SPW / 2; idx += 512) ((lds_f*)(sm + S::OFF_IMG))[idx] = 0.f;
#ifdef NIC_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t4)::"memory");
#endif
    __syncthreads();

    // ---------------- launch-lifetime accumulators: the weight-gradient tiles this wave owns
    f32x16 accW[S::NACC];                // slot j >> 1: the tile (wave & 3) of the phases j this wave's half owns (LdsQ); XSLOT: its dW1 tile when there are 8
    f32x4 accT = f32x4(0.f);             // HALF: dW1's last 16 columns, rows 16 (wave & 3).. (the waves of TAIL_HALF)
    f32x4 accWOq = f32x4(0.f), accBH = f32x4(0.f);   // accBH: row k = db of hidden layer k (4x4x4 MFMAs against ones in row k only)
#pragma unroll
    for (int k = 0; k < S::NACC; ++k) accW[k] = f32x16(0.f);
    float accBO[3] = {0.f, 0.f, 0.f}, accLoss = 0.f;
    const int T4 = wave & 3, kh = wave >> 2;
    const int to = T4 >> 1, tk = T4 & 1;
    lds_bf* const img0 = sm + S::OFF_IMG;
    auto barrier = [&]() { wg_lds_barrier(); };
    const NoiseSrc nsrc = noise_with_step(p.noise, p.step_dev);

#ifdef NIC_STAMPS
    // phases: 0 encode + noise | 1 forward layers | 2 dW_out, dA_last | 3 + 2 j: phase j up to its barrier, 4 + 2 j: its barrier wait + owned dW
    // (j = 0 .. NH - 1) | 11 dX + grid sums | 12 setup + gather | 13 flush | 14 barrier + dW1 | 15 round-end barrier
    unsigned long long stamp_sum[NIC_NPH];
#pragma unroll
    for (int i = 0; i < NIC_NPH; ++i) stamp_sum[i] = 0;
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
    if (NL == 3) { stamp_sum[9] = stamp_t1 - stamp_t0; stamp_sum[7] = stamp_t2 - stamp_t1; stamp_sum[8] = stamp_t3 - stamp_t2; stamp_sum[5] = stamp_t4 - stamp_t3; stamp_sum[6] = stamp_last - stamp_t4; }      // 9: prologue; 5 / 6: absolute start / end of the wave (3 layers: phases 5 .. 10 are free)
#endif
    // ---------------- XCD-aware persistent walk, workgroup-synchronous rounds of 8 units (one per wave), two segments: fused_train16.hpp
    const int xcd = blockIdx.x & 7, nb8 = gridDim.x >> 3;
  for (int seg = 0; seg < 2; ++seg) {
    const int seg_tile0 = seg ? (int)p.seg_split : 0;
    const int seg_tiles = seg ? (int)p.n_tiles - (int)p.seg_split : (int)p.seg_split;
    const int rg = seg ? p.rg_log2 : p.rg0_log2;
    if (seg_tiles <= 0) continue;                                       // launch-uniform
    const int n_units = seg_tiles << rg;
    const int chunk = (((n_units + 7) >> 3) + 7) & ~7;
    const int t_begin = xcd * chunk;
    const int t_end = t_begin + chunk < n_units ? t_begin + chunk : n_units;
    const int lstride = nb8 * 8;
    const int base0 = t_begin + (int)(blockIdx.x >> 3) * 8;
    const int n_my = base0 < t_end ? (t_end - base0 + lstride - 1) / lstride : 0;
    const int tiles_per_crop = (int)p.tiles_per_crop, tiles_main = (int)p.tiles_main;
    const int rounds_unit = (p.niter * p.passes) >> rg;
    const int nph = rounds_unit >= NIC_PHASES ? NIC_PHASES : (rounds_unit >= 2 ? 2 : 1);
    const int shift = (TRAIN && NIC_STAGGER && (NIC_STAGGER_RG || rg == 0)) ? (int)((blockIdx.x >> 3) & (nph - 1)) * (rounds_unit / nph) : 0;
    const bool packed = p.pk_nc > 0;                                    // launch-uniform
    const int pk[3] = {p.pk_bx, p.pk_by, packed ? p.pk_nc / (p.pk_bx * p.pk_by) : 1};

    for (int kk = 0; kk < n_my + (shift ? 1 : 0); ++kk) {
        const int base = base0 + (kk < n_my ? kk : 0) * lstride;
        const bool tile_ok = base + wave < t_end;
        const int unit = tile_ok ? base + wave : t_end - 1;
        const int tile = seg_tile0 + (unit >> rg);
        int it_len = rounds_unit, it_begin = (int)(unit & ((1 << rg) - 1)) * rounds_unit;
        if (shift) {
            if (kk == 0) { it_begin += shift; it_len -= shift; }
            else if (kk == n_my) it_len = shift;
        }
        // ---------- macro-tile -> this lane's cell (absolute block coordinates) and crop
        int lw = 4;
        int org[3] = {0, 0, 0}, blk[3] = {0, 0, 0}, pk_lc[3] = {0, 0, 0};
        const int crop = tile / tiles_per_crop;
        float g1s[I::NG1V];                                                  // G1 gradient sums of the cell
        f32x4 dxacc[NDX];                                                    // tiles 0 .. NG0T-1: the cell's G0 gradient sums (persistent over the rounds)
        uint32_t blk_off0, blk_off1;
        uint32_t mo0[Q::LEVELS], mo1[Q::LEVELS];                             // multi-level: the cell offsets of every pair
        CellRawQ<Q> raw;
        {
            const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4;
            int tt = tile - crop * tiles_per_crop;
            int boff[3] = {0, 0, 0};
            if (packed) {
                // packed tiling: 16 consecutive blocks of the crop's block list (x fastest); numbers past the end land outside the crop
                const int ci = tt * 16 + n16;
                const int cy = ci / pk[0];
                pk_lc[0] = ci - cy * pk[0];
                pk_lc[2] = D == 3 ? cy / pk[1] : 0;
                pk_lc[1] = D == 3 ? cy - pk_lc[2] * pk[1] : cy;
            } else if (p.edge_lw < 0 || tt < tiles_main) {
                boff[1] = tt % p.tiles_y;                 // regular tile: 16 x 1 x 1 cells; y fastest, then z, then the 16-cell column along x
                tt /= p.tiles_y;
                boff[2] = D == 3 ? tt % p.tiles_z : 0;
                boff[0] = (D == 3 ? tt / p.tiles_z : tt) * 16;
            } else {
                int e = tt - tiles_main;                  // edge tile: 2^lw x (16 >> lw) x 1 cells
                lw = p.edge_lw;
                boff[2] = D == 3 ? e % p.tiles_z : 0;
                if (D == 3) e /= p.tiles_z;
                boff[1] = e * (16 >> lw);
                boff[0] = p.full_x * 16;
            }
            const int lc[3] = {packed ? pk_lc[0] : n16 & ((1 << lw) - 1), packed ? pk_lc[1] : n16 >> lw, packed ? pk_lc[2] : 0};
#pragma unroll
            for (int a = 0; a < D; ++a) {
                org[a] = origin_of(p, crop * D + a);
                blk[a] = (org[a] >> p.lm) + boff[a] + lc[a];
            }
#pragma unroll
            for (int i = 0; i < I::NG1V; ++i) g1s[i] = 0.f;
#pragma unroll
            for (int t = 0; t < NDX; ++t) dxacc[t] = f32x4(0.f);
            const int qb[3] = {blk[0] << p.lm, blk[1] << p.lm, blk[2] << p.lm};
            if constexpr (ML) {
                cell_offsets_ml<Q>(p, qb, mo0, mo1);
                blk_off0 = mo0[0]; blk_off1 = mo1[0];
                gather_cell_ml<Q, true, true>(p, mo0, mo1, g, raw);
            } else {
                cell_offsets<Q>(p, qb, blk_off0, blk_off1);
                gather_cell_q<Q, true, true>(p, blk_off0, blk_off1, g, raw);
            }
        }
        STAMP(12);

        for (int it = it_begin; it < it_begin + it_len; ++it) {
            // ================= forward =================
            bf16x8 dpk[NH + 1][2];                                            // GELU derivatives of every hidden activation, bf16, packed like the B fragments
            float dz3[3];
            float kf[3];                                                      // G1 interpolation fractions of the sample
            int qs[2] = {0, 0};                                               // multi-level: its coordinates (the fractions of every pair are re-derived in the backward pass)
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4;
                // ---------- which sample does this lane own in this round
                bool valid = tile_ok;
                int64_t n;
                int q[3] = {0, 0, 0};
                {
                    const int m1 = (1 << p.lm) - 1;
                    const int pass = it >> (p.lm * D), its = it & (p.niter - 1);
                    int j[3];
                    if (D == 2) { j[0] = its >> p.lm; j[1] = its & m1; j[2] = 0; }
                    else { j[0] = its >> (2 * p.lm); j[1] = (its >> p.lm) & m1; j[2] = its & m1; }
                    int idx[3] = {0, 0, 0};
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const int ext = p.d.extent[a];
                        const int i = (blk[a] << p.lm) + j[a] - org[a];
                        valid = valid && i >= 0 && i < ext;
                        idx[a] = i < 0 ? 0 : (i >= ext ? ext - 1 : i);
                        q[a] = org[a] + idx[a];
                    }
                    int64_t lin = (int64_t)idx[0] * p.d.extent[1] + idx[1];
                    if (D == 3) lin = lin * p.d.extent[2] + idx[2];
                    n = ((int64_t)crop * p.passes + pass) * p.n_per_crop + lin;
                }
                // ---------- target (or incoming dY) of the sample: fetched now, used after the forward pass
                float tgt[3] = {0.f, 0.f, 0.f};
                if (!TRAIN) {
                } else if (MODE == MODE_TRAIN_IMG) {
                    int64_t off = (int64_t)q[0] * p.timg_s[0] + (int64_t)q[1] * p.timg_s[1];
                    if (D == 3) off += (int64_t)q[2] * p.timg_s[2];
                    uint32_t rgbx = 0u;
                    if (p.timg_u8 == 2) rgbx = reinterpret_cast<const uint32_t*>(p.timg)[off];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (p.timg_u8) {
                            const float u = p.timg_u8 == 2 ? (float)((rgbx >> (8 * c)) & 255u) : (float)(reinterpret_cast<const uint8_t*>(p.timg) + c * p.timg_cs)[off];
                            const float t0 = mul_rn(u, p.timg_rcp);
                            tgt[c] = fmaf(fmaf(-t0, p.timg_den, u), p.timg_rcp, t0);
                        } else {
                            tgt[c] = (reinterpret_cast<const float*>(p.timg) + c * p.timg_cs)[off];
                        }
                    }
                } else {
                    const float* tp = (MODE == MODE_TRAIN_MSE ? p.target : p.dy) + n * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) tgt[c] = tp[c];
                }
                // ---------- input slots
                float xs[NS];
                if constexpr (ML) { encode_ml<Q>(p, q, g, xs, raw); qs[0] = q[0]; qs[1] = q[1]; }
                else encode_q<Q>(p, q, g, xs, kf, raw);
                add_noise_q<Q>(nsrc, (uint64_t)(p.d.sample_base + n), n, g, xs);
                STAMP(0);
                lds_bf* const imgw = img0 + wave * S::SPW;
                lds_cbf* const w_row = opaque((lds_cbf*)(sm + (LOFF ? lo_rH : n16 * LDH + 8 * g)));                       // 64-column weight images: row n16, columns 8 g ..
                lds_cf* const b_row = opaque(Bs + 4 * g);
                f32x4 z[4];
                // ---------- layer 1: Z1[o][n] = sum_rho W1[o][rho] X[rho][n] + b1[o]
                {
                    lds_cbf* const w1_row = opaque((lds_cbf*)(sm + S::OFF_W1 + (LOFF ? lo_r1 : n16 * LD1 + 8 * g)));
                    lds_bf* const x_st = LOFF ? opaque(img0 + lo_rX) : opaque(imgw + n16 * LDX + 8 * g);                                // fragment stores: row n, columns 32 s + 8 g
#pragma unroll
                    for (int t = 0; t < 4; ++t) z[t] = ld4(&b_row[16 * t]);
#pragma unroll
                    for (int s = 0; s < KF; ++s) {
                        const float xv[8] = {xs[8 * s], xs[8 * s + 1], xs[8 * s + 2], xs[8 * s + 3], xs[8 * s + 4], xs[8 * s + 5], xs[8 * s + 6], xs[8 * s + 7]};
                        const bf16x8 bf = cvt8<F16>(xv);
                        if (TRAIN) st_frag(&x_st[32 * s], bf);
                        kstep_b<4, false, KPF, F16>(z, bf, [&](int t) { return ld_frag(&w1_row[16 * t * LD1 + 32 * s]); });
                    }
                    if constexpr (HALF) {   // slots 8 KF .. 8 KF + 3: compact columns 32 KF + 4 g + j
                        lds_cbf* const w1_row2 = opaque((lds_cbf*)(sm + S::OFF_W1 + 32 * KF + (LOFF ? lo_r1 - 4 * g : n16 * LD1 + 4 * g)));
                        lds_bf* const x_st2 = LOFF ? opaque(img0 + 32 * KF + lo_rX - 4 * g) : opaque(imgw + n16 * LDX + 32 * KF + 4 * g);
                        const float xv[8] = {xs[8 * KF], xs[8 * KF + 1], xs[8 * KF + 2], xs[8 * KF + 3], 0.f, 0.f, 0.f, 0.f};
                        const bf16x8 bf = cvt8<F16>(xv);
                        const s16x8 bh = __builtin_bit_cast(s16x8, bf);
                        if (TRAIN) *reinterpret_cast<lds_s16x4*>(x_st2) = s16x4{bh[0], bh[1], bh[2], bh[3]};
                        if constexpr (NIC_Q16_HALF16 == 1 || (NIC_Q16_HALF16 == 2 && NL == 5)) {
                            // the compact half k-step IS the operand layout of v_mfma_f32_16x16x16_bf16 (fused_train16.hpp): no zero-padded fragments
                            // (measured: 5 layers 2.365 -> 2.348 ms; 3 layers and method 4 unchanged - with SLP vectorisation still on it LOST 1.4 % there)
                            const s16x4 bq = {bh[0], bh[1], bh[2], bh[3]};
#pragma unroll
                            for (int t = 0; t < 4; ++t) z[t] = mm16h<F16>(*reinterpret_cast<lds_cs16x4*>(&w1_row2[16 * t * LD1]), bq, z[t]);
                        } else {
                            kstep_b<4, false, KPF, F16>(z, bf, [&](int t) { return half_frag(*reinterpret_cast<lds_cs16x4*>(&w1_row2[16 * t * LD1])); });
                        }
                    }
                }
                // ---------- hidden layers: the B fragments of layer k + 1 are the image A_k of its weight gradient
                // GELU two row tiles at a time, packed at once: a k-step's activations and derivatives are 8 + 8 registers only until they are 4 + 4
                bf16x8 af[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f32x4 a4[2], d4[2];
                    gelu_q(z[2 * s], a4[0], d4[0]);
                    gelu_q(z[2 * s + 1], a4[1], d4[1]);
                    af[s] = cvt_pair<F16>(a4[0], a4[1]);
                    dpk[0][s] = cvt_pair<F16>(d4[0], d4[1]);
                    if (PIN) pin(dpk[0][s]);
                }
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    lds_bf* const a_st = LOFF ? opaque(img0 + S::OFF_A + k * S::ASZ + lo_rZ) : opaque(imgw + S::OFF_A + k * S::ASZ + n16 * LDZ + 8 * g);
#pragma unroll
                    for (int t = 0; t < 4; ++t) z[t] = ld4(&b_row[(k + 1) * kH + 16 * t]);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (TRAIN) st_frag(&a_st[32 * s], af[s]);
                        kstep_b<4, false, KPF, F16>(z, af[s], [&](int t) { return ld_frag(&w_row[S::OFF_WH + k * S::WSZ + 16 * t * LDH + 32 * s]); });
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f32x4 a4[2], d4[2];
                        gelu_q(z[2 * s], a4[0], d4[0]);
                        gelu_q(z[2 * s + 1], a4[1], d4[1]);
                        af[s] = cvt_pair<F16>(a4[0], a4[1]);
                        dpk[k + 1][s] = cvt_pair<F16>(d4[0], d4[1]);
                        if (PIN) pin(dpk[k + 1][s]);
                    }
                }
                // ---------- output layer (rows 0..2 of a 16-row tile; quarter 0 holds the sample's 3 outputs); its input fragments are the
                // image dW_out contracts with (the DZ region is free until the first dZ is stored)
                float yv[3];
                {
                    lds_cbf* const wo_row = opaque((lds_cbf*)(sm + S::OFF_WO + (n16 < 3 ? n16 : 3) * LDH + 8 * g));
                    lds_bf* const dz_st = LOFF ? opaque(img0 + S::OFF_DZ + S::dz_buf(0) * S::ASZ + lo_rZ) : opaque(imgw + S::OFF_DZ + S::dz_buf(0) * S::ASZ + n16 * LDZ + 8 * g);
                    f32x4 z3;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (TRAIN) st_frag(&dz_st[32 * s], af[s]);
                        z3 = mm16<F16>(ld_frag(&wo_row[32 * s]), af[s], s == 0 ? f32x4(0.f) : z3);
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) yv[c] = sigmoid_f(z3[c] + ((lds_cf*)Bs)[(NH + 1) * kH + c]);
                }
                const bool own = valid && g == 0;
                if (p.y != nullptr && own) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) p.y[n * 3 + c] = yv[c];
                }
                if (!TRAIN) {
                    if (p.y_u8 != nullptr && own) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) p.y_u8[n * 3 + c] = (uint8_t)(int)floorf(add_rn(mul_rn(yv[c], 255.0f), 0.5f));      // quantize_to_bit (models.py:39-40)
                    }
                    continue;
                }
                // ---------- dZ_out
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float gr;
                    if (MODE == MODE_TRAIN_MSE || MODE == MODE_TRAIN_IMG) {
                        const float diff = own ? yv[c] - tgt[c] : 0.f;
                        accLoss += diff * diff;
                        gr = (F16 ? p.grad_scale * p.dz_scale : p.grad_scale) * diff;       // fp16 products: dZ is carried as 2^k dZ (FusedParams::dz_scale)
                    } else {
                        gr = own ? (F16 ? tgt[c] * p.dz_scale : tgt[c]) : 0.f;
                    }
                    dz3[c] = gr * yv[c] * (1.0f - yv[c]);
                    accBO[c] += dz3[c];
                }
                if (g == 0) {
                    lds_bf* const d3_st = opaque(imgw + S::OFF_D3 + 4 * (n16 & 3) + (n16 >> 2));
#pragma unroll
                    for (int c = 0; c < 3; ++c) d3_st[c * 16] = to16<F16>(dz3[c]);
                }
            }
            wave_lds_fence();
            STAMP(1);
            // ================= backward =================
            f32x4 dzc[4];                                                   // dZ of the pre-activation of the layer at hand; ends as layer 1's dZ
            {
                const int ln = opaque_i(lane), q4 = (ln & 15) >> 2, p4 = ln & 3, g = ln >> 4;
                lds_bf* const imgw = img0 + wave * S::SPW;
                lds_cbf* const dz_b44 = LOFF ? opaque((lds_cbf*)(img0 + S::OFF_DZ + S::dz_buf(0) * S::ASZ + lo_b44)) : opaque((lds_cbf*)(imgw + S::OFF_DZ + S::dz_buf(0) * S::ASZ + 4 * q4 * LDZ + 16 * g + 4 * p4));
                {   // dW_out[c][pos = lane] += sum_n dZ_out[c][n] a_last[pos][n]: 4x4x4 MFMAs over the wave's own 16 samples
                    lds_cbf* const d3_a44 = opaque((lds_cbf*)(imgw + S::OFF_D3 + (ln & 3) * 16));
                    s16x4 bh[4], ah[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bh[r] = tr4(&dz_b44[r * LDZ]);
                        ah[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[4 * r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) accWOq = mm4<F16>(ah[r], bh[r], accWOq);
                }
                {   // dA_last = W_out^T dZ_out (k = c: quarter 0 carries dZ_out in elements 0..2), dZ = dA * gelu'
                    lds_cbf* const wo_tr = opaque((lds_cbf*)(sm + S::OFF_WO + q4 * LDH + 8 * p4));
                    const float dzv[8] = {dz3[0], dz3[1], dz3[2], 0.f, 0.f, 0.f, 0.f, 0.f};
                    const bf16x8 bf = cvt8<F16>(dzv);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const s16x4 a = tr4(&wo_tr[32 * (t >> 1) + 4 * (t & 1)]);        // rows c = 0..3; the k = 4..31 part meets zeros
                        const f32x4 dl = mm16<F16>(join8(a, a), bf, f32x4(0.f));
                        dzc[t] = dl * unpack4<F16>(dpk[NH][t >> 1], t & 1);
                    }
                }
            }
            STAMP(2);
#pragma unroll
            for (int k = NH - 1; k >= 0; --k) {
                // phase j: hidden layer k: a_k -> z -> a_{k+1};  dzc = dZ of z
                const int j = NH - 1 - k;
                const int DZO = S::OFF_DZ + S::dz_buf(1 + j) * S::ASZ;
                f32x4 acc[4];
                {
                    const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
                    lds_bf* const imgw = img0 + wave * S::SPW;
                    if (S::DZB == 1) wave_lds_fence();                         // one buffer: this wave's dW_out / db reads of it are issued before it is overwritten
                    lds_cbf* const wh_tr = opaque((lds_cbf*)(sm + S::OFF_WH + k * S::WSZ + (LOFF ? lo_tH : (4 * g + q4) * LDH + 8 * p4)));
                    lds_bf* const dz_st = LOFF ? opaque(img0 + DZO + lo_rZ) : opaque(imgw + DZO + n16 * LDZ + 8 * g);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 bf = cvt_pair<F16>(dzc[2 * s], dzc[2 * s + 1]);
                        st_frag(&dz_st[32 * s], bf);
                        auto la = [&](int t) {
                            const int co = 32 * (t >> 1) + 4 * (t & 1);
                            return join8(tr4(&wh_tr[32 * s * LDH + co]), tr4(&wh_tr[(32 * s + 16) * LDH + co]));
                        };
                        if (s == 0) kstep_b<4, true, KPF, F16>(acc, bf, la);
                        else kstep_b<4, false, KPF, F16>(acc, bf, la);
                    }
                    wave_lds_fence();
                    {   // db[pos = lane] += sum_n dZ[pos][n]: 4x4x4 MFMAs against a block of ones
                        lds_cbf* const dz_b44 = LOFF ? opaque((lds_cbf*)(img0 + DZO + lo_b44)) : opaque((lds_cbf*)(imgw + DZO + 4 * q4 * LDZ + 16 * g + 4 * p4));
                        const short one = (ln & 3) == k ? (short)(F16 ? 0x3C00 : 0x3F80) : (short)0;      // A[i][.] = 1 for output row i = k only
                        const s16x4 ones = {one, one, one, one};
                        s16x4 bh[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) bh[r] = tr4(&dz_b44[r * LDZ]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) accBH = mm4<F16>(ones, bh[r], accBH);
                    }
                }
                STAMP(3 + 2 * j);
                barrier();                                                     // every wave's dZ image of this phase is in place
                if (kh == (j & 1)) {   // dW_hidden[k] tile (to, tk) += sum over the samples of all eight waves of dZ[o][n] a_k[i][n]
                    const int ln = opaque_i(lane), q4 = (ln & 15) >> 2, p4 = ln & 3, h32 = ln >> 5, cg = (ln >> 4) & 1;
                    lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + DZO + 32 * to + (LOFF ? lo_tZ : (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
                    lds_cbf* const a_t32 = opaque((lds_cbf*)(img0 + S::OFF_A + k * S::ASZ + 32 * tk + (LOFF ? lo_tZ : (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const bf16x8 a = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                        const bf16x8 b = join8(tr4(&a_t32[v * S::SPW]), tr4(&a_t32[v * S::SPW + LDZ]));
                        accW[j >> 1] = mm32<F16>(a, b, accW[j >> 1]);
                    }
                }
                if (S::DZB == 1) barrier();                                    // one buffer: everyone is done reading before it is replaced
                STAMP(4 + 2 * j);
#pragma unroll
                for (int t = 0; t < 4; ++t) dzc[t] = acc[t] * unpack4<F16>(dpk[k][t >> 1], t & 1);
            }
            // ---------- phase NH, layer 1: dX = W1^T dZ1 for the grid slots (tile t = slots 4t .. 4t+3); tiles 0 .. NG0T-1 (the G0 channels) keep
            // their running sums over the rounds in the product's C operand; the dZ1 fragments are the dZ1 image of dW1
            constexpr int DZ1 = S::OFF_DZ + S::dz_buf(1 + NH) * S::ASZ;
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
                lds_cbf* const w1_tr = opaque((lds_cbf*)(sm + S::OFF_W1 + (LOFF ? lo_t1 : (4 * g + q4) * LD1 + 8 * p4)));
                lds_bf* const dz_st = LOFF ? opaque(img0 + DZ1 + lo_rZ) : opaque(img0 + wave * S::SPW + DZ1 + n16 * LDZ + 8 * g);
                if (S::DZB == 1) wave_lds_fence();
#pragma unroll
                for (int t = NG0T; t < NDX; ++t) dxacc[t] = f32x4(0.f);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 bf = cvt_pair<F16>(dzc[2 * s], dzc[2 * s + 1]);
                    st_frag(&dz_st[32 * s], bf);
                    kstep_b<NDX, false, KPF, F16>(dxacc, bf, [&](int t) {
                        const int co = 32 * (t >> 1) + 4 * (t & 1);
                        return join8(tr4(&w1_tr[32 * s * LD1 + co]), tr4(&w1_tr[(32 * s + 16) * LD1 + co]));
                    });
                }
                // G1 slots I::NG0V .. + GQ - 1: registers of tile NG0T (and the next one when GQ = 4 .. never: NG0V is a multiple of 4, GQ <= 4)
                if constexpr (ML) {
#pragma unroll
                    for (int l = 0; l < Q::LEVELS; ++l) {
                        const int e = p.d.log2_step - 2 * l;
                        const Axis ax = axis_coords(qs[0], e), ay = axis_coords(qs[1], e);
                        const G1FactorsT<2> gf = g1_factors<2>(p.d.g1_weight_mode, ax.k1, ay.k1, 0.f);
#pragma unroll
                        for (int c8 = 0; c8 < 4; ++c8) {
                            const float w = g1_corner_factor<2>(gf, c8);
#pragma unroll
                            for (int cc = 0; cc < Q::GQ; ++cc) {
                                const int sl = I::NG0V + l * Q::GQ + cc;
                                g1s[(l * 4 + c8) * Q::GQ + cc] = fmaf(dxacc[sl >> 2][sl & 3], w, g1s[(l * 4 + c8) * Q::GQ + cc]);
                            }
                        }
                    }
                } else {
                const G1FactorsT<D> gf = g1_factors<D>(p.d.g1_weight_mode, kf[0], kf[1], kf[2]);
#pragma unroll
                for (int c8 = 0; c8 < Q::K1; ++c8) {
                    const float w = g1_corner_factor<D>(gf, c8);
#pragma unroll
                    for (int cc = 0; cc < Q::GQ; ++cc) g1s[c8 * Q::GQ + cc] = fmaf(dxacc[(I::NG0V + cc) >> 2][(I::NG0V + cc) & 3], w, g1s[c8 * Q::GQ + cc]);
                }
                }
            }
            // not hoisted: the next round's raw values are fetched HERE - in flight across the dW1 phase and the round-end barrier, where
            // few registers are live, and spread over the waves' drift - instead of at the head of the round, where all eight waves of the
            // CU would queue 36+ scattered loads each on its one texture-address path at the same moment (+6.5 K cycles per round)
            if (!HG0 || !HG1) {
                const int g = opaque_i(lane) >> 4;
                if constexpr (ML) gather_cell_ml<Q, !HG0, !HG1>(p, mo0, mo1, g, raw);
                else gather_cell_q<Q, !HG0, !HG1>(p, blk_off0, blk_off1, g, raw);
            }
            STAMP(11);
            barrier();
            {
                const int ln = opaque_i(lane), g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3, h32 = ln >> 5, cg = (ln >> 4) & 1;
                if constexpr (S::ML) {
                    // multi-level layouts: wave w owns tiles w, w + 8 (row half tile / KF, column block tile % KF) over all eight source waves
#pragma unroll
                    for (int i = 0; i < I::T1W; ++i) {
                        const int tile1 = wave + 8 * i;
                        if (tile1 < I::NT1) {                                  // wave-uniform
                            const int t1o = tile1 / KF, t1k = tile1 - t1o * KF;
                            lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + DZ1 + 32 * t1o + (LOFF ? lo_tZ : (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
                            lds_cbf* const x_t32 = opaque((lds_cbf*)(img0 + 32 * t1k + (LOFF ? lo_tX : (4 * q4 + 2 * h32) * LDX + 16 * cg + 4 * p4)));
#pragma unroll
                            for (int v = 0; v < 8; ++v) {
                                const bf16x8 a = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                                const bf16x8 b = join8(tr4(&x_t32[v * S::SPW]), tr4(&x_t32[v * S::SPW + LDX]));
                                accW[S::D1SLOT + i] = mm32<F16>(a, b, accW[S::D1SLOT + i]);
                            }
                        }
                    }
                } else if constexpr (I::NT1 <= 4) {
                    if (kh == (NH & 1) && T4 < I::NT1) {
                        // dW1, the columns of the full k-steps: tile T4 = (row half T4 / KF, column block T4 % KF) over all eight source waves
                        const int t1o = T4 / KF, t1k = T4 - t1o * KF;
                        lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + DZ1 + 32 * t1o + (LOFF ? lo_tZ : (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
                        lds_cbf* const x_t32 = opaque((lds_cbf*)(img0 + 32 * t1k + (LOFF ? lo_tX : (4 * q4 + 2 * h32) * LDX + 16 * cg + 4 * p4)));
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            const bf16x8 a = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                            const bf16x8 b = join8(tr4(&x_t32[v * S::SPW]), tr4(&x_t32[v * S::SPW + LDX]));
                            accW[NH >> 1] = mm32<F16>(a, b, accW[NH >> 1]);
                        }
                    }
                } else if (wave < I::NT1) {
                    // 6 or 8 tiles: wave w owns tile w = (row half w / KF, column block w % KF) over all eight source waves
                    const int t1o = wave / KF, t1k = wave - t1o * KF;
                    lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + DZ1 + 32 * t1o + (LOFF ? lo_tZ : (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
                    lds_cbf* const x_t32 = opaque((lds_cbf*)(img0 + 32 * t1k + (LOFF ? lo_tX : (4 * q4 + 2 * h32) * LDX + 16 * cg + 4 * p4)));
                    f32x16 c = kh == (NH & 1) ? accW[NH >> 1] : accW[S::XSLOT];
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const bf16x8 a = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                        const bf16x8 b = join8(tr4(&x_t32[v * S::SPW]), tr4(&x_t32[v * S::SPW + LDX]));
                        c = mm32<F16>(a, b, c);
                    }
                    if (kh == (NH & 1)) accW[NH >> 1] = c;
                    else accW[S::XSLOT] = c;
                }
                if constexpr (HALF) {
                    if (kh == S::TAIL_HALF) {
                        // the last 16 columns as a 16x16 tile (rows 16 T4 ..): four k-steps of 32 samples; quarter G reads source wave
                        // 2 uu + (G >> 1), samples 4 q' + 2 (G & 1) + rd
                        lds_cbf* const dz_t16 = opaque((lds_cbf*)(img0 + (g >> 1) * S::SPW + DZ1 + (4 * q4 + 2 * (g & 1)) * LDZ + 16 * T4 + 4 * p4));
                        lds_cbf* const x_t16 = opaque((lds_cbf*)(img0 + (g >> 1) * S::SPW + (4 * q4 + 2 * (g & 1)) * LDX + 32 * KF + 4 * p4));
#pragma unroll
                        for (int uu = 0; uu < 4; ++uu) {
                            const bf16x8 a = join8(tr4(&dz_t16[2 * uu * S::SPW]), tr4(&dz_t16[2 * uu * S::SPW + LDZ]));
                            const bf16x8 b = join8(tr4(&x_t16[2 * uu * S::SPW]), tr4(&x_t16[2 * uu * S::SPW + LDX]));
                            accT = mm16<F16>(a, b, accT);
                        }
                    }
                }
            }
            STAMP(14);
            barrier();   // all reads of dZ1 / X / A_k done before the next round overwrites them
            STAMP(15);
        }  // rounds of one macro-tile

        // ---------- flush of the cell's gradient sums
        if constexpr (TRAIN && ML) {
            flush_ml<Q, NDX>(p, mo0, mo1, dxacc, g1s, opaque_i(lane), F16 ? p.dz_unscale : 1.0f);
        } else if constexpr (TRAIN) {
            const int ln = opaque_i(lane), g = ln >> 4;
            combine_g1_lanes_q<Q>(g1s, blk_off1, blk, ln, lw, packed, pk, pk_lc);
            bool flush = true;
            constexpr int REGION = S::SPW / 2;                             // floats per wave
            if (NIC_GROUP_SUM && rg > 0) {                                  // segment-uniform: groups of one macro-tile sit in one workgroup
                lds_f* const reg0 = (lds_f*)img0;
                static_assert((I::NG0V + I::NG1V) * 64 * 2 <= S::SCRATCH, "group-sum scratch");
                int leader = wave;
                if (tile_ok)
                    for (int w = wave - 1; w >= 4 * kh; --w)
                        if (seg_tile0 + ((base + w) >> rg) == tile) leader = w;
                if (leader != wave) {
                    lds_f* const mine = opaque(reg0 + wave * REGION + ln);
#pragma unroll
                    for (int i = 0; i < I::NG0V; ++i) mine[i * 64] = dxacc[i >> 2][i & 3];
#pragma unroll
                    for (int i = 0; i < I::NG1V; ++i) mine[(I::NG0V + i) * 64] = g1s[i];
                }
                barrier();
                if (leader == wave && tile_ok) {
                    for (int w = wave + 1; w < 4 * kh + 4; ++w) {
                        if (base + w >= t_end || seg_tile0 + ((base + w) >> rg) != tile) break;
                        lds_cf* const theirs = opaque(reg0 + w * REGION + ln);
#pragma unroll
                        for (int i = 0; i < I::NG0V; ++i) dxacc[i >> 2][i & 3] += theirs[i * 64];
#pragma unroll
                        for (int i = 0; i < I::NG1V; ++i) g1s[i] += theirs[(I::NG0V + i) * 64];
                    }
                }
                barrier();
                flush = leader == wave;
            }
            constexpr bool SHARE_XY = !Q::TETRA;                            // method 4's tetrahedra share no corner along x or y
            if (SHARE_XY && NIC_Q16_PREADD && flush && !packed) preadd_x_q<NG0T>(dxacc, blk_off0, ln);
            if (SHARE_XY && NIC_Q16_PREADD >= 2 && rg == 0 && !packed && (p.preadd_y || !NIC_PREADD_Y_SMALL)) {    // segment-uniform
                static_assert((I::NG0V + 1) * 64 * 2 <= S::SCRATCH, "pre-add scratch");
                preadd_y_q<NG0T>(dxacc, blk_off0, (uint32_t)p.g0.nx, ln, wave, (lds_f*)img0, REGION, barrier);
            }
            if (flush) {
                const uint32_t pb0 = (uint32_t)p.g0.plane * 4u, pb1 = (uint32_t)p.g1.plane * 4u;
#pragma unroll
                for (int e = 0; e < Q::NG0; ++e) {
                    uint32_t nz0 = 0u;
#pragma unroll
                    for (int c = 0; c < Q::C; ++c) nz0 |= __builtin_bit_cast(uint32_t, dxacc[(e * Q::C + c) >> 2][(e * Q::C + c) & 3]);
                    if ((nz0 << 1) != 0u) {
                        int dx, dy, dz;
                        q_g0_corner<Q>(g, e, dx, dy, dz);
                        uint32_t ob = (blk_off0 + (uint32_t)p.g0.at(dx, dy, dz)) * 4u;
                        char* gbase = reinterpret_cast<char*>(p.g0_grad);
#pragma unroll
                        for (int c = 0; c < Q::C; ++c, ob += pb0) atomicAdd(reinterpret_cast<float*>(gbase + ob), F16 ? dxacc[(e * Q::C + c) >> 2][(e * Q::C + c) & 3] * p.dz_unscale : dxacc[(e * Q::C + c) >> 2][(e * Q::C + c) & 3]);
                    }
                }
                uint32_t nz1 = 0u;
#pragma unroll
                for (int i = 0; i < I::NG1V; ++i) nz1 |= __builtin_bit_cast(uint32_t, g1s[i]);
                if ((nz1 << 1) != 0u) {
#pragma unroll
                    for (int c8 = 0; c8 < Q::K1; ++c8) {
                        int dx, dy, dz;
                        q_g1_corner<Q>(c8, dx, dy, dz);
                        uint32_t ob = (blk_off1 + (uint32_t)p.g1.at(dx, dy, dz)) * 4u + (uint32_t)(Q::GQ * g) * pb1;
                        char* gbase = reinterpret_cast<char*>(p.g1_grad);
#pragma unroll
                        for (int cc = 0; cc < Q::GQ; ++cc, ob += pb1) atomicAdd(reinterpret_cast<float*>(gbase + ob), F16 ? g1s[c8 * Q::GQ + cc] * p.dz_unscale : g1s[c8 * Q::GQ + cc]);
                    }
                }
            }
        }
        STAMP(13);
    }  // macro-tile loop
  }  // segments
#ifdef NIC_STAMPS
    auto dump_stamps = [&]() {
        if (NL == 3) {
            __builtin_amdgcn_s_waitcnt(0);                                      // the record stores have left
            unsigned long long t_;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
            stamp_sum[10] = t_ - stamp_last;
        }
        if (lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.partials + (size_t)gridDim.x * S::REC) + ((size_t)blockIdx.x * 8 + wave) * 16;   // behind the records (nic_workspace_bytes leaves 1 MiB)
#pragma unroll
            for (int i = 0; i < NIC_NPH; ++i) dst[i] = stamp_sum[i];
        }
    };
    if (!TRAIN) dump_stamps();
#endif

    if (!TRAIN) return;
    // ---------------- one record per workgroup
    float* rec = p.partials + (int64_t)blockIdx.x * S::REC;
    const float us = F16 ? p.dz_unscale : 1.0f;                      // fp16 products: every sum of dZ products carries the loss scale 2^k; 2^-k is exact
#pragma unroll
    for (int k = 0; k < S::NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[S::REC_W + (wave * S::NACC + k) * 1024 + r * 64 + lane] = accW[k][r] * us;
    if constexpr (HALF) {
        if (kh == S::TAIL_HALF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[S::REC_TAIL + wave * 256 + r * 64 + lane] = accT[r] * us;
        }
    }
    float* tail = rec + S::REC_WAVE + wave * S::WTAIL;
#pragma unroll
    for (int k = 0; k < NH; ++k) tail[k * 64 + lane] = accBH[k] * us;
#pragma unroll
    for (int c = 0; c < 3; ++c) tail[NH * 64 + 64 * c + lane] = accWOq[c] * us;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = c < 3 ? accBO[c] * us : accLoss;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) tail[NH * 64 + 192 + c] = v;
    }
#ifdef NIC_STAMPS
    dump_stamps();
#endif
}

// =====================================================================================================
// Fixed-order reduction of the records of fused_q16_kernel.  Output index space, layer by layer:
// dW1 (+ db1 on its constant-one column) in RECORD order - the 2 KF 32x32 tiles of the full k-steps, element by element, then the 64 x 16 block of the
// half k-step - | (W_hidden [64][64] | b_hidden [64]) x NH | W_out [3][64] | b_out [3] | loss.
// (Round 4: the dW1 part used to walk nn.Linear order and SEARCH the slot of every channel (rho_of_channel: up to KP evaluations of slot_channel per
// thread) - 66 us for the multi-level layouts, whose runs of consecutive channels are 4 long, against 14 us for the same records read in record order.)
template <class Q>
__host__ __device__ constexpr int reduce_q16_w1_outputs() { return QInfo<Q>::NT1 * 1024 + (QInfo<Q>::HALF ? kH * 16 : 0); }
template <class Q, int NL>
__host__ __device__ constexpr int reduce_q16_outputs() { return reduce_q16_w1_outputs<Q>() + (NL - 2) * (kH * kH + kH) + 3 * kH + 3 + 1; }
template <class Q, int NL>
__global__ void __launch_bounds__(256) reduce_q16_kernel(const float* partials, int n_rec, nic_mlp_grads gr, float* loss, float loss_scale, const StepTail tl) {
    if (tail_block(tl)) return;                                   // a streaming block of the optimiser tail (nic_adam.hpp)
    using S = LdsQ<Q, NL>;
    using I = QInfo<Q>;
    constexpr int NH = S::NH, KF = I::KF;
    constexpr int N_W1 = reduce_q16_w1_outputs<Q>(), N_HID = kH * kH + kH;
    constexpr int N_OUT = reduce_q16_outputs<Q, NL>();
    constexpr int OUTS = 256 / NIC_RQ_SLICES;                     // outputs per block
    __shared__ float red[NIC_RQ_SLICES][OUTS];
    const int slice = threadIdx.x / OUTS, oi = threadIdx.x % OUTS;
    const int gid = blockIdx.x * OUTS + oi;
    const bool live = gid < N_OUT;
    int nsrc = 0, off0 = 0, stride = 0;
    float* dst = nullptr;
    auto tile32 = [](int i, int j) { return ((i & 3) + 4 * (i >> 3)) * 64 + j + 32 * ((i >> 2) & 1); };      // element (row i, column j) of a 32x32 accumulator tile
    if (live) {
        int t = gid;
        if (t < N_W1) {
            int po, r;
            nsrc = 1;
            if (t < I::NT1 * 1024) {
                // element e of 32x32 tile `tile` = (row half to, column block tk): register e >> 6 of lane e & 63 is row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), column lane & 31
                const int tile = t >> 10, e = t & 1023, reg = e >> 6, ln = e & 63;
                po = 32 * (tile / KF) + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
                r = 32 * (tile % KF) + (ln & 31);
                // phase NH: the tile's owner wave and accumulator slot (LdsQ)
                int w, slot;
                if (S::ML) { w = tile & 7; slot = S::D1SLOT + (tile >> 3); }
                else if (I::NT1 <= 4) { w = 4 * (NH & 1) + tile; slot = NH >> 1; }
                else { w = tile; slot = (w >> 2) == (NH & 1) ? NH >> 1 : S::XSLOT; }
                off0 = S::REC_W + (w * S::NACC + slot) * 1024 + e;
            } else {
                // the half k-step's 16 columns: 16x16 tiles, register v >> 6 of lane v & 63 is row (v >> 6) + 4 ((v >> 4) & 3), column v & 15
                const int u = t - I::NT1 * 1024, q = u >> 8, v = u & 255;
                po = 16 * q + 4 * ((v >> 4) & 3) + (v >> 6);
                r = 32 * KF + (v & 15);
                off0 = S::REC_TAIL + (4 * S::TAIL_HALF + q) * 256 + v;
            }
            const int o = hid16(po), ch = r < I::KP ? I::channel_of_rho(r) : kSlotZero;
            if (ch >= 0) dst = gr.w[0] ? gr.w[0] + o * Q::CIN + ch : nullptr;
            else if (ch == kSlotOne) dst = gr.b[0] ? gr.b[0] + o : nullptr;                  // db1 rode through dW1 on the constant-one column
        } else {
            t -= N_W1;
            if (t < NH * N_HID) {
                const int k = t / N_HID, u = t - k * N_HID;
                if (u < kH * kH) {
                    const int o = u / kH, i = u - o * kH;
                    const int po = pos16(o), pi = pos16(i);
                    const int j = NH - 1 - k;                                   // the layer's phase: owner half j & 1, slot j >> 1
                    off0 = S::REC_W + ((4 * (j & 1) + 2 * (po >> 5) + (pi >> 5)) * S::NACC + (j >> 1)) * 1024 + tile32(po & 31, pi & 31);
                    nsrc = 1;
                    dst = gr.w[1 + k] ? gr.w[1 + k] + u : nullptr;
                } else {
                    off0 = S::REC_WAVE + k * 64 + pos16(u - kH * kH); nsrc = 8; stride = S::WTAIL;
                    dst = gr.b[1 + k] ? gr.b[1 + k] + (u - kH * kH) : nullptr;
                }
            } else {
                t -= NH * N_HID;
                nsrc = 8; stride = S::WTAIL;
                if (t < 3 * kH) {
                    const int c = t / kH, i = t - c * kH;
                    off0 = S::REC_WAVE + NH * 64 + 64 * c + pos16(i);
                    dst = gr.w[NL - 1] ? gr.w[NL - 1] + t : nullptr;
                } else if (t < 3 * kH + 3) {
                    off0 = S::REC_WAVE + NH * 64 + 192 + (t - 3 * kH);
                    dst = gr.b[NL - 1] ? gr.b[NL - 1] + (t - 3 * kH) : nullptr;
                } else {
                    off0 = S::REC_WAVE + NH * 64 + 195;
                    dst = loss;
                }
            }
        }
    }
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // fixed summation tree: bit-stable for a given grid size
    const int per = (n_rec + NIC_RQ_SLICES - 1) / NIC_RQ_SLICES;
    const int w_lo = slice * per, w_hi = (w_lo + per < n_rec) ? w_lo + per : n_rec;
    if (live && dst != nullptr) {
        for (int k = 0; k < nsrc; ++k) {
            const float* src = partials + off0 + k * stride;
            int w = w_lo;
            for (; w + 8 <= w_hi; w += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) part[j] += src[(int64_t)(w + j) * S::REC];
            }
            for (; w < w_hi; ++w) part[0] += src[(int64_t)w * S::REC];
        }
    }
    red[slice][oi] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    __syncthreads();
    if (slice != 0 || !live || dst == nullptr) return;
    float acc = red[0][oi];
#pragma unroll
    for (int sl = 1; sl < NIC_RQ_SLICES; ++sl) acc += red[sl][oi];
    if (gid == N_OUT - 1) *dst = acc * loss_scale;
    else tail_store(tl, dst, acc);
}

}  // namespace nic
