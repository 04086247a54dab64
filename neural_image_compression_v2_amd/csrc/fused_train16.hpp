// Two waves per SIMD: the 2D split-bf16 training kernel as 8 waves x 16 samples (fused_train16_kernel<Layout, MODE>).
//
// fused_kernel (fused_kernel.hpp) carries 32 samples per wave on 32x32x16 MFMAs and needs ~490 registers: ONE wave per SIMD, which
// issues a vector instruction every ~5 cycles where the SIMD could execute one every 2.  This kernel halves the
// tile: a wave owns 16 samples, every chained product runs on v_mfma_f32_16x16x32_bf16 (D: lane l = (sample n = l & 15, quarter
// g = l >> 4) holds rows 4g .. 4g+3 of a 16-row tile), a workgroup is 8 waves = two per SIMD, <= 256 registers per lane.
//
// Layout rules (all checked against the fp32 kernel and the CPU oracle by tests/test_gpu_parity.py):
//   * quarter g of sample n owns 24 input slots: G0 corner g (12 channels), G1 channels 3g..3g+2 (blended), one zero, PE rows
//     3g..3g+2, LOD (g = 0) / the constant one that carries b1 (g = 1), zeros.  Slots 8s..8s+7 of the four quarters are the B
//     operand of k-step s as they stand (k = 8g + j): the internal input row of (slot, g) is rho = 32 (slot >> 3) + 8g + (slot & 7),
//     the third k-step (slots 16..19) is stored compactly as rho = 64 + 4g + (slot - 16): 80 columns;
//   * hidden activations: register r of row tile t of quarter g = hidden unit 16t + 4g + r; as the B operand of the next product
//     the eight registers of tiles 2s, 2s+1 are k-step s, i.e. hidden unit h sits at POSITION pos(h) = 32 (t >> 1) + 8g + 4 (t & 1) + r
//     of the contraction.  Every LDS image whose columns are hidden units (W2, W3, the a1 / a2 / dZ images) is stored in position
//     order, so a forward A fragment is ONE ds_read_b128 and a fragment store ONE ds_write_b128;
//   * transposed products (dA2, dA1, dX) read their A fragments from the same weight images with ds_read_b64_tr_b16;
//   * weight-gradient products contract over the samples of all 8 waves ([sample][position] bf16 images, hi + lo, of every wave):
//     dW2 and the first 64 columns of dW1 as 32x32 tiles on v_mfma_f32_32x32x16_bf16, wave w owning tile w & 3 over the samples of
//     waves 4 (w >> 2) .. + 3 (two partial sums per tile, added by reduce16_kernel); the last 16 columns of dW1 as 16x16 tiles
//     (wave w: rows 16 (w & 3).., k-steps 2 (w >> 2), + 1); dW3 and db2 per wave over its own samples on 4x4x4 MFMAs.
//   * everything else - cell-major lanes, per-cell register sums flushed once with fp32 atomics, G1 lane combine, staggered flush
//     phases, round groups, edge tiles, XCD-aware persistent walk, fixed-order partial reduction - is the scheme of fused_kernel.hpp.
#pragma once
#include "fused_kernel.hpp"


#ifndef NIC_T16_HALF16
#define NIC_T16_HALF16 1
#endif
#ifndef NIC_T16_LOFF
#define NIC_T16_LOFF 26        // rZ | b44 | tZ (bits 1, 3, 4): 4K launch 2.120 -> 2.078 ms (interleaved A/B, 4 rounds); all six offsets 2.078 with 3 spilled registers, rZ alone nothing
#endif
#ifndef NIC_T16_PIN
#define NIC_T16_PIN 0        // bit 0: pin d1, bit 1: pin d2, bit 2: pin d2 in the sinusoidal-PE layout only (all measured: no gain in this kernel, see below)
#endif
namespace nic {

struct Lds16 {
    static constexpr int LD1 = 80, LD2 = 80, LD3 = 80;     // weight images (bf16 elements): 160-byte rows, conflict-free b128 row reads
    static constexpr int LDZ = 72, LDX = 88;               // wave images: conflict-free fragment stores and 32x32x16 transposed reads
    static constexpr int KP = 80;                          // compact input columns (rho)
    static constexpr int OFF_W1 = 0;                       // hi [64][LD1], lo [64][LD1]
    static constexpr int W1LO = kH * LD1;
    static constexpr int OFF_W2 = OFF_W1 + 2 * kH * LD1;
    static constexpr int W2LO = kH * LD2;
    static constexpr int OFF_W3 = OFF_W2 + 2 * kH * LD2;   // hi [4][LD3] (row 3 = zeros), lo [4][LD3]
    static constexpr int W3LO = 4 * LD3;
    static constexpr int OFF_B = OFF_W3 + 2 * 4 * LD3;     // fp32: b2 [64], b3 [4]  (136 bf16 elements, padded to 144)
    static constexpr int OFF_IMG = OFF_B + 144;
    // per wave: DZ hi, lo | A1 hi, lo | X hi, lo | D3 hi [4][16], lo
    static constexpr int DZLO = 16 * LDZ, OFF_A1 = 2 * 16 * LDZ, OFF_X = 4 * 16 * LDZ, XLO = 16 * LDX, OFF_D3 = OFF_X + 2 * 16 * LDX;
    static constexpr int SPW = OFF_D3 + 128;
    static constexpr int TOTAL = OFF_IMG + 8 * SPW;         // bf16 elements
    static_assert(TOTAL * 2 <= 163840, "LDS");
    static_assert(OFF_IMG % 8 == 0 && SPW % 8 == 0 && OFF_B % 8 == 0, "16-byte alignment");
    // per-workgroup record of decoder-gradient partials (floats)
    static constexpr int REC_W2 = 0, REC_W1 = 8 * 1024, REC_TAIL = 16 * 1024, REC_WAVE = REC_TAIL + 8 * 256, REC = REC_WAVE + 8 * 320;
};

// decoder-input channel of slot sigma of quarter g (image_compression.py:94-96 channel order)
__host__ __device__ constexpr int slot16_channel(int s, int g) {
    if (s < 12) return 12 * g + s;                      // G0 corner g, channel s
    if (s < 15) return 48 + 3 * g + (s - 12);           // G1 channel 3g + ..
    if (s == 15) return kSlotZero;
    if (s < 19) return 60 + 3 * g + (s - 16);           // PE row 3g + ..
    if (s == 19) return g == 0 ? 72 : (g == 1 ? kSlotOne : kSlotZero);
    return kSlotZero;
}
// channel (or kSlotOne / kSlotZero) of compact input column rho
__host__ __device__ constexpr int channel_of_rho16(int rho) {
    if (rho < 64) return slot16_channel(8 * (rho >> 5) + (rho & 7), (rho >> 3) & 3);
    return slot16_channel(16 + (rho & 3), (rho - 64) >> 2);
}
__host__ __device__ constexpr int rho16_of_channel(int ch) {       // ch: 0..72, or kSlotOne
    for (int rho = 0; rho < Lds16::KP; ++rho)
        if (channel_of_rho16(rho) == ch) return rho;
    return -1;
}
// position of hidden unit h in the contraction order of the chained products, and its inverse
__host__ __device__ constexpr int pos16(int h) { return 32 * (h >> 5) + 8 * ((h >> 2) & 3) + 4 * ((h >> 4) & 1) + (h & 3); }
__host__ __device__ constexpr int hid16(int pos) { return 16 * (2 * (pos >> 5) + ((pos >> 2) & 1)) + 4 * ((pos >> 3) & 3) + (pos & 3); }

typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
typedef __attribute__((address_space(3))) u32x4 lds_u4;
__device__ __forceinline__ bf16x8 ld_frag(lds_cbf* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<lds_cu4*>(p)); }       // ds_read_b128
__device__ __forceinline__ void st_frag(lds_bf* p, const bf16x8& f) { *reinterpret_cast<lds_u4*>(p) = __builtin_bit_cast(u32x4, f); }   // ds_write_b128
__device__ __forceinline__ bf16x8 half_frag(s16x4 a) {               // k = 0..3 real, 4..7 zero
    const s16x8 v = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, v);
}
// The compact half k-step (input columns 64 .. 79: four slots per lane quarter) is exactly the operand layout of v_mfma_f32_16x16x16_bf16 (quarter g holds
// k = 4g .. 4g+3): no zero-padded 8-element fragments to assemble (4 register moves per fragment: ~ 40 per round in the split kernel) and half the pipe time.
struct Half2 {
    s16x4 hi, lo;
};
__device__ __forceinline__ f32x4 mfma16h_bf(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16h_split(const Half2& a, const Half2& b, f32x4 c) {
    c = mfma16h_bf(a.lo, b.hi, c);
    c = mfma16h_bf(a.hi, b.lo, c);
    return mfma16h_bf(a.hi, b.hi, c);
}
__device__ __forceinline__ f32x4 mfma16_split(const Frag2& a, const Frag2& b, f32x4 c) {
    c = mfma16_bf(a.lo, b.hi, c);
    c = mfma16_bf(a.hi, b.lo, c);
    return mfma16_bf(a.hi, b.hi, c);
}
__device__ __forceinline__ Frag2 split_pair(const f32x4& a, const f32x4& b) {       // registers of row tiles 2s, 2s + 1 = k-step s
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return split8(x);
}
__device__ __forceinline__ lds_bf* opaque(lds_bf* p) {
#ifndef NIC_NO_OPAQUE_PTR
    asm volatile("" : "+v"(p));
#endif
    return p;
}

// acc[t] += A_t x B for the NT row tiles of one k-step; the A fragments run PF tiles ahead of the MFMAs that consume them
#ifndef NIC_T16_PF
#define NIC_T16_PF 2
#endif
// ZERO: the first k-step of a product starts from C = 0 (an inline constant of the MFMA: no accumulator initialisation moves)
template <int NT, bool ZERO = false, class LoadA>
__device__ __forceinline__ void kstep16(f32x4 (&acc)[NT], const Frag2& bf, LoadA&& load_a) {
    constexpr int PF = NIC_T16_PF < NT ? NIC_T16_PF : NT;
    Frag2 af[PF];
#pragma unroll
    for (int t = 0; t < PF; ++t) af[t] = load_a(t);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = mfma16_split(af[t % PF], bf, ZERO ? f32x4(0.f) : acc[t]);
        if (t + PF < NT) af[t % PF] = load_a(t + PF);
    }
}

// raw grid values of the lane's cell: G0 corner g (12 channels), G1 channels 3g..3g+2 at the 4 corners - gathered once per macro-tile
// (every sample a lane handles inside one macro-tile lies in the same G0 cell and the same G1 cell)
struct CellRaw16 {
    float g0[kC];
    float g1[4 * 3];     // [corner q][cc]
};
struct GridAcc16 {
    float g1[4 * 3];     // [corner q][cc]; the G0 sums live in the dX accumulator tiles
};

// Lane ids are re-derived from an opaque copy of the lane number at the start of every phase: everything computed from them
// (LDS addresses, PE constants, select masks) is loop-invariant, and left alone the optimiser hoists it all out of the round loop
// and the macro-tile loop - about 70 registers more than the 256 that two waves per SIMD allow, spilled to scratch.
__device__ __forceinline__ int opaque_i(int v) {
#ifndef NIC_NO_OPAQUE_LANE
    asm volatile("" : "+v"(v));
#endif
    return v;
}

// Addressing of the once-per-macro-tile gathers and atomics: one wave-uniform tensor base (scalar registers) + a 32-bit byte offset
// per lane that walks the channel planes (the per-plane scalar bases of fused_kernel cost ~60 scalar registers, which this kernel
// does not have: they were spilled to vector lanes and read back with v_readlane in every round).
// one grid element at byte offset ob: fp32, or 16-bit storage widened to fp32 (p.grid_kind, launch-uniform)
// kind 3: the stored uint8 codec (save4fp, models.py:61-64), dequantised exactly like load4fp (models.py:68-71; see grid_elem)
__device__ __forceinline__ float grid_load16(const char* base, uint32_t ob, int kind, const FusedParams& p) {
    if (kind == 0) return *reinterpret_cast<const float*>(base + ob);
    if (kind == 3) {
        const float n = (float)*reinterpret_cast<const uint8_t*>(base + ob) - p.dq_sub;
        const float q = mul_rn(n, p.dq_rcp);
        return fmaf(fmaf(-q, p.dq_den, n), p.dq_rcp, q);
    }
    const uint32_t hbits = *reinterpret_cast<const uint16_t*>(base + ob);
    if (kind == 1) return __builtin_bit_cast(float, hbits << 16);                          // bfloat16
    return (float)__builtin_bit_cast(_Float16, (uint16_t)hbits);                            // IEEE half
}
__device__ __forceinline__ void gather_cell16(const FusedParams& p, uint32_t off0, uint32_t off1, int g, CellRaw16& raw) {
    const int kind = p.grid_kind;
    const uint32_t eb = kind == 0 ? 4u : (kind == 3 ? 1u : 2u);   // bytes per stored element
    const uint32_t pb0 = (uint32_t)p.g0.plane * eb, pb1 = (uint32_t)p.g1.plane * eb;
    {
        uint32_t ob = (off0 + (uint32_t)p.g0.at(g >> 1, g & 1, 0)) * eb;
        const char* base = reinterpret_cast<const char*>(p.g0.p);
#pragma unroll
        for (int c = 0; c < kC; ++c, ob += pb0) raw.g0[c] = grid_load16(base, ob, kind, p);
    }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        uint32_t ob = (off1 + (uint32_t)p.g1.at(c4 >> 1, c4 & 1, 0)) * eb + (uint32_t)(3 * g) * pb1;
        const char* base = reinterpret_cast<const char*>(p.g1.p);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc, ob += pb1) raw.g1[c4 * 3 + cc] = grid_load16(base, ob, kind, p);
    }
}

// the lane's 19 real slots for the sample at absolute coordinates q: xs[0..11] G0 corner g, xs[12..14] G1 channels 3g.. (blended),
// xs[15..17] PE rows 3g.., xs[18] LOD (g = 0) / the constant one (g = 1).  Slot 15 and slots 20..23 are zero and never materialised.
template <class L>
__device__ __forceinline__ void encode16(const FusedParams& p, const int (&q)[3], int g, float (&xs)[20], EncCtx& cx, const CellRaw16& raw) {
    const nic_path_desc& d = p.d;
    const int e = d.log2_step;
    const Axis ax = axis_coords(q[0], e), ay = axis_coords(q[1], e);
    cx.kx = ax.k1; cx.ky = ay.k1; cx.kz = 0.f;
#pragma unroll
    for (int c = 0; c < kC; ++c) xs[c] = raw.g0[c];
    // --- PE rows 3g .. 3g+2: dimension g >> 1, rows 3 (g & 1) + i of its block (utils.py:198-227); LOD / the constant one
    {
        const float c = (g >> 1) ? ay.t1 : ax.t1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r = 3 * (g & 1) + i;
            float v;
            if (L::PE == NIC_PE_TRIANGULAR) {
                v = tri_pe_row(c, r, kP);
            } else {
                const int k = r >> 1;
                const float dv = k == 0 ? d.pe_div[0] : (k == 1 ? d.pe_div[1] : d.pe_div[2]);
                float sv, cv;
                sincos_cw(mul_rn(c, dv), sv, cv);
                v = (r & 1) ? cv : sv;
            }
            xs[15 + i] = v;
        }
        xs[18] = g == 0 ? d.lod_value : (g == 1 ? 1.0f : 0.f);
    }
    // --- G1 blend with the reference's factor order (fp_def.py:141-144)
    const G1FactorsT<2> gf = g1_factors<2>(d.g1_weight_mode, cx.kx, cx.ky, 0.f);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        float sum = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const uint32_t b = (gf.bits >> (3 * c4)) & 7u;
            float v = raw.g1[c4 * 3 + cc];
            v = mul_rn(v, (b & 1u) ? gf.fx[1] : gf.fx[0]);
            v = mul_rn(v, (b & 2u) ? gf.fy[1] : gf.fy[0]);
            sum = c4 == 0 ? v : add_rn(sum, v);
        }
        xs[12 + cc] = sum;
    }
}
// xs index of slot sigma
__host__ __device__ constexpr int xs_of_slot(int s) { return s < 15 ? s : s - 1; }

// The in-kernel noise of the 2D kernels (nic_device.hpp::noise_field, oracle/nic_oracle.py::kernel_noise): quarter g consumes
// exactly generator block g of its sample - fields 0..11 the G0 channels, 12..14 the G1 channels, 15..17 the PE rows, 18 the LOD.
template <class L>
__device__ __forceinline__ void add_noise16(const NoiseSrc& ns, uint64_t sample_global, int64_t n_local, int g, float (&xs)[20]) {
    if (ns.mode == NIC_NOISE_NONE) return;
    if (ns.mode == NIC_NOISE_TENSOR) {
        const float* row = ns.tensor + n_local * L::CIN;
#pragma unroll
        for (int s = 0; s < 20; ++s) {
            if (s == 15) continue;
            const int ch0 = slot16_channel(s, 0);
            if (s == 19) { xs[18] += g == 0 ? row[72] : 0.f; continue; }
            const int stride = slot16_channel(s, 1) - ch0;        // the channel is affine in g for every other real slot
            xs[xs_of_slot(s)] += row[ch0 + stride * g];
        }
        return;
    }
    const U4 b = noise_block(ns, sample_global, g);
#pragma unroll
    for (int f = 0; f < 18; ++f) xs[f] += noise_field(ns, b, f);
    const float nl = noise_field(ns, b, 18);
    xs[18] += g == 0 ? nl : 0.f;
}

// G1 sums of the lanes whose G0 cells share a G1 cell are added across lanes before the flush (see combine_g1_lanes)
__device__ __forceinline__ void combine_g1_lanes16(GridAcc16& ga, uint32_t off1, const int (&blk)[3], int lane, int lw) {
    const int T[2] = {1 << lw, 16 >> lw};
    const int STR[2] = {1, 1 << lw};
    const int pl = lane & 15;
    const int lc[2] = {pl & (T[0] - 1), pl >> lw};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const bool odd = blk[a] & 1;
        const int pc = lc[a] + (odd ? -1 : 1);
        const bool inb = pc >= 0 && pc < T[a];
        const int partner = inb ? lane + (odd ? -STR[a] : STR[a]) : lane;
        const bool pair = inb && (uint32_t)__shfl((int)off1, partner) == off1;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float pv = __shfl(ga.g1[i], partner);
            ga.g1[i] = pair ? (odd ? 0.f : ga.g1[i] + pv) : ga.g1[i];
        }
    }
}

// Neighbour pre-add of a quarter-layout kernel's G0 sums along x, inside a wave: the dx = 1 corner of cell n - 1 IS the dx = 0 corner of
// cell n whenever their cell offsets differ by one element (same row of the grid, no clamping in between), so the lane of quarter (1, dy)
// hands its 12 sums to the lane of quarter (0, dy) of the next cell and issues nothing: 34 instead of 64 atomics per channel and 16 x 1 tile
// (the flush is bound by the memory-side atomic units whenever many workgroups flush a small grid at the same time).
__device__ __forceinline__ void preadd_x16(f32x4 (&dxacc)[4], uint32_t off0, int ln) {
    const int n16 = ln & 15, g = ln >> 4;
    const bool recv = g < 2;
    const bool inb = recv ? n16 >= 1 : n16 <= 14;
    const int partner = inb ? (recv ? ln + 31 : ln - 31) : ln;
    const uint32_t poff = (uint32_t)__shfl((int)off0, partner);
    const bool pair = inb && (recv ? off0 == poff + 1u : poff == off0 + 1u);
#pragma unroll
    for (int c = 0; c < kC; ++c) {
        const float pv = __shfl(dxacc[c >> 2][c & 3], partner);
        dxacc[c >> 2][c & 3] = pair ? (recv ? dxacc[c >> 2][c & 3] + pv : 0.f) : dxacc[c >> 2][c & 3];
    }
}
// .. and along y, across the NW waves of the workgroup: consecutive waves hold consecutive macro-tiles, i.e. rows y, y + 1, .. of the same
// 16 columns, and the dy = 1 corners of wave w are the dy = 0 corners of wave w + 1 whenever the cell offsets differ by one grid row.  Sums
// and offsets go through the wave regions of the LDS (`region` floats per wave, free between the last barrier of a unit and the first
// store of the next); two workgroup barriers, executed by every wave.
template <int NW, class Barrier>
__device__ __forceinline__ void preadd_y16(f32x4 (&dxacc)[4], uint32_t off0, uint32_t row, int ln, int wave, lds_f* reg0, int region, Barrier&& barrier) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    const int g = ln >> 4;
    lds_f* const mine = opaque(reg0 + wave * region + ln);
    const bool up = (g & 1) != 0;                                  // dy = 1 quarters hand over, dy = 0 quarters receive
    ((lds_u32_t*)mine)[12 * 64] = off0;
    if (up) {
#pragma unroll
        for (int i = 0; i < 12; ++i) mine[i * 64] = dxacc[i >> 2][i & 3];
    }
    barrier();
    const int pw = up ? wave + 1 : wave - 1;
    const bool inw = pw >= 0 && pw < NW;
    lds_cf* const theirs = opaque(reg0 + (inw ? pw : wave) * region + (up ? ln - 16 : ln + 16));
    const uint32_t poff = ((const lds_u32_t*)theirs)[12 * 64];
    const bool pair = inw && (up ? poff == off0 + row : off0 == poff + row);
    if (!up) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float pv = theirs[i * 64];
            dxacc[i >> 2][i & 3] += pair ? pv : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) dxacc[i >> 2][i & 3] = pair ? 0.f : dxacc[i >> 2][i & 3];
    }
    barrier();
}

// =====================================================================================================
#ifndef NIC_T16_PREADD
#define NIC_T16_PREADD 2          // 0: every lane flushes its own sums; 1: pre-add along x inside a wave; 2: and along y across the waves of a workgroup
#endif
#ifndef NIC_T16_LB
#define NIC_T16_LB 512
#endif
// (measured and dropped, DESIGN 8: a scheduling barrier between the phases of a round +4 %; per-half LDS-counter barriers with skewed halves; static /
//  alternating / time-sliced wave priorities; the noise as a vector behind the mode switch; the generator branch first)
constexpr int MODE_TRAIN_RGBX = 4;       // this kernel's own mode: MODE_TRAIN_IMG with an interleaved uint8 RGBX target (nic_target_image.is_u8 == 2)
template <class L, int MODE>
__global__ void __launch_bounds__(NIC_T16_LB) fused_train16_kernel(FusedParams p) {
    using S = Lds16;
    static_assert(L::DIM == 2 && L::CIN == 73, "the 16-sample training kernel is built for the 2D slot layouts");
    static_assert(MODE != MODE_INFER, "training kernel");
    constexpr int LD1 = S::LD1, LD2 = S::LD2, LD3 = S::LD3, LDZ = S::LDZ, LDX = S::LDX;
    __shared__ __attribute__((aligned(16))) __bf16 smem16[S::TOTAL];
    lds_bf* const sm = (lds_bf*)smem16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: scalar registers, scalar branches

    // ---------------- prologue: decoder weights -> bf16 hi / lo images (W1 in compact rho order with b1 in the column of the
    // constant-one slot; W2 / W3 columns in position order), per-wave images zeroed
    stage_all<kH * LD1, 512>(tid,
        [&](int idx) {
            const int o = idx / LD1, rho = idx - o * LD1;
            const int ch = channel_of_rho16(rho);
            const float* src = ch >= 0 ? &p.W[0][o * L::CIN + ch] : &p.b[0][o];
            const float v = *src;
            return (ch >= 0 || ch == kSlotOne) ? v : 0.f;
        },
        [&](int idx, float v) {
            const __bf16 hi = (__bf16)v;
            sm[S::OFF_W1 + idx] = hi;
            sm[S::OFF_W1 + S::W1LO + idx] = (__bf16)(v - (float)hi);
        });
    stage_all<kH * LD2, 512>(tid,
        [&](int idx) {
            const int o = idx / LD2, ps = idx - o * LD2;
            const float v = p.W[1][o * kH + hid16(ps < kH ? ps : 0)];
            return ps < kH ? v : 0.f;
        },
        [&](int idx, float v) {
            const __bf16 hi = (__bf16)v;
            sm[S::OFF_W2 + idx] = hi;
            sm[S::OFF_W2 + S::W2LO + idx] = (__bf16)(v - (float)hi);
        });
    stage_all<4 * LD3, 512>(tid,
        [&](int idx) {
            const int c = idx / LD3, ps = idx - c * LD3;
            return (c < 3 && ps < kH) ? p.W[2][c * kH + hid16(ps)] : 0.f;
        },
        [&](int idx, float v) {
            const __bf16 hi = (__bf16)v;
            sm[S::OFF_W3 + idx] = hi;
            sm[S::OFF_W3 + S::W3LO + idx] = (__bf16)(v - (float)hi);
        });
    lds_f* const Bs = (lds_f*)(sm + S::OFF_B);                   // b2 [64] in natural order, b3 [4]
    if (tid < kH) Bs[tid] = p.b[1][tid];
    if (tid >= kH && tid < kH + 4) Bs[tid] = tid - kH < 3 ? p.b[2][tid - kH] : 0.f;
    for (int idx = tid; idx < 8 * S::SPW / 2; idx += 512) ((lds_f*)(sm + S::OFF_IMG))[idx] = 0.f;
    __syncthreads();
    const NoiseSrc nsrc = noise_with_step(p.noise, p.step_dev);

    // ---------------- launch-lifetime accumulators: the weight-gradient tiles this wave owns
    f32x16 accW2 = f32x16(0.f);          // dW2 tile (wave & 3), samples of waves 4 (wave >> 2) ..
    f32x16 accW1 = f32x16(0.f);          // dW1 tile (wave & 3) of the first 64 input columns, same samples
    f32x4 accT = f32x4(0.f);             // dW1 columns 64..79, rows 16 (wave & 3).., k-steps 2 (wave >> 2), + 1
    f32x4 accW3q = f32x4(0.f), accB2q = f32x4(0.f);
    float accB3[3] = {0.f, 0.f, 0.f}, accLoss = 0.f;
    const int T4 = wave & 3, kh = wave >> 2;
    const int to = T4 >> 1, tk = T4 & 1;
    lds_bf* const img0 = sm + S::OFF_IMG;

#ifdef NIC_STAMPS
    unsigned long long stamp_sum[NIC_NPH];
#pragma unroll
    for (int i = 0; i < NIC_NPH; ++i) stamp_sum[i] = 0;
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

    // ---------------- XCD-aware persistent walk, workgroup-synchronous rounds of 8 units (one per wave); see fused_kernel.hpp
    // (work units are counted in 32 bits: the host refuses launches with 2^30 or more of them)
    // Two segments (balance_units): macro-tiles [0, seg_split) as whole work units - as many as fill every wave of the launch the same
    // number of times - and the remainder dealt out in 2^rg_log2 groups of rounds, so that the last, partly filled step of a small
    // launch costs a fraction of a unit (the reference's default step: 2 120 macro-tiles on 2 048 waves = one step of 16 rounds and
    // 72 macro-tiles in 8 groups of 2 rounds, instead of 5 steps of 4).
    // lane-dependent parts of the round's LDS addresses kept in registers for the whole launch (NIC_T16_LOFF, a bit per offset: this kernel sits at
    // 252 - 256 registers, so only the offsets that pay for their register are hoisted; fused_q16.hpp has the full set and the measurement)
    constexpr int LOFF = NIC_T16_LOFF;
    int lo_rW = 0, lo_rZ = 0, lo_tW = 0, lo_b44 = 0, lo_tZ = 0, lo_rX = 0;
    {
        const int n16 = lane & 15, g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h32 = lane >> 5, cg = (lane >> 4) & 1;
        if constexpr ((LOFF & 1) != 0) lo_rW = opaque_i(n16 * S::LD1 + 8 * g);
        if constexpr ((LOFF & 2) != 0) lo_rZ = opaque_i(wave * S::SPW + n16 * S::LDZ + 8 * g);
        if constexpr ((LOFF & 4) != 0) lo_tW = opaque_i((4 * g + q4) * S::LD1 + 8 * p4);
        if constexpr ((LOFF & 8) != 0) lo_b44 = opaque_i(wave * S::SPW + 4 * q4 * S::LDZ + 16 * g + 4 * p4);
        if constexpr ((LOFF & 16) != 0) lo_tZ = opaque_i((4 * (wave >> 2)) * S::SPW + (4 * q4 + 2 * h32) * S::LDZ + 16 * cg + 4 * p4);
        if constexpr ((LOFF & 32) != 0) lo_rX = opaque_i(wave * S::SPW + S::OFF_X + n16 * S::LDX + 8 * g);
    }
    const int xcd = blockIdx.x & 7, nb8 = gridDim.x >> 3;
  for (int seg = 0; seg < 2; ++seg) {
    const int seg_tile0 = seg ? (int)p.seg_split : 0;
    const int seg_tiles = seg ? (int)p.n_tiles - (int)p.seg_split : (int)p.seg_split;
    const int rg = seg ? p.rg_log2 : 0;
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
    const int shift = (NIC_STAGGER && (NIC_STAGGER_RG || rg == 0)) ? (int)((blockIdx.x >> 3) & (nph - 1)) * (rounds_unit / nph) : 0;

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
        int org[3] = {0, 0, 0}, blk[3] = {0, 0, 0};
        const int crop = tile / tiles_per_crop;
        GridAcc16 gacc;
        f32x4 dxacc[4];                                                      // tiles 0..2: the cell's G0 gradient sums (persistent over the rounds)
        uint32_t blk_off0, blk_off1;
        CellRaw16 raw;
        {
            const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4;
            const int tt = tile - crop * tiles_per_crop;
            int boff[2];
            if (p.edge_lw < 0 || tt < tiles_main) {
                boff[1] = tt % p.tiles_y;                 // regular tile: 16 x 1 cells
                boff[0] = (tt / p.tiles_y) * 16;
            } else {
                lw = p.edge_lw;                           // edge tile: 2^lw x (16 >> lw) cells
                boff[1] = (tt - tiles_main) * (16 >> lw);
                boff[0] = p.full_x * 16;
            }
            const int lc[2] = {n16 & ((1 << lw) - 1), n16 >> lw};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                org[a] = origin_of(p, crop * 2 + a);
                blk[a] = (org[a] >> p.lm) + boff[a] + lc[a];
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) gacc.g1[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) dxacc[t] = f32x4(0.f);
            const int qb[3] = {blk[0] << p.lm, blk[1] << p.lm, 0};
            cell_offsets<L>(p, qb, blk_off0, blk_off1);
            gather_cell16(p, blk_off0, blk_off1, g, raw);
        }
        STAMP(12);   // macro-tile setup, raw gathers issued

        for (int it = it_begin; it < it_begin + it_len; ++it) {
            // ================= forward =================
            f32x4 a1[4], d1[4], a2[4], d2[4];
            float dz3[3];
            float kx1, ky1;                                                   // G1 interpolation fractions of the sample (for the G1 gradient weights)
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4;
                // ---------- which sample does this lane own in this round
                bool valid = tile_ok;
                int64_t n;
                int q[3] = {0, 0, 0};
                {
                    const int m1 = (1 << p.lm) - 1;
                    const int pass = it >> (p.lm * 2), its = it & (p.niter - 1);
                    const int j[2] = {its >> p.lm, its & m1};
                    const int ext[2] = {p.d.extent[0], p.d.extent[1]};
                    int idx[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int i = (blk[a] << p.lm) + j[a] - org[a];
                        valid = valid && i >= 0 && i < ext[a];
                        idx[a] = i < 0 ? 0 : (i >= ext[a] ? ext[a] - 1 : i);
                        q[a] = org[a] + idx[a];
                    }
                    n = ((int64_t)crop * p.passes + pass) * p.n_per_crop + (int64_t)idx[0] * ext[1] + idx[1];
                }
                // ---------- target (or incoming dY) of the sample: fetched now, used after the forward pass
                float tgt[3] = {0.f, 0.f, 0.f};
                if (MODE == MODE_TRAIN_RGBX) {
                    // interleaved uint8 RGBX (one dword per pixel, fewer than 2^31 of them: checked by the host): one 32-bit offset, one load,
                    // byte converts; u / den correctly rounded as in the planar path
                    const uint32_t off = (uint32_t)q[0] * (uint32_t)p.timg_s[0] + (uint32_t)q[1] * (uint32_t)p.timg_s[1];
                    const uint32_t rgbx = reinterpret_cast<const uint32_t*>(p.timg)[off];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float u = (float)((rgbx >> (8 * c)) & 255u);
                        const float t0 = mul_rn(u, p.timg_rcp);
                        tgt[c] = fmaf(fmaf(-t0, p.timg_den, u), p.timg_rcp, t0);
                    }
                } else if (MODE == MODE_TRAIN_IMG) {
                    const int64_t off = (int64_t)q[0] * p.timg_s[0] + (int64_t)q[1] * p.timg_s[1];
                    uint32_t rgbx = 0u;
                    if (p.timg_u8 == 2) rgbx = reinterpret_cast<const uint32_t*>(p.timg)[off];      // interleaved: one load for the three targets
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
                float xs[20];
                {
                    EncCtx cx;
                    encode16<L>(p, q, g, xs, cx, raw);
                    kx1 = cx.kx; ky1 = cx.ky;
                    add_noise16<L>(nsrc, (uint64_t)(p.d.sample_base + n), n, g, xs);
                }
                STAMP(0);    // coordinates, blend, PE, noise
                lds_bf* const imgw = img0 + wave * S::SPW;
                // ---------- layer 1: Z1[o][n] = sum_rho W1[o][rho] X[rho][n]   (b1 rides on the constant-one slot)
                {
                    lds_cbf* const w_row = opaque((lds_cbf*)(sm + ((LOFF & 1) ? lo_rW : n16 * LD1 + 8 * g)));                    // W1 and W2 (LD1 == LD2): row n16, columns 8 g ..
                    lds_cbf* const w1_row2 = opaque((lds_cbf*)(sm + S::OFF_W1 + 64 + ((LOFF & 1) ? lo_rW - 4 * g : n16 * LD1 + 4 * g)));
                    lds_bf* const x_st = (LOFF & 32) ? opaque(img0 + lo_rX) : opaque(imgw + S::OFF_X + n16 * LDX + 8 * g);                     // fragment stores: row n, columns 32 s + 8 g
                    lds_bf* const x_st2 = (LOFF & 32) ? opaque(img0 + 64 + lo_rX - 4 * g) : opaque(imgw + S::OFF_X + n16 * LDX + 64 + 4 * g);
                    f32x4 z[4];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        // slots 8s .. 8s+7 (slot 15 is the zero slot)
                        const float xv[8] = {xs[8 * s], xs[8 * s + 1], xs[8 * s + 2], xs[8 * s + 3], xs[8 * s + 4], xs[8 * s + 5], xs[8 * s + 6],
                                             s == 0 ? xs[7] : 0.f};
                        const Frag2 bf = split8(xv);
                        st_frag(&x_st[32 * s], bf.hi);
                        st_frag(&x_st[S::XLO + 32 * s], bf.lo);
                        auto la = [&](int t) {
                            Frag2 a;
                            a.hi = ld_frag(&w_row[S::OFF_W1 + 16 * t * LD1 + 32 * s]);
                            a.lo = ld_frag(&w_row[S::OFF_W1 + S::W1LO + 16 * t * LD1 + 32 * s]);
                            return a;
                        };
                        if (s == 0) kstep16<4, true>(z, bf, la);
                        else kstep16<4>(z, bf, la);
                    }
                    {   // k-step 2: slots 16..19, compact columns 64 + 4 g + j
                        const float xv[8] = {xs[15], xs[16], xs[17], xs[18], 0.f, 0.f, 0.f, 0.f};
                        const Frag2 bf = split8(xv);
                        const s16x8 bh = __builtin_bit_cast(s16x8, bf.hi), bl = __builtin_bit_cast(s16x8, bf.lo);
                        *reinterpret_cast<lds_s16x4*>(x_st2) = s16x4{bh[0], bh[1], bh[2], bh[3]};
                        *reinterpret_cast<lds_s16x4*>(x_st2 + S::XLO) = s16x4{bl[0], bl[1], bl[2], bl[3]};
#if NIC_T16_HALF16
                        const Half2 bq = {s16x4{bh[0], bh[1], bh[2], bh[3]}, s16x4{bl[0], bl[1], bl[2], bl[3]}};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const Half2 a = {*reinterpret_cast<lds_cs16x4*>(&w1_row2[16 * t * LD1]), *reinterpret_cast<lds_cs16x4*>(&w1_row2[S::W1LO + 16 * t * LD1])};
                            z[t] = mfma16h_split(a, bq, z[t]);
                        }
#else
                        kstep16<4>(z, bf, [&](int t) {
                            Frag2 a;
                            a.hi = half_frag(*reinterpret_cast<lds_cs16x4*>(&w1_row2[16 * t * LD1]));
                            a.lo = half_frag(*reinterpret_cast<lds_cs16x4*>(&w1_row2[S::W1LO + 16 * t * LD1]));
                            return a;
                        });
#endif
                    }
                    // (tile-outer order - the GELU of row tile t beside the MFMAs of tile t + 1 - measured 3.5 % slower: 8 more registers
                    //  spill, and the packed-fp32 GELU chains stop the matrix pipe rather than run beside it)
#pragma unroll
                    for (int t = 0; t < 4; ++t) gelu_and_grad4(z[t], a1[t], d1[t]);
                    // (pinning the derivatives - fused_q16.hpp::pin - costs this kernel 3.7 % for d1 and 0.3 % for d2 with the triangular PE, where nothing
                    //  spills: the compiler's sinking of their last steps spreads vector work into the backward pass.  The sinusoidal-PE layout spills
                    //  6 - 9 registers; pinning d2 there removes them and changes nothing: 2.236 against 2.230 ms)
                    if (NIC_T16_PIN & 1) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d1[t]));
                    }
                    // ---------- layer 2
                    lds_cf* const b2_row = opaque(Bs + 4 * g);
                    lds_bf* const a1_st = (LOFF & 2) ? opaque(img0 + S::OFF_A1 + lo_rZ) : opaque(imgw + S::OFF_A1 + n16 * LDZ + 8 * g);
#pragma unroll
                    for (int t = 0; t < 4; ++t) z[t] = ld4(&b2_row[16 * t]);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const Frag2 bf = split_pair(a1[2 * s], a1[2 * s + 1]);
                        st_frag(&a1_st[32 * s], bf.hi);
                        st_frag(&a1_st[S::DZLO + 32 * s], bf.lo);
                        kstep16<4>(z, bf, [&](int t) {
                            Frag2 a;
                            a.hi = ld_frag(&w_row[S::OFF_W2 + 16 * t * LD2 + 32 * s]);
                            a.lo = ld_frag(&w_row[S::OFF_W2 + S::W2LO + 16 * t * LD2 + 32 * s]);
                            return a;
                        });
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) gelu_and_grad4(z[t], a2[t], d2[t]);
                    if ((NIC_T16_PIN & 2) || ((NIC_T16_PIN & 4) && L::PE == NIC_PE_SINUSOIDAL)) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d2[t]));
                    }
                }
                // ---------- layer 3 (rows 0..2 of a 16-row tile; quarter 0 holds the sample's 3 outputs); the a2 fragments are also
                // the a2 image of dW3 (the DZ region is free until dZ2 is stored)
                float yv[3];
                {
                    lds_cbf* const w3_row = opaque((lds_cbf*)(sm + S::OFF_W3 + (n16 < 3 ? n16 : 3) * LD3 + 8 * g));
                    lds_bf* const dz_st = (LOFF & 2) ? opaque(img0 + lo_rZ) : opaque(imgw + n16 * LDZ + 8 * g);
                    f32x4 z3;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        Frag2 af;
                        af.hi = ld_frag(&w3_row[32 * s]);
                        af.lo = ld_frag(&w3_row[S::W3LO + 32 * s]);
                        const Frag2 bf = split_pair(a2[2 * s], a2[2 * s + 1]);
                        st_frag(&dz_st[32 * s], bf.hi);
                        st_frag(&dz_st[S::DZLO + 32 * s], bf.lo);
                        z3 = mfma16_split(af, bf, s == 0 ? f32x4(0.f) : z3);
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) yv[c] = sigmoid_f(z3[c] + ((lds_cf*)Bs)[kH + c]);
                }
                STAMP(1);    // layers 1 - 3 with their image stores and GELUs
                const bool own = valid && g == 0;
                if (p.y != nullptr && own) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) p.y[n * 3 + c] = yv[c];
                }
                // ---------- dZ3
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float gr;
                    if (MODE == MODE_TRAIN_MSE || MODE == MODE_TRAIN_IMG || MODE == MODE_TRAIN_RGBX) {
                        const float diff = own ? yv[c] - tgt[c] : 0.f;
                        accLoss += diff * diff;
                        gr = p.grad_scale * diff;
                    } else {
                        gr = own ? tgt[c] : 0.f;
                    }
                    dz3[c] = gr * yv[c] * (1.0f - yv[c]);
                    accB3[c] += dz3[c];
                }
                if (g == 0) {
                    lds_bf* const d3_st = opaque(imgw + S::OFF_D3 + 4 * (n16 & 3) + (n16 >> 2));
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const __bf16 hi = (__bf16)dz3[c];
                        d3_st[c * 16] = hi;
                        d3_st[64 + c * 16] = (__bf16)(dz3[c] - (float)hi);
                    }
                }
            }
            wave_lds_fence();
            // ================= backward =================
            f32x4 dz1[4];
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
                lds_bf* const imgw = img0 + wave * S::SPW;
                // 4x4x4 operands (own samples): B = column `lane` of samples 4 q' + r; A = dZ3[c = lane & 3][those samples]
                lds_cbf* const dz_b44 = (LOFF & 8) ? opaque((lds_cbf*)(img0 + lo_b44)) : opaque((lds_cbf*)(imgw + 4 * q4 * LDZ + 16 * g + 4 * p4));
                // ---------- dW3[c][pos = lane] += sum_n dZ3[c][n] a2[pos][n]: 4x4x4 MFMAs over the wave's own 16 samples
                {
                    lds_cbf* const d3_a44 = opaque((lds_cbf*)(imgw + S::OFF_D3 + (ln & 3) * 16));
                    s16x4 bh[4], bl[4], ah[4], al[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bh[r] = tr4(&dz_b44[r * LDZ]);
                        bl[r] = tr4(&dz_b44[S::DZLO + r * LDZ]);
                        ah[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[4 * r]);
                        al[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[64 + 4 * r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accW3q = mfma4_bf(al[r], bh[r], accW3q);
                        accW3q = mfma4_bf(ah[r], bl[r], accW3q);
                        accW3q = mfma4_bf(ah[r], bh[r], accW3q);
                    }
                }
                // ---------- dA2 = W3^T dZ3 (k = c: quarter 0 carries dZ3 in elements 0..2, everything else is zero), dZ2 = dA2 * gelu'(Z2)
                f32x4 dz2[4];
                {
                    lds_cbf* const w3_tr = opaque((lds_cbf*)(sm + S::OFF_W3 + q4 * LD3 + 8 * p4));
                    const float dzv[8] = {dz3[0], dz3[1], dz3[2], 0.f, 0.f, 0.f, 0.f, 0.f};
                    const Frag2 bf = split8(dzv);
                    const s16x8 b3h = __builtin_bit_cast(s16x8, bf.hi), b3l = __builtin_bit_cast(s16x8, bf.lo);
                    const Half2 bq3 = {s16x4{b3h[0], b3h[1], b3h[2], b3h[3]}, s16x4{b3l[0], b3l[1], b3l[2], b3l[3]}};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const s16x4 ahh = tr4(&w3_tr[32 * (t >> 1) + 4 * (t & 1)]);        // rows c = 0..3; the k = 4..31 part meets zeros
                        const s16x4 all = tr4(&w3_tr[S::W3LO + 32 * (t >> 1) + 4 * (t & 1)]);
#if NIC_T16_HALF16
                        // k = c < 4 lives in quarter 0 of either MFMA shape: the 16x16x16 one takes the transposed read as it is (no doubled fragment)
                        const Half2 aq = {ahh, all};
                        dz2[t] = mfma16h_split(aq, bq3, f32x4(0.f)) * d2[t];
#else
                        Frag2 af;
                        af.hi = join8(ahh, ahh);
                        af.lo = join8(all, all);
                        dz2[t] = mfma16_split(af, bf, f32x4(0.f)) * d2[t];
#endif
                    }
                }
                STAMP(2);    // dZ3 image, dW3, dA2
                wave_lds_fence();                                              // the dW3 reads of the DZ region are issued: it may be overwritten
                // ---------- dA1 = W2^T dZ2; the split dZ2 fragments are the dZ2 image of the weight-gradient product
                {
                    lds_cbf* const w2_tr = opaque((lds_cbf*)(sm + S::OFF_W2 + ((LOFF & 4) ? lo_tW : (4 * g + q4) * LD2 + 8 * p4)));
                    lds_bf* const dz_st = (LOFF & 2) ? opaque(img0 + lo_rZ) : opaque(imgw + n16 * LDZ + 8 * g);
                    f32x4 acc[4];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const Frag2 bf = split_pair(dz2[2 * s], dz2[2 * s + 1]);
                        st_frag(&dz_st[32 * s], bf.hi);
                        st_frag(&dz_st[S::DZLO + 32 * s], bf.lo);
                        auto la = [&](int t) {
                            const int co = 32 * (t >> 1) + 4 * (t & 1);
                            Frag2 a;
                            a.hi = join8(tr4(&w2_tr[32 * s * LD2 + co]), tr4(&w2_tr[(32 * s + 16) * LD2 + co]));
                            a.lo = join8(tr4(&w2_tr[S::W2LO + 32 * s * LD2 + co]), tr4(&w2_tr[S::W2LO + (32 * s + 16) * LD2 + co]));
                            return a;
                        };
                        if (s == 0) kstep16<4, true>(acc, bf, la);
                        else kstep16<4>(acc, bf, la);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) dz1[t] = acc[t] * d1[t];
                }
                wave_lds_fence();
                // ---------- db2[pos = lane] += sum_n dZ2[pos][n]: 4x4x4 MFMAs against a block of ones
                {
                    const s16x4 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
                    s16x4 bh[4], bl[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bl[r] = tr4(&dz_b44[S::DZLO + r * LDZ]);
                        bh[r] = tr4(&dz_b44[r * LDZ]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accB2q = mfma4_bf(ones, bl[r], accB2q);
                        accB2q = mfma4_bf(ones, bh[r], accB2q);
                    }
                }
            }
            STAMP(3);    // dA1 (+ dZ2 image), db2
            wg_lds_barrier();
            STAMP(4);    // wait at barrier 1
            // ---------- dW2 tile (to, tk) += sum over the samples of waves 4 kh .. 4 kh + 3 of dZ2[o][n] a1[k][n]
            {
                const int ln = opaque_i(lane), q4 = (ln & 15) >> 2, p4 = ln & 3, h32 = ln >> 5, cg = (ln >> 4) & 1;
                // 32x32x16 operands: samples 4 q' + 2 (lane >> 5) + rd of a source wave, columns 32 tile + 16 cg + 4 p4
                lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + 32 * to + ((LOFF & 16) ? lo_tZ : (4 * kh) * S::SPW + (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
                lds_cbf* const a1_t32 = opaque((lds_cbf*)(img0 + S::OFF_A1 + 32 * tk + ((LOFF & 16) ? lo_tZ : (4 * kh) * S::SPW + (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    Frag2 af, bf;
                    af.hi = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                    bf.lo = join8(tr4(&a1_t32[v * S::SPW + S::DZLO]), tr4(&a1_t32[v * S::SPW + S::DZLO + LDZ]));
                    af.lo = join8(tr4(&dz_t32[v * S::SPW + S::DZLO]), tr4(&dz_t32[v * S::SPW + S::DZLO + LDZ]));
                    bf.hi = join8(tr4(&a1_t32[v * S::SPW]), tr4(&a1_t32[v * S::SPW + LDZ]));
                    accW2 = mfma_split(af, bf, accW2);
                }
            }
            STAMP(5);    // dW2 MFMAs
            wg_lds_barrier();                       // everyone is done reading dZ2 before dZ1 replaces it
            STAMP(6);    // wait at barrier 2
            // ---------- dX = W1^T dZ1 for the grid slots (tiles 0..3 = slots 0..15); the split dZ1 fragments are the dZ1 image.
            // Tiles 0..2 (the G0 channels) keep their running sums over the rounds in the product's C operand.
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
                lds_cbf* const w1_tr = opaque((lds_cbf*)(sm + S::OFF_W1 + ((LOFF & 4) ? lo_tW : (4 * g + q4) * LD1 + 8 * p4)));
                lds_bf* const dz_st = (LOFF & 2) ? opaque(img0 + lo_rZ) : opaque(img0 + wave * S::SPW + n16 * LDZ + 8 * g);
                dxacc[3] = f32x4(0.f);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const Frag2 bf = split_pair(dz1[2 * s], dz1[2 * s + 1]);
                    st_frag(&dz_st[32 * s], bf.hi);
                    st_frag(&dz_st[S::DZLO + 32 * s], bf.lo);
                    kstep16<4>(dxacc, bf, [&](int t) {
                        const int co = 32 * (t >> 1) + 4 * (t & 1);
                        Frag2 a;
                        a.hi = join8(tr4(&w1_tr[32 * s * LD1 + co]), tr4(&w1_tr[(32 * s + 16) * LD1 + co]));
                        a.lo = join8(tr4(&w1_tr[S::W1LO + 32 * s * LD1 + co]), tr4(&w1_tr[S::W1LO + (32 * s + 16) * LD1 + co]));
                        return a;
                    });
                }
                const G1FactorsT<2> gf = g1_factors<2>(p.d.g1_weight_mode, kx1, ky1, 0.f);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float w = g1_corner_factor<2>(gf, c4);
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) gacc.g1[c4 * 3 + cc] = fmaf(dxacc[3][cc], w, gacc.g1[c4 * 3 + cc]);
                }
            }
            STAMP(7);    // dX (+ dZ1 image), grid-gradient accumulation
            wg_lds_barrier();
            STAMP(8);    // wait at barrier 3
            {
                const int ln = opaque_i(lane), g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3, h32 = ln >> 5, cg = (ln >> 4) & 1;
                lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + 32 * to + ((LOFF & 16) ? lo_tZ : (4 * kh) * S::SPW + (4 * q4 + 2 * h32) * LDZ + 16 * cg + 4 * p4)));
                lds_cbf* const x_t32 = opaque((lds_cbf*)(img0 + (4 * kh) * S::SPW + S::OFF_X + (4 * q4 + 2 * h32) * LDX + 32 * tk + 16 * cg + 4 * p4));
                // ---------- dW1: columns 0..63 as the 32x32 tile (to, tk) over the same four source waves ...
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    Frag2 af, bf;
                    af.hi = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                    bf.lo = join8(tr4(&x_t32[v * S::SPW + S::XLO]), tr4(&x_t32[v * S::SPW + S::XLO + LDX]));
                    af.lo = join8(tr4(&dz_t32[v * S::SPW + S::DZLO]), tr4(&dz_t32[v * S::SPW + S::DZLO + LDZ]));
                    bf.hi = join8(tr4(&x_t32[v * S::SPW]), tr4(&x_t32[v * S::SPW + LDX]));
                    accW1 = mfma_split(af, bf, accW1);
                }
                // ... columns 64..79 as a 16x16 tile (rows 16 T4 ..): two k-steps of 32 samples; quarter G reads source wave
                // 4 kh + 2 uu + (G >> 1), samples 4 q' + 2 (G & 1) + rd
                lds_cbf* const dz_t16 = opaque((lds_cbf*)(img0 + (4 * kh + (g >> 1)) * S::SPW + (4 * q4 + 2 * (g & 1)) * LDZ + 16 * T4 + 4 * p4));
                lds_cbf* const x_t16 = opaque((lds_cbf*)(img0 + (4 * kh + (g >> 1)) * S::SPW + S::OFF_X + (4 * q4 + 2 * (g & 1)) * LDX + 64 + 4 * p4));
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    Frag2 af, bf;
                    af.hi = join8(tr4(&dz_t16[2 * uu * S::SPW]), tr4(&dz_t16[2 * uu * S::SPW + LDZ]));
                    af.lo = join8(tr4(&dz_t16[2 * uu * S::SPW + S::DZLO]), tr4(&dz_t16[2 * uu * S::SPW + S::DZLO + LDZ]));
                    bf.hi = join8(tr4(&x_t16[2 * uu * S::SPW]), tr4(&x_t16[2 * uu * S::SPW + LDX]));
                    bf.lo = join8(tr4(&x_t16[2 * uu * S::SPW + S::XLO]), tr4(&x_t16[2 * uu * S::SPW + S::XLO + LDX]));
                    accT = mfma16_split(af, bf, accT);
                }
            }
            STAMP(9);    // dW1 MFMAs
            wg_lds_barrier();   // all reads of dZ1 / X done before the next round overwrites them
            STAMP(10);   // wait at barrier 4
        }  // rounds of one macro-tile

        // ---------- flush of the cell's gradient sums
        {
            const int ln = opaque_i(lane), g = ln >> 4;
            combine_g1_lanes16(gacc, blk_off1, blk, ln, lw);
            bool flush = true;
            if (NIC_GROUP_SUM && rg > 0) {                                  // segment-uniform: groups of one macro-tile sit in one workgroup
                lds_f* const reg0 = (lds_f*)img0;
                constexpr int REGION = S::SPW / 2;                         // floats per wave
                static_assert(24 * 64 <= REGION, "group-sum scratch");
                int leader = wave;
                if (tile_ok)
                    for (int w = wave - 1; w >= 4 * kh; --w)
                        if (seg_tile0 + ((base + w) >> rg) == tile) leader = w;
                if (leader != wave) {
                    lds_f* const mine = opaque(reg0 + wave * REGION + ln);
#pragma unroll
                    for (int i = 0; i < 12; ++i) mine[i * 64] = dxacc[i >> 2][i & 3];
#pragma unroll
                    for (int i = 0; i < 12; ++i) mine[(12 + i) * 64] = gacc.g1[i];
                }
                wg_lds_barrier();
                if (leader == wave && tile_ok) {
                    for (int w = wave + 1; w < 4 * kh + 4; ++w) {
                        if (base + w >= t_end || seg_tile0 + ((base + w) >> rg) != tile) break;
                        lds_cf* const theirs = opaque(reg0 + w * REGION + ln);
#pragma unroll
                        for (int i = 0; i < 12; ++i) dxacc[i >> 2][i & 3] += theirs[i * 64];
#pragma unroll
                        for (int i = 0; i < 12; ++i) gacc.g1[i] += theirs[(12 + i) * 64];
                    }
                }
                wg_lds_barrier();
                flush = leader == wave;
            }
            if (flush && NIC_T16_PREADD) preadd_x16(dxacc, blk_off0, ln);
            if (NIC_T16_PREADD >= 2 && rg == 0 && (p.preadd_y || !NIC_PREADD_Y_SMALL)) {      // segment-uniform
                static_assert(13 * 64 <= S::SPW / 2, "pre-add scratch");
                preadd_y16<8>(dxacc, blk_off0, (uint32_t)p.g0.nx, ln, wave, (lds_f*)img0, S::SPW / 2, [&]() { wg_lds_barrier(); });
            }
            if (flush) {
                // one predicate per lane and grid instead of one per value: a lane whose 12 sums are all exact zeros (cell outside the
                // crop; G1 sums handed to the partner lane) issues nothing - 2 exec-mask regions per flush instead of 24
                const uint32_t pb0 = (uint32_t)p.g0.plane * 4u, pb1 = (uint32_t)p.g1.plane * 4u;
                uint32_t nz0 = 0u, nz1 = 0u;
#pragma unroll
                for (int c = 0; c < kC; ++c) {
                    nz0 |= __builtin_bit_cast(uint32_t, dxacc[c >> 2][c & 3]);
                    nz1 |= __builtin_bit_cast(uint32_t, gacc.g1[c]);
                }
                if ((nz0 << 1) != 0u) {
                    uint32_t ob = (blk_off0 + (uint32_t)p.g0.at(g >> 1, g & 1, 0)) * 4u;
                    char* gbase = reinterpret_cast<char*>(p.g0_grad);
#pragma unroll
                    for (int c = 0; c < kC; ++c, ob += pb0) atomicAdd(reinterpret_cast<float*>(gbase + ob), dxacc[c >> 2][c & 3]);
                }
                if ((nz1 << 1) != 0u) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        uint32_t ob = (blk_off1 + (uint32_t)p.g1.at(c4 >> 1, c4 & 1, 0)) * 4u + (uint32_t)(3 * g) * pb1;
                        char* gbase = reinterpret_cast<char*>(p.g1_grad);
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc, ob += pb1) atomicAdd(reinterpret_cast<float*>(gbase + ob), gacc.g1[c4 * 3 + cc]);
                    }
                }
            }
        }
        STAMP(13);   // grid-gradient flush (atomics)
    }  // macro-tile loop
  }  // segments
#ifdef NIC_STAMPS
    if (lane == 0) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.partials + (size_t)gridDim.x * S::REC) + ((size_t)blockIdx.x * 8 + wave) * 16;   // behind the records (nic_workspace_bytes leaves 1 MiB)
#pragma unroll
        for (int i = 0; i < NIC_NPH; ++i) dst[i] = stamp_sum[i];
    }
#endif

    // ---------------- one record per workgroup
    float* rec = p.partials + (int64_t)blockIdx.x * S::REC;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        rec[S::REC_W2 + wave * 1024 + r * 64 + lane] = accW2[r];
        rec[S::REC_W1 + wave * 1024 + r * 64 + lane] = accW1[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rec[S::REC_TAIL + wave * 256 + r * 64 + lane] = accT[r];
    float* tail = rec + S::REC_WAVE + wave * 320;
    tail[lane] = accB2q[0];
#pragma unroll
    for (int c = 0; c < 3; ++c) tail[64 + 64 * c + lane] = accW3q[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = c < 3 ? accB3[c] : accLoss;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) tail[256 + c] = v;
    }
}

// =====================================================================================================
// Fixed-order reduction of the per-workgroup records of fused_train16_kernel into the decoder gradients (nn.Linear layouts) and
// the loss.  Threads are numbered in RECORD order (kR16_SRC of them, see below): the 32 threads of a slice read 32 consecutive floats
// of every record (the first version numbered them in output order and gathered - 4-byte reads scattered over a tile, 21 us for the
// 21.5 MB of a 256-workgroup launch), the scattered side is the 9 092 stores.
//   [0, 4096)      dW1 columns 0..63: tile T4 = 2 to + tk, register r, lane - partial sums of the two halves 4 x 1024 floats apart
//   [4096, 8192)   dW2, the same
//   [8192, 9216)   dW1 columns 64..79: row group po >> 4, register r, lane - two partial sums 4 x 256 apart
//   [9216, 9536)   per-wave tails: db2 [64] | dW3 [3][64] | db3 [3] | loss | unused - 8 partial sums 320 apart
constexpr int kR16_SRC = 4096 + 4096 + 1024 + 320;
#ifndef NIC_R16_SLICES
#define NIC_R16_SLICES 8       // see NIC_RQ_SLICES (fused_q16.hpp)
#endif
template <class L>
__global__ void __launch_bounds__(256) reduce16_kernel(const float* partials, int n_rec, nic_mlp_grads gr, float* loss, float loss_scale, const StepTail tl) {
    if (tail_block(tl)) return;                                   // a streaming block of the optimiser tail (nic_adam.hpp)
    using S = Lds16;
    constexpr int SL = NIC_R16_SLICES, OUTS = 256 / SL;           // record slices summed in parallel x outputs per block
    __shared__ float red[SL][OUTS];
    const int slice = threadIdx.x / OUTS, oi = threadIdx.x % OUTS;
    const int gid = blockIdx.x * OUTS + oi;
    const bool live = gid < kR16_SRC;
    int nsrc = 0, off0 = 0, stride = 0;
    float* dst = nullptr;
    bool is_loss = false;
    if (live) {
        if (gid < 8192) {
            // element of a 32x32 accumulator tile: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
            const int u = gid & 4095, T4 = u >> 10, r = (u >> 6) & 15, l = u & 63;
            const int po = 32 * (T4 >> 1) + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = 32 * (T4 & 1) + (l & 31);
            const int o = hid16(po);
            nsrc = 2; stride = 4 * 1024;
            if (gid < 4096) {
                off0 = S::REC_W1 + u;
                const int ch = channel_of_rho16(col);
                if (ch >= 0) dst = gr.w[0] ? gr.w[0] + o * L::CIN + ch : nullptr;
                else if (ch == kSlotOne) dst = gr.b[0] ? gr.b[0] + o : nullptr;
            } else {
                off0 = S::REC_W2 + u;
                dst = gr.w[1] ? gr.w[1] + o * kH + hid16(col) : nullptr;
            }
        } else if (gid < 9216) {
            // 16x16 tile: register r of lane l is row r + 4 (l >> 4), column l & 15
            const int u = gid - 8192, wq = u >> 8, r = (u >> 6) & 3, l = u & 63;
            const int o = hid16(16 * wq + r + 4 * (l >> 4)), ch = channel_of_rho16(64 + (l & 15));
            off0 = S::REC_TAIL + u; nsrc = 2; stride = 4 * 256;
            if (ch >= 0) dst = gr.w[0] ? gr.w[0] + o * L::CIN + ch : nullptr;
            else if (ch == kSlotOne) dst = gr.b[0] ? gr.b[0] + o : nullptr;
        } else {
            const int u = gid - 9216;
            off0 = S::REC_WAVE + u; nsrc = 8; stride = 320;
            if (u < 64) dst = gr.b[1] ? gr.b[1] + hid16(u) : nullptr;
            else if (u < 256) dst = gr.w[2] ? gr.w[2] + ((u - 64) >> 6) * kH + hid16(u & 63) : nullptr;
            else if (u < 259) dst = gr.b[2] ? gr.b[2] + (u - 256) : nullptr;
            else if (u == 259) { dst = loss; is_loss = true; }
        }
    }
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // fixed summation tree: bit-stable for a given grid size
    const int per = (n_rec + SL - 1) / SL;
    const int w_lo = slice * per, w_hi = (w_lo + per < n_rec) ? w_lo + per : n_rec;
    if (live && dst != nullptr) {
        for (int k = 0; k < nsrc; ++k) {
            const float* src = partials + off0 + k * stride;
            int w = w_lo;
            for (; w + 8 <= w_hi; w += 8) {                          // (16 loads in flight per thread measured slower: 21.6 us against 14.0)
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
    for (int sl = 1; sl < SL; ++sl) acc += red[sl][oi];
    if (is_loss) *dst = acc * loss_scale;
    else tail_store(tl, dst, acc);
}

}  // namespace nic
