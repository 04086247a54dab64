// Depth as a template parameter: fused_mlpn_kernel<Layout, MODE, NL> - the 2D fused step (and decode) for a decoder of NL Linear
// layers, Linear(Cin, 64) - GELU - [Linear(64, 64) - GELU] x (NL - 2) - Linear(64, 3) - Sigmoid.  NL = 3 is the reference's
// ColorDecoder (image_compression.py:57-64: depth is hard-coded there); NL = 5 is the "4 x 64" decoder BASELINE.json's north star
// names (102 912 FLOP per sample).  Split-bf16 products (hi + lo operand pairs, fp32 accumulation) throughout.
//
// The quarter layout, position order, images and fragment rules are those of fused_train16.hpp; what changes:
//   * 4 waves per workgroup (one per SIMD, up to 512 registers): every GELU derivative of every hidden layer stays in registers
//     for the backward pass (32 registers per hidden layer), and NL - 2 more 64 x 64 weight images (20 KB each) have to fit the LDS;
//   * the backward pass walks the hidden layers with ONE pair of wave images (dZ of the layer's output, the layer's input
//     re-split from registers): per layer  store images -> barrier -> owned 32x32 weight-gradient tile over the four waves'
//     samples -> barrier;  wave w owns tile w of every 64 x 64 layer and of dW1's first 64 columns, and row tile w of dW1's
//     last 16 columns;
//   * MODE_INFER: the forward pass alone (decode_image for an NL-layer decoder).
#pragma once
#include "fused_train16.hpp"

#ifndef NIC_MLPN_PIN
#define NIC_MLPN_PIN 1        // pin the GELU derivatives (fused_q16.hpp::pin): 430 -> 343 registers with 5 layers, 4K launch 4.98 -> 4.86 ms
#endif
namespace nic {

template <int NL>
struct LdsN {
    static constexpr int NH = NL - 2;                         // hidden 64 x 64 layers
    static constexpr int LD = 80, LDZ = Lds16::LDZ, LDX = Lds16::LDX;
    static constexpr int WSZ = 2 * kH * LD;                   // one 64-row weight image, hi + lo
    static constexpr int LO = kH * LD;
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_WH = OFF_W1 + WSZ;               // NH images
    static constexpr int OFF_WO = OFF_WH + NH * WSZ;          // hi [4][LD] (row 3 = zeros), lo [4][LD]
    static constexpr int WOLO = 4 * LD;
    static constexpr int OFF_B = OFF_WO + 2 * 4 * LD;         // fp32: hidden biases [NH][64], output bias [4]
    static constexpr int OFF_IMG = OFF_B + 2 * (NH * kH + 4);
    static constexpr int SPW = Lds16::SPW;                    // wave region: DZ hi, lo | AP hi, lo | X hi, lo | D3 (Lds16 offsets)
    static constexpr int TOTAL = OFF_IMG + 4 * SPW;
    static constexpr int TOTAL_INFER = OFF_IMG;
    static_assert(TOTAL * 2 <= 163840 && OFF_IMG % 8 == 0 && OFF_B % 8 == 0, "LDS");
    // per-workgroup record (floats): NH hidden tiles x 4 waves | dW1 main x 4 | dW1 tail x 4 | wave tails
    static constexpr int REC_WH = 0, REC_W1 = NH * 4 * 1024, REC_TAIL = REC_W1 + 4 * 1024, REC_WAVE = REC_TAIL + 4 * 256;
    static constexpr int WTAIL = NH * 64 + 192 + 4;           // db_hidden [NH][64], dW_out [3][64], db_out [3], loss
    static constexpr int REC = REC_WAVE + 4 * WTAIL;
};

template <class L, int MODE, int NL>
__global__ void __launch_bounds__(256) fused_mlpn_kernel(FusedParams p) {
    using S = LdsN<NL>;
    constexpr int NH = S::NH;
    constexpr bool TRAIN = MODE != MODE_INFER;
    static_assert(L::DIM == 2 && L::CIN == 73 && (NL == 3 || NL == 5), "2D layouts, 3 or 5 Linear layers");
    constexpr int LD = S::LD, LDZ = S::LDZ, LDX = S::LDX;
    constexpr int DZLO = Lds16::DZLO, OFF_AP = Lds16::OFF_A1, OFF_X = Lds16::OFF_X, XLO = Lds16::XLO, OFF_D3 = Lds16::OFF_D3;
    __shared__ __attribute__((aligned(16))) __bf16 smemn[TRAIN ? S::TOTAL : S::TOTAL_INFER];
    lds_bf* const sm = (lds_bf*)smemn;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---------------- prologue: weight images (W1 in compact rho order with b1 in the column of the constant-one slot; hidden and
    // output layers with their columns in position order)
    stage_all<kH * LD, 256>(tid,
        [&](int idx) {
            const int o = idx / LD, rho = idx - o * LD;
            const int ch = channel_of_rho16(rho);
            const float* src = ch >= 0 ? &p.W[0][o * L::CIN + ch] : &p.b[0][o];
            const float v = *src;
            return (ch >= 0 || ch == kSlotOne) ? v : 0.f;
        },
        [&](int idx, float v) {
            const __bf16 hi = (__bf16)v;
            sm[S::OFF_W1 + idx] = hi;
            sm[S::OFF_W1 + S::LO + idx] = (__bf16)(v - (float)hi);
        });
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const float* Wk = p.W[1 + k];
        stage_all<kH * LD, 256>(tid,
            [&](int idx) {
                const int o = idx / LD, ps = idx - o * LD;
                const float v = Wk[o * kH + hid16(ps < kH ? ps : 0)];
                return ps < kH ? v : 0.f;
            },
            [&](int idx, float v) {
                const __bf16 hi = (__bf16)v;
                sm[S::OFF_WH + k * S::WSZ + idx] = hi;
                sm[S::OFF_WH + k * S::WSZ + S::LO + idx] = (__bf16)(v - (float)hi);
            });
    }
    stage_all<4 * LD, 256>(tid,
        [&](int idx) {
            const int c = idx / LD, ps = idx - c * LD;
            return (c < 3 && ps < kH) ? p.W[NL - 1][c * kH + hid16(ps)] : 0.f;
        },
        [&](int idx, float v) {
            const __bf16 hi = (__bf16)v;
            sm[S::OFF_WO + idx] = hi;
            sm[S::OFF_WO + S::WOLO + idx] = (__bf16)(v - (float)hi);
        });
    lds_f* const Bs = (lds_f*)(sm + S::OFF_B);                   // hidden biases [NH][64] in natural order, output bias [4]
    for (int idx = tid; idx < NH * kH; idx += 256) Bs[idx] = p.b[1 + idx / kH][idx % kH];
    if (tid < 4) Bs[NH * kH + tid] = tid < 3 ? p.b[NL - 1][tid] : 0.f;
    if (TRAIN)
        for (int idx = tid; idx < 4 * S::SPW / 2; idx += 256) ((lds_f*)(sm + S::OFF_IMG))[idx] = 0.f;
    __syncthreads();

    // ---------------- launch-lifetime accumulators: the weight-gradient tiles this wave owns
    f32x16 accWH[NH];                    // hidden layer k: tile `wave` (rows 32 (wave >> 1).., columns 32 (wave & 1)..) over the four waves' samples
    f32x16 accW1 = f32x16(0.f);          // dW1, first 64 input columns: tile `wave`
    f32x4 accT = f32x4(0.f);             // dW1 columns 64..79: rows 16 wave ..
    f32x4 accWOq = f32x4(0.f), accBH[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { accWH[k] = f32x16(0.f); accBH[k] = f32x4(0.f); }
    float accBO[3] = {0.f, 0.f, 0.f}, accLoss = 0.f;
    const int to = wave >> 1, tk = wave & 1;
    lds_bf* const img0 = sm + S::OFF_IMG;

    // ---------------- XCD-aware persistent walk, workgroup-synchronous rounds of 4 units (one per wave); see fused_kernel.hpp
    const int xcd = blockIdx.x & 7, nb8 = gridDim.x >> 3;
    const int n_units = (int)p.n_tiles << p.rg_log2;
    const int chunk = (((n_units + 7) >> 3) + 3) & ~3;
    const int t_begin = xcd * chunk;
    const int t_end = t_begin + chunk < n_units ? t_begin + chunk : n_units;
    const int lstride = nb8 * 4;
    const int base0 = t_begin + (int)(blockIdx.x >> 3) * 4;
    const int n_my = base0 < t_end ? (t_end - base0 + lstride - 1) / lstride : 0;
    const int tiles_per_crop = (int)p.tiles_per_crop, tiles_main = (int)p.tiles_main;
    const int rounds_unit = (p.niter * p.passes) >> p.rg_log2;
    const int nph = rounds_unit >= NIC_PHASES ? NIC_PHASES : (rounds_unit >= 2 ? 2 : 1);
    const int shift = (TRAIN && NIC_STAGGER && (NIC_STAGGER_RG || p.rg_log2 == 0)) ? (int)((blockIdx.x >> 3) & (nph - 1)) * (rounds_unit / nph) : 0;

    for (int kk = 0; kk < n_my + (shift ? 1 : 0); ++kk) {
        const int base = base0 + (kk < n_my ? kk : 0) * lstride;
        const bool tile_ok = base + wave < t_end;
        const int unit = tile_ok ? base + wave : t_end - 1;
        const int tile = unit >> p.rg_log2;
        int it_len = rounds_unit, it_begin = (int)(unit & ((1 << p.rg_log2) - 1)) * rounds_unit;
        if (shift) {
            if (kk == 0) { it_begin += shift; it_len -= shift; }
            else if (kk == n_my) it_len = shift;
        }
        // ---------- macro-tile -> this lane's cell (absolute block coordinates) and crop
        int lw = 4;
        int org[3] = {0, 0, 0}, blk[3] = {0, 0, 0};
        const int crop = tile / tiles_per_crop;
        GridAcc16 gacc;
        f32x4 dxacc[4];
        uint32_t blk_off0, blk_off1;
        CellRaw16 raw;
        {
            const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4;
            const int tt = tile - crop * tiles_per_crop;
            int boff[2];
            if (p.edge_lw < 0 || tt < tiles_main) {
                boff[1] = tt % p.tiles_y;
                boff[0] = (tt / p.tiles_y) * 16;
            } else {
                lw = p.edge_lw;
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

        for (int it = it_begin; it < it_begin + it_len; ++it) {
            // ================= forward =================
            f32x4 a[NH + 1][4], d[NH + 1][4];
            float dz3[3];
            float kx1, ky1;
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4;
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
                    for (int ax = 0; ax < 2; ++ax) {
                        const int i = (blk[ax] << p.lm) + j[ax] - org[ax];
                        valid = valid && i >= 0 && i < ext[ax];
                        idx[ax] = i < 0 ? 0 : (i >= ext[ax] ? ext[ax] - 1 : i);
                        q[ax] = org[ax] + idx[ax];
                    }
                    n = ((int64_t)crop * p.passes + pass) * p.n_per_crop + (int64_t)idx[0] * ext[1] + idx[1];
                }
                float tgt[3] = {0.f, 0.f, 0.f};
                if (MODE == MODE_TRAIN_IMG) {
                    const int64_t off = (int64_t)q[0] * p.timg_s[0] + (int64_t)q[1] * p.timg_s[1];
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
                } else if (TRAIN) {
                    const float* tp = (MODE == MODE_TRAIN_MSE ? p.target : p.dy) + n * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) tgt[c] = tp[c];
                }
                float xs[20];
                {
                    EncCtx cx;
                    encode16<L>(p, q, g, xs, cx, raw);
                    kx1 = cx.kx; ky1 = cx.ky;
                    add_noise16<L>(p.noise, (uint64_t)(p.d.sample_base + n), n, g, xs);
                }
                lds_bf* const imgw = img0 + wave * S::SPW;
                lds_cbf* const w_row = opaque((lds_cbf*)(sm + n16 * LD + 8 * g));                     // every 64-row weight image: row n16, columns 8 g ..
                // ---------- layer 1
                {
                    lds_cbf* const w1_row2 = opaque((lds_cbf*)(sm + S::OFF_W1 + n16 * LD + 64 + 4 * g));
                    lds_bf* const x_st = opaque(imgw + OFF_X + n16 * LDX + 8 * g);
                    lds_bf* const x_st2 = opaque(imgw + OFF_X + n16 * LDX + 64 + 4 * g);
                    f32x4 z[4];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float xv[8] = {xs[8 * s], xs[8 * s + 1], xs[8 * s + 2], xs[8 * s + 3], xs[8 * s + 4], xs[8 * s + 5], xs[8 * s + 6],
                                             s == 0 ? xs[7] : 0.f};
                        const Frag2 bf = split8(xv);
                        if (TRAIN) {
                            st_frag(&x_st[32 * s], bf.hi);
                            st_frag(&x_st[XLO + 32 * s], bf.lo);
                        }
                        auto la = [&](int t) {
                            Frag2 af;
                            af.hi = ld_frag(&w_row[S::OFF_W1 + 16 * t * LD + 32 * s]);
                            af.lo = ld_frag(&w_row[S::OFF_W1 + S::LO + 16 * t * LD + 32 * s]);
                            return af;
                        };
                        if (s == 0) kstep16<4, true>(z, bf, la);
                        else kstep16<4>(z, bf, la);
                    }
                    {
                        const float xv[8] = {xs[15], xs[16], xs[17], xs[18], 0.f, 0.f, 0.f, 0.f};
                        const Frag2 bf = split8(xv);
                        if (TRAIN) {
                            const s16x8 bh = __builtin_bit_cast(s16x8, bf.hi), bl = __builtin_bit_cast(s16x8, bf.lo);
                            *reinterpret_cast<lds_s16x4*>(x_st2) = s16x4{bh[0], bh[1], bh[2], bh[3]};
                            *reinterpret_cast<lds_s16x4*>(x_st2 + XLO) = s16x4{bl[0], bl[1], bl[2], bl[3]};
                        }
#if NIC_T16_HALF16
                        {   // v_mfma_f32_16x16x16_bf16 on the compact half k-step (fused_train16.hpp)
                            const s16x8 qh = __builtin_bit_cast(s16x8, bf.hi), ql = __builtin_bit_cast(s16x8, bf.lo);
                            const Half2 bq = {s16x4{qh[0], qh[1], qh[2], qh[3]}, s16x4{ql[0], ql[1], ql[2], ql[3]}};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const Half2 aq = {*reinterpret_cast<lds_cs16x4*>(&w1_row2[16 * t * LD]), *reinterpret_cast<lds_cs16x4*>(&w1_row2[S::LO + 16 * t * LD])};
                                z[t] = mfma16h_split(aq, bq, z[t]);
                            }
                        }
#else
                        kstep16<4>(z, bf, [&](int t) {
                            Frag2 af;
                            af.hi = half_frag(*reinterpret_cast<lds_cs16x4*>(&w1_row2[16 * t * LD]));
                            af.lo = half_frag(*reinterpret_cast<lds_cs16x4*>(&w1_row2[S::LO + 16 * t * LD]));
                            return af;
                        });
#endif
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) gelu_and_grad4(z[t], a[0][t], d[0][t]);
#if NIC_MLPN_PIN
#pragma unroll
                    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[0][t]));
#endif
                }
                // ---------- hidden layers
                lds_cf* const b_row = opaque(Bs + 4 * g);
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    f32x4 z[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) z[t] = ld4(&b_row[k * kH + 16 * t]);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const Frag2 bf = split_pair(a[k][2 * s], a[k][2 * s + 1]);
                        kstep16<4>(z, bf, [&](int t) {
                            Frag2 af;
                            af.hi = ld_frag(&w_row[S::OFF_WH + k * S::WSZ + 16 * t * LD + 32 * s]);
                            af.lo = ld_frag(&w_row[S::OFF_WH + k * S::WSZ + S::LO + 16 * t * LD + 32 * s]);
                            return af;
                        });
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) gelu_and_grad4(z[t], a[k + 1][t], d[k + 1][t]);
#if NIC_MLPN_PIN
#pragma unroll
                    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(d[k + 1][t]));
#endif
                }
                // ---------- output layer (rows 0..2 of a 16-row tile; quarter 0 holds the sample's 3 outputs); training: the fragments
                // of its input are also the image dW_out contracts with (the DZ region is free until the first dZ is stored)
                float yv[3];
                {
                    lds_cbf* const wo_row = opaque((lds_cbf*)(sm + S::OFF_WO + (n16 < 3 ? n16 : 3) * LD + 8 * g));
                    lds_bf* const dz_st = opaque(imgw + n16 * LDZ + 8 * g);
                    f32x4 z3;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        Frag2 af;
                        af.hi = ld_frag(&wo_row[32 * s]);
                        af.lo = ld_frag(&wo_row[S::WOLO + 32 * s]);
                        const Frag2 bf = split_pair(a[NH][2 * s], a[NH][2 * s + 1]);
                        if (TRAIN) {
                            st_frag(&dz_st[32 * s], bf.hi);
                            st_frag(&dz_st[DZLO + 32 * s], bf.lo);
                        }
                        z3 = mfma16_split(af, bf, s == 0 ? f32x4(0.f) : z3);
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) yv[c] = sigmoid_f(z3[c] + ((lds_cf*)Bs)[NH * kH + c]);
                }
                const bool own = valid && g == 0;
                if (p.y != nullptr && own) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) p.y[n * 3 + c] = yv[c];
                }
                if (!TRAIN) {
                    if (p.y_u8 != nullptr && own) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) p.y_u8[n * 3 + c] = (uint8_t)(int)floorf(add_rn(mul_rn(yv[c], 255.0f), 0.5f));
                    }
                    continue;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float gr;
                    if (MODE == MODE_TRAIN_MSE || MODE == MODE_TRAIN_IMG) {
                        const float diff = own ? yv[c] - tgt[c] : 0.f;
                        accLoss += diff * diff;
                        gr = p.grad_scale * diff;
                    } else {
                        gr = own ? tgt[c] : 0.f;
                    }
                    dz3[c] = gr * yv[c] * (1.0f - yv[c]);
                    accBO[c] += dz3[c];
                }
                if (g == 0) {
                    lds_bf* const d3_st = opaque(imgw + OFF_D3 + 4 * (n16 & 3) + (n16 >> 2));
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
            f32x4 dzc[4];                                                   // dZ of the pre-activation feeding a[k + 1]; ends as layer 1's dZ
            {
                const int ln = opaque_i(lane), q4 = (ln & 15) >> 2, p4 = ln & 3, g = ln >> 4;
                lds_bf* const imgw = img0 + wave * S::SPW;
                lds_cbf* const dz_b44 = opaque((lds_cbf*)(imgw + 4 * q4 * LDZ + 16 * g + 4 * p4));
                {   // dW_out[c][pos = lane] += sum_n dZ3[c][n] a_last[pos][n]
                    lds_cbf* const d3_a44 = opaque((lds_cbf*)(imgw + OFF_D3 + (ln & 3) * 16));
                    s16x4 bh[4], bl[4], ah[4], al[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bh[r] = tr4(&dz_b44[r * LDZ]);
                        bl[r] = tr4(&dz_b44[DZLO + r * LDZ]);
                        ah[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[4 * r]);
                        al[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[64 + 4 * r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accWOq = mfma4_bf(al[r], bh[r], accWOq);
                        accWOq = mfma4_bf(ah[r], bl[r], accWOq);
                        accWOq = mfma4_bf(ah[r], bh[r], accWOq);
                    }
                }
                {   // dA_last = W_out^T dZ3, dZ = dA * gelu'
                    lds_cbf* const wo_tr = opaque((lds_cbf*)(sm + S::OFF_WO + q4 * LD + 8 * p4));
                    const float dzv[8] = {dz3[0], dz3[1], dz3[2], 0.f, 0.f, 0.f, 0.f, 0.f};
                    const Frag2 bf = split8(dzv);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const s16x4 ahh = tr4(&wo_tr[32 * (t >> 1) + 4 * (t & 1)]);
                        const s16x4 all = tr4(&wo_tr[S::WOLO + 32 * (t >> 1) + 4 * (t & 1)]);
                        Frag2 af;
                        af.hi = join8(ahh, ahh);
                        af.lo = join8(all, all);
                        dzc[t] = mfma16_split(af, bf, f32x4(0.f)) * d[NH][t];
                    }
                }
            }
#pragma unroll
            for (int k = NH - 1; k >= 0; --k) {
                // hidden layer k: a[k] -> z -> a[k + 1];  dzc = dZ of z
                f32x4 acc[4];
                {
                    const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
                    lds_bf* const imgw = img0 + wave * S::SPW;
                    wave_lds_fence();                                          // this wave's reads of its DZ region are issued: it may be overwritten
                    lds_cbf* const wh_tr = opaque((lds_cbf*)(sm + S::OFF_WH + k * S::WSZ + (4 * g + q4) * LD + 8 * p4));
                    lds_bf* const dz_st = opaque(imgw + n16 * LDZ + 8 * g);
                    lds_bf* const ap_st = opaque(imgw + OFF_AP + n16 * LDZ + 8 * g);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const Frag2 bf = split_pair(dzc[2 * s], dzc[2 * s + 1]);
                        st_frag(&dz_st[32 * s], bf.hi);
                        st_frag(&dz_st[DZLO + 32 * s], bf.lo);
                        const Frag2 ab = split_pair(a[k][2 * s], a[k][2 * s + 1]);     // the layer's input, as the image of its weight gradient
                        st_frag(&ap_st[32 * s], ab.hi);
                        st_frag(&ap_st[DZLO + 32 * s], ab.lo);
                        auto la = [&](int t) {
                            const int co = 32 * (t >> 1) + 4 * (t & 1);
                            Frag2 af;
                            af.hi = join8(tr4(&wh_tr[32 * s * LD + co]), tr4(&wh_tr[(32 * s + 16) * LD + co]));
                            af.lo = join8(tr4(&wh_tr[S::LO + 32 * s * LD + co]), tr4(&wh_tr[S::LO + (32 * s + 16) * LD + co]));
                            return af;
                        };
                        if (s == 0) kstep16<4, true>(acc, bf, la);
                        else kstep16<4>(acc, bf, la);
                    }
                    wave_lds_fence();
                    {   // db[pos = lane] += sum_n dZ[pos][n]
                        lds_cbf* const dz_b44 = opaque((lds_cbf*)(imgw + 4 * q4 * LDZ + 16 * g + 4 * p4));
                        const s16x4 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
                        s16x4 bh[4], bl[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            bl[r] = tr4(&dz_b44[DZLO + r * LDZ]);
                            bh[r] = tr4(&dz_b44[r * LDZ]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            accBH[k] = mfma4_bf(ones, bl[r], accBH[k]);
                            accBH[k] = mfma4_bf(ones, bh[r], accBH[k]);
                        }
                    }
                }
                wg_lds_barrier();
                {   // dW_hidden[k] tile (to, tk) += sum over the four waves' samples of dZ[o][n] a[k][i][n]
                    const int ln = opaque_i(lane), q4 = (ln & 15) >> 2, p4 = ln & 3, h32 = ln >> 5, cg = (ln >> 4) & 1;
                    lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + (4 * q4 + 2 * h32) * LDZ + 32 * to + 16 * cg + 4 * p4));
                    lds_cbf* const ap_t32 = opaque((lds_cbf*)(img0 + OFF_AP + (4 * q4 + 2 * h32) * LDZ + 32 * tk + 16 * cg + 4 * p4));
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        Frag2 af, bf;
                        af.hi = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                        bf.lo = join8(tr4(&ap_t32[v * S::SPW + DZLO]), tr4(&ap_t32[v * S::SPW + DZLO + LDZ]));
                        af.lo = join8(tr4(&dz_t32[v * S::SPW + DZLO]), tr4(&dz_t32[v * S::SPW + DZLO + LDZ]));
                        bf.hi = join8(tr4(&ap_t32[v * S::SPW]), tr4(&ap_t32[v * S::SPW + LDZ]));
                        accWH[k] = mfma_split(af, bf, accWH[k]);
                    }
                }
                wg_lds_barrier();                                              // everyone is done reading before the images are replaced
#pragma unroll
                for (int t = 0; t < 4; ++t) dzc[t] = acc[t] * d[k][t];
            }
            // ---------- layer 1: dX = W1^T dZ1 for the grid slots; the split dZ1 fragments are the dZ1 image
            {
                const int ln = opaque_i(lane), n16 = ln & 15, g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3;
                lds_cbf* const w1_tr = opaque((lds_cbf*)(sm + S::OFF_W1 + (4 * g + q4) * LD + 8 * p4));
                lds_bf* const dz_st = opaque(img0 + wave * S::SPW + n16 * LDZ + 8 * g);
                dxacc[3] = f32x4(0.f);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const Frag2 bf = split_pair(dzc[2 * s], dzc[2 * s + 1]);
                    st_frag(&dz_st[32 * s], bf.hi);
                    st_frag(&dz_st[DZLO + 32 * s], bf.lo);
                    kstep16<4>(dxacc, bf, [&](int t) {
                        const int co = 32 * (t >> 1) + 4 * (t & 1);
                        Frag2 af;
                        af.hi = join8(tr4(&w1_tr[32 * s * LD + co]), tr4(&w1_tr[(32 * s + 16) * LD + co]));
                        af.lo = join8(tr4(&w1_tr[S::LO + 32 * s * LD + co]), tr4(&w1_tr[S::LO + (32 * s + 16) * LD + co]));
                        return af;
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
            wg_lds_barrier();
            {
                const int ln = opaque_i(lane), g = ln >> 4, q4 = (ln & 15) >> 2, p4 = ln & 3, h32 = ln >> 5, cg = (ln >> 4) & 1;
                lds_cbf* const dz_t32 = opaque((lds_cbf*)(img0 + (4 * q4 + 2 * h32) * LDZ + 32 * to + 16 * cg + 4 * p4));
                lds_cbf* const x_t32 = opaque((lds_cbf*)(img0 + OFF_X + (4 * q4 + 2 * h32) * LDX + 32 * tk + 16 * cg + 4 * p4));
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    Frag2 af, bf;
                    af.hi = join8(tr4(&dz_t32[v * S::SPW]), tr4(&dz_t32[v * S::SPW + LDZ]));
                    bf.lo = join8(tr4(&x_t32[v * S::SPW + XLO]), tr4(&x_t32[v * S::SPW + XLO + LDX]));
                    af.lo = join8(tr4(&dz_t32[v * S::SPW + DZLO]), tr4(&dz_t32[v * S::SPW + DZLO + LDZ]));
                    bf.hi = join8(tr4(&x_t32[v * S::SPW]), tr4(&x_t32[v * S::SPW + LDX]));
                    accW1 = mfma_split(af, bf, accW1);
                }
                // columns 64..79 as a 16x16 tile (rows 16 wave ..): two k-steps of 32 samples; quarter G reads source wave 2 uu + (G >> 1),
                // samples 4 q' + 2 (G & 1) + rd
                lds_cbf* const dz_t16 = opaque((lds_cbf*)(img0 + (g >> 1) * S::SPW + (4 * q4 + 2 * (g & 1)) * LDZ + 16 * wave + 4 * p4));
                lds_cbf* const x_t16 = opaque((lds_cbf*)(img0 + (g >> 1) * S::SPW + OFF_X + (4 * q4 + 2 * (g & 1)) * LDX + 64 + 4 * p4));
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    Frag2 af, bf;
                    af.hi = join8(tr4(&dz_t16[2 * uu * S::SPW]), tr4(&dz_t16[2 * uu * S::SPW + LDZ]));
                    af.lo = join8(tr4(&dz_t16[2 * uu * S::SPW + DZLO]), tr4(&dz_t16[2 * uu * S::SPW + DZLO + LDZ]));
                    bf.hi = join8(tr4(&x_t16[2 * uu * S::SPW]), tr4(&x_t16[2 * uu * S::SPW + LDX]));
                    bf.lo = join8(tr4(&x_t16[2 * uu * S::SPW + XLO]), tr4(&x_t16[2 * uu * S::SPW + XLO + LDX]));
                    accT = mfma16_split(af, bf, accT);
                }
            }
            wg_lds_barrier();
        }  // rounds of one macro-tile

        if (TRAIN) {
            const int ln = opaque_i(lane), g = ln >> 4;
            combine_g1_lanes16(gacc, blk_off1, blk, ln, lw);
            bool flush = true;
            if (NIC_GROUP_SUM && p.rg_log2 > 0) {
                lds_f* const reg0 = (lds_f*)img0;
                constexpr int REGION = S::SPW / 2;
                int leader = wave;
                if (tile_ok)
                    for (int w = wave - 1; w >= 0; --w)
                        if (((base + w) >> p.rg_log2) == tile) leader = w;
                if (leader != wave) {
                    lds_f* const mine = opaque(reg0 + wave * REGION + ln);
#pragma unroll
                    for (int i = 0; i < 12; ++i) mine[i * 64] = dxacc[i >> 2][i & 3];
#pragma unroll
                    for (int i = 0; i < 12; ++i) mine[(12 + i) * 64] = gacc.g1[i];
                }
                wg_lds_barrier();
                if (leader == wave && tile_ok) {
                    for (int w = wave + 1; w < 4; ++w) {
                        if (base + w >= t_end || ((base + w) >> p.rg_log2) != tile) break;
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
            if (NIC_T16_PREADD >= 2 && p.rg_log2 == 0) {                    // launch-uniform
                static_assert(13 * 64 <= S::SPW / 2, "pre-add scratch");
                preadd_y16<4>(dxacc, blk_off0, (uint32_t)p.g0.nx, ln, wave, (lds_f*)img0, S::SPW / 2, [&]() { wg_lds_barrier(); });
            }
            if (flush) {
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
    }  // macro-tile loop

    if (!TRAIN) return;
    // ---------------- one record per workgroup
    float* rec = p.partials + (int64_t)blockIdx.x * S::REC;
#pragma unroll
    for (int k = 0; k < NH; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[S::REC_WH + (k * 4 + wave) * 1024 + r * 64 + lane] = accWH[k][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) rec[S::REC_W1 + wave * 1024 + r * 64 + lane] = accW1[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) rec[S::REC_TAIL + wave * 256 + r * 64 + lane] = accT[r];
    float* tail = rec + S::REC_WAVE + wave * S::WTAIL;
#pragma unroll
    for (int k = 0; k < NH; ++k) tail[k * 64 + lane] = accBH[k][0];
#pragma unroll
    for (int c = 0; c < 3; ++c) tail[NH * 64 + 64 * c + lane] = accWOq[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = c < 3 ? accBO[c] : accLoss;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) tail[NH * 64 + 192 + c] = v;
    }
}

// Fixed-order reduction of the records of fused_mlpn_kernel.  Output index space, layer by layer:
// W1 [64][73] | b1 [64] | (W_hidden [64][64] | b_hidden [64]) x NH | W_out [3][64] | b_out [3] | loss.
template <class L, int NL>
__global__ void __launch_bounds__(256) reducen_kernel(const float* partials, int n_rec, nic_mlp_grads gr, float* loss, float loss_scale, const StepTail tl) {
    if (tail_block(tl)) return;                                   // a streaming block of the optimiser tail (nic_adam.hpp)
    using S = LdsN<NL>;
    constexpr int NH = S::NH;
    constexpr int N_W1 = kH * L::CIN, N_HID = kH * kH + kH;
    constexpr int N_OUT = N_W1 + kH + NH * N_HID + 3 * kH + 3 + 1;
    __shared__ float red[8][32];
    const int slice = threadIdx.x >> 5;
    const int gid = blockIdx.x * 32 + (threadIdx.x & 31);
    const bool live = gid < N_OUT;
    int nsrc = 0, off0 = 0, stride = 0;
    float* dst = nullptr;
    auto tile32 = [](int i, int j) { return ((i & 3) + 4 * (i >> 3)) * 64 + j + 32 * ((i >> 2) & 1); };
    if (live) {
        int t = gid;
        if (t < N_W1 + kH) {
            int o, ch;
            if (t < N_W1) { o = t / L::CIN; ch = t - o * L::CIN; dst = gr.w[0] ? gr.w[0] + t : nullptr; }
            else { o = t - N_W1; ch = kSlotOne; dst = gr.b[0] ? gr.b[0] + o : nullptr; }
            const int rho = rho16_of_channel(ch), po = pos16(o);
            if (rho < 64) {
                off0 = S::REC_W1 + (2 * (po >> 5) + (rho >> 5)) * 1024 + tile32(po & 31, rho & 31);
                nsrc = 1;
            } else {
                const int m = po & 15;
                off0 = S::REC_TAIL + (po >> 4) * 256 + (m & 3) * 64 + 16 * (m >> 2) + (rho - 64);
                nsrc = 1;
            }
        } else {
            t -= N_W1 + kH;
            if (t < NH * N_HID) {
                const int k = t / N_HID, u = t - k * N_HID;
                if (u < kH * kH) {
                    const int o = u / kH, i = u - o * kH;
                    const int po = pos16(o), pi = pos16(i);
                    off0 = S::REC_WH + (k * 4 + 2 * (po >> 5) + (pi >> 5)) * 1024 + tile32(po & 31, pi & 31);
                    nsrc = 1;
                    dst = gr.w[1 + k] ? gr.w[1 + k] + u : nullptr;
                } else {
                    off0 = S::REC_WAVE + k * 64 + pos16(u - kH * kH); nsrc = 4; stride = S::WTAIL;
                    dst = gr.b[1 + k] ? gr.b[1 + k] + (u - kH * kH) : nullptr;
                }
            } else {
                t -= NH * N_HID;
                nsrc = 4; stride = S::WTAIL;
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
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int per = (n_rec + 7) >> 3;
    const int w_lo = slice * per, w_hi = (w_lo + per < n_rec) ? w_lo + per : n_rec;
    if (live) {
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
    red[slice][threadIdx.x & 31] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    __syncthreads();
    if (slice != 0 || !live || dst == nullptr) return;
    float acc = red[0][threadIdx.x];
#pragma unroll
    for (int sl = 1; sl < 8; ++sl) acc += red[sl][threadIdx.x];
    if (gid == N_OUT - 1) *dst = acc * loss_scale;
    else tail_store(tl, dst, acc);
}

}  // namespace nic
