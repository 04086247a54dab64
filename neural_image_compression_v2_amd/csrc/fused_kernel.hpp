// Fused encode + decoder MLP forward / backward for gfx950 (CDNA4): fused_kernel<Layout, SRC, MODE, GT, PREC>.
//
//   SRC   SRC_ENCODE: the input rows are built from the grids in-kernel (gathers, blend, PE, noise);  SRC_MEMORY: read from [N, Cin]
//   MODE  MODE_INFER | MODE_TRAIN_MSE (target tensor) | MODE_TRAIN_IMG (target read from the resident image) | MODE_TRAIN_DY (incoming dY)
//   GT    grid element type: float, or uint8_t = the stored codec, dequantised in the gather (inference)
//   PREC  PREC_F32: every matrix product on v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain);
//         PREC_SPLIT (2D training, every layout's inference): every matrix product on the bf16 matrix pipe with hi + lo bf16
//         operand pairs and fp32 accumulation - see "split-bf16 matrix products" below; the fp32-input MFMA blocks the wave's
//         vector issue, the bf16 one does not;  PREC_CHAIN (3D training): only the four chained products that way.
//
// Work decomposition (both precisions)
//   * one wave = one tile of 32 samples at a time; a 256-thread workgroup is 4 waves that share the decoder weights in LDS and,
//     in training, split the ownership of the weight-gradient tiles (4 LDS barriers per round);
//   * persistent grid (<= 1 workgroup per CU for training: a wave keeps the decoder-gradient accumulator tiles it owns in
//     registers for the whole launch), XCD-aware work order: blocks b and b+8 share an XCD / L2, so each XCD walks one
//     contiguous range of work units and neighbouring units re-use the same grid cells out of that XCD's L2;
//   * sample-on-lane orientation: every activation matrix is [features, 32 samples] with the sample on the MFMA column
//     (lane & 31).  A 32x32 accumulator register r of lane-half h then holds feature row ROW(r,h) of the lane's own sample,
//     which is precisely the B operand the next product wants: the forward layers and the backward dA products chain
//     through registers with no LDS traffic.  Only the weight-gradient products, which contract over the sample index, go
//     through wave-private LDS images ([feature][sample] fp32, or [sample][feature] bf16 read with ds_read_b64_tr_b16);
//   * lanes map to grid CELLS, not to consecutive samples: a wave owns a 16 x 2 (x 1 in 3D) block of G0 cells, lanes x-fastest,
//     and walks the m^D samples inside a cell (m = 1/step_number: 16 rounds at mip 0) one per round.  All samples of a lane
//     share their G0 corners and their G1 cell, so the input gradient - which leaves the last backward product in the
//     registers of the lane that knows the slot's grid address - is summed in registers over the rounds and scattered ONCE
//     per cell with fp32 atomics (the reference's index_put_(accumulate=True), image_compression.py:265), the G1 sums of the
//     lanes that share a G1 cell combined across lanes first.  (Per-sample atomics ran at the contended memory-side rate;
//     LDS float atomics at about one lane per clock - both measured and dropped, see DESIGN.md.)
//   * decoder-gradient accumulators are written once per workgroup to a workspace and summed in a fixed order by
//     reduce_partials_kernel (bit-stable decoder gradients and loss for a given grid size).
#pragma once
#include "nic_device.hpp"
#include "nic_adam.hpp"
#ifndef NIC_GROUP_SUM
#define NIC_GROUP_SUM 1
#endif
#ifndef NIC_STAGGER_RG
#define NIC_STAGGER_RG 1      // also when the rounds of a macro-tile are dealt out in groups (small launches)
#endif
#ifndef NIC_PHASES
#define NIC_PHASES 4
#endif
#ifndef NIC_GX_F32
#define NIC_GX_F32 1
#endif
#ifndef NIC_GX
#define NIC_GX 1
#endif
#ifndef NIC_STAGGER
#define NIC_STAGGER 1
#endif
#ifndef NIC_HOIST_INFER
#define NIC_HOIST_INFER -1  // inference: -1 = by layout (4-corner G0 layouts: both grids; 3D method 3: G0 only - 96 raw values per lane spill at 2 waves per SIMD)
#endif
#ifndef NIC_HOIST_3D
#define NIC_HOIST_3D 0
#endif
#ifndef NIC_HOIST_SPLIT
#define NIC_HOIST_SPLIT 3
#endif
#ifndef NIC_HOIST_TRAIN
#define NIC_HOIST_TRAIN -1     // -1: per layout (below); bit 0: G0 raw values gathered once per macro-tile, bit 1: G1
#endif

namespace nic {

enum { SRC_ENCODE = 0, SRC_MEMORY = 1 };
enum { MODE_INFER = 0, MODE_TRAIN_MSE = 1, MODE_TRAIN_DY = 2, MODE_TRAIN_IMG = 3 };   // IMG: MSE against a resident image

// Pin the GELU derivatives where the forward pass computes them (fused_q16.hpp::pin has the story): the compiler otherwise sinks the last steps of each
// derivative to its use in the backward pass and carries three values per derivative through the round.  32-sample kernels, measured (interleaved A/B):
// spilled registers method 3 130 -> 16 (split) / 115 -> 1 (fp32), method 4 112 -> 4, 2D 20 -> 0; 128^3 split steps 1.42 -> 0.94 ms (method 3), 1.04 -> 0.81
// (method 4); the fp32 4K launch 4.58 -> 4.40 ms.  (fused_train16, which does not spill, LOSES 3.9 % with the same pin: there the sinking spreads vector
// work into the backward pass.)
#ifndef NIC_PREADD_Y_SMALL
#define NIC_PREADD_Y_SMALL 1      // 1: the y pre-add only where FusedParams.preadd_y says so (small launches); 0: always (round 2's behaviour)
#endif
#ifndef NIC_FK_PIN
#define NIC_FK_PIN 1
#endif
// one level pair of a multi-level launch (fused_q16.hpp::QML)
struct MlPair {
    GridView g0, g1;
    float* g0_grad;
    float* g1_grad;
};
struct FusedParams {
    nic_path_desc d;
    GridView g0, g1;
    float* g0_grad;
    float* g1_grad;
    MlPair ml[NIC_ML_MAX_LEVELS];      // multi-level layouts only: pair l (pair 0 repeats g0 / g1 / their gradients); pair l's step is 2^(d.log2_step - 2 l)
    const int32_t* origins;            // device memory - or null: the origins ride in the kernel arguments (NIC_FLAG_ORIGINS_HOST: a step of a host loop
    int32_t org_inl[NIC_ORIGINS_INLINE_MAX * 3];   // whose origins are host values needs no upload and no dependent global load in front of its gathers)
    const float* W[NIC_MAX_LINEAR];
    const float* b[NIC_MAX_LINEAR];
    int n_linear;
    NoiseSrc noise;
    const float* x;        // SRC_MEMORY: [n, cin]
    float* dx;             // SRC_MEMORY training: [n, cin] or null
    const float* target;   // MODE_TRAIN_MSE: [N, 3]
    const float* dy;       // MODE_TRAIN_DY:  [N, 3]
    // MODE_TRAIN_IMG: the target of sample (crop, i) is image[c][origin + i] of a resident [3, (S2,) S1, S0]-strided image (the
    // reference's dataset tensor, image_compression.py:37-47): no [N, 3] target tensor is ever materialised
    const void* timg;
    int64_t timg_cs, timg_s[3];      // element strides: channel, sample axes 0..2
    int timg_u8;                     // 1: uint8 codes, target = u / timg_den (ToTensor: 255; the 3D loader: 256)
    float timg_den, timg_rcp;
    float* y;              // [N, 3] or null
    uint8_t* y_u8;         // [N, 3] or null: quantize_to_bit(y) as bytes (models.py:39-40)
    float dq_sub, dq_den, dq_rcp;   // uint8 grids (decode from the stored codec): value = (u - dq_sub) / dq_den (models.py:68-71)
    int grid_u8;
    int grid_kind;                  // 0: fp32 grids; 1: bfloat16, 2: IEEE half storage, widened in the gather (fused_train16 / fused_mlpn kernels)
    float* partials;       // workspace, [n_waves][REC]
    int64_t n_total, n_per_crop, n_tiles, tiles_per_crop;   // tiles = macro-tiles of TX x TY x TZ cell blocks (SRC_MEMORY: 32 rows)
    int tiles_y, tiles_z;
    // Edge tiles: when the last column of macro-tiles along x would hold at most 8 of its 16 cell blocks (always so for an unaligned
    // crop of a power-of-two extent: 64 + 1 blocks), that column is covered by tiles of 2^edge_lw x 32 / 2^edge_lw blocks instead
    // (1 x 32, 2 x 16, 4 x 8 or 8 x 4): 3 tiles instead of 33 for the 65 x 65 blocks of an unaligned 256^2 crop.  Tiles
    // [0, tiles_main) of a crop are the regular ones (full_x columns), the rest the edge tiles; edge_lw < 0: none.
    int64_t tiles_main;
    int full_x, edge_lw;
    int lm, niter;         // cell block = 2^lm samples per axis (lm = max(0, -log2_step)); niter = 2^(lm * dim) rounds per macro-tile and pass
    int passes;            // training: a macro-tile runs niter * passes rounds (nic_path_desc.passes); round it = pass (it / niter), sample it % niter
    int rg_log2;           // the rounds of a macro-tile are dealt out in 2^rg_log2 groups (work units): small launches balance better
    int pk_bx, pk_by, pk_nc;   // fused_kernel, packed tiling (pk_nc > 0): a macro-tile is 32 CONSECUTIVE cell blocks of the crop's pk_bx x pk_by x .. block
                           // list (x fastest), pk_nc blocks per crop - crops whose block counts are far from multiples of the 16 x 2 wave block
                           // (the 9 x 9 x 9 blocks of an unaligned 32^3 crop: 23 macro-tiles instead of 45)
    int64_t seg_split;     // macro-tiles [0, seg_split) run in 2^rg0_log2 groups of rounds (fused_train16: whole units), the rest in 2^rg_log2 groups; 0 = one segment
    int rg0_log2;
    float grad_scale;      // 2 * loss_scale
    int f16;               // plain 16-bit products on IEEE half operands (NIC_FLAG_FP16) instead of bfloat16
    float dz_scale, dz_unscale;   // fp16 products (NIC_FLAG_FP16): dZ is carried as dz_scale x dZ, dz_scale = 2^k chosen by the host so that 2 loss_scale 2^k is O(1)
                           // (gradients of a mean over millions of samples are far below the fp16 range); every sum of dZ products is multiplied by
                           // dz_unscale = 2^-k where it leaves the kernel (records, grid-gradient atomics): exact.  1 / 1 in every other mode
    // hipGraph-captured training loops (nic_fused_forward_backward_img_dev): the step number lives in DEVICE memory and is added to the
    // noise offset at kernel start, so one captured launch serves every step (fused_train16 / fused_q16 kernels; null everywhere else)
    const int64_t* step_dev;
    // two-waves-per-SIMD kernels: pre-add the G0 sums along y across the waves of a workgroup before the flush (two barriers + an LDS exchange per
    // macro-tile: halves the atomics - what small launches are bound by - and costs a launch that fills the chip many times over ~ 0.9 %): set by the host
    // for launches of at most 4 macro-tiles per wave of the chip
    int preadd_y;
};

// origin component `idx` (= crop * DIM + axis) of the launch
__device__ __forceinline__ int origin_of(const FusedParams& p, int idx) { return p.origins != nullptr ? p.origins[idx] : p.org_inl[idx]; }
// the launch's noise source with the device-side step added to its offset (uniform: scalar registers)
__device__ __forceinline__ NoiseSrc noise_with_step(const NoiseSrc& base, const int64_t* step_dev) {
    NoiseSrc ns = base;
    if (step_dev != nullptr) {
        const uint64_t o = (((uint64_t)ns.off_hi << 32) | ns.off_lo) + (uint64_t)*step_dev;
        ns.off_lo = (uint32_t)o;
        ns.off_hi = (uint32_t)(o >> 32);
    }
    return ns;
}

// Prologue staging: thread `tid` of NT first ISSUES all of its global loads (elements tid, tid + NT, ...), then converts and stores:
// one memory round trip for the whole image instead of one per element (the element-at-a-time loops made 20 - 40 dependent round
// trips, 20+ us per launch - a tenth of the reference's default step).
template <int N, int NT, class Load, class Store>
__device__ __forceinline__ void stage_all(int tid, Load&& ld, Store&& st) {
    constexpr int IT = (N + NT - 1) / NT;
    float v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + NT * i;
        v[i] = (N % NT == 0 || idx < N) ? ld(idx) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + NT * i;
        if (N % NT == 0 || idx < N) st(idx, v[i]);
    }
}

template <class L>
struct Lds {
    static constexpr int KPAD = 2 * L::NSLOT;
    static constexpr int KT = (L::NSLOT + 15) / 16;          // 32-row tiles of the padded input
    static constexpr int LD1 = KPAD + 4;                      // row strides are 4 * odd: conflict-free b128 rows
    static constexpr int LD2 = kH + 4;
    static constexpr int LDT = 36;                            // transposed scratch: 32 samples + 4
    static constexpr int NACC = 2 * KT + 4;                   // MFMA accumulator tiles: dW1 [2][KT], dW2 [2][2]
    static constexpr int TAIL = 64 + 192 + 4;                 // db2[64], dW3[3][64], db3[3], loss
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = OFF_W1 + kH * LD1;
    static constexpr int OFF_W3 = OFF_W2 + kH * LD2;
    static constexpr int OFF_B2 = OFF_W3 + 4 * LD2;           // W3 rows 0..2 + one zero row (k padding of dA2)
    static constexpr int OFF_B3 = OFF_B2 + kH;
    static constexpr int OFF_SCR = OFF_B3 + 16;
    static constexpr int SCR_PER_WAVE = 128 * LDT;            // SA (64 rows) + SB (64 rows), wave-private
    static constexpr int TOTAL_INFER = OFF_SCR;
    // Where the 160 KiB allow it (2D layouts), every wave also keeps a permanent transposed image of its input X
    // ([KPAD rows][32 samples]): the input registers die right after layer 1 and dW1 needs no re-staging of X.
    static constexpr int XIMG_PER_WAVE = KPAD * LDT;           // rows beyond KPAD of the last 32-row tile do not exist
    static constexpr bool XIMG = (OFF_SCR + 4 * (SCR_PER_WAVE + XIMG_PER_WAVE)) * 4 <= 163840;
    static constexpr int OFF_XIMG = OFF_SCR + 4 * SCR_PER_WAVE;
    static constexpr int TOTAL_TRAIN = OFF_XIMG + (XIMG ? 4 * XIMG_PER_WAVE : 0);
    // Decoder-gradient bookkeeping.  The 4 waves of a workgroup split OWNERSHIP of the dW output tiles: each wave
    // contracts its tiles over the transposed operands of all 4 waves (K = 128 samples per round), so a wave carries
    // 3-4 accumulator tiles (48-64 registers) through the launch instead of all NACC (160-192).  The odd col-tile of dW1
    // (KT odd: 2D and method 4) is contracted by every wave over its own samples only ("partial" tiles, summed later).
    static constexpr int KTF = KT & ~1;                       // dW1 col tiles handled in cross-wave chunks of two
    static constexpr int PART = KT & 1;
    // When the odd col tile holds at most 16 real rows (2D: rho 64..79) it is contracted with 16x16x4 MFMAs - 4 row tiles of
    // [16 o][16 rho], half the matrix-pipe time of two 32x32 tiles that are half padding, and 16 accumulator registers
    // instead of 32.
    static constexpr bool PART16 = PART && XIMG && (KPAD - 32 * (KT - 1) <= 16);
    static constexpr int NSLOT_REC = NACC + (PART ? (PART16 ? 4 : 8) : 0);   // + per-wave partial tiles: [4][1024] or [4][2][1024]
    static constexpr int REC = NSLOT_REC * 1024 + 4 * 320;    // floats per WORKGROUP record; tails per wave: db2, dW3, db3, loss
    // PREC_SPLIT: the wave-private region holds bf16 images with the SAMPLE as the row (32 rows), hi and lo of each: DZ (a2, then
    // dZ2, then dZ1: 64 columns), A1 (64 columns), X (KPAD columns), and dZ3 as [4][32].  Every consumer reads them with
    // ds_read_b64_tr_b16 (sizes in bf16 elements).
    // row strides of 34 / 42 dwords (= 2 mod 4): the fragment stores (32 lanes, 32 rows, one 8-byte chunk) are conflict-free, the
    // transposed reads 2-way (72 / 88 elements had both 2-way: 39 % of the LDS-active cycles were conflicts)
    static constexpr int LDZB = kH + 4, LDXB = KPAD + 4;
    static constexpr int BZ = 32 * LDZB, BX = 32 * LDXB;
    static constexpr int SPW = 4 * BZ + 2 * BX + 256;         // per wave
    static constexpr int W3B = 2 * 16 * LD2;                  // W3 as a bf16 image, hi + lo: 16 rows (3 real), LD2 elements per row
    static constexpr int OFF_IMG = OFF_SCR + W3B / 2;         // (floats) the wave images start behind it
    static constexpr int TOTAL_INFER_SPLIT = OFF_IMG;
    static constexpr int TOTAL_SPLIT = OFF_IMG + 4 * SPW / 2;
    static_assert(SPW % 8 == 0, "16-byte aligned wave regions");
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// 16x16x4: A row / B col = lane & 15, k = lane >> 4; D register r of lane l = row 4 (l >> 4) + r, col l & 15
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// ---- split-bf16 matrix products (PREC_SPLIT): an fp32 operand x is carried as hi + lo, hi = bf16(x), lo = bf16(x - hi) (16
// significant bits), and a product as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 x 32 cycles per
// K = 16 instead of 8 x 64 for the fp32 form, and - unlike the fp32 form - the bf16 pipe runs beside the wave's
// own VALU work.  Operand layouts (checked with exact integer data by ab/micro/bf16_chain.hip): accumulator registers 8s..8s+7
// of a tile, converted pairwise, are the B fragment of k-step s (element j of lane half g = row 16s + 8(j>>2) + 4g + (j&3));
// the A fragment of a natural [row][k] bf16 image is two 8-byte reads at k = 16s + 4g and + 8 (row-wise), or two
// ds_read_b64_tr_b16 of the 4 x 16 blocks at rows 16s + 4g and + 8 (the transposed product), from ONE image.
// PREC_CHAIN: only the four chained products (layer 1, layer 2, dA1, dX) in split bf16; the weight-gradient products, layer 3 and
// dA2 stay fp32 on the fp32 [feature][sample] images - for the 3D training kernels, whose split images do not fit the LDS
enum { PREC_F32 = 0, PREC_SPLIT = 1, PREC_CHAIN = 2 };
// between the k-steps of a split product the scheduler is left free: the next step's operand split (VALU) and LDS reads run
// beside the current step's bf16 MFMAs (3.51 -> 3.42 ms against a scheduling barrier per step)
#ifndef NIC_DW_SB
#define NIC_DW_SB __builtin_amdgcn_sched_barrier(0)   // keeps the one-step-ahead order of the weight-gradient loops
#endif
#ifndef NIC_CHAIN_DW
#define NIC_CHAIN_DW 1        // PREC_CHAIN: the weight-gradient products too run in split bf16, their operands split on read from the fp32 images
#endif
#ifndef NIC_CHAIN_DW_PART
#define NIC_CHAIN_DW_PART 1
#endif
#ifndef NIC_SPLIT_SB
#define NIC_SPLIT_SB ((void)0)
#endif
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) __bf16 lds_bf;
typedef __attribute__((address_space(3))) const __bf16 lds_cbf;
typedef __attribute__((address_space(3))) const s16x4 lds_cs16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ lds_cbf* opaque(lds_cbf* p) {
#ifndef NIC_NO_OPAQUE_PTR
    asm volatile("" : "+v"(p));
#endif
    return p;
}
__device__ __forceinline__ f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
struct Frag2 {
    bf16x8 hi, lo;
};
// 8 fp32 values -> hi / lo fragments.  x - hi is exact in fp32, so hi + lo carries 16 significant bits of x.
__device__ __forceinline__ Frag2 split8(const float (&x)[8]) {
    u32x4 hp, lp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));       // v_cvt_pk_bf16_f32, RNE
        const f32x2 r = {x[2 * i] - __builtin_bit_cast(float, h << 16), x[2 * i + 1] - __builtin_bit_cast(float, h & 0xFFFF0000u)};
        hp[i] = h;
        lp[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    }
    Frag2 f;
    f.hi = __builtin_bit_cast(bf16x8, hp);
    f.lo = __builtin_bit_cast(bf16x8, lp);
    return f;
}
__device__ __forceinline__ Frag2 split_acc(const f32x16& t, int s) {            // registers 8s .. 8s+7 of an accumulator tile
    const float x[8] = {t[8 * s], t[8 * s + 1], t[8 * s + 2], t[8 * s + 3], t[8 * s + 4], t[8 * s + 5], t[8 * s + 6], t[8 * s + 7]};
    return split8(x);
}
__device__ __forceinline__ bf16x8 join8(s16x4 a, s16x4 b) {
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// A fragment, row-wise: p = &image[row][k0 + 4g]; elements k0 + 4g .. + 3 and k0 + 8 + 4g .. + 3
__device__ __forceinline__ bf16x8 frag_row(lds_cbf* p) {
    return join8(*reinterpret_cast<lds_cs16x4*>(p), *reinterpret_cast<lds_cs16x4*>(p + 8));
}
// A fragment of the transposed product: p = &image[r0 + 4g + q][c0 + 16cg + 4p'] (lane 4q + p' of its 16-lane group), ld = row stride
template <int LD>
__device__ __forceinline__ bf16x8 frag_tr(lds_cbf* p) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * LD));
    return join8(a, b);
}
__device__ __forceinline__ void store_frag(lds_bf* p, const bf16x8& f) {        // elements 0..3 at p, 4..7 at p + 8
    const s16x8 v = __builtin_bit_cast(s16x8, f);
    *reinterpret_cast<lds_s16x4*>(p) = s16x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<lds_s16x4*>(p + 8) = s16x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ s16x4 tr4(lds_cbf* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p); }
__device__ __forceinline__ f32x4 mfma16_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma4_bf(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0); }
// the same for the [sample][feature] images: base row r and r + 1, each read combining rows +0, +8, +16, +24 (set up in p)
template <int LD>
__device__ __forceinline__ bf16x8 frag_trs(lds_cbf* p) {
    return join8(tr4(p), tr4(p + LD));
}
// acc += A (hi, lo) x B (hi, lo) without the lo x lo term (2^-18 relative)
__device__ __forceinline__ f32x16 mfma_split(const Frag2& a, const Frag2& b, f32x16 c) {
    c = mfma_bf(a.lo, b.hi, c);
    c = mfma_bf(a.hi, b.lo, c);
    return mfma_bf(a.hi, b.hi, c);
}

// LDS traffic between lanes of ONE wave: DS operations of a wave complete in order, so only the
// compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Phase timing for a DIAGNOSTIC build (-DNIC_STAMPS, see ab/stamps.py): s_memtime at phase boundaries, per-phase sums kept in
// scalars and dumped by lane 0 of every wave to the unused tail of the workspace.  The shipped library is built without it.
#ifdef NIC_STAMPS
#define NIC_NPH 16
#define STAMP(ph)                                                                                   \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        stamp_sum[ph] += t_ - stamp_last;                                                           \
        stamp_last = t_;                                                                            \
    } while (0)
#else
#define STAMP(ph) do { } while (0)
#endif

// Workgroup barrier for LDS hand-offs.  NOT __syncthreads(): that also drains vmcnt, i.e. waits for the previous tile's
// fire-and-forget gradient atomics (thousands of cycles); only this wave's LDS traffic has to be complete here.
__device__ __forceinline__ void wg_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Makes a per-lane LDS base pointer opaque to the optimiser at this point of the tile loop: everything
// derived from it (base + compile-time constant) stays in the loop body, where instruction selection
// folds the constant into the DS instruction's 16-bit offset field.  Without this, loop-invariant code
// motion hoists one VGPR per unrolled LDS access out of the loop (hundreds) and spills them.
typedef __attribute__((address_space(3))) float lds_f;            // 32-bit LDS pointers: stay DS instructions
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
__device__ __forceinline__ lds_f* opaque(lds_f* p) {
#ifndef NIC_NO_OPAQUE_PTR
    asm volatile("" : "+v"(p));
#endif
    return p;
}
__device__ __forceinline__ f32x4 ld4(lds_cf* p) { return *reinterpret_cast<lds_cf4*>(p); }
// PREC_CHAIN weight gradients: 8 consecutive samples of one row of a [feature][sample] fp32 image, split on the way to the bf16 pipe
// (the images keep their fp32 layout and size - the [sample][feature] bf16 images of PREC_SPLIT do not fit beside 3D weights)
struct Row8 {
    f32x4 a, b;
};
__device__ __forceinline__ Row8 ld_row8(lds_cf* p) { return Row8{ld4(p), ld4(p + 4)}; }
__device__ __forceinline__ Frag2 split_row8(const Row8& r) {
    const float x[8] = {r.a[0], r.a[1], r.a[2], r.a[3], r.b[0], r.b[1], r.b[2], r.b[3]};
    return split8(x);
}
// ds_add_f32 (no return): wave-private accumulation, workgroup scope is plenty
__device__ __forceinline__ void lds_add(lds_f* p, float v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__host__ __device__ constexpr int ROWC(int r) { return (r & 3) + 8 * (r >> 2); }   // ROW(r,h) = ROWC(r) + 4h

// what the scatter needs to recompute a slot's grid address
struct EncCtx {
    uint32_t off0, off1;    // element offsets of corner (0,0,0) in G0 / G1 (channel 0); a channel plane is < 2^32 elements
    float kx, ky, kz;       // G1 interpolation fractions
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- G1 interpolation factors.  The reference multiplies corner q by one factor per axis, k or 1-k,
// chosen by a bit triple (see g1_ref_weight_bits): fx[bit], fy[bit], fz[bit].  NIC_G1_UNWEIGHTED (the
// reference's step_number == 2 case) is the same arithmetic with every factor = 1, which is bit-identical
// to not multiplying.  Everything here is selects on launch-uniform flags: no branches in the tile loop.
// c ? a1 : a0 for two values LOADED from one struct (a local one, or the kernel arguments): written as a plain select, the optimiser folds it into
// ONE load with a computed offset - which keeps a local struct in memory (scratch, or "promoted" to LDS) and makes every thread copy a by-value
// kernel-argument struct to scratch first (method 4's plain-bf16 kernels: 632 bytes per thread).  The empty asm hides the loads from that fold.
__device__ __forceinline__ float pick_loaded(bool c, float a1, float a0) {
    asm("" : "+v"(a1));
    asm("" : "+v"(a0));
    return c ? a1 : a0;
}
// (not "if (__builtin_constant_p(c)) return c ? a1 : a0;" for the 2D kernels, whose corner bits are compile-time constants: with it the 2D split
//  inference kernel returned outputs that changed from run to run by 1e-3 - the callers say which form they need instead)

template <int DIM>
struct G1FactorsT;
template <>
struct G1FactorsT<2> {          // the corner bits are compile-time constants: every select below folds away
    float fx[2], fy[2], fz[2];
    uint32_t bits;          // 3 bits per corner q: bx | by<<1 | bz<<2
};
template <>
struct G1FactorsT<3> {
    // 3D: the weight mode - the bits - is a launch parameter.  Separate members and pick_loaded, not float[2] and a plain select: "b ? f[1] : f[0]" is
    // folded into an indexed load f[b], which keeps the struct in memory - the 3D kernels then carried it in LDS (28 bytes per thread: "promote
    // alloca to LDS"; in scratch where the LDS was full) and, for the flat thread id that addressing needs, read the workgroup size from the dispatch
    // packet in HOST memory at every launch (stamps: 20 K cycles of the method-3 prologue)
    float fx0, fx1, fy0, fy1, fz0, fz1;
    uint32_t bits;
    __device__ __forceinline__ float x(uint32_t b) const { return pick_loaded((b & 1u) != 0, fx1, fx0); }
    __device__ __forceinline__ float y(uint32_t b) const { return pick_loaded((b & 2u) != 0, fy1, fy0); }
    __device__ __forceinline__ float z(uint32_t b) const { return pick_loaded((b & 4u) != 0, fz1, fz0); }
};
template <int DIM>
__device__ __forceinline__ G1FactorsT<DIM> g1_factors(int mode, float kx, float ky, float kz) {
    G1FactorsT<DIM> f;
    const bool unw = mode == NIC_G1_UNWEIGHTED;
    if constexpr (DIM == 2) {
        f.fx[0] = unw ? 1.0f : 1.0f - kx; f.fx[1] = unw ? 1.0f : kx;
        f.fy[0] = unw ? 1.0f : 1.0f - ky; f.fy[1] = unw ? 1.0f : ky;
        f.fz[0] = unw ? 1.0f : 1.0f - kz; f.fz[1] = unw ? 1.0f : kz;
    } else {
        f.fx0 = unw ? 1.0f : 1.0f - kx; f.fx1 = unw ? 1.0f : kx;
        f.fy0 = unw ? 1.0f : 1.0f - ky; f.fy1 = unw ? 1.0f : ky;
        f.fz0 = unw ? 1.0f : 1.0f - kz; f.fz1 = unw ? 1.0f : kz;
    }
    // corner offsets: 2D q -> (dx = q>>1, dy = q&1); 3D q -> (dx = q>>2, dy = (q>>1)&1, dz = q&1)
    constexpr uint32_t nat2 = (0u) | (2u << 3) | (1u << 6) | (3u << 9);
    constexpr uint32_t nat3 = (0u) | (4u << 3) | (2u << 6) | (6u << 9) | (1u << 12) | (5u << 15) | (3u << 18) | (7u << 21);
    constexpr uint32_t ref3 = (0u) | (4u << 3) | (2u << 6) | (1u << 9) | (3u << 12) | (5u << 15) | (6u << 18) | (7u << 21);   // Q1
    f.bits = DIM == 2 ? nat2 : (mode == NIC_G1_REFERENCE ? ref3 : nat3);
    return f;
}
template <int DIM>
__device__ __forceinline__ float g1_corner_factor(const G1FactorsT<DIM>& f, int q) {      // d(blend)/d(corner q)
    const uint32_t b = (f.bits >> (3 * q)) & 7u;
    if constexpr (DIM == 2) {
        float w = ((b & 1u) ? f.fx[1] : f.fx[0]) * ((b & 2u) ? f.fy[1] : f.fy[0]);
        return w;
    } else {
        float w = f.x(b) * f.y(b);
        w *= f.z(b);
        return w;
    }
}
// blend of one channel: ((g * fx) * fy) (* fz), corners added left to right (fp_def.py:141-144, 176-183;
// image_compression.py:95) with individually rounded products and sums
template <int DIM>
__device__ __forceinline__ float g1_blend(const float* pc, const GridView& g, const G1FactorsT<DIM>& f) {
    constexpr int K1 = DIM == 2 ? 4 : 8;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < K1; ++q) {
        const int dx = DIM == 2 ? (q >> 1) : ((q >> 2) & 1), dy = DIM == 2 ? (q & 1) : ((q >> 1) & 1), dz = DIM == 2 ? 0 : (q & 1);
        const uint32_t b = (f.bits >> (3 * q)) & 7u;
        float v = pc[g.at(dx, dy, dz)];
        if constexpr (DIM == 2) {
            v = mul_rn(v, (b & 1u) ? f.fx[1] : f.fx[0]);
            v = mul_rn(v, (b & 2u) ? f.fy[1] : f.fy[0]);
        } else {
            v = mul_rn(v, f.x(b));
            v = mul_rn(v, f.y(b));
            v = mul_rn(v, f.z(b));
        }
        sum = q == 0 ? v : add_rn(sum, v);
    }
    return sum;
}

// One grid element.  GT = float: the training grids.  GT = uint8_t: the stored codec (save4fp, models.py:61-64), dequantised
// on the fly exactly like load4fp (models.py:68-71): (u - (2^(b-1) - 1)) / (2^b - 1), correctly rounded - the numerator is an
// exact small integer, the quotient comes from the reciprocal plus one residual correction (checked against true division for
// every byte and every bit depth in tests/test_oracle_golden.py) - so decoding from bytes is bit-identical to fp_load + the
// fp32 kernel.
template <class GT>
__device__ __forceinline__ float grid_elem(const FusedParams& p, const float* base, int64_t plane_off, uint32_t voff) {
    if constexpr (sizeof(GT) == 1) {
        const float n = (float)(reinterpret_cast<const uint8_t*>(base) + plane_off)[voff] - p.dq_sub;
        const float q = mul_rn(n, p.dq_rcp);
        return fmaf(fmaf(-q, p.dq_den, n), p.dq_rcp, q);
    } else {
        // wave-uniform plane base (scalar registers) + a 32-bit BYTE offset per lane: the `global_load_dword v, v_off, s[base]` form,
        // no 64-bit vector address arithmetic per load (the host checks that the byte offsets fit 32 bits)
        const char* b = reinterpret_cast<const char*>(base + plane_off);
        return *reinterpret_cast<const float*>(b + (uint32_t)(voff * 4u));
    }
}

// corner offsets of G0 slot group e (the lane-half h fixes dx = h), per layout
template <class L>
__device__ __forceinline__ void g0_corner(int e, int h, int& dx, int& dy, int& dz) {
    dx = h;
    if (L::DIM == 2) { dy = e; dz = 0; }
    else if (L::K0 == 8) { dy = (e >> 1) & 1; dz = e & 1; }   // corners 4h + e, bit1 = dy, bit0 = dz
    else { dy = e; dz = h ^ e; }                              // tetra corners 2h + e: dz = dx ^ dy
}

// element offsets of the (clamped) G0 / G1 cells of absolute sample coordinate q
template <class L>
__device__ __forceinline__ void cell_offsets(const FusedParams& p, const int (&q)[3], uint32_t& off0, uint32_t& off1) {
    constexpr int D = L::DIM;
    const int e = p.d.log2_step;
    const Axis ax = axis_coords(q[0], e), ay = axis_coords(q[1], e);
    Axis az;
    if (D == 3) az = axis_coords(q[2], e);
    else { az.i0 = az.i1 = 0; az.t1 = az.k1 = 0.f; }
    off0 = (uint32_t)p.g0.at(clampi(ax.i0, 0, p.g0.nx - 2), clampi(ay.i0, 0, p.g0.ny - 2), D == 3 ? clampi(az.i0, 0, p.g0.nz - 2) : 0);
    off1 = (uint32_t)p.g1.at(clampi(ax.i1, 0, p.g1.nx - 2), clampi(ay.i1, 0, p.g1.ny - 2), D == 3 ? clampi(az.i1, 0, p.g1.nz - 2) : 0);
}

// Fills the lane's slots (see Layout<> in nic_device.hpp) for the sample at ABSOLUTE coordinates q = origin + index.
// raw grid values of one cell: every sample of a lane's macro-tile lies in the same G0 cell and the same G1 cell, so they can be
// gathered once per macro-tile instead of once per round (where the register budget allows: NIC_HOIST_*)
template <class L>
struct CellRaw {
    static constexpr int NG0 = L::K0 / 2 * kC;
    static constexpr int K1 = L::DIM == 2 ? 4 : 8;
    float g0[NG0];              // [corner e of this half][channel]
    float g1[K1 * (kC / 2)];    // [corner q][channel of this half]
};
template <class L, class GT, bool HG0, bool HG1>
__device__ __forceinline__ void gather_cell(const FusedParams& p, uint32_t off0, uint32_t off1, int h, CellRaw<L>& raw) {
    constexpr int D = L::DIM;
    if (HG0) {
#pragma unroll
        for (int e = 0; e < CellRaw<L>::NG0 / kC; ++e) {
            int dx, dy, dz;
            g0_corner<L>(e, h, dx, dy, dz);
            const uint32_t voff = off0 + (uint32_t)p.g0.at(dx, dy, dz);
#pragma unroll
            for (int c = 0; c < kC; ++c) raw.g0[e * kC + c] = grid_elem<GT>(p, p.g0.p, (int64_t)c * p.g0.plane, voff);
        }
    }
    if (HG1) {
#pragma unroll
        for (int q = 0; q < CellRaw<L>::K1; ++q) {
            const int dx = D == 2 ? (q >> 1) : ((q >> 2) & 1), dy = D == 2 ? (q & 1) : ((q >> 1) & 1), dz = D == 2 ? 0 : (q & 1);
            const uint32_t voff = off1 + (uint32_t)p.g1.at(dx, dy, dz) + (uint32_t)(kC / 2 * h) * (uint32_t)p.g1.plane;
#pragma unroll
            for (int cc = 0; cc < kC / 2; ++cc) raw.g1[q * (kC / 2) + cc] = grid_elem<GT>(p, p.g1.p, (int64_t)cc * p.g1.plane, voff);
        }
    }
}

template <class L, class GT, bool HG0 = false, bool HG1 = false>
__device__ __forceinline__ void encode_slots(const FusedParams& p, const int (&q)[3], int h, float (&xs)[L::NSLOT], EncCtx& cx,
                                             const CellRaw<L>* raw = nullptr) {
    constexpr int D = L::DIM;
    const nic_path_desc& d = p.d;
    const int e = d.log2_step;
    Axis ax = axis_coords(q[0], e);
    Axis ay = axis_coords(q[1], e);
    Axis az;
    if (D == 3) az = axis_coords(q[2], e);
    else { az.i0 = az.i1 = 0; az.t1 = az.k1 = 0.f; }
    // memory safety: a corner index never leaves the grid, whatever the origins hold
    const int x0 = clampi(ax.i0, 0, p.g0.nx - 2), y0 = clampi(ay.i0, 0, p.g0.ny - 2), z0 = D == 3 ? clampi(az.i0, 0, p.g0.nz - 2) : 0;
    const int x1 = clampi(ax.i1, 0, p.g1.nx - 2), y1 = clampi(ay.i1, 0, p.g1.ny - 2), z1 = D == 3 ? clampi(az.i1, 0, p.g1.nz - 2) : 0;
    cx.off0 = (uint32_t)p.g0.at(x0, y0, z0);
    cx.off1 = (uint32_t)p.g1.at(x1, y1, z1);
    cx.kx = ax.k1; cx.ky = ay.k1; cx.kz = az.k1;
    constexpr int NG0 = L::K0 / 2 * kC;             // G0 slots per half
    constexpr int K1 = D == 2 ? 4 : 8;
    // Order: issue the G1 gathers, issue the G0 gathers, compute the slots that need no memory (PE, LOD, constants), THEN blend
    // G1 - the blend is the first consumer of a gather, everything before it runs under the loads' latency.
    // --- G1 corner values: channels (kC/2)*h + cc.  Plane base wave-uniform; the lane-half part is in voff
    float g1v[K1 * (kC / 2)];
    {
        uint32_t voff[K1];
#pragma unroll
        for (int q = 0; q < K1; ++q) {
            const int dx = D == 2 ? (q >> 1) : ((q >> 2) & 1), dy = D == 2 ? (q & 1) : ((q >> 1) & 1), dz = D == 2 ? 0 : (q & 1);
            voff[q] = cx.off1 + (uint32_t)p.g1.at(dx, dy, dz) + (uint32_t)(kC / 2 * h) * (uint32_t)p.g1.plane;
        }
#pragma unroll
        for (int cc = 0; cc < kC / 2; ++cc)
#pragma unroll
            for (int q = 0; q < K1; ++q)
                g1v[q * (kC / 2) + cc] = HG1 ? raw->g1[q * (kC / 2) + cc] : grid_elem<GT>(p, p.g1.p, (int64_t)cc * p.g1.plane, voff[q]);
    }
    // --- G0 raw corners.  Address = (uniform channel-plane base in SGPRs) + (one 32-bit lane offset per corner): the 12
    // channel loads of a corner share ONE offset register (96 address registers otherwise)
#pragma unroll
    for (int e = 0; e < NG0 / kC; ++e) {
        int dx, dy, dz;
        g0_corner<L>(e, h, dx, dy, dz);
        const uint32_t voff = cx.off0 + (uint32_t)p.g0.at(dx, dy, dz);
#pragma unroll
        for (int c = 0; c < kC; ++c)
            xs[e * kC + c] = HG0 ? raw->g0[e * kC + c] : grid_elem<GT>(p, p.g0.p, (int64_t)c * p.g0.plane, voff);   // plane base: wave-uniform
    }
    __builtin_amdgcn_sched_barrier(0);
    // --- slots without memory: PE rows, LOD, the constant one, zero padding.  The channel of a slot is affine
    // in h, so its kind is known at compile time per half; only PE row / axis are lane-half dependent.
    float pdiv[kP / 2];
#pragma unroll
    for (int i = 0; i < kP / 2; ++i) pdiv[i] = d.pe_div[i];
#pragma unroll
    for (int s = NG0 + kC / 2; s < L::NSLOT; ++s) {
        constexpr int PE0 = (L::K0 + 1) * kC;                 // first PE channel
        const int ch0 = L::slot_channel(s, 0), ch1 = L::slot_channel(s, 1);
        const bool pe0 = ch0 >= PE0 && ch0 < L::CIN - 1, pe1 = ch1 >= PE0 && ch1 < L::CIN - 1;
        float v0 = ch0 == kSlotOne ? 1.0f : (ch0 == L::CIN - 1 ? d.lod_value : 0.f);
        float v1 = ch1 == kSlotOne ? 1.0f : (ch1 == L::CIN - 1 ? d.lod_value : 0.f);
        if (pe0 || pe1) {
            const int pr = (h ? ch1 : ch0) - PE0;             // PE row 0 .. P*D-1 of this lane-half
            const int a = pr / kP, r = pr - a * kP;
            const float c = a == 0 ? ax.t1 : (a == 1 ? ay.t1 : az.t1);
            float v;
            if (L::PE == NIC_PE_TRIANGULAR) {
                v = tri_pe_row(c, r, kP);
            } else {
                const int k = r >> 1;
                const float dv = k == 0 ? pdiv[0] : (k == 1 ? pdiv[1] : pdiv[2]);
                float sv, cv;
                sincos_cw(mul_rn(c, dv), sv, cv);
                v = (r & 1) ? cv : sv;
            }
            if (pe0) v0 = v;
            if (pe1) v1 = v;
        }
        xs[s] = h ? v1 : v0;
    }
    __builtin_amdgcn_sched_barrier(0);
    // --- G1 blend with the reference's factor order
    const G1FactorsT<D> gf = g1_factors<D>(d.g1_weight_mode, cx.kx, cx.ky, cx.kz);
#pragma unroll
    for (int cc = 0; cc < kC / 2; ++cc) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < K1; ++q) {
            const uint32_t b = (gf.bits >> (3 * q)) & 7u;
            float v = g1v[q * (kC / 2) + cc];
            if constexpr (D == 2) {
                v = mul_rn(v, (b & 1u) ? gf.fx[1] : gf.fx[0]);
                v = mul_rn(v, (b & 2u) ? gf.fy[1] : gf.fy[0]);
            } else {
                v = mul_rn(v, gf.x(b));
                v = mul_rn(v, gf.y(b));
                v = mul_rn(v, gf.z(b));
            }
            sum = q == 0 ? v : add_rn(sum, v);
        }
        xs[NG0 + cc] = sum;
    }
}

// In-kernel noise (NIC_NOISE_KERNEL).  The values of a sample come in two streams, one per lane half: value v of stream h
// belongs to the v-th REAL slot (slot_channel >= 0) of that half, in slot order - the same v for both halves wherever both are
// real - and lives in generator block NB * h + (v >> 4), position v & 15.  A lane therefore runs ceil(real slots / 16) blocks
// (3 in 2D instead of the 4 of a by-channel numbering) and needs no selects.  oracle/nic_oracle.py::kernel_noise restates the
// (channel -> stream, v) map from the slot layouts.
template <class L>
struct NoiseMap {
    __host__ __device__ static constexpr int count(int h) {
        int n = 0;
        for (int s = 0; s < L::NSLOT; ++s) n += L::slot_channel(s, h) >= 0 ? 1 : 0;
        return n;
    }
    struct Table { int v[2][L::NSLOT]; };                                      // v of slot s in stream h
    __host__ __device__ static constexpr Table make() {
        Table t{};
        for (int h = 0; h < 2; ++h) {
            int n = 0;
            for (int s = 0; s < L::NSLOT; ++s) {
                t.v[h][s] = n;
                n += L::slot_channel(s, h) >= 0 ? 1 : 0;
            }
        }
        return t;
    }
    static constexpr int NV = count(0) > count(1) ? count(0) : count(1);
    static constexpr int NB = (NV + 15) / 16;                                 // blocks per stream
};

// adds the noise of the slot's reference channel (image_compression.py:250: every real channel)
template <class L>
__device__ __forceinline__ void add_noise(const NoiseSrc& ns, uint64_t sample_global, int64_t n_local, int h, float (&xs)[L::NSLOT]) {
    if (ns.mode == NIC_NOISE_NONE) return;
    if (ns.mode == NIC_NOISE_TENSOR) {
        const float* row = ns.tensor + n_local * L::CIN;
#pragma unroll
        for (int s = 0; s < L::NSLOT; ++s) {
            const int ch = L::slot_channel(s, h);
            if (ch >= 0) xs[s] += row[ch];
        }
        return;
    }
    if constexpr (L::DIM == 2) {
        // 2D: one block per G0 corner (noise_field, nic_device.hpp); lane half h holds corners 2h, 2h + 1.  Slot -> (corner of the
        // half, field) is the same map for both halves.
        const U4 b0 = noise_block(ns, sample_global, 2 * h), b1 = noise_block(ns, sample_global, 2 * h + 1);
#pragma unroll
        for (int s = 0; s < L::NSLOT; ++s) {
            int e, f;
            if (s < 24) { e = s / 12; f = s % 12; }
            else if (s < 30) { e = (s - 24) / 3; f = 12 + (s - 24) % 3; }
            else if (s >= 32 && s < 38) { e = (s - 32) / 3; f = 15 + (s - 32) % 3; }
            else if (s == 38) { e = 0; f = 18; }
            else continue;
            const float v = noise_field(ns, e ? b1 : b0, f);
            xs[s] += (s == 38) ? (h == 0 ? v : 0.f) : v;                          // slot 38 of half 1 is the bias carrier
        }
        return;
    }
    using M = NoiseMap<L>;
    constexpr typename M::Table tab = M::make();
    U4 blk[M::NB];
#pragma unroll
    for (int j = 0; j < M::NB; ++j) blk[j] = noise_block(ns, sample_global, M::NB * h + j);
#pragma unroll
    for (int s = 0; s < L::NSLOT; ++s) {
        const bool r0 = L::slot_channel(s, 0) >= 0, r1 = L::slot_channel(s, 1) >= 0;
        if (!r0 && !r1) continue;
        const int v0 = tab.v[0][s], v1 = tab.v[1][s];
        if (r0 && r1 && v0 == v1) {
            xs[s] += noise_from_block(ns, blk[v0 >> 4], v0 & 15);
        } else {                                                              // real in one half only, or different positions
            const float n0 = r0 ? noise_from_block(ns, blk[v0 >> 4], v0 & 15) : 0.f;
            const float n1 = r1 ? noise_from_block(ns, blk[v1 >> 4], v1 & 15) : 0.f;
            xs[s] += h ? n1 : n0;
        }
    }
}

// Gradients of the lane's grid slots.  Every sample a lane handles inside one macro-tile lies in the same G0 cell (and G1
// cell), so the slot gradients are summed in registers over the rounds (GridAcc) and scattered once per cell.
template <class L>
struct GridAcc {
    static constexpr int NG0 = L::K0 / 2 * kC;
    static constexpr int K1 = L::DIM == 2 ? 4 : 8;
    float g0[NG0];               // [corner e of this half][channel]
    float g1[K1 * (kC / 2)];     // [corner q][channel of this half]
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < NG0; ++i) g0[i] = 0.f;
#pragma unroll
        for (int i = 0; i < K1 * (kC / 2); ++i) g1[i] = 0.f;
    }
};

template <class L, int NT>
__device__ __forceinline__ void accumulate_grid_grads(const FusedParams& p, const EncCtx& cx, const f32x16 (&dxacc)[NT], GridAcc<L>& ga,
                                                      int carried = 0) {
    constexpr int D = L::DIM;
    constexpr int NG0 = GridAcc<L>::NG0, K1 = GridAcc<L>::K1;
#pragma unroll
    for (int s = 0; s < NG0; ++s) ga.g0[s] = s < carried ? dxacc[s >> 4][s & 15] : ga.g0[s] + dxacc[s >> 4][s & 15];   // carried: the tile already holds the sum
    const G1FactorsT<D> gf = g1_factors<D>(p.d.g1_weight_mode, cx.kx, cx.ky, cx.kz);
#pragma unroll
    for (int q = 0; q < K1; ++q) {
        const float w = g1_corner_factor<D>(gf, q);
#pragma unroll
        for (int cc = 0; cc < kC / 2; ++cc) {
            const int s = NG0 + cc;
            ga.g1[q * (kC / 2) + cc] = fmaf(dxacc[s >> 4][s & 15], w, ga.g1[q * (kC / 2) + cc]);
        }
    }
}

// The 2^D G0 cells inside one G1 cell sit in 2^D lanes of the wave and would hit the same G1 nodes in the same atomic
// instruction (same-address atomics of one instruction serialise in L2: measured, the G1 half of the flush cost 2.4x the G0
// half).  They are summed across lanes first - partner = the lane whose block coordinate differs in the last bit along one
// axis, when it is inside the wave's block and really has the same G1 cell (clamped cells, odd alignment) - and only the
// even-coordinate lane keeps the sum; the others are left with exact zeros, which the flush skips.
template <class L>
__device__ __forceinline__ void combine_g1_lanes(GridAcc<L>& ga, uint32_t off1, const int (&blk)[3], int lane, int lw, const int (&pk)[3] = {0, 0, 0},
                                                 const int (&pk_lc)[3] = {0, 0, 0}) {
    constexpr int D = L::DIM;
    static_assert(L::TX * L::TY == 32 && L::TZ == 1, "a wave block is 2^lw x 32 / 2^lw x 1 cell blocks");
    const bool packed = pk[0] > 0;                            // packed tiling: lane pl holds block (pl + 32 tile) of the crop's list
    const int T[3] = {packed ? pk[0] : 1 << lw, packed ? pk[1] : 32 >> lw, packed ? pk[2] : 1};   // lw = 4 for the regular tiles (TX x TY), smaller for edge tiles
    const int STR[3] = {1, packed ? pk[0] : 1 << lw, packed ? pk[0] * pk[1] : 32};
    const int pl = lane & 31;
    const int lc[3] = {packed ? pk_lc[0] : pl & (T[0] - 1), packed ? pk_lc[1] : pl >> lw, packed ? pk_lc[2] : 0};
#pragma unroll
    for (int a = 0; a < D; ++a) {
        const bool odd = blk[a] & 1;
        const int pc = lc[a] + (odd ? -1 : 1);
        const int ppl = pl + (odd ? -STR[a] : STR[a]);         // the partner must sit in the same wave block
        const bool inb = pc >= 0 && pc < T[a] && ppl >= 0 && ppl < 32;
        const int partner = inb ? lane + (odd ? -STR[a] : STR[a]) : lane;
        const bool pair = inb && (uint32_t)__shfl((int)off1, partner) == off1;
#pragma unroll
        for (int i = 0; i < GridAcc<L>::K1 * (kC / 2); ++i) {
            const float pv = __shfl(ga.g1[i], partner);
            ga.g1[i] = pair ? (odd ? 0.f : ga.g1[i] + pv) : ga.g1[i];
        }
    }
}

// Neighbour pre-add of the G0 sums before the flush (2D and method 3: a lane half is one dx): the dx = 1 corners of block n - 1 are the
// dx = 0 corners of block n whenever the two cell offsets differ by one element, the dy = 1 corners of a block are the dy = 0 corners of
// the block one grid row up - inside a 16 x 2 wave block that is the lane 16 further (packed tiling: bx further).  A lane hands such sums
// to the lane that addresses the same nodes and issues nothing for them: 17 x 3 instead of 32 x 4 node visits per channel and wave block
// in 2D.  The offset tests reject clamped cells, row wraps and blocks of other crops.  Method 4's tetrahedra share no corner along x or y.
#ifndef NIC_PREADD32
#define NIC_PREADD32 1
#endif
template <class L>
__device__ __forceinline__ void preadd_g0_lanes(GridAcc<L>& ga, uint32_t off0, int lane, int str_y, uint32_t row) {
    if (!NIC_PREADD32 || (L::DIM == 3 && L::K0 == 4)) return;
    constexpr int NG0 = GridAcc<L>::NG0, E = NG0 / kC;
    const int pl = lane & 31, h = lane >> 5;
    {   // x: lane (pl, h = 1) -> lane (pl + 1, h = 0), same corner index e
        const bool recv = h == 0;
        const bool inb = recv ? pl >= 1 : pl <= 30;
        const int partner = inb ? (recv ? lane + 31 : lane - 31) : lane;
        const uint32_t poff = (uint32_t)__shfl((int)off0, partner);
        const bool pair = inb && (recv ? off0 == poff + 1u : poff == off0 + 1u);
#pragma unroll
        for (int i = 0; i < NG0; ++i) {
            const float pv = __shfl(ga.g0[i], partner);
            ga.g0[i] = pair ? (recv ? ga.g0[i] + pv : 0.f) : ga.g0[i];
        }
    }
    {   // y: corner e with dy = 1 of lane pl -> corner e - (dy bit) of lane pl + str_y (same half)
        const bool has_up = pl + str_y < 32, has_dn = pl - str_y >= 0;
        const uint32_t off_up = (uint32_t)__shfl((int)off0, has_up ? lane + str_y : lane);
        const uint32_t off_dn = (uint32_t)__shfl((int)off0, has_dn ? lane - str_y : lane);
        const bool send = has_up && off_up == off0 + row, recv = has_dn && off0 == off_dn + row;
        constexpr int DYB = L::DIM == 2 ? 1 : 2;              // the dy bit of the corner index e
        float pv[NG0 / 2];
#pragma unroll
        for (int e = 0, k = 0; e < E; ++e) {
            if (!(e & DYB)) continue;
#pragma unroll
            for (int c = 0; c < kC; ++c, ++k) pv[k] = __shfl(ga.g0[e * kC + c], has_dn ? lane - str_y : lane);
        }
#pragma unroll
        for (int e = 0, k = 0; e < E; ++e) {
            if (!(e & DYB)) continue;
#pragma unroll
            for (int c = 0; c < kC; ++c, ++k) {
                ga.g0[(e ^ DYB) * kC + c] += recv ? pv[k] : 0.f;
                ga.g0[e * kC + c] = send ? 0.f : ga.g0[e * kC + c];
            }
        }
    }
}

// one fp32 atomic per (corner, channel) of the lane's cell; exact zeros (cells outside the crop, zero weights) are skipped
template <class L>
__device__ __forceinline__ void flush_grid_grads(const FusedParams& p, uint32_t off0, uint32_t off1, int h, const GridAcc<L>& ga) {
    constexpr int D = L::DIM;
    constexpr int NG0 = GridAcc<L>::NG0, K1 = GridAcc<L>::K1;
#pragma unroll
    for (int e = 0; e < NG0 / kC; ++e) {
        int dx, dy, dz;
        g0_corner<L>(e, h, dx, dy, dz);
        const uint32_t voff = off0 + (uint32_t)p.g0.at(dx, dy, dz);
#pragma unroll
        for (int c = 0; c < kC; ++c) {
            const float v = ga.g0[e * kC + c];
            float* plane = p.g0_grad + (int64_t)c * p.g0.plane;               // wave-uniform
            if (v != 0.f) atomicAdd(plane + voff, v);
        }
    }
#pragma unroll
    for (int q = 0; q < K1; ++q) {
        const int dx = D == 2 ? (q >> 1) : ((q >> 2) & 1), dy = D == 2 ? (q & 1) : ((q >> 1) & 1), dz = D == 2 ? 0 : (q & 1);
        const uint32_t voff = off1 + (uint32_t)p.g1.at(dx, dy, dz) + (uint32_t)(kC / 2 * h) * (uint32_t)p.g1.plane;
#pragma unroll
        for (int cc = 0; cc < kC / 2; ++cc) {
            const float v = ga.g1[q * (kC / 2) + cc];
            float* plane = p.g1_grad + (int64_t)cc * p.g1.plane;
            if (v != 0.f) atomicAdd(plane + voff, v);
        }
    }
}

// =====================================================================================================
template <class L, int SRC, int MODE, class GT = float, int PREC = PREC_F32>
__global__ void __launch_bounds__(256, MODE == MODE_INFER ? 2 : 1) fused_kernel(FusedParams p) {
    using S = Lds<L>;
    constexpr bool TRAIN = MODE != MODE_INFER;
    constexpr bool SPLIT = PREC == PREC_SPLIT;
    constexpr bool CHAIN = PREC != PREC_F32;                // bf16 weight images + split chained products
    constexpr bool L3MM = SPLIT && MODE != MODE_INFER;      // layer 3 on the matrix pipe: pays where the a2 fragments exist anyway (training)
    constexpr int NGT = SRC == SRC_ENCODE ? (L::NGRID + 15) / 16 : Lds<L>::KT;     // dX row tiles: the grid slots / every slot
    static_assert(!CHAIN || (L::NSLOT % 8 == 0 && SRC == SRC_ENCODE), "split-bf16: kernels that encode from the grids, k-steps of 8 slots");
    static_assert(PREC != PREC_CHAIN || TRAIN, "PREC_CHAIN is a training mode");
    constexpr int KT = S::KT, LD1 = S::LD1, LD2 = S::LD2, LDT = S::LDT;
    static_assert(!SPLIT || !TRAIN || (S::TOTAL_SPLIT * 4 <= 163840 && S::PART16), "split-bf16 training layout is built for the 2D slot layouts");
    __shared__ __attribute__((aligned(16))) float smem[TRAIN ? (SPLIT ? S::TOTAL_SPLIT : S::TOTAL_TRAIN) : (SPLIT ? S::TOTAL_INFER_SPLIT : S::TOTAL_INFER)];
    lds_f* const sm = (lds_f*)smem;
    lds_f* const W1s = sm + S::OFF_W1;
    lds_f* const W2s = sm + S::OFF_W2;
    lds_f* const W3s = sm + S::OFF_W3;
    lds_f* const B2s = sm + S::OFF_B2;
    lds_f* const B3s = sm + S::OFF_B3;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;

    // ---------------- prologue: decoder weights -> LDS (first layer permuted into slot order, bias in the
    // column of the constant-one slot), scratch zeroed
    // SPLIT: the same two regions hold bf16 hi / lo images ([64][LD] elements each, natural row / column order) instead
    lds_bf* const W1b = (lds_bf*)W1s;
    lds_bf* const W2b = (lds_bf*)W2s;
    stage_all<kH * LD1, 256>(tid,
        [&](int idx) {
            const int o = idx / LD1, rho = idx - o * LD1;
            float v = 0.f;
            if (rho < S::KPAD) {
                const int ch = channel_of_rho<L>(rho);
                if (ch >= 0) v = p.W[0][o * L::CIN + ch];
                else if (ch == kSlotOne) v = p.b[0][o];
            }
            return v;
        },
        [&](int idx, float v) {
            if (CHAIN) {
                const __bf16 hi = (__bf16)v;
                W1b[idx] = hi;
                W1b[kH * LD1 + idx] = (__bf16)(v - (float)hi);
            } else {
                W1s[idx] = v;
            }
        });
    stage_all<kH * LD2, 256>(tid,
        [&](int idx) {
            const int o = idx / LD2, k = idx - o * LD2;
            return k < kH ? p.W[1][o * kH + k] : 0.f;
        },
        [&](int idx, float v) {
            if (CHAIN) {
                const __bf16 hi = (__bf16)v;
                W2b[idx] = hi;
                W2b[kH * LD2 + idx] = (__bf16)(v - (float)hi);
            } else {
                W2s[idx] = v;
            }
        });
    lds_bf* const W3b = (lds_bf*)(sm + S::OFF_SCR);            // SPLIT only
    stage_all<16 * LD2, 256>(tid,
        [&](int idx) {
            const int c = idx / LD2, k = idx - c * LD2;
            return (c < 3 && k < kH) ? p.W[2][c * kH + k] : 0.f;
        },
        [&](int idx, float v) {
            if (idx < 4 * LD2) W3s[idx] = v;
            if (SPLIT) {
                const __bf16 hi = (__bf16)v;
                W3b[idx] = hi;
                W3b[16 * LD2 + idx] = (__bf16)(v - (float)hi);
            }
        });
    if (tid < kH) B2s[tid] = p.b[1][tid];
    if (tid < 16) B3s[tid] = tid < 3 ? p.b[2][tid] : 0.f;
    if (TRAIN)
        for (int idx = tid + (SPLIT ? S::OFF_IMG : S::OFF_SCR); idx < (SPLIT ? S::TOTAL_SPLIT : S::TOTAL_TRAIN); idx += 256) smem[idx] = 0.f;
    __syncthreads();
    lds_f* const SCR0 = sm + S::OFF_SCR;                        // wave 0's scratch; wave w's is SCR_PER_WAVE * w further

    lds_f* const SA = sm + S::OFF_SCR + (TRAIN ? wave * S::SCR_PER_WAVE : 0);
    lds_f* const SB = SA + 64 * LDT;

    // ---------------- launch-lifetime accumulators (training): the dW tiles this wave OWNS
    constexpr int NCH = S::KTF / 2;                             // cross-wave dW1 chunks (2 col tiles each)
    f32x16 accW2o = f32x16(0.f);                                // dW2 tile (to = wave >> 1, tk = wave & 1)
    f32x16 accW1o[NCH > 0 ? NCH : 1];                           // dW1 tiles (to = wave & 1, tk = 2c + (wave >> 1))
    f32x16 accW1p[2];                                           // partial dW1 tiles (to = 0, 1; tk = KT - 1), own samples
    f32x4 accW1q[4];                                            // PART16: the same as 4 row tiles of 16 x 16
#pragma unroll
    for (int i = 0; i < 4; ++i) accW1q[i] = f32x4(0.f);
#pragma unroll
    for (int c = 0; c < (NCH > 0 ? NCH : 1); ++c) accW1o[c] = f32x16(0.f);
    accW1p[0] = accW1p[1] = f32x16(0.f);
    float accW3[3] = {0.f, 0.f, 0.f};     // dW3[c][k = lane]
    float accB2 = 0.f;                    // db2[o = lane]
    f32x4 accW3q = f32x4(0.f), accB2q = f32x4(0.f);   // SPLIT: the same from 4x4x4 MFMAs (register c of lane k / every register of lane o)
    float accB3[3] = {0.f, 0.f, 0.f}, accLoss = 0.f;
    const int to2 = wave >> 1, tk2 = wave & 1;                  // ownership
    const int to1 = wave & 1, tk1 = wave >> 1;

    // ---------------- XCD-aware persistent walk, in workgroup-synchronous rounds of 4 tiles (one per wave)
    const int xcd = blockIdx.x & 7, nb8 = gridDim.x >> 3;
    // work unit = (macro-tile, group of rounds); units of one macro-tile are consecutive, so they land on the waves of one workgroup
    const int lstride = nb8 * 4;

#ifdef NIC_STAMPS
    unsigned long long stamp_sum[NIC_NPH];
#pragma unroll
    for (int i = 0; i < NIC_NPH; ++i) stamp_sum[i] = 0;
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    // Staggered flushes (training): every wave of the chip does the same work per round, so without this all 1 024 waves reach the
    // grid-gradient flush of their macro-tile in the same few thousand cycles - and a no-return atomic instruction that costs its
    // wave ~75 ticks when a half or a quarter of the chip flushes costs ~420 when everybody does (ab/micro/atomic_burst.hip).
    // Workgroups therefore run in four phases: phase f starts its first unit at round f * rounds / 4 and comes back for the rounds it
    // skipped after its last unit (one extra, partial unit: same work for every wave, no idle time), which keeps the phases
    // f * rounds / 4 apart for the whole launch.
    // Two segments (balance_units): macro-tiles [0, seg_split) dealt out in 2^rg0_log2 groups of rounds - as many of them as fill every
    // wave of the launch a whole number of times - and the remainder in 2^rg_log2 (usually more) groups, so that the last, partly filled
    // step of a small launch costs a fraction of a unit.  seg_split = 0: one segment, as before.
  for (int seg = 0; seg < 2; ++seg) {
    const int64_t seg_tile0 = seg ? p.seg_split : 0;
    const int64_t seg_tiles = seg ? p.n_tiles - p.seg_split : p.seg_split;
    const int rg = seg ? p.rg_log2 : p.rg0_log2;
    if (seg_tiles <= 0) continue;                                        // launch-uniform
    const int64_t n_units = seg_tiles << rg;
    const int64_t chunk = (((n_units + 7) >> 3) + 3) & ~(int64_t)3;      // a multiple of 4: the groups of a macro-tile stay in one workgroup
    const int64_t t_begin = xcd * chunk;
    const int64_t t_end = t_begin + chunk < n_units ? t_begin + chunk : n_units;
    const int64_t base0 = t_begin + (int64_t)(blockIdx.x >> 3) * 4;
    const int64_t n_my = base0 < t_end ? (t_end - base0 + lstride - 1) / lstride : 0;
    const int rounds_unit = (SRC == SRC_ENCODE ? p.niter * p.passes : 1) >> rg;
    const int nph = rounds_unit >= NIC_PHASES ? NIC_PHASES : (rounds_unit >= 2 ? 2 : 1);      // phases (a power of two)
    const int shift = (NIC_STAGGER && TRAIN && SRC == SRC_ENCODE && ((NIC_STAGGER_RG && L::DIM == 2) || rg == 0)) ? (int)((blockIdx.x >> 3) & (nph - 1)) * (rounds_unit / nph) : 0;   // 3D, rounds dealt out in groups: the extra partial unit per wave costs more than the phases gain (the reference's sweep shape 380 -> 361 us, 323 -> 294 us)
    for (int64_t kk = 0; kk < n_my + (shift ? 1 : 0); ++kk) {
        const int64_t base = base0 + (kk < n_my ? kk : 0) * lstride;
        // a wave without a tile in the last round still takes part (barriers, owned dW tiles): it recomputes the range's
        // last tile with every lane masked, which contributes exact zeros everywhere
        const bool tile_ok = base + wave < t_end;
        const int64_t unit = tile_ok ? base + wave : t_end - 1;
        const int64_t tile = seg_tile0 + (unit >> rg);
        int it_len = rounds_unit, it_begin = (int)(unit & ((1 << rg) - 1)) * rounds_unit;
        if (shift) {
            if (kk == 0) { it_begin += shift; it_len -= shift; }          // the first unit without its first `shift` rounds ..
            else if (kk == n_my) it_len = shift;                         // .. which are done at the very end
        }
        // ---------- macro-tile -> this lane's cell block (absolute block coordinates) and crop
        int crop = 0, lw = 4;                                       // lw: log2 of the lane block's x extent (TX = 16)
        static_assert(L::TX == 16, "lw");
        int org[3] = {0, 0, 0}, blk[3] = {0, 0, 0}, pk_lc[3] = {0, 0, 0};
        if (SRC == SRC_ENCODE) {
            crop = (int)(tile / p.tiles_per_crop);
            int tt = (int)(tile - (int64_t)crop * p.tiles_per_crop);
            // regular tile: TX x TY blocks, x fastest (x is the grids' contiguous axis); edge tile: 2^lw x 32 / 2^lw blocks
            int boff[3];                                                          // the tile's first block inside the crop
            if (p.pk_nc > 0) {
                // packed tiling: block number -> (x, y, z); numbers past the crop's list land outside the crop (masked like any such block)
                const int ci = tt * 32 + pl;
                const int cy = ci / p.pk_bx;
                pk_lc[0] = ci - cy * p.pk_bx;
                pk_lc[2] = L::DIM == 3 ? cy / p.pk_by : 0;
                pk_lc[1] = L::DIM == 3 ? cy - pk_lc[2] * p.pk_by : cy;
                boff[0] = boff[1] = boff[2] = 0;
            } else if (p.edge_lw < 0 || tt < p.tiles_main) {
                int tc[3];
                if (L::DIM == 2) {
                    tc[1] = tt % p.tiles_y; tc[0] = tt / p.tiles_y; tc[2] = 0;
                } else {
                    tc[2] = tt % p.tiles_z; tt /= p.tiles_z;
                    tc[1] = tt % p.tiles_y; tc[0] = tt / p.tiles_y;
                }
                boff[0] = tc[0] * L::TX; boff[1] = tc[1] * L::TY; boff[2] = tc[2] * L::TZ;
            } else {
                int e = tt - (int)p.tiles_main;
                lw = p.edge_lw;
                boff[2] = L::DIM == 3 ? e % p.tiles_z : 0;
                if (L::DIM == 3) e /= p.tiles_z;
                boff[1] = e * (32 >> lw);
                boff[0] = p.full_x * L::TX;
            }
            const int lc[3] = {p.pk_nc > 0 ? pk_lc[0] : pl & ((1 << lw) - 1), p.pk_nc > 0 ? pk_lc[1] : pl >> lw, p.pk_nc > 0 ? pk_lc[2] : 0};
#pragma unroll
            for (int a = 0; a < L::DIM; ++a) {
                org[a] = origin_of(p, crop * L::DIM + a);
                blk[a] = (org[a] >> p.lm) + boff[a] + lc[a];
            }
        }
        GridAcc<L> gacc;
        uint32_t blk_off0 = 0, blk_off1 = 0;
        // which raw grid values are gathered once per macro-tile: inference has the registers for both grids
        constexpr int HOIST_INFER = NIC_HOIST_INFER >= 0 ? NIC_HOIST_INFER : (L::DIM == 2 ? 3 : (SPLIT ? 1 : (L::K0 == 4 ? 3 : 1)));   // measured per layout
        // training: the 2D kernels have the registers for a cell's raw values since the build keeps MFMA results out of the
        // accumulator file where vector code consumes them: G0 (24 values) -1.8 % split, -1 % fp32; G0 + G1 (48) another -1.8 % in the
        // split kernel although 9 values spill (the fp32 kernel with both: the compiler gives up); 3D: none
        constexpr int HOIST_TRAIN = NIC_HOIST_TRAIN >= 0 ? NIC_HOIST_TRAIN : (L::DIM == 2 ? (SPLIT ? NIC_HOIST_SPLIT : 3) : NIC_HOIST_3D);   // 2D fp32: both grids since the derivatives are pinned (no spill; 4.404 -> 4.380 ms)
        constexpr bool HG0 = SRC == SRC_ENCODE && ((TRAIN ? HOIST_TRAIN : HOIST_INFER) & 1) != 0;
        constexpr bool HG1 = SRC == SRC_ENCODE && ((TRAIN ? HOIST_TRAIN : HOIST_INFER) & 2) != 0;
        CellRaw<L> raw;
        if (SRC == SRC_ENCODE && (TRAIN || HG0 || HG1)) {
            const int qb[3] = {blk[0] << p.lm, blk[1] << p.lm, blk[2] << p.lm};
            cell_offsets<L>(p, qb, blk_off0, blk_off1);
            if (TRAIN) gacc.clear();
            gather_cell<L, GT, HG0, HG1>(p, blk_off0, blk_off1, h, raw);
        }

        STAMP(12);   // macro-tile setup
      for (int it = it_begin; it < it_begin + it_len; ++it) {                   // one sample of the cell block per round
        // ---------- per-lane LDS bases (every access below is base[compile-time constant])
        // ---------- per-lane LDS bases (every access below is base[compile-time constant])
        lds_cf* const w1_row = opaque(W1s + pl * LD1 + 4 * h);      // A rows of layer 1 (b128 along k)
        lds_cf* const w2_row = opaque(W2s + pl * LD2 + 4 * h);
        lds_cf* const w3_row = opaque(W3s + 4 * h);
        lds_cf* const b2_row = opaque(B2s + 4 * h);
        lds_cf* const w3_col = opaque(W3s + h * LD2 + pl);          // A columns (b32 along the out index)
        lds_cf* const w2_col = opaque(W2s + 4 * h * LD2 + pl);
        lds_cf* const w1_col = opaque(W1s + 4 * h * LD1 + pl);
        lds_f* const sa_st = opaque(SA + 4 * h * LDT + pl);              // transposed stores: row ROWC(r)+4h, col pl
        lds_f* const sb_st = opaque(SB + 4 * h * LDT + pl);
        lds_cf* const sa_rd = opaque(SA + pl * LDT + 16 * h);       // operand reads: row pl, samples 16h..
        lds_cf* const sb_rd = opaque(SB + pl * LDT + 16 * h);
        lds_cf* const sa_lane = opaque(SA + lane * LDT);            // row passes: lane walks row `lane`
        lds_cf* const sb_lane = opaque(SB + lane * LDT);
        lds_cf* const sa_bc = opaque(SA);                           // broadcast reads of rows 0..2
        // SPLIT: row-wise fragments [row pl][k + 4h]; transposed fragments: lane 4q + p' of its 16-lane group points at
        // [row 4h + q][col 16 cg + 4 p'] (cg = which half of the 32 output rows the group covers)
        lds_cbf* w1b_row = nullptr, *w2b_row = nullptr, *w1b_tr = nullptr, *w2b_tr = nullptr;
        // SPLIT images (see Lds<>): store bases [sample pl][4h ..] of the wave's own images; transposed-read bases of wave 0's
        // images for the 32x32x16 fragments (lane 4q + p' of a 16-lane group: [row 4h + q][col 16 cg + 4 p']), of the wave's own
        // images for the 16x16x32 fragments ([row 8 (lane >> 4) + q][col 4 p']) and the 4x4x4 B operand ([row q][col 16 (lane >> 4) + 4 p'])
        lds_bf* dz_st = nullptr, *a1_st = nullptr, *x_st = nullptr, *d3_st = nullptr;
        lds_cbf* dz_tr = nullptr, *a1_tr = nullptr, *x_tr = nullptr, *dz_p16 = nullptr, *x_p16 = nullptr, *dz_b44 = nullptr, *d3_a44 = nullptr;
        if constexpr (SPLIT && TRAIN) {
            lds_bf* const img0 = (lds_bf*)(sm + S::OFF_IMG);
            lds_bf* const imgw = img0 + wave * S::SPW;
            const int q4 = (lane & 15) >> 2, p4 = lane & 3, cg = (lane >> 4) & 1, g16 = lane >> 4;
            dz_st = (lds_bf*)opaque((lds_cbf*)(imgw + pl * S::LDZB + 4 * h));
            a1_st = dz_st + 2 * S::BZ;
            x_st = (lds_bf*)opaque((lds_cbf*)(imgw + 4 * S::BZ + pl * S::LDXB + 4 * h));
            // Which 4 samples a transposed read combines is free (A and B operands use the same rule): rows r, r + 8, r + 16,
            // r + 24 sit 16 bank pairs apart under the 34 / 42-dword strides, so the reads are conflict-free (4 consecutive rows
            // were 2-way).  The 8 base rows r of a wave's 32 samples: 32x32x16: r = 4 (k-step) + 2 (lane half) + (read 0 / 1);
            // 16x16x32: r = 2 (lane >> 4) + read; 4x4x4: r = MFMA index (the dZ3 image is stored in that order: [c][4 r + q]).
            d3_st = (lds_bf*)opaque((lds_cbf*)(imgw + 4 * S::BZ + 2 * S::BX + 4 * (pl & 7) + (pl >> 3)));
            dz_tr = opaque((lds_cbf*)(img0 + (8 * q4 + 2 * h) * S::LDZB + 16 * cg + 4 * p4));
            a1_tr = dz_tr + 2 * S::BZ;
            x_tr = opaque((lds_cbf*)(img0 + 4 * S::BZ + (8 * q4 + 2 * h) * S::LDXB + 16 * cg + 4 * p4));
            dz_p16 = opaque((lds_cbf*)(imgw + (8 * q4 + 2 * g16) * S::LDZB + 4 * p4));
            x_p16 = opaque((lds_cbf*)(imgw + 4 * S::BZ + (8 * q4 + 2 * g16) * S::LDXB + 32 * (KT - 1) + 4 * p4));
            dz_b44 = opaque((lds_cbf*)(imgw + 8 * q4 * S::LDZB + 16 * g16 + 4 * p4));
            d3_a44 = opaque((lds_cbf*)(imgw + 4 * S::BZ + 2 * S::BX + (lane & 3) * 32));
        }
        lds_cbf* w3b_row = nullptr, *w3b_tr = nullptr;
        if constexpr (SPLIT) {
            w3b_row = opaque((lds_cbf*)W3b + (pl & 15) * LD2 + 4 * h);            // rows 16 .. 31 of the A operand re-read rows 0 .. 15
            w3b_tr = opaque((lds_cbf*)W3b + (4 * h + ((lane & 15) >> 2)) * LD2 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
        }
        if constexpr (CHAIN) {
            w1b_row = opaque((lds_cbf*)W1b + pl * LD1 + 4 * h);
            w2b_row = opaque((lds_cbf*)W2b + pl * LD2 + 4 * h);
            w1b_tr = opaque((lds_cbf*)W1b + (4 * h + ((lane & 15) >> 2)) * LD1 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
            w2b_tr = opaque((lds_cbf*)W2b + (4 * h + ((lane & 15) >> 2)) * LD2 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
        }

        // ---------- which sample does this lane own in this round
        bool valid;
        int64_t n;          // sample index inside this launch
        int q[3] = {0, 0, 0};
        if (SRC == SRC_ENCODE) {
            const int m1 = (1 << p.lm) - 1;
            const int pass = it >> (p.lm * L::DIM), its = it & (p.niter - 1);      // niter = 2^(lm * dim)
            int j[3];
            if (L::DIM == 2) { j[1] = its & m1; j[0] = its >> p.lm; j[2] = 0; }
            else { j[2] = its & m1; j[1] = (its >> p.lm) & m1; j[0] = its >> (2 * p.lm); }
            const int ez = L::DIM == 3 ? p.d.extent[2] : 1;
            const int ext[3] = {p.d.extent[0], p.d.extent[1], ez};
            int idx[3] = {0, 0, 0};
            valid = tile_ok;
#pragma unroll
            for (int a = 0; a < L::DIM; ++a) {
                const int i = (blk[a] << p.lm) + j[a] - org[a];          // index inside the crop
                valid = valid && i >= 0 && i < ext[a];
                idx[a] = i < 0 ? 0 : (i >= ext[a] ? ext[a] - 1 : i);     // masked lanes stay inside the crop
                q[a] = org[a] + idx[a];
            }
            n = ((int64_t)crop * p.passes + pass) * p.n_per_crop + ((int64_t)idx[0] * ext[1] + idx[1]) * ez + idx[2];
        } else {
            n = tile * 32 + pl;
            valid = tile_ok && n < p.n_total;
            n = valid ? n : p.n_total - 1;
        }

        // ---------- the sample's target (or incoming dY): fetched now, used after the forward pass.  n is a valid sample index
        // for every lane (masked lanes are clamped), so the loads are unconditional and all in flight together
        float tgt[3] = {0.f, 0.f, 0.f};
        if (MODE == MODE_TRAIN_IMG) {
            int64_t off = (int64_t)q[0] * p.timg_s[0] + (int64_t)q[1] * p.timg_s[1];
            if (L::DIM == 3) off += (int64_t)q[2] * p.timg_s[2];
            uint32_t rgbx = 0u;
            if (p.timg_u8 == 2) rgbx = reinterpret_cast<const uint32_t*>(p.timg)[off];      // interleaved: one load for the three targets
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (p.timg_u8) {
                    const float u = p.timg_u8 == 2 ? (float)((rgbx >> (8 * c)) & 255u) : (float)(reinterpret_cast<const uint8_t*>(p.timg) + c * p.timg_cs)[off];
                    const float t0 = mul_rn(u, p.timg_rcp);                               // correctly rounded u / den, as in grid_elem
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
        // ---------- input slots
        float xs[L::NSLOT];
        EncCtx cx;
        if (SRC == SRC_ENCODE) {
            encode_slots<L, GT, HG0, HG1>(p, q, h, xs, cx, &raw);
            add_noise<L>(p.noise, (uint64_t)(p.d.sample_base + n), n, h, xs);
        } else {
            const float* row = p.x + n * L::CIN;
#pragma unroll
            for (int s = 0; s < L::NSLOT; ++s) {
                const int ch = L::slot_channel(s, h);
                xs[s] = ch >= 0 ? row[ch] : (ch == kSlotOne ? 1.0f : 0.f);
            }
        }

        STAMP(0);    // coordinates, gathers, blend, PE, noise
        if (TRAIN && S::XIMG && !SPLIT) {
            // permanent transposed image of X for dW1 (read by all four waves after the round's barriers)
            lds_f* const xi_st = opaque(sm + S::OFF_XIMG + wave * S::XIMG_PER_WAVE + 4 * h * LDT + pl);
#pragma unroll
            for (int s = 0; s < L::NSLOT; ++s) xi_st[(32 * (s >> 4) + ROWC(s & 15)) * LDT] = xs[s];
        }
        // ---------- layer 1: Z1[o][s] = sum_rho W1p[o][rho] X[rho][s]   (bias rides on the constant-one slot)
        f32x16 a1[2], d1[2];
        {
            f32x16 z[2] = {f32x16(0.f), f32x16(0.f)};
            if constexpr (CHAIN) {
                // k-step b = slots 8b .. 8b+7 of both lane halves = internal rows 16b + 8(j>>2) + 4g + (j&3)
                auto loadw = [&](int b, Frag2 (&af)[2]) {
#pragma unroll
                    for (int to = 0; to < 2; ++to) {
                        af[to].hi = frag_row(&w1b_row[32 * to * LD1 + 16 * b]);
                        af[to].lo = frag_row(&w1b_row[kH * LD1 + 32 * to * LD1 + 16 * b]);
                    }
                };
                Frag2 afq[2][2];                                           // weight fragments of step b + 1 in flight during step b
                loadw(0, afq[0]);
#pragma unroll
                for (int b = 0; b < L::NSLOT / 8; ++b) {
                    if (b + 1 < L::NSLOT / 8) loadw(b + 1, afq[(b + 1) & 1]);
                    const float xv[8] = {xs[8 * b], xs[8 * b + 1], xs[8 * b + 2], xs[8 * b + 3], xs[8 * b + 4], xs[8 * b + 5], xs[8 * b + 6], xs[8 * b + 7]};
                    const Frag2 bf = split8(xv);
                    if (TRAIN && SPLIT) {                                // the X image of the weight-gradient product: columns 16b + 4h.. and + 8
                        store_frag(&x_st[16 * b], bf.hi);
                        store_frag(&x_st[S::BX + 16 * b], bf.lo);
                    }
#pragma unroll
                    for (int to = 0; to < 2; ++to) z[to] = mfma_split(afq[b & 1][to], bf, z[to]);
                    NIC_SPLIT_SB;
                }
            } else {
            // every MFMA phase fetches its LDS operands one step (4 MFMAs) ahead, so the LDS latency of step i + 1 runs
            // under the MFMAs of step i instead of in front of them
            constexpr int NST = 2 * (L::NSLOT / 4);                     // step = (4 slots, one row tile)
            auto w1_at = [&](int st) {
                const int sig = 4 * (st >> 1);
                return ld4(&w1_row[32 * (st & 1) * LD1 + 32 * (sig >> 4) + 8 * ((sig & 15) >> 2)]);
            };
            f32x4 aq[2];
            aq[0] = w1_at(0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) aq[(st + 1) & 1] = w1_at(st + 1);
                const int sig = 4 * (st >> 1), to = st & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // a k-step whose slot is zero padding in BOTH lane halves contributes nothing: no MFMA
                    if (L::slot_channel(sig + j, 0) == kSlotZero && L::slot_channel(sig + j, 1) == kSlotZero) continue;
                    z[to] = mfma32(aq[st & 1][j], xs[sig + j], z[to]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
#pragma unroll
            for (int to = 0; to < 2; ++to)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4_t zq = {z[to][4 * r4], z[to][4 * r4 + 1], z[to][4 * r4 + 2], z[to][4 * r4 + 3]};
                    f32x4_t aq4, dq4;
                    gelu_and_grad4(zq, aq4, dq4);
                    if (TRAIN && NIC_FK_PIN) asm volatile("" : "+v"(dq4));      // keep the derivative, not the three values it is made of (fused_q16.hpp::pin)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a1[to][4 * r4 + j] = aq4[j];
                        d1[to][4 * r4 + j] = TRAIN ? dq4[j] : 0.f;
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------- layer 2
        f32x16 a2[2], d2[2];
        float part3[3] = {0.f, 0.f, 0.f};
        {
            f32x16 z[2];
#pragma unroll
            for (int to = 0; to < 2; ++to)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 bb = ld4(&b2_row[32 * to + 8 * r4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[to][4 * r4 + j] = bb[j];
                }
            if constexpr (CHAIN) {
                auto loadw = [&](int ks, Frag2 (&af)[2]) {
#pragma unroll
                    for (int to = 0; to < 2; ++to) {
                        af[to].hi = frag_row(&w2b_row[32 * to * LD2 + 16 * ks]);
                        af[to].lo = frag_row(&w2b_row[kH * LD2 + 32 * to * LD2 + 16 * ks]);
                    }
                };
                Frag2 afq[2][2];
                loadw(0, afq[0]);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                         // k-step (t, s): hidden rows 32t + 16s + ..
                    if (ks + 1 < 4) loadw(ks + 1, afq[(ks + 1) & 1]);
                    const Frag2 bf = split_acc(a1[ks >> 1], ks & 1);
                    if (TRAIN && SPLIT) {
                        store_frag(&a1_st[16 * ks], bf.hi);
                        store_frag(&a1_st[S::BZ + 16 * ks], bf.lo);
                    }
#pragma unroll
                    for (int to = 0; to < 2; ++to) z[to] = mfma_split(afq[ks & 1][to], bf, z[to]);
                    NIC_SPLIT_SB;
                }
            } else {
            auto w2_at = [&](int st) { return ld4(&w2_row[32 * (st & 1) * LD2 + 32 * (st >> 3) + 8 * ((st >> 1) & 3)]); };
            f32x4 aq[2];
            aq[0] = w2_at(0);
#pragma unroll
            for (int st = 0; st < 16; ++st) {                            // step = (t, r4, to)
                if (st + 1 < 16) aq[(st + 1) & 1] = w2_at(st + 1);
                const int t = st >> 3, r4 = (st >> 1) & 3, to = st & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) z[to] = mfma32(aq[st & 1][j], a1[t][4 * r4 + j], z[to]);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            // layer 3 rides along: only 3 outputs - a 32-row MFMA tile would be 90 % padding, so each lane dots its 32 hidden
            // values with the matching W3 columns (broadcast LDS reads, issued before the group's GELU so their latency runs
            // under its arithmetic) and the two lane halves are added
#pragma unroll
            for (int to = 0; to < 2; ++to)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 w3q[3];
                    if (!L3MM) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) w3q[c] = ld4(&w3_row[c * LD2 + 32 * to + 8 * r4]);
                    }
                    const f32x4_t zq = {z[to][4 * r4], z[to][4 * r4 + 1], z[to][4 * r4 + 2], z[to][4 * r4 + 3]};
                    f32x4_t aq4, dq4;
                    gelu_and_grad4(zq, aq4, dq4);
                    if (TRAIN && NIC_FK_PIN) asm volatile("" : "+v"(dq4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a2[to][4 * r4 + j] = aq4[j];
                        d2[to][4 * r4 + j] = TRAIN ? dq4[j] : 0.f;
                        if (!L3MM) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) part3[c] = fmaf(w3q[c][j], aq4[j], part3[c]);
                        }
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        float yv[3];
        if constexpr (L3MM) {
            // layer 3 as one more chained split product: A = the W3 image (16 rows, 3 real; the other row tile re-reads them),
            // B = the a2 fragments - which are also the a2 image of dW3.  Rows 0..2 of the result sit in registers 0..2 of the
            // lane that owns the sample (lane half 0).
            f32x16 z3 = f32x16(0.f);
            Frag2 afq[4];                                                  // all four W3 fragments up front (16 registers)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                afq[ks].hi = frag_row(&w3b_row[16 * ks]);
                afq[ks].lo = frag_row(&w3b_row[16 * LD2 + 16 * ks]);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const Frag2 bf = split_acc(a2[ks >> 1], ks & 1);
                if (TRAIN) {                                             // the DZ region is free until dZ2 is stored
                    store_frag(&dz_st[16 * ks], bf.hi);
                    store_frag(&dz_st[S::BZ + 16 * ks], bf.lo);
                }
                z3 = mfma_split(afq[ks], bf, z3);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) yv[c] = sigmoid_f(z3[c] + B3s[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) yv[c] = sigmoid_f(part3[c] + __shfl_xor(part3[c], 32) + B3s[c]);
        }
        STAMP(1);    // X^T store, layers 1 + 2 (MFMA), their GELUs, layer 3
        if (p.y != nullptr && valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) p.y[n * 3 + c] = yv[c];
        }
        if (!TRAIN && p.y_u8 != nullptr && valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) p.y_u8[n * 3 + c] = (uint8_t)(int)floorf(add_rn(mul_rn(yv[c], 255.0f), 0.5f));   // y in (0, 1): 0 .. 255
        }
        if (!TRAIN) continue;

        __builtin_amdgcn_sched_barrier(0);
        // ---------- dZ3 (lane-half 0 owns the sample's 3 outputs)
        float dz3[3];
        {
            const bool own = valid && h == 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float g;
                if (MODE == MODE_TRAIN_MSE || MODE == MODE_TRAIN_IMG) {
                    const float diff = own ? yv[c] - tgt[c] : 0.f;
                    accLoss += diff * diff;
                    g = p.grad_scale * diff;
                } else {
                    g = own ? tgt[c] : 0.f;
                }
                dz3[c] = g * yv[c] * (1.0f - yv[c]);
                accB3[c] += dz3[c];
            }
        }
        if constexpr (SPLIT && SRC == SRC_ENCODE) {
        // ================= split-bf16 backward: every product on the bf16 matrix pipe, operands from the [sample][feature] images
        // ---------- (the a2 image was stored with layer 3) dZ3 -> its [c][s] image
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const __bf16 hi = (__bf16)dz3[c];
                d3_st[c * 32] = hi;
                d3_st[128 + c * 32] = (__bf16)(dz3[c] - (float)hi);
            }
        }
        wave_lds_fence();
        // ---------- dW3[c][k = lane] += sum_s dZ3[c][s] A2[k][s]: 4x4x4 MFMAs (16 blocks of 4 columns: lane l <-> column l),
        // A = dZ3[c = lane & 3][4 samples] (the same in every block), B = column l of 4 sample rows of the a2 image
        {
            s16x4 bh[8], bl[8], ah[8], al[8];                              // 32 small reads in flight, then 24 MFMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {                                  // samples r, r + 8, r + 16, r + 24
                bh[r] = tr4(&dz_b44[r * S::LDZB]);
                bl[r] = tr4(&dz_b44[S::BZ + r * S::LDZB]);
                ah[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[4 * r]);
                al[r] = *reinterpret_cast<lds_cs16x4*>(&d3_a44[128 + 4 * r]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                accW3q = mfma4_bf(al[r], bh[r], accW3q);
                accW3q = mfma4_bf(ah[r], bl[r], accW3q);
                accW3q = mfma4_bf(ah[r], bh[r], accW3q);
            }
        }
        // ---------- dA2 = W3^T dZ3: one k-step (rows c = 0..2 of 16; lane half 0 carries dZ3 in elements 0..2), dZ2 = dA2 * gelu'(Z2)
        f32x16 dz2[2];
        {
            const float dzv[8] = {dz3[0], dz3[1], dz3[2], 0.f, 0.f, 0.f, 0.f, 0.f};      // lane half 1: zeros (dZ3 is masked there)
            const Frag2 bf = split8(dzv);
#pragma unroll
            for (int tk = 0; tk < 2; ++tk) {
                Frag2 af;
                af.hi = frag_tr<LD2>(&w3b_tr[32 * tk]);
                af.lo = frag_tr<LD2>(&w3b_tr[16 * LD2 + 32 * tk]);
                dz2[tk] = mfma_split(af, bf, f32x16(0.f)) * d2[tk];
            }
        }
        STAMP(2);    // dZ3 image, dW3, dA2
        wave_lds_fence();                                              // the dW3 reads of the DZ region are issued: it may be overwritten
        // ---------- dA1 = W2^T dZ2; the split dZ2 fragments are also the dZ2 image of the weight-gradient product
        f32x16 dz1[2];
        {
            f32x16 acc[2] = {f32x16(0.f), f32x16(0.f)};
            auto loadw = [&](int ks, Frag2 (&af)[2]) {
#pragma unroll
                for (int tk = 0; tk < 2; ++tk) {
                    af[tk].hi = frag_tr<LD2>(&w2b_tr[16 * ks * LD2 + 32 * tk]);
                    af[tk].lo = frag_tr<LD2>(&w2b_tr[kH * LD2 + 16 * ks * LD2 + 32 * tk]);
                }
            };
            Frag2 afq[2][2];                                               // weight fragments of step k + 1 in flight during step k
            loadw(0, afq[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) loadw(ks + 1, afq[(ks + 1) & 1]);
                const Frag2 bf = split_acc(dz2[ks >> 1], ks & 1);
                store_frag(&dz_st[16 * ks], bf.hi);
                store_frag(&dz_st[S::BZ + 16 * ks], bf.lo);
#pragma unroll
                for (int tk = 0; tk < 2; ++tk) acc[tk] = mfma_split(afq[ks & 1][tk], bf, acc[tk]);
                NIC_DW_SB;
            }
            dz1[0] = acc[0] * d1[0];
            dz1[1] = acc[1] * d1[1];
        }
        wave_lds_fence();
        // ---------- db2[o = lane] += sum_s dZ2[o][s]: 4x4x4 MFMAs against a block of ones (every output row is the column sum)
        {
            const s16x4 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
            s16x4 bh[8], bl[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                bl[r] = tr4(&dz_b44[S::BZ + r * S::LDZB]);
                bh[r] = tr4(&dz_b44[r * S::LDZB]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                accB2q = mfma4_bf(ones, bl[r], accB2q);
                accB2q = mfma4_bf(ones, bh[r], accB2q);
            }
        }
        STAMP(3);    // dA1 (+ dZ2 image), db2
        wg_lds_barrier();
        STAMP(4);    // wait at barrier 1
        // ---------- dW2 tile (to2, tk2) += sum over the four waves' samples of dZ2[o][s] A1[k][s]
        {
            auto load2 = [&](int ksw, Frag2& af, Frag2& bf) {             // k-step = (source wave, 16 samples)
                const int off = (ksw >> 1) * S::SPW + 4 * (ksw & 1) * S::LDZB;
                // issue order = reverse order of first use (mfma_split starts with lo x hi): one s_waitcnt per step, and an
                // s_waitcnt costs a lone wave an issue slot like any other instruction
                af.hi = frag_trs<S::LDZB>(&dz_tr[off + 32 * to2]);
                bf.lo = frag_trs<S::LDZB>(&a1_tr[off + S::BZ + 32 * tk2]);
                af.lo = frag_trs<S::LDZB>(&dz_tr[off + S::BZ + 32 * to2]);
                bf.hi = frag_trs<S::LDZB>(&a1_tr[off + 32 * tk2]);
            };
            Frag2 af[2], bf[2];                                            // fragments of step k + 1 are in flight during step k
            load2(0, af[0], bf[0]);
#pragma unroll
            for (int ksw = 0; ksw < 8; ++ksw) {
                if (ksw + 1 < 8) load2(ksw + 1, af[(ksw + 1) & 1], bf[(ksw + 1) & 1]);
                accW2o = mfma_split(af[ksw & 1], bf[ksw & 1], accW2o);
                NIC_DW_SB;
            }
        }
        STAMP(5);    // dW2 MFMAs (owned tile, 4 sources)
        wg_lds_barrier();                                              // everyone is done reading dZ2 before dZ1 replaces it
        STAMP(6);    // wait at barrier 2
        // ---------- dX = W1p^T dZ1 for the grid slots; the split dZ1 fragments are the dZ1 image
        {
            f32x16 dxacc[NGT];
#pragma unroll
            for (int tg = 0; tg < NGT; ++tg) dxacc[tg] = f32x16(0.f);
            // the first 16 G0 slots are a whole accumulator tile: their running sums over the rounds ride in the matrix pipe's C
            // operand (no vector add, and the sums never have to leave the accumulator registers)
            constexpr bool GX = NIC_GX && GridAcc<L>::NG0 >= 16;
            if (GX) {
#pragma unroll
                for (int s = 0; s < 16; ++s) dxacc[0][s] = gacc.g0[s];
            }
            auto loadw = [&](int ks, Frag2 (&af)[NGT]) {
#pragma unroll
                for (int tg = 0; tg < NGT; ++tg) {
                    af[tg].hi = frag_tr<LD1>(&w1b_tr[16 * ks * LD1 + 32 * tg]);
                    af[tg].lo = frag_tr<LD1>(&w1b_tr[kH * LD1 + 16 * ks * LD1 + 32 * tg]);
                }
            };
            Frag2 afq[2][NGT];
            loadw(0, afq[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) loadw(ks + 1, afq[(ks + 1) & 1]);
                const Frag2 bf = split_acc(dz1[ks >> 1], ks & 1);
                store_frag(&dz_st[16 * ks], bf.hi);
                store_frag(&dz_st[S::BZ + 16 * ks], bf.lo);
#pragma unroll
                for (int tg = 0; tg < NGT; ++tg) dxacc[tg] = mfma_split(afq[ks & 1][tg], bf, dxacc[tg]);
                NIC_DW_SB;
            }
            accumulate_grid_grads<L, NGT>(p, cx, dxacc, gacc, GX ? 16 : 0);  // masked lanes carry exact zeros (dZ3 = 0)
        }
        STAMP(7);    // dX (+ dZ1 image), grid-gradient accumulation
        wg_lds_barrier();
        STAMP(8);    // wait at barrier 3
        // ---------- dW1 tiles (to1, 2c + tk1) += sum over the four waves' samples of dZ1[o][s] X[rho][s]
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) {
            auto load1 = [&](int ksw, Frag2& af, Frag2& bf) {
                const int offz = (ksw >> 1) * S::SPW + 4 * (ksw & 1) * S::LDZB, offx = (ksw >> 1) * S::SPW + 4 * (ksw & 1) * S::LDXB;
                af.hi = frag_trs<S::LDZB>(&dz_tr[offz + 32 * to1]);
                bf.lo = frag_trs<S::LDXB>(&x_tr[offx + S::BX + 64 * c2 + 32 * tk1]);
                af.lo = frag_trs<S::LDZB>(&dz_tr[offz + S::BZ + 32 * to1]);
                bf.hi = frag_trs<S::LDXB>(&x_tr[offx + 64 * c2 + 32 * tk1]);
            };
            Frag2 af[2], bf[2];
            load1(0, af[0], bf[0]);
#pragma unroll
            for (int ksw = 0; ksw < 8; ++ksw) {
                if (ksw + 1 < 8) load1(ksw + 1, af[(ksw + 1) & 1], bf[(ksw + 1) & 1]);
                accW1o[c2] = mfma_split(af[ksw & 1], bf[ksw & 1], accW1o[c2]);
                NIC_DW_SB;
            }
        }
        // odd last col tile (16 real rows): 16x16x32 over this wave's own 32 samples, 4 row tiles of 16
        {
            auto frag16 = [&](lds_cbf* ptr, int ld) { return join8(tr4(ptr), tr4(ptr + ld)); };
            Frag2 bf;
            bf.hi = frag16(&x_p16[0], S::LDXB);
            bf.lo = frag16(&x_p16[S::BX], S::LDXB);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                Frag2 af;
                af.hi = frag16(&dz_p16[16 * ot], S::LDZB);
                af.lo = frag16(&dz_p16[S::BZ + 16 * ot], S::LDZB);
                accW1q[ot] = mfma16_bf(af.lo, bf.hi, accW1q[ot]);
                accW1q[ot] = mfma16_bf(af.hi, bf.lo, accW1q[ot]);
                accW1q[ot] = mfma16_bf(af.hi, bf.hi, accW1q[ot]);
            }
        }
        STAMP(9);    // dW1 MFMAs (owned + partial tiles)
        wg_lds_barrier();                  // all reads of dZ1 / X done before the next round overwrites them
        STAMP(10);   // wait at barrier 4
        } else {
        // ---------- dW3[c][k] += sum_s dZ3[c][s] A2[k][s]: lane k walks row k of the transposed A2 image
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) sa_st[c * LDT] = dz3[c];          // h == 0: sa_st = SA + pl
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sb_st[(32 * t + ROWC(r)) * LDT] = a2[t][r];
        wave_lds_fence();
        {
            f32x4 avq[2], zvq[2][3];                                       // operands of group g + 1 are in flight during group g
            avq[0] = ld4(&sb_lane[0]);
#pragma unroll
            for (int c = 0; c < 3; ++c) zvq[0][c] = ld4(&sa_bc[c * LDT]);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) {
                    avq[(g + 1) & 1] = ld4(&sb_lane[4 * (g + 1)]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) zvq[(g + 1) & 1][c] = ld4(&sa_bc[c * LDT + 4 * (g + 1)]);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accW3[c] = fmaf(zvq[g & 1][c][j], avq[g & 1][j], accW3[c]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------- dA2 = W3^T dZ3 (K = 3, padded to 4), dZ2 = dA2 * gelu'(Z2)
        f32x16 dz2[2];
        {
            const float s1 = __shfl(dz3[1], pl);
            const float b0 = h ? s1 : dz3[0];
            const float b1 = h ? 0.f : dz3[2];
#pragma unroll
            for (int tk = 0; tk < 2; ++tk) {
                f32x16 acc = f32x16(0.f);
                acc = mfma32(w3_col[32 * tk], b0, acc);
                acc = mfma32(w3_col[2 * LD2 + 32 * tk], b1, acc);
                dz2[tk] = acc * d2[tk];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);    // layer 3, y store, dZ3, dW3 row pass, dA2
        // ---------- dW2[o][k] += sum_s dZ2[o][s] A1[k][s]: every wave publishes its transposed operands, then contracts the
        // ONE tile it owns over the samples of all four waves
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sa_st[(32 * t + ROWC(r)) * LDT] = dz2[t][r];
                sb_st[(32 * t + ROWC(r)) * LDT] = a1[t][r];
            }
        wave_lds_fence();
        {                                                              // db2[o = lane] += sum_s dZ2[o][s]   (own samples)
            f32x4 zq8[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) zq8[g] = ld4(&sa_lane[4 * g]);     // all eight reads in flight, then the sums
#pragma unroll
            for (int g = 0; g < 8; ++g) accB2 += (zq8[g][0] + zq8[g][1]) + (zq8[g][2] + zq8[g][3]);
        }
        STAMP(3);    // dZ2^T / A1^T stores, db2 row pass
        wg_lds_barrier();
        STAMP(4);    // wait at barrier 1
        {
            lds_cf* const sa_o = opaque(SCR0 + pl * LDT + 16 * h + 32 * to2 * LDT);
            lds_cf* const sb_o = opaque(SCR0 + 64 * LDT + pl * LDT + 16 * h + 32 * tk2 * LDT);
            if constexpr (CHAIN && NIC_CHAIN_DW) {
                // split on read: k-step = (source wave, samples 16h + 8u .. + 7 of lane half h); the rows of step k + 1 are in flight
                // while step k is split and multiplied
                Row8 aq[2], bq[2];
                aq[0] = ld_row8(&sa_o[0]);
                bq[0] = ld_row8(&sb_o[0]);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 1 < 8) {
                        aq[(ks + 1) & 1] = ld_row8(&sa_o[((ks + 1) >> 1) * S::SCR_PER_WAVE + 8 * ((ks + 1) & 1)]);
                        bq[(ks + 1) & 1] = ld_row8(&sb_o[((ks + 1) >> 1) * S::SCR_PER_WAVE + 8 * ((ks + 1) & 1)]);
                    }
                    accW2o = mfma_split(split_row8(aq[ks & 1]), split_row8(bq[ks & 1]), accW2o);
                }
            } else {
            f32x4 aq[2], bq[2];
            aq[0] = ld4(&sa_o[0]);
            bq[0] = ld4(&sb_o[0]);
#pragma unroll
            for (int st = 0; st < 16; ++st) {                            // step = (source wave, 4 samples)
                if (st + 1 < 16) {
                    aq[(st + 1) & 1] = ld4(&sa_o[((st + 1) >> 2) * S::SCR_PER_WAVE + 4 * ((st + 1) & 3)]);
                    bq[(st + 1) & 1] = ld4(&sb_o[((st + 1) >> 2) * S::SCR_PER_WAVE + 4 * ((st + 1) & 3)]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) accW2o = mfma32(aq[st & 1][j], bq[st & 1][j], accW2o);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
        STAMP(5);    // dW2 MFMAs (owned tile, 4 sources)
        wg_lds_barrier();                                              // everyone is done reading before the operands are replaced
        STAMP(6);    // wait at barrier 2
        __builtin_amdgcn_sched_barrier(0);
        // ---------- dA1 = W2^T dZ2, dZ1 = dA1 * gelu'(Z1)
        f32x16 dz1[2];
        {
            f32x16 acc[2] = {f32x16(0.f), f32x16(0.f)};
            if constexpr (CHAIN) {
                auto loadw = [&](int ks, Frag2 (&af)[2]) {
#pragma unroll
                    for (int tk = 0; tk < 2; ++tk) {
                        af[tk].hi = frag_tr<LD2>(&w2b_tr[16 * ks * LD2 + 32 * tk]);
                        af[tk].lo = frag_tr<LD2>(&w2b_tr[kH * LD2 + 16 * ks * LD2 + 32 * tk]);
                    }
                };
                Frag2 afq[2][2];
                loadw(0, afq[0]);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                         // contraction rows o = 16 ks + ..
                    if (ks + 1 < 4) loadw(ks + 1, afq[(ks + 1) & 1]);
                    const Frag2 bf = split_acc(dz2[ks >> 1], ks & 1);
#pragma unroll
                    for (int tk = 0; tk < 2; ++tk) acc[tk] = mfma_split(afq[ks & 1][tk], bf, acc[tk]);
                    NIC_DW_SB;
                }
            } else {
            // A operands (columns of W2) are fetched one step (2 k-steps = 4 MFMAs) ahead: the LDS latency of step i + 1
            // runs under the MFMAs of step i instead of in front of them
            float wq[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int tk = 0; tk < 2; ++tk) wq[0][2 * u + tk] = w2_col[ROWC(u) * LD2 + 32 * tk];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (st + 1 < 16) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int rr = 2 * (st + 1) + u, o = 32 * (rr >> 4) + ROWC(rr & 15);
#pragma unroll
                        for (int tk = 0; tk < 2; ++tk) wq[(st + 1) & 1][2 * u + tk] = w2_col[o * LD2 + 32 * tk];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int rr = 2 * st + u;
#pragma unroll
                    for (int tk = 0; tk < 2; ++tk) acc[tk] = mfma32(wq[st & 1][2 * u + tk], dz2[rr >> 4][rr & 15], acc[tk]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            dz1[0] = acc[0] * d1[0];
            dz1[1] = acc[1] * d1[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------- dW1[o][rho] += sum_s dZ1[o][s] X[rho][s]
        if (S::XIMG) {
            // X^T of every wave is already in LDS: publish dZ1^T, then each wave contracts the tiles it owns
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sa_st[(32 * t + ROWC(r)) * LDT] = dz1[t][r];
            STAMP(7);    // dA1 MFMAs, dZ1, dZ1^T store
            wg_lds_barrier();
            STAMP(8);    // wait at barrier 3
            lds_cf* const sa_o = opaque(SCR0 + pl * LDT + 16 * h + 32 * to1 * LDT);
            lds_cf* const xi_o = opaque(sm + S::OFF_XIMG + pl * LDT + 16 * h + 32 * tk1 * LDT);
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2) {
                f32x4 aq[2], bq[2];
                aq[0] = ld4(&sa_o[0]);
                bq[0] = ld4(&xi_o[64 * c2 * LDT]);
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    if (st + 1 < 16) {
                        aq[(st + 1) & 1] = ld4(&sa_o[((st + 1) >> 2) * S::SCR_PER_WAVE + 4 * ((st + 1) & 3)]);
                        bq[(st + 1) & 1] = ld4(&xi_o[((st + 1) >> 2) * S::XIMG_PER_WAVE + 64 * c2 * LDT + 4 * ((st + 1) & 3)]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) accW1o[c2] = mfma32(aq[st & 1][j], bq[st & 1][j], accW1o[c2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (S::PART16) {
                // odd last col tile, 16 real rows: 4 row tiles of 16x16x4 over this wave's own samples; lane (i, kg) feeds
                // row / col i with samples 8 kg + j for MFMA j
                const int i16 = lane & 15, kg = lane >> 4;
                lds_cf* const xi_q = opaque(sm + S::OFF_XIMG + wave * S::XIMG_PER_WAVE + (32 * (KT - 1) + i16) * LDT + 8 * kg);
                lds_cf* const sa_q = opaque(SA + i16 * LDT + 8 * kg);
                f32x4 bq2[2], aq2[4][2];
                bq2[0] = ld4(&xi_q[0]);
                bq2[1] = ld4(&xi_q[4]);
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    aq2[ot][0] = ld4(&sa_q[16 * ot * LDT]);
                    aq2[ot][1] = ld4(&sa_q[16 * ot * LDT + 4]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot) accW1q[ot] = mfma16(aq2[ot][j >> 2][j & 3], bq2[j >> 2][j & 3], accW1q[ot]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (S::PART) {
                // odd last col tile: both row tiles over this wave's own samples
                constexpr int ROWS_LAST = S::KPAD - 32 * (KT - 1);         // lanes past the image re-read a valid row: their columns
                const int prow = pl < ROWS_LAST ? pl : pl - ROWS_LAST;     // (rho >= KPAD) are discarded by the reduction
                lds_cf* const xi_w = opaque(sm + S::OFF_XIMG + wave * S::XIMG_PER_WAVE + prow * LDT + 16 * h + 32 * (KT - 1) * LDT);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = ld4(&xi_w[4 * g]);
#pragma unroll
                    for (int to = 0; to < 2; ++to) {
                        const f32x4 a = ld4(&sa_rd[32 * to * LDT + 4 * g]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) accW1p[to] = mfma32(a[j], b[j], accW1p[to]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            STAMP(9);    // dW1 MFMAs (owned + partial tiles)
            wg_lds_barrier();                  // all reads of dZ1^T / X^T done before the next round overwrites them
            STAMP(10);   // wait at barrier 4
        } else {
        // (X staged two 32-row tiles at a time through SB)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa_st[(32 * t + ROWC(r)) * LDT] = dz1[t][r];
#pragma unroll
        for (int c2 = 0; c2 < (KT + 1) / 2; ++c2) {
#pragma unroll
            for (int s = 32 * c2; s < 32 * c2 + 32 && s < L::NSLOT; ++s) sb_st[(32 * ((s >> 4) & 1) + ROWC(s & 15)) * LDT] = xs[s];
            if (c2 < NCH) {
                // full chunk: owned tile (to1, 2 c2 + tk1) over the four waves' samples
                wg_lds_barrier();
                lds_cf* const sa_o = opaque(SCR0 + pl * LDT + 16 * h + 32 * to1 * LDT);
                lds_cf* const sb_o = opaque(SCR0 + 64 * LDT + pl * LDT + 16 * h + 32 * tk1 * LDT);
                if constexpr (CHAIN && NIC_CHAIN_DW) {
                    Row8 aq[2], bq[2];
                    aq[0] = ld_row8(&sa_o[0]);
                    bq[0] = ld_row8(&sb_o[0]);
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        if (ks + 1 < 8) {
                            aq[(ks + 1) & 1] = ld_row8(&sa_o[((ks + 1) >> 1) * S::SCR_PER_WAVE + 8 * ((ks + 1) & 1)]);
                            bq[(ks + 1) & 1] = ld_row8(&sb_o[((ks + 1) >> 1) * S::SCR_PER_WAVE + 8 * ((ks + 1) & 1)]);
                        }
                        accW1o[c2 < NCH ? c2 : 0] = mfma_split(split_row8(aq[ks & 1]), split_row8(bq[ks & 1]), accW1o[c2 < NCH ? c2 : 0]);
                    }
                } else {
#pragma unroll
                for (int src = 0; src < 4; ++src) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 a = ld4(&sa_o[src * S::SCR_PER_WAVE + 4 * g]);
                        const f32x4 b = ld4(&sb_o[src * S::SCR_PER_WAVE + 4 * g]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) accW1o[c2 < NCH ? c2 : 0] = mfma32(a[j], b[j], accW1o[c2 < NCH ? c2 : 0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                wg_lds_barrier();
            } else {
                // odd last col tile: both row tiles over this wave's own samples
                wave_lds_fence();
                if constexpr (CHAIN && NIC_CHAIN_DW && NIC_CHAIN_DW_PART) {
                    const Row8 b0 = ld_row8(&sb_rd[0]), b1 = ld_row8(&sb_rd[8]);
                    const Row8 a00 = ld_row8(&sa_rd[0]), a01 = ld_row8(&sa_rd[8]);
                    const Row8 a10 = ld_row8(&sa_rd[32 * LDT]), a11 = ld_row8(&sa_rd[32 * LDT + 8]);
                    const Frag2 bf0 = split_row8(b0), bf1 = split_row8(b1);
                    // (the round's products start from zero and are added to the running sums: accumulating into accW1p directly
                    //  crashes the compiler's 'AMDGPU Rewrite AGPR-Copy-MFMA' pass on the method-4 kernel, ROCm 7.2)
                    f32x16 t0 = mfma_split(split_row8(a00), bf0, f32x16(0.f));
                    f32x16 t1 = mfma_split(split_row8(a10), bf0, f32x16(0.f));
                    t0 = mfma_split(split_row8(a01), bf1, t0);
                    t1 = mfma_split(split_row8(a11), bf1, t1);
                    accW1p[0] += t0;
                    accW1p[1] += t1;
                } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = ld4(&sb_rd[4 * g]);
#pragma unroll
                    for (int to = 0; to < 2; ++to) {
                        const f32x4 a = ld4(&sa_rd[32 * to * LDT + 4 * g]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) accW1p[to] = mfma32(a[j], b[j], accW1p[to]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                wave_lds_fence();
            }
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------- dX = W1p^T dZ1 for the slots that need it
        if (SRC == SRC_ENCODE || p.dx != nullptr) {
            f32x16 dxacc[NGT];
#pragma unroll
            for (int tg = 0; tg < NGT; ++tg) dxacc[tg] = f32x16(0.f);
            // whole accumulator tiles of G0 slots carry their running sums in the product's C operand (see the split block)
            // (measured per layout: method 4 +2 %, 2D fp32 neutral; method 3 with its three G0 tiles 6 % slower - left alone)
            constexpr int GXT = (NIC_GX_F32 && SRC == SRC_ENCODE && GridAcc<L>::NG0 < 32) ? GridAcc<L>::NG0 / 16 : 0;
#pragma unroll
            for (int tg = 0; tg < GXT; ++tg)
#pragma unroll
                for (int s = 0; s < 16; ++s) dxacc[tg][s] = gacc.g0[16 * tg + s];
            if constexpr (CHAIN) {
                auto loadw = [&](int ks, Frag2 (&af)[NGT]) {
#pragma unroll
                    for (int tg = 0; tg < NGT; ++tg) {
                        af[tg].hi = frag_tr<LD1>(&w1b_tr[16 * ks * LD1 + 32 * tg]);
                        af[tg].lo = frag_tr<LD1>(&w1b_tr[kH * LD1 + 16 * ks * LD1 + 32 * tg]);
                    }
                };
                Frag2 afq[2][NGT];
                loadw(0, afq[0]);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4) loadw(ks + 1, afq[(ks + 1) & 1]);
                    const Frag2 bf = split_acc(dz1[ks >> 1], ks & 1);
#pragma unroll
                    for (int tg = 0; tg < NGT; ++tg) dxacc[tg] = mfma_split(afq[ks & 1][tg], bf, dxacc[tg]);
                    NIC_DW_SB;
                }
            } else {
            // same one-step-ahead operand fetch as dA1
            float wq[2][2 * NGT];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int tg = 0; tg < NGT; ++tg) wq[0][NGT * u + tg] = w1_col[ROWC(u) * LD1 + 32 * tg];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (st + 1 < 16) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int rr = 2 * (st + 1) + u, o = 32 * (rr >> 4) + ROWC(rr & 15);
#pragma unroll
                        for (int tg = 0; tg < NGT; ++tg) wq[(st + 1) & 1][NGT * u + tg] = w1_col[o * LD1 + 32 * tg];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int rr = 2 * st + u;
#pragma unroll
                    for (int tg = 0; tg < NGT; ++tg) dxacc[tg] = mfma32(wq[st & 1][NGT * u + tg], dz1[rr >> 4][rr & 15], dxacc[tg]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if (SRC == SRC_ENCODE) {
                accumulate_grid_grads<L, NGT>(p, cx, dxacc, gacc, 16 * GXT);   // masked lanes carry exact zeros (dZ3 = 0)
            } else if (valid) {
                float* row = p.dx + n * L::CIN;
#pragma unroll
                for (int s = 0; s < L::NSLOT; ++s) {
                    const int ch = L::slot_channel(s, h);
                    if (ch >= 0) row[ch] = dxacc[s >> 4][s & 15];
                }
            }
        }
        }
        STAMP(11);   // dX MFMAs, grid-gradient accumulation
      }  // rounds of one macro-tile
        if (SRC == SRC_ENCODE && TRAIN) {
            {
                const int pkd[3] = {p.pk_nc > 0 ? p.pk_bx : 0, p.pk_by, p.pk_nc > 0 ? p.pk_nc / (p.pk_bx * p.pk_by) : 0};
                combine_g1_lanes<L>(gacc, blk_off1, blk, lane, lw, pkd, pk_lc);
            }
            bool flush = true;
            // Small launches deal the rounds of a macro-tile out in groups (work units) on neighbouring waves.  Left alone, those
            // waves flush the SAME nodes in the same few thousand cycles (4 / 8 / 16 groups of the default 8 x 256^2 step: 0.35 / 0.54 /
            // 0.96 ms against 0.30 with 2).  The groups that sit in one workgroup are summed through LDS first - the wave regions are
            // free between the last barrier of a unit and the first store of the next - and the lowest wave flushes for all.
            constexpr int GV = GridAcc<L>::NG0 + GridAcc<L>::K1 * (kC / 2);
            constexpr int REGION = SPLIT ? S::SPW / 2 : S::SCR_PER_WAVE;                  // floats per wave
            if (NIC_GROUP_SUM && GV * 64 <= REGION && rg > 0) {                            // segment-uniform
                lds_f* const reg0 = sm + (SPLIT ? S::OFF_IMG : S::OFF_SCR);
                int leader = wave;
                if (tile_ok)
                    for (int w = wave - 1; w >= 0; --w)
                        if (seg_tile0 + ((base + w) >> rg) == tile) leader = w;
                if (leader != wave) {
                    lds_f* const mine = opaque(reg0 + wave * REGION + lane);
#pragma unroll
                    for (int i = 0; i < GridAcc<L>::NG0; ++i) mine[i * 64] = gacc.g0[i];
#pragma unroll
                    for (int i = 0; i < GridAcc<L>::K1 * (kC / 2); ++i) mine[(GridAcc<L>::NG0 + i) * 64] = gacc.g1[i];
                }
                wg_lds_barrier();
                if (leader == wave && tile_ok) {
                    for (int w = wave + 1; w < 4; ++w) {
                        if (base + w >= t_end || seg_tile0 + ((base + w) >> rg) != tile) break;
                        lds_cf* const theirs = opaque(reg0 + w * REGION + lane);
#pragma unroll
                        for (int i = 0; i < GridAcc<L>::NG0; ++i) gacc.g0[i] += theirs[i * 64];
#pragma unroll
                        for (int i = 0; i < GridAcc<L>::K1 * (kC / 2); ++i) gacc.g1[i] += theirs[(GridAcc<L>::NG0 + i) * 64];
                    }
                }
                wg_lds_barrier();
                flush = leader == wave;
            }
            if (flush) {
                preadd_g0_lanes<L>(gacc, blk_off0, lane, p.pk_nc > 0 ? p.pk_bx : (1 << lw), (uint32_t)p.g0.nx);
                flush_grid_grads<L>(p, blk_off0, blk_off1, h, gacc);
            }
        }
        STAMP(13);   // grid-gradient flush (atomics)
    }  // macro-tile loop
  }  // segments
#ifdef NIC_STAMPS
    if (lane == 0) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.partials + (size_t)gridDim.x * S::REC) + ((size_t)blockIdx.x * 4 + wave) * 16;   // behind the records (nic_workspace_bytes leaves 1 MiB)
#pragma unroll
        for (int i = 0; i < NIC_NPH; ++i) dst[i] = stamp_sum[i];
    }
#endif

    if (!TRAIN) return;
    // ---------------- flush: ONE record per workgroup; every wave writes the tiles it owns, its partial tiles and its tail
    float* rec = p.partials + (int64_t)blockIdx.x * S::REC;
    {
        const int a2 = 2 * KT + 2 * to2 + tk2;
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[(a2 * 16 + r) * 64 + lane] = accW2o[r];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int a1i = to1 * KT + 2 * c + tk1;
#pragma unroll
            for (int r = 0; r < 16; ++r) rec[(a1i * 16 + r) * 64 + lane] = accW1o[c][r];
        }
        if (S::PART16) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) rec[(S::NACC + wave) * 1024 + (ot * 4 + r) * 64 + lane] = accW1q[ot][r];
        } else if (S::PART) {
#pragma unroll
            for (int to = 0; to < 2; ++to)
#pragma unroll
                for (int r = 0; r < 16; ++r) rec[((S::NACC + 2 * wave + to) * 16 + r) * 64 + lane] = accW1p[to][r];
        }
    }
    float* tail = rec + S::NSLOT_REC * 1024 + wave * 320;
    tail[lane] = SPLIT ? accB2q[0] : accB2;
#pragma unroll
    for (int c = 0; c < 3; ++c) tail[64 + 64 * c + lane] = SPLIT ? accW3q[c] : accW3[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = c < 3 ? accB3[c] : accLoss;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) tail[256 + c] = v;
    }
}

// =====================================================================================================
// Fixed-order reduction of the per-wave records into the decoder gradients (nn.Linear layouts) and loss.
template <class L>
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* partials, int n_waves /* records */, nic_mlp_grads g, float* loss, float loss_scale, const StepTail tl) {
    if (tail_block(tl)) return;                                   // a streaming block of the optimiser tail (nic_adam.hpp)
    using S = Lds<L>;
    constexpr int KT = S::KT;
    // a block = 32 outputs x 8 slices of the record list (a serial walk over all records per output left the launch
    // latency-bound: 52 us for 24 MB); the slices are combined through LDS in slice order - still one fixed summation tree
    // per grid size
    __shared__ float red[8][32];
    const int slice = threadIdx.x >> 5;
    const int gid = blockIdx.x * 32 + (threadIdx.x & 31);
    const bool live = gid < S::NACC * 1024 + S::TAIL;
    // which floats of a record feed output gid: one slot, or (partial dW1 tiles, tails) one slot per wave
    int nsrc = 1, off0 = gid, stride = 0;
    if (gid < S::NACC * 1024) {
        const int a = gid >> 10;
        if (S::PART && a < 2 * KT && a % KT == KT - 1) {          // dW1 tile (to, KT-1): 4 per-wave partials
            if (S::PART16) {
                // element (o, rho = 32 (KT-1) + col) of the 16x16 tiles: tile o >> 4, D register o & 3 of lane 16 ((o & 15) >> 2) + col
                const int r = (gid >> 6) & 15, ln = gid & 63;
                const int o = 32 * (a / KT) + ROW(r, ln >> 5), col = ln & 31;
                nsrc = col < 16 ? 4 : 0;
                off0 = S::NACC * 1024 + ((o >> 4) * 4 + (o & 3)) * 64 + 16 * ((o & 15) >> 2) + (col & 15);
                stride = 1024;
            } else {
                nsrc = 4; off0 = (S::NACC + a / KT) * 1024 + (gid & 1023); stride = 2 * 1024;
            }
        }
    } else {
        nsrc = 4; off0 = S::NSLOT_REC * 1024 + (gid - S::NACC * 1024); stride = 320;
    }
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // fixed summation tree: bit-stable for a given grid size
    const int per = (n_waves + 7) >> 3;                              // records per slice
    const int w_lo = slice * per, w_hi = (w_lo + per < n_waves) ? w_lo + per : n_waves;
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
    if (slice != 0 || !live) return;
    float acc = red[0][threadIdx.x];
#pragma unroll
    for (int sl = 1; sl < 8; ++sl) acc += red[sl][threadIdx.x];
    if (gid < S::NACC * 1024) {
        const int a = gid >> 10, r = (gid >> 6) & 15, lane = gid & 63;
        const int row = ROW(r, lane >> 5), col = lane & 31;
        if (a < 2 * KT) {                                       // dW1 tile (to, tk)
            const int to = a / KT, tk = a % KT;
            const int o = 32 * to + row, rho = 32 * tk + col;
            const int ch = rho < S::KPAD ? channel_of_rho<L>(rho) : kSlotZero;
            if (ch >= 0) { if (g.w[0]) tail_store(tl, &g.w[0][o * L::CIN + ch], acc); }
            else if (ch == kSlotOne) { if (g.b[0]) tail_store(tl, &g.b[0][o], acc); }
        } else {                                                // dW2 tile (to, tk)
            const int q = a - 2 * KT;
            if (g.w[1]) tail_store(tl, &g.w[1][(32 * (q >> 1) + row) * kH + 32 * (q & 1) + col], acc);
        }
    } else {
        const int t = gid - S::NACC * 1024;
        if (t < 64) { if (g.b[1]) tail_store(tl, &g.b[1][t], acc); }
        else if (t < 256) { if (g.w[2]) tail_store(tl, &g.w[2][t - 64], acc); }            // [3][64] row-major
        else if (t < 259) { if (g.b[2]) tail_store(tl, &g.b[2][t - 256], acc); }
        else if (loss) *loss = acc * loss_scale;
    }
}

}  // namespace nic
