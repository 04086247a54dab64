// C ABI of the fused encode + decoder kernels (include/nicv2_hip.h): argument checks, launch geometry,
// workspace bookkeeping.  No allocation, no synchronisation: capture-safe.
#include "fused_launch.hpp"
#include "fused_t16.hpp"
#include <stdlib.h>
#include <string.h>

using namespace nic;

namespace {

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// non-default channel counts (FEATURE_PYRAMID_CHANNELS, PE_CHANNELS: var2.py:68-69) exist on the plain-bf16 kernels (NIC_FLAG_BF16), 3 Linear layers:
// (layout, C, P) - one translation unit each (fused_qc_*.hip)
#define NIC_CP_LIST(X)                                                                                                       \
    X(1, 4, 6) X(1, 8, 6) X(1, 16, 6) X(1, 12, 4) X(1, 12, 8) X(2, 4, 6) X(2, 8, 6) X(2, 16, 6) X(2, 12, 4) X(2, 12, 8)     \
    X(3, 4, 6) X(3, 8, 6) X(4, 4, 6) X(4, 8, 6) X(4, 16, 6)
bool cp_default(const nic_path_desc* d) { return d->channels == kC && d->pe_channels == kP; }
bool cp_listed(int layout, int c, int pch) {
#define X(L, C, P) if (layout == L && c == C && pch == P) return true;
    NIC_CP_LIST(X)
#undef X
    return false;
}
// layout id (see nic_device.hpp) or a negative error
int pick_layout(const nic_path_desc* d) {
    if (!d) return NIC_E_NULL;
    if (d->hidden != kH) return NIC_E_UNSUPPORTED;                       // HIDDEN_LAYER_CHANNELS: 64 only (every tile shape of the kernels hangs on it)
    if (!cp_default(d)) {
        if (!(d->flags & NIC_FLAG_BF16)) return NIC_E_UNSUPPORTED;      // other channel counts: the plain-bf16 kernels
        int layout = NIC_E_UNSUPPORTED;
        if (d->dim == 2 && d->method == 1) layout = d->pe_mode == NIC_PE_TRIANGULAR ? 1 : (d->pe_mode == NIC_PE_SINUSOIDAL ? 2 : NIC_E_UNSUPPORTED);
        else if (d->dim == 3 && d->method == 3 && d->pe_mode == NIC_PE_TRIANGULAR) layout = 3;
        else if (d->dim == 3 && d->method == 4 && d->pe_mode == NIC_PE_SINUSOIDAL) layout = 4;
        return (layout > 0 && cp_listed(layout, d->channels, d->pe_channels)) ? layout : NIC_E_UNSUPPORTED;
    }
    if (d->dim == 2 && d->method == 1) return d->pe_mode == NIC_PE_TRIANGULAR ? 1 : (d->pe_mode == NIC_PE_SINUSOIDAL ? 2 : NIC_E_UNSUPPORTED);
    if (d->dim == 3 && d->method == 3) return d->pe_mode == NIC_PE_TRIANGULAR ? 3 : NIC_E_UNSUPPORTED;   // fp_def.py:169
    if (d->dim == 3 && d->method == 4) return d->pe_mode == NIC_PE_SINUSOIDAL ? 4 : NIC_E_UNSUPPORTED;   // fp_def.py:208
    return NIC_E_UNSUPPORTED;
}
int layout_of_cin(int cin) { return cin == 73 ? 1 : (cin == 127 ? 3 : (cin == 79 ? 4 : NIC_E_UNSUPPORTED)); }

// 2D training steps with split-bf16 products run on the 8-wave / 16-sample kernel (fused_train16.hpp: two waves per SIMD).
// NIC_T16=0 in the environment keeps them on the 4-wave / 32-sample fused_kernel (A/B timing, and the reference the parity tests
// compare the new kernel with).
bool use_t16(int layout, const nic_path_desc* d) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("NIC_T16");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    return enabled && (layout == 1 || layout == 2) && (d->flags & NIC_FLAG_SPLIT_BF16) != 0 && (d->flags & NIC_FLAG_SPLIT_TILE32) == 0;
}
FusedInfo info_t16() { return FusedInfo{0, train16_record_floats(), 16, 1, 1, 73, 8}; }

FusedInfo info_of(int layout) {
    switch (layout) {
        case 1: return fused_info<1>();
        case 2: return fused_info<2>();
        case 3: return fused_info<3>();
        default: return fused_info<4>();
    }
}
int launch(int layout, int src, int mode, const FusedParams& p, int grid, hipStream_t s) {
    switch (layout) {
        case 1: return launch_fused<1>(src, mode, p, grid, s);
        case 2: return launch_fused<2>(src, mode, p, grid, s);
        case 3: return launch_fused<3>(src, mode, p, grid, s);
        default: return launch_fused<4>(src, mode, p, grid, s);
    }
}
int reduce(int layout, const float* partials, int n_waves, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
    switch (layout) {
        case 1: return launch_reduce<1>(partials, n_waves, g, loss, loss_scale, s);
        case 2: return launch_reduce<2>(partials, n_waves, g, loss, loss_scale, s);
        case 3: return launch_reduce<3>(partials, n_waves, g, loss, loss_scale, s);
        default: return launch_reduce<4>(partials, n_waves, g, loss, loss_scale, s);
    }
}

// ---- the optimiser tail of the call in progress (nic_path_desc.tail -> nic_adam.hpp::StepTail), parked around the reduce dispatch
thread_local hipEvent_t g_kernel_end = nullptr;               // nic_mark_kernel_end: one-shot
void mark_kernel_end(hipStream_t s) {
    if (g_kernel_end) { (void)hipEventRecord(g_kernel_end, s); g_kernel_end = nullptr; }
}
thread_local const StepTail* g_tail = nullptr;
thread_local int64_t g_tail_blocks = 0;
struct TailScope {
    StepTail tl;
    bool on = false;
    // builds the table of d->tail (null: nothing parked); rc != 0: the tail is malformed
    int open(const nic_path_desc* d, const nic_mlp_grads* grads, const int64_t* step_dev) {
        const nic_step_tail* t = d->tail;
        if (!t) return NIC_OK;
        if (!t->tensors || !grads) return NIC_E_NULL;
        if (t->count < 1 || t->count > NIC_ADAM_MAX_TENSORS || t->n_stream < 0 || t->n_stream > t->count) return NIC_E_ARG;
        if ((t->sched != nullptr) != (step_dev != nullptr) || (t->sched && t->sched_rows < 1)) return NIC_E_ARG;
        // a decoder entry's gradient is one of the buffers the reduction writes, whole
        for (int i = t->n_stream; i < t->count; ++i) {
            bool found = false;
            for (int k = 0; k < NIC_MAX_LINEAR; ++k) found = found || (t->tensors[i].grad != nullptr && (t->tensors[i].grad == grads->w[k] || t->tensors[i].grad == grads->b[k]));
            if (!found) return NIC_E_ARG;
        }
        int64_t blocks = 0;
        const int rc = adam_build_table(t->tensors, t->count, t->n_stream, t->beta1, t->beta2, t->eps, t->sched, t->sched_rows, step_dev, tl.t, tl.n_stream, blocks);
        if (rc) return rc;
        g_tail = &tl; g_tail_blocks = blocks; on = true;
        return NIC_OK;
    }
    ~TailScope() { if (on) { g_tail = nullptr; g_tail_blocks = 0; } }
};
}  // namespace
namespace nic {
TailLaunch tail_for(int reduce_blocks) {
    TailLaunch r;
    if (g_tail) { r.tl = *g_tail; r.tl.reduce_blocks = reduce_blocks; r.blocks = (unsigned)(reduce_blocks + g_tail_blocks); }
    else { r.tl.t.count = 0; r.tl.t.sched = nullptr; r.tl.n_stream = 0; r.tl.reduce_blocks = 0x7fffffff; r.blocks = (unsigned)reduce_blocks; }
    return r;
}
}  // namespace nic
namespace {

int check_geometry(const nic_path_desc* d, bool training = false) {
    if (d->num_crops < 1) return NIC_E_SHAPE;
    if (!training && d->tail) return NIC_E_ARG;                                  // the optimiser tail: training entry points only
    if ((d->flags & NIC_FLAG_ORIGINS_HOST) && d->num_crops > NIC_ORIGINS_INLINE_MAX) return NIC_E_ARG;
    if (d->max_workgroups < 0) return NIC_E_ARG;
    if (d->passes < 0 || (!training && d->passes > 1)) return NIC_E_ARG;           // passes: training entry points only
    for (int a = 0; a < d->dim; ++a)
        if (d->extent[a] < 1 || d->g0_nodes[a] < 2 || d->g1_nodes[a] < 2) return NIC_E_SHAPE;
    // the kernels address a grid element as (wave-uniform channel-plane base) + (32-bit byte offset of the lane): a G0 channel
    // plane, and 10 G1 channel planes (the lane-quarter term of the G1 offsets), must stay below 4 GiB
    int64_t n0 = 1, n1 = 1;
    for (int a = 0; a < d->dim; ++a) { n0 *= d->g0_nodes[a]; n1 *= d->g1_nodes[a]; }
    if (n0 >= (int64_t)1 << 30 || 10 * n1 >= (int64_t)1 << 30) return NIC_E_UNSUPPORTED;
    if (d->log2_step < -8 || d->log2_step > 8) return NIC_E_ARG;
    if (d->g1_weight_mode < 0 || d->g1_weight_mode > 2 || d->noise_mode < 0 || d->noise_mode > 2) return NIC_E_ARG;
    return NIC_OK;
}

// persistent grid: a multiple of 8 blocks (one slice of the tile range per XCD), at most `per_cu` per CU
// Small launches: a wave does whole work units, so with few macro-tiles the time is ceil(units / waves) x unit length - the default
// 8 x 256^2 step (1 080 macro-tiles of 16 rounds on 1 024 waves) would take 2 x 16 rounds.  The rounds of a macro-tile are therefore
// dealt out in 2 or 4 groups (work units on neighbouring waves of ONE workgroup: their grid-gradient sums are added through LDS and
// flushed once) when that shortens the longest wave: 3 x 8 or 5 x 4 rounds here.  Every unit pays its setup, the LDS sum and a share of
// the flush (~ a quarter of a round); more than 4 groups would put the groups of one macro-tile into different workgroups, which then
// flush the same nodes at the same time (measured: 8 / 16 groups 0.54 / 0.96 ms against 0.30).
#ifndef NIC_RG_MAX
#define NIC_RG_MAX 2
#endif
#ifndef NIC_RG_SEG0
#define NIC_RG_SEG0 3        // segment 0 may use 8 groups (a macro-tile's groups then sit in two workgroups and flush twice)
#endif
#ifndef NIC_RG_FILL
#define NIC_RG_FILL 6
#endif
// workgroups a launch may use: one (two: inference) per CU, or nic_path_desc.max_workgroups when the caller shares the chip
int64_t wg_cap(int per_cu, int max_wg) {
    int64_t cap = (int64_t)cu_count() * per_cu / 8 * 8;
    if (max_wg > 0) {
        const int64_t lim = max_wg / 8 * 8 < 8 ? 8 : max_wg / 8 * 8;
        if (lim < cap) cap = lim;
    }
    return cap;
}
// fused_train16 (two_seg): the macro-tiles that fill every wave a whole number of times run as whole units (segment 0), only the
// remainder is dealt out in groups (segment 1, up to 16 = single rounds: 8 crops 188 -> 174 us against 8 groups, although a macro-tile's
// groups then sit in two workgroups and flush twice).  Setup + LDS sum + flush of a
// unit cost that kernel 0.8 of a round (stamps: 7.6 K + 9 K cycles against 20.4 K).
#ifndef NIC_RG_MAX_SEG
#define NIC_RG_MAX_SEG 4
#endif
#ifndef NIC_Q16_RG0
#define NIC_Q16_RG0 0        // measured: 8-group segment 0 (12 rounds + 2 unit overheads instead of 3) changes nothing on the 3D sweep step (209.5 against 205.5 us, 155.8 against 154)
#endif
void balance_units(FusedParams& p, int per_cu, int waves_per_wg = 4, bool two_seg = false, bool allow_seg0 = false, int rg0_max = NIC_RG_MAX, int q0_max = 0) {
    const int64_t waves = wg_cap(per_cu, p.d.max_workgroups) * waves_per_wg;
    const int rounds = p.niter * p.passes;                // niter is a power of two: the groups stay equal with any number of passes
    p.rg_log2 = 0;
    p.rg0_log2 = 0;
    p.seg_split = 0;
    double best = 0.0;
    if (two_seg) {
        // fused_train16: segment 0 = whole units (q0_max = 0).  fused_q16 (q0_max = NIC_Q16_RG0: 8 groups - a macro-tile's groups then fill one
        // workgroup round and flush twice, once per half): segment 0 may be dealt out in groups as well, when that fills every wave once more -
        // the reference's 3D sweep step (8 x 46 packed macro-tiles x 64 rounds on 2 048 waves) runs as 256 macro-tiles in 8-round units + 112
        // in 4-round units = 12 rounds and TWO unit overheads per wave instead of three 4-round units (setup + gather + flush of a unit cost
        // those kernels 1.3 - 1.5 rounds: stamps 9 - 13 K + 13 - 14 K cycles against 17 K).
        const double oh = q0_max > 0 ? 1.3 : 0.8;
        auto rest_cost = [&](int64_t rest, int& rg_best) {
            double bc = 0.0;
            rg_best = 0;
            for (int rg = 0; rg <= NIC_RG_MAX_SEG && (p.niter >> rg) >= 1 && rest > 0; ++rg) {
                const double cost = (double)(((rest << rg) + waves - 1) / waves) * ((double)(rounds >> rg) + oh);
                if (rg == 0 || cost < bc * 0.98) { bc = cost; rg_best = rg; }
            }
            return bc;
        };
        for (int rg0 = 0; rg0 <= q0_max && (p.niter >> rg0) >= 1; ++rg0) {
            const int64_t full = (p.n_tiles << rg0) / waves;
            if (rg0 > 0 && full == 0) continue;
            const int64_t t0 = (full * waves) >> rg0, rest = p.n_tiles - t0;
            int r1 = 0;
            const double cost = (double)full * ((double)(rounds >> rg0) + oh) + rest_cost(rest, r1);
            if (rg0 == 0 || cost < best * 0.98) { best = cost; p.seg_split = t0; p.rg0_log2 = rg0; p.rg_log2 = r1; }
        }
        return;
    }
    // The 32-sample kernels (mlpn_two_seg: fused_mlpn keeps one segment): segment 0 = the macro-tiles that fill every wave a whole number of
    // times when dealt out in 2^rg0 groups (rg0 <= NIC_RG_MAX: the groups of a macro-tile stay in one workgroup), segment 1 = the remainder in
    // 2^rg1 groups, where launches that cannot give every wave one unit may go on splitting (up to 64 groups) as long as the units still fit
    // one step: a 64^3 volume is 128 macro-tiles x 64 rounds - 8 groups fill the 1 024 waves once with 8-round units; the 8 x 23 packed
    // macro-tiles of the reference's 3D sweeps run as one step of 8-round units (128 macro-tiles) + one of 4-round units (56).
    auto rest_cost = [&](int64_t rest, int& rg_best) {
        double bc = 0.0;
        rg_best = 0;
        for (int rg = 0; (rg <= NIC_RG_MAX || (rg <= NIC_RG_FILL && (rest << rg) <= waves)) && (p.niter >> rg) >= 1; ++rg) {
            const double cost = (double)(((rest << rg) + waves - 1) / waves) * ((double)(rounds >> rg) + 0.25);
            if (rg == 0 || cost < bc * 0.98) { bc = cost; rg_best = rg; }          // finer only when it pays at least 2 %
        }
        return bc;
    };
    int rg1 = 0;
    best = rest_cost(p.n_tiles, rg1);                                             // one segment
    p.rg_log2 = rg1;
    p.rg0_log2 = 0;
    if (!allow_seg0) return;
    for (int rg0 = 0; rg0 <= rg0_max && (p.niter >> rg0) >= 1; ++rg0) {
        const int64_t full = (p.n_tiles << rg0) / waves;
        if (full == 0) continue;
        const int64_t t0 = (full * waves) >> rg0, rest = p.n_tiles - t0;
        int r1 = 0;
        const double cost = (double)full * ((double)(rounds >> rg0) + 0.25) + (rest > 0 ? rest_cost(rest, r1) : 0.0);
        if (cost < best * 0.98) { best = cost; p.seg_split = t0; p.rg0_log2 = rg0; p.rg_log2 = r1; }
    }
}

int grid_for(int64_t n_tiles, int per_cu, int waves_per_wg = 4, int max_wg = 0) {
    int64_t want = (n_tiles + waves_per_wg - 1) / waves_per_wg;                 // one tile per wave
    want = (want + 7) / 8 * 8;
    const int64_t cap = wg_cap(per_cu, max_wg);
    if (want > cap) want = cap;
    if (want < 8) want = 8;
    return (int)want;
}

void fill_encode(FusedParams& p, const nic_path_desc* d, const FusedInfo& fi, const float* g0, const float* g1, const int32_t* origins,
                 const float* noise, bool allow_packed = false) {
    p.d = *d;
    p.g0.p = g0; p.g0.nx = d->g0_nodes[0]; p.g0.ny = d->g0_nodes[1]; p.g0.nz = d->dim == 3 ? d->g0_nodes[2] : 1;
    p.g0.plane = (int64_t)p.g0.nx * p.g0.ny * p.g0.nz;
    p.g1.p = g1; p.g1.nx = d->g1_nodes[0]; p.g1.ny = d->g1_nodes[1]; p.g1.nz = d->dim == 3 ? d->g1_nodes[2] : 1;
    p.g1.plane = (int64_t)p.g1.nx * p.g1.ny * p.g1.nz;
    p.origins = origins;
    if (d->flags & NIC_FLAG_ORIGINS_HOST) {                 // host values: by value in the kernel arguments (check_geometry bounds num_crops)
        p.origins = nullptr;
        for (int i = 0; i < d->num_crops * d->dim && i < NIC_ORIGINS_INLINE_MAX * 3; ++i) p.org_inl[i] = origins[i];
    }
    const int ez = d->dim == 3 ? d->extent[2] : 1;
    p.d.extent[2] = ez;
    p.n_per_crop = (int64_t)d->extent[0] * d->extent[1] * ez;
    p.passes = d->passes > 1 ? d->passes : 1;
    p.n_total = p.n_per_crop * d->num_crops * p.passes;
    // macro-tiles of fi.tx x fi.ty x fi.tz CELL BLOCKS; a cell block = m samples per axis, m = 1 / step_number (1 when the step
    // is >= 1).  The crop origin is not known on the host, so the block count per axis is the unaligned upper bound unless the
    // caller vouches for cell-aligned origins (NIC_FLAG_ORIGINS_ALIGNED); blocks (lanes) that fall outside the crop are masked
    // in the kernel.
    p.lm = d->log2_step < 0 ? -d->log2_step : 0;
    const int m = 1 << p.lm;
    p.niter = 1;
    for (int a = 0; a < d->dim; ++a) p.niter *= m;
    const bool aligned = (d->flags & NIC_FLAG_ORIGINS_ALIGNED) != 0;
    auto blocks = [&](int extent) { return (extent + m - 1) / m + ((m > 1 && !aligned) ? 1 : 0); };
    const int bx = blocks(d->extent[0]), by = blocks(d->extent[1]);
    const int ty = (by + fi.ty - 1) / fi.ty, tz = d->dim == 3 ? (blocks(ez) + fi.tz - 1) / fi.tz : 1;
    // the remainder column along x: as regular tiles when it holds more than half a tile's blocks, otherwise as edge tiles of
    // 2^lw x (32 >> lw) blocks (see FusedParams)
    p.full_x = bx / fi.tx;
    const int rem = bx - p.full_x * fi.tx;
    p.edge_lw = -1;
    int64_t edge_tiles = 0;
    if (rem > fi.tx / 2) {
        p.full_x += 1;
    } else if (rem > 0) {
        p.edge_lw = 0;
        while ((1 << p.edge_lw) < rem) ++p.edge_lw;
        const int eh = (fi.tx * fi.ty) >> p.edge_lw;                   // blocks along y per edge tile
        edge_tiles = (int64_t)((by + eh - 1) / eh) * tz;
    }
    p.tiles_y = ty; p.tiles_z = tz;
    p.tiles_main = (int64_t)p.full_x * ty * tz;
    p.tiles_per_crop = p.tiles_main + edge_tiles;
    // packed tiling (the 32-sample kernels): 32 consecutive blocks of the crop's block list per macro-tile, when that needs at least
    // 15 % fewer macro-tiles than the 16 x 2 wave blocks (block counts far from multiples of 16 and 2: small unaligned crops)
    p.pk_nc = 0;
    {
        const int bz = d->dim == 3 ? blocks(ez) : 1;
        const int tsz = fi.tx * fi.ty * fi.tz;                                   // 32 blocks (fused_kernel) or 16 (fused_q16_kernel)
        const int64_t nc = (int64_t)bx * by * bz, pt = (nc + tsz - 1) / tsz;
        if (allow_packed && (tsz == 32 || tsz == 16) && nc < ((int64_t)1 << 24) && pt * 115 <= p.tiles_per_crop * 100) {
            p.pk_bx = bx; p.pk_by = by; p.pk_nc = (int)nc;
            p.edge_lw = -1;
            p.tiles_main = pt;
            p.tiles_per_crop = pt;
        }
    }
    p.n_tiles = p.tiles_per_crop * d->num_crops;
    p.rg_log2 = 0;
    p.noise.mode = d->noise_mode;
    p.noise.tensor = noise;
    p.noise.k0 = (uint32_t)d->noise_seed; p.noise.k1 = (uint32_t)(d->noise_seed >> 32);
    p.noise.off_lo = (uint32_t)d->noise_offset; p.noise.off_hi = (uint32_t)(d->noise_offset >> 32);
    p.noise.scale = ldexpf(1.0f, -d->num_bits);
    p.grad_scale = 2.0f * d->loss_scale;
    p.f16 = 0;
    p.dz_scale = p.dz_unscale = 1.0f;
}
// NIC_FLAG_FP16: the loss scale of dZ (FusedParams::dz_scale).  auto_from_loss: 2^k with 2 loss_scale 2^k in [4, 8) - the output layer's dZ is then at most
// ~ 2 and everything behind it stays O(1): far from both ends of the half range
void set_f16(FusedParams& p, const nic_path_desc* d, bool auto_from_loss) {
    p.f16 = 1;
    int k = d->dz_scale_log2;
    if (k == 0 && auto_from_loss && p.grad_scale > 0.f) {
        int e = 0;
        frexpf(p.grad_scale, &e);                                     // grad_scale = m 2^e, m in [0.5, 1)
        k = 3 - e;
        if (k < 0) k = 0;
    }
    if (k > 60) k = 60;
    if (k < -60) k = -60;
    p.dz_scale = ldexpf(1.0f, k);
    p.dz_unscale = ldexpf(1.0f, -k);
}
int mlp_depth(const nic_mlp* m) { return m->n_linear == 0 ? 3 : m->n_linear; }
bool depth_unsupported(const nic_mlp* m) { const int n = mlp_depth(m); return n != 3 && n != 5 && n >= 2 && n <= NIC_MAX_LINEAR; }
void fill_mlp(FusedParams& p, const nic_mlp* m) {
    for (int i = 0; i < NIC_MAX_LINEAR; ++i) { p.W[i] = m->w[i]; p.b[i] = m->b[i]; }
    p.n_linear = mlp_depth(m);
}
bool mlp_ok(const nic_mlp* m) {
    if (!m) return false;
    const int n = mlp_depth(m);
    if (n != 3 && n != 5) return false;
    for (int i = 0; i < n; ++i) if (!m->w[i] || !m->b[i]) return false;
    return true;
}
int grid_kind_of(const nic_path_desc* d) { return (d->flags & NIC_FLAG_GRID_BF16) ? 1 : ((d->flags & NIC_FLAG_GRID_FP16) ? 2 : 0); }
// the depth-generic kernels serve n_linear = 5, n_linear = 3 on request (NIC_FLAG_MLPN), and the decode from 16-bit grids: 2D layouts,
// split-bf16 products
int use_mlpn(int layout, const nic_path_desc* d, const nic_mlp* m, bool grid_u8, bool& yes, bool inference = false) {
    const int n = mlp_depth(m);
    if ((d->flags & NIC_FLAG_GRID_BF16) && (d->flags & NIC_FLAG_GRID_FP16)) return NIC_E_ARG;
    yes = n == 5 || (d->flags & NIC_FLAG_MLPN) != 0 || (inference && grid_kind_of(d) != 0);
    if (grid_kind_of(d) != 0 && ((layout != 1 && layout != 2) || !(d->flags & NIC_FLAG_SPLIT_BF16) || (d->flags & NIC_FLAG_SPLIT_TILE32) || grid_u8))
        return NIC_E_UNSUPPORTED;                                  // 16-bit grid storage: the quarter-layout 2D kernels only
    if (!yes) return NIC_OK;
    if ((layout != 1 && layout != 2) || !(d->flags & NIC_FLAG_SPLIT_BF16) || grid_u8) return NIC_E_UNSUPPORTED;
    return NIC_OK;
}
FusedInfo info_mlpn(int n_linear) { return FusedInfo{0, mlpn_record_floats(n_linear), 16, 1, 1, 73, 4}; }
// plain-bf16 kernels (fused_q16.hpp): every layout, 3 or 5 Linear layers
int q16_rec(int layout, int n_linear) {
    switch (layout) {
        case 1: return q16_record_floats<1>(n_linear);
        case 2: return q16_record_floats<2>(n_linear);
        case 3: return q16_record_floats<3>(n_linear);
        default: return q16_record_floats<4>(n_linear);
    }
}
int q16_rec_cp(int layout, int c, int pch) {
#define X(L, C, P) if (layout == L && c == C && pch == P) return q16_record_floats_cp<L, C, P>();
    NIC_CP_LIST(X)
#undef X
    return 0;
}
FusedInfo info_q16(int layout, int n_linear, const nic_path_desc* d = nullptr) {
    const int c = d ? d->channels : kC, pch = d ? d->pe_channels : kP;
    const int cin = layout <= 2 ? 5 * c + 2 * pch + 1 : (layout == 3 ? 9 * c + 19 : 5 * c + 19);
    const bool def = c == kC && pch == kP;
    return FusedInfo{0, def ? q16_rec(layout, n_linear) : q16_rec_cp(layout, c, pch), 16, 1, 1, cin, 8};
}
int launch_q16_any(int layout, int n_linear, int mode, const FusedParams& p, int grid, hipStream_t s) {
    if (!cp_default(&p.d)) {
        if (n_linear != 3) return NIC_E_UNSUPPORTED;
#define X(L, C, P) if (layout == L && p.d.channels == C && p.d.pe_channels == P) return launch_q16_cp<L, C, P>(mode, p, grid, s);
        NIC_CP_LIST(X)
#undef X
        return NIC_E_UNSUPPORTED;
    }
    switch (layout) {
        case 1: return launch_q16<1>(n_linear, mode, p, grid, s);
        case 2: return launch_q16<2>(n_linear, mode, p, grid, s);
        case 3: return launch_q16<3>(n_linear, mode, p, grid, s);
        default: return launch_q16<4>(n_linear, mode, p, grid, s);
    }
}
int reduce_q16_any(int layout, int n_linear, const nic_path_desc* d, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
    if (!cp_default(d)) {
#define X(L, C, P) if (layout == L && d->channels == C && d->pe_channels == P) return reduce_q16_cp<L, C, P>(partials, n_rec, g, loss, loss_scale, s);
        NIC_CP_LIST(X)
#undef X
        return NIC_E_UNSUPPORTED;
    }
    switch (layout) {
        case 1: return reduce_q16<1>(n_linear, partials, n_rec, g, loss, loss_scale, s);
        case 2: return reduce_q16<2>(n_linear, partials, n_rec, g, loss, loss_scale, s);
        case 3: return reduce_q16<3>(n_linear, partials, n_rec, g, loss, loss_scale, s);
        default: return reduce_q16<4>(n_linear, partials, n_rec, g, loss, loss_scale, s);
    }
}
// multi-level layouts (fused_q16.hpp::QML): (levels, C, n_linear) with P = 6, one translation unit each (fused_ml_*.hip)
#define NIC_ML_LIST(X) X(2, 4, 3) X(3, 4, 3) X(5, 4, 3) X(2, 4, 5) X(3, 4, 5) X(2, 12, 3) X(3, 12, 3)
int ml_rec(int lv, int c, int nl) {
#define X(L, C, N) if (lv == L && c == C && nl == N) return ml_record_floats<L, C, N>();
    NIC_ML_LIST(X)
#undef X
    return 0;
}
int launch_ml_any(int lv, int c, int nl, int pe, int mode, const FusedParams& p, int grid, hipStream_t s) {
#define X(L, C, N) if (lv == L && c == C && nl == N) return launch_ml<L, C, N>(pe, mode, p, grid, s);
    NIC_ML_LIST(X)
#undef X
    return NIC_E_UNSUPPORTED;
}
int reduce_ml_any(int lv, int c, int nl, int pe, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
#define X(L, C, N) if (lv == L && c == C && nl == N) return reduce_ml<L, C, N>(pe, partials, n_rec, g, loss, loss_scale, s);
    NIC_ML_LIST(X)
#undef X
    return NIC_E_UNSUPPORTED;
}
FusedParams zero_params() {
    FusedParams p;
    ::memset(static_cast<void*>(&p), 0, sizeof(p));
    p.passes = 1;
    return p;
}

int fused_train(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, const nic_mlp* mlp, const float* noise,
                const float* target, const float* dy, float* y, float* loss, float* g0_grad, float* g1_grad, const nic_mlp_grads* grads,
                void* workspace, size_t workspace_bytes, void* stream, const nic_target_image* img = nullptr, const int64_t* step_dev = nullptr) {
    const int layout = pick_layout(d);
    if (layout < 0) return layout;
    if (mlp && depth_unsupported(mlp)) return NIC_E_UNSUPPORTED;      // the fused kernels exist for 3 and 5 Linear layers
    int rc = check_geometry(d, true);
    if (rc) return rc;
    if (!g0 || !g1 || !origins || !mlp_ok(mlp) || !g0_grad || !g1_grad || !grads || !workspace) return NIC_E_NULL;
    if ((target != nullptr) + (dy != nullptr) + (img != nullptr) != 1) return NIC_E_ARG;
    if (img) {
        if (!img->data) return NIC_E_NULL;
        for (int a = 0; a < d->dim; ++a)
            if (img->size[a] < d->extent[a]) return NIC_E_SHAPE;
        if (img->is_u8 && !(img->den > 0.f)) return NIC_E_ARG;
    }
    if (d->noise_mode == NIC_NOISE_TENSOR && !noise) return NIC_E_NULL;
    const bool f16 = (d->flags & NIC_FLAG_FP16) != 0;                  // .. on IEEE half operands
    const bool q16 = (d->flags & NIC_FLAG_BF16) != 0 || f16;           // plain 16-bit products: every layout on the quarter kernels
    if (q16 && (d->flags & NIC_FLAG_GRID_BF16) && (d->flags & NIC_FLAG_GRID_FP16)) return NIC_E_ARG;
    if (f16 && !cp_default(d)) return NIC_E_UNSUPPORTED;
    bool mlpn = false;
    if (!q16) {
        rc = use_mlpn(layout, d, mlp, false, mlpn);
        if (rc) return rc;
    }
    const bool t16 = !q16 && !mlpn && use_t16(layout, d);
    if (q16 && !cp_default(d) && mlp_depth(mlp) != 3) return NIC_E_UNSUPPORTED;
    const FusedInfo fi = q16 ? info_q16(layout, mlp_depth(mlp), d) : (mlpn ? info_mlpn(mlp_depth(mlp)) : (t16 ? info_t16() : info_of(layout)));
    FusedParams p = zero_params();
    fill_encode(p, d, fi, g0, g1, origins, noise, q16 || (!mlpn && !t16));
    fill_mlp(p, mlp);
    p.g0_grad = g0_grad; p.g1_grad = g1_grad;
    p.target = target; p.dy = dy; p.y = y;
    if (img) {
        // [3, S0, S1(, S2)] contiguous: the last sample axis is the unit-stride one
        const int64_t s2 = d->dim == 3 ? img->size[2] : 1;
        p.timg = img->data;
        p.timg_s[2] = d->dim == 3 ? 1 : 0;
        p.timg_s[1] = s2;
        p.timg_s[0] = (int64_t)img->size[1] * s2;
        p.timg_cs = (int64_t)img->size[0] * p.timg_s[0];
        if (img->is_u8 < 0 || img->is_u8 > 2) return NIC_E_ARG;
        p.timg_u8 = img->is_u8;
        if (img->is_u8 == 2 && (int64_t)img->size[0] * p.timg_s[0] >= ((int64_t)1 << 31)) return NIC_E_UNSUPPORTED;   // RGBX pixels are addressed in 32 bits
        p.timg_den = img->is_u8 ? img->den : 1.0f;
        p.timg_rcp = 1.0f / p.timg_den;
    }
    p.grid_kind = grid_kind_of(d);
    if (f16) set_f16(p, d, target != nullptr || img != nullptr);
    if (p.grid_kind != 0 && !(mlpn || t16 || q16)) return NIC_E_UNSUPPORTED;
    if (step_dev != nullptr && !(t16 || q16)) return NIC_E_UNSUPPORTED;                // the device-side step: the two-waves-per-SIMD kernels
    if (step_dev != nullptr && (d->flags & NIC_FLAG_ORIGINS_HOST)) return NIC_E_ARG;     // .. reads the origins the device sampler wrote
    p.step_dev = step_dev;
    p.partials = (float*)workspace;
    const int wpw = (t16 || q16) ? 8 : 4;                     // waves per workgroup = work units per workgroup round
    static const bool two_seg = []() { const char* e = getenv("NIC_TWO_SEG"); return !(e && e[0] == '0'); }();   // NIC_TWO_SEG=0: one segment (A/B timing)
    // (8 groups in segment 0 - a macro-tile's groups in two workgroups, two flushes: the reference's 3D sweep shape 0.368 -> 0.332 ms with
    //  method 4; method 3, twice the sums per lane, lost 5 % until its flush pre-added neighbouring sums and gains 6 % since)
    balance_units(p, 1, wpw, (t16 || q16) && two_seg, !t16 && !q16 && !mlpn && two_seg, NIC_RG_SEG0, q16 ? NIC_Q16_RG0 : 0);
    p.preadd_y = p.n_tiles <= (int64_t)4 * wpw * cu_count() ? 1 : 0;
    const int64_t units0 = p.seg_split << p.rg0_log2, units1 = (p.n_tiles - p.seg_split) << p.rg_log2;
    const int64_t units_max = units0 > units1 ? units0 : units1;
    if (units_max >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;                    // the kernels count work units in 32 bits
    const int grid = grid_for(units_max, 1, wpw, d->max_workgroups);
    const int n_rec = grid;                                   // one record per workgroup
    if (workspace_bytes < (size_t)n_rec * fi.rec * sizeof(float)) return NIC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int mode = img ? MODE_TRAIN_IMG : (target ? MODE_TRAIN_MSE : MODE_TRAIN_DY);
    TailScope tail;                                           // nic_path_desc.tail: Adam rides on the reduce launch below
    rc = tail.open(d, grads, step_dev);
    if (rc) return rc;
    if (q16) {
        rc = launch_q16_any(layout, p.n_linear, mode, p, grid, s);
        if (rc) return rc;
        mark_kernel_end(s);
        return reduce_q16_any(layout, p.n_linear, d, p.partials, n_rec, *grads, (target || img) ? loss : nullptr, d->loss_scale, s);
    }
    if (mlpn) {
        rc = launch_mlpn(layout, p.n_linear, mode, p, grid, s);
        if (rc) return rc;
        mark_kernel_end(s);
        return launch_reducen(layout, p.n_linear, p.partials, n_rec, *grads, (target || img) ? loss : nullptr, d->loss_scale, s);
    }
    if (t16) {
        rc = launch_train16(layout, mode, p, grid, s);
        if (rc) return rc;
        mark_kernel_end(s);
        return launch_reduce16(layout, p.partials, n_rec, *grads, (target || img) ? loss : nullptr, d->loss_scale, s);
    }
    rc = launch(layout, SRC_ENCODE, mode, p, grid, s);
    if (rc) return rc;
    mark_kernel_end(s);
    return reduce(layout, p.partials, n_rec, *grads, (target || img) ? loss : nullptr, d->loss_scale, s);
}

// geometry, pairs and decoder of a multi-level launch -> FusedParams (regular 16 x 1 cell tiles of pair 0, whole macro-tiles: the flush of the
// multi-level kernels sums runs of lanes, which the edge-tile / packed / grouped schedules of the single-pair kernels would break up)
int fill_ml(FusedParams& p, const nic_path_desc* d, const nic_ml_pairs* pr, const int32_t* origins, const nic_mlp* mlp, const float* noise, bool training,
            int& rec) {
    if (!d || !pr || !origins || !mlp) return NIC_E_NULL;
    if (d->dim != 2 || d->method != 1 || d->hidden != kH || d->pe_channels != kP) return NIC_E_UNSUPPORTED;
    if (d->pe_mode != NIC_PE_TRIANGULAR && d->pe_mode != NIC_PE_SINUSOIDAL) return NIC_E_UNSUPPORTED;
    if (pr->levels < 2 || pr->levels > NIC_ML_MAX_LEVELS) return NIC_E_UNSUPPORTED;
    if (!mlp_ok(mlp)) return depth_unsupported(mlp) ? NIC_E_UNSUPPORTED : NIC_E_NULL;
    rec = ml_rec(pr->levels, d->channels, mlp_depth(mlp));
    if (rec == 0) return NIC_E_UNSUPPORTED;
    if (d->flags & (NIC_FLAG_GRID_BF16 | NIC_FLAG_GRID_FP16)) return NIC_E_UNSUPPORTED;       // fp32 grids
    nic_path_desc d2 = *d;
    for (int a = 0; a < 2; ++a) { d2.g0_nodes[a] = pr->g0_nodes[0][a]; d2.g1_nodes[a] = pr->g1_nodes[0][a]; }
    int rc = check_geometry(&d2, training);
    if (rc) return rc;
    if (d->log2_step - 2 * (pr->levels - 1) < -16) return NIC_E_ARG;
    if (d->noise_mode == NIC_NOISE_TENSOR && !noise) return NIC_E_NULL;
    for (int l = 0; l < pr->levels; ++l) {
        if (!pr->g0[l] || !pr->g1[l] || (training && (!pr->g0_grad[l] || !pr->g1_grad[l]))) return NIC_E_NULL;
        for (int a = 0; a < 2; ++a)
            if (pr->g0_nodes[l][a] < 2 || pr->g1_nodes[l][a] < 2) return NIC_E_SHAPE;
        if ((int64_t)pr->g0_nodes[l][0] * pr->g0_nodes[l][1] >= (int64_t)1 << 30 || 10 * (int64_t)pr->g1_nodes[l][0] * pr->g1_nodes[l][1] >= (int64_t)1 << 30)
            return NIC_E_UNSUPPORTED;
    }
    const int cin = pr->levels * (5 * d->channels + 2 * d->pe_channels) + 1;
    fill_encode(p, &d2, FusedInfo{0, rec, 16, 1, 1, cin, 8}, pr->g0[0], pr->g1[0], origins, noise, false);
    if (p.edge_lw >= 0) {                                              // the remainder column along x as one more column of regular tiles
        p.full_x += 1;
        p.edge_lw = -1;
        p.tiles_main = (int64_t)p.full_x * p.tiles_y * p.tiles_z;
        p.tiles_per_crop = p.tiles_main;
        p.n_tiles = p.tiles_per_crop * d->num_crops;
    }
    fill_mlp(p, mlp);
    for (int l = 0; l < pr->levels; ++l) {
        MlPair& m = p.ml[l];
        m.g0.p = pr->g0[l]; m.g0.nx = pr->g0_nodes[l][0]; m.g0.ny = pr->g0_nodes[l][1]; m.g0.nz = 1; m.g0.plane = (int64_t)m.g0.nx * m.g0.ny;
        m.g1.p = pr->g1[l]; m.g1.nx = pr->g1_nodes[l][0]; m.g1.ny = pr->g1_nodes[l][1]; m.g1.nz = 1; m.g1.plane = (int64_t)m.g1.nx * m.g1.ny;
        m.g0_grad = training ? pr->g0_grad[l] : nullptr;
        m.g1_grad = training ? pr->g1_grad[l] : nullptr;
    }
    p.g0_grad = p.ml[0].g0_grad; p.g1_grad = p.ml[0].g1_grad;
    p.grid_kind = 0;
    return NIC_OK;
}

}  // namespace

extern "C" {

int nic_mark_kernel_end(void* hip_event) {
    g_kernel_end = (hipEvent_t)hip_event;
    return NIC_OK;
}

int nic_fused_ml_forward_backward(const nic_path_desc* d, const nic_ml_pairs* pairs, const int32_t* origins, const nic_mlp* mlp, const float* noise,
                                  const float* target, float* y, float* loss, const nic_mlp_grads* grads, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (!target || !loss || !grads || !workspace) return NIC_E_NULL;
    FusedParams p = zero_params();
    int rec = 0;
    int rc = fill_ml(p, d, pairs, origins, mlp, noise, true, rec);
    if (rc) return rc;
    p.target = target; p.y = y;
    p.partials = (float*)workspace;
    p.rg_log2 = 0; p.rg0_log2 = 0; p.seg_split = 0; p.preadd_y = 0;    // whole macro-tiles, one segment
    if (p.n_tiles >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;
    const int grid = grid_for(p.n_tiles, 1, 8, d->max_workgroups);
    if (workspace_bytes < (size_t)grid * rec * sizeof(float)) return NIC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    TailScope tail;
    rc = tail.open(d, grads, nullptr);
    if (rc) return rc;
    rc = launch_ml_any(pairs->levels, d->channels, p.n_linear, d->pe_mode, MODE_TRAIN_MSE, p, grid, s);
    if (rc) return rc;
    mark_kernel_end(s);
    return reduce_ml_any(pairs->levels, d->channels, p.n_linear, d->pe_mode, p.partials, grid, *grads, loss, d->loss_scale, s);
}

int nic_fused_ml_forward(const nic_path_desc* d, const nic_ml_pairs* pairs, const int32_t* origins, const nic_mlp* mlp, float* y, void* stream) {
    if (!y) return NIC_E_NULL;
    if (d && d->noise_mode != NIC_NOISE_NONE) return NIC_E_ARG;        // decoding never adds noise (image_compression.py:307-346)
    FusedParams p = zero_params();
    int rec = 0;
    int rc = fill_ml(p, d, pairs, origins, mlp, nullptr, false, rec);
    if (rc) return rc;
    p.y = y;
    balance_units(p, 1, 8);
    if ((p.n_tiles << p.rg_log2) >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;
    p.seg_split = 0;
    return launch_ml_any(pairs->levels, d->channels, p.n_linear, d->pe_mode, MODE_INFER, p, grid_for(p.n_tiles << p.rg_log2, 1, 8, d->max_workgroups),
                         (hipStream_t)stream);
}

size_t nic_workspace_bytes(const nic_path_desc* d) {
    int rec = fused_info<3>().rec;                                   // the largest record
    if (d) {
        const int layout = pick_layout(d);
        if (layout > 0) rec = info_of(layout).rec;
    }
    if (train16_record_floats() > rec) rec = train16_record_floats();
    if (mlpn_record_floats(5) > rec) rec = mlpn_record_floats(5);
    for (int l = 1; l <= 4; ++l)
        if (q16_rec(l, 5) > rec) rec = q16_rec(l, 5);
#define X(L, C, P) if (q16_record_floats_cp<L, C, P>() > rec) rec = q16_record_floats_cp<L, C, P>();
    NIC_CP_LIST(X)
#undef X
#define X(L, C, N) if (ml_record_floats<L, C, N>() > rec) rec = ml_record_floats<L, C, N>();
    NIC_ML_LIST(X)
#undef X
    const size_t fused = (size_t)(cu_count() / 8 * 8) * rec * sizeof(float) + (1u << 20);   // one record per workgroup, at most one workgroup per CU (+ 1 MiB: diagnostic builds)
    const size_t psnr = 1024 * sizeof(double);
    return fused > psnr ? fused : psnr;
}

int nic_fused_forward(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, const nic_mlp* mlp,
                      const float* noise, float* y, void* stream) {
    const int layout = pick_layout(d);
    if (layout < 0) return layout;
    if (mlp && depth_unsupported(mlp)) return NIC_E_UNSUPPORTED;      // the fused kernels exist for 3 and 5 Linear layers
    int rc = check_geometry(d);
    if (rc) return rc;
    if (!g0 || !g1 || !origins || !mlp_ok(mlp) || !y) return NIC_E_NULL;
    if (d->noise_mode == NIC_NOISE_TENSOR && !noise) return NIC_E_NULL;
    if (d->flags & (NIC_FLAG_BF16 | NIC_FLAG_FP16)) {
        // plain 16-bit products: the forward pass of fused_q16_kernel (every layout and channel count those kernels serve, 16-bit grids included)
        if ((d->flags & NIC_FLAG_GRID_BF16) && (d->flags & NIC_FLAG_GRID_FP16)) return NIC_E_ARG;
        if (!cp_default(d) && (mlp_depth(mlp) != 3 || (d->flags & NIC_FLAG_FP16))) return NIC_E_UNSUPPORTED;
        FusedParams p = zero_params();
        fill_encode(p, d, info_q16(layout, mlp_depth(mlp), d), g0, g1, origins, noise, true);
        if (d->flags & NIC_FLAG_FP16) set_f16(p, d, false);
        fill_mlp(p, mlp);
        p.y = y;
        p.grid_kind = grid_kind_of(d);
        balance_units(p, 1, 8);
        if ((p.n_tiles << p.rg_log2) >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;
        p.seg_split = 0;                                              // one segment: every macro-tile in 2^rg_log2 groups
        return launch_q16_any(layout, p.n_linear, MODE_INFER, p, grid_for(p.n_tiles << p.rg_log2, 1, 8, d->max_workgroups), (hipStream_t)stream);
    }
    bool mlpn = false;
    rc = use_mlpn(layout, d, mlp, false, mlpn, true);
    if (rc) return rc;
    const FusedInfo fi = mlpn ? info_mlpn(mlp_depth(mlp)) : info_of(layout);
    FusedParams p = zero_params();
    fill_encode(p, d, fi, g0, g1, origins, noise, !mlpn);
    fill_mlp(p, mlp);
    p.y = y;
    p.grid_kind = grid_kind_of(d);
    if (mlpn) {                                                       // one workgroup per CU (84 KB of weight images)
        balance_units(p, 1, 4);
        if ((p.n_tiles << p.rg_log2) >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;     // the kernels count work units in 32 bits
        return launch_mlpn(layout, p.n_linear, MODE_INFER, p, grid_for(p.n_tiles << p.rg_log2, 1, 4, d->max_workgroups), (hipStream_t)stream);
    }
    balance_units(p, 2);
    if ((p.n_tiles << p.rg_log2) >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;
    return launch(layout, SRC_ENCODE, MODE_INFER, p, grid_for(p.n_tiles << p.rg_log2, 2, 4, d->max_workgroups), (hipStream_t)stream);
}

int nic_fused_forward_u8(const nic_path_desc* d, const uint8_t* g0_u8, const uint8_t* g1_u8, const int32_t* origins, const nic_mlp* mlp,
                         float* y, uint8_t* y_u8, void* stream) {
    const int layout = pick_layout(d);
    if (layout < 0) return layout;
    // the stored-codec kernels are the 32-sample / depth-generic families, built for C = 12, P = 6 (with NIC_FLAG_BF16 pick_layout also answers for
    // the plain-bf16 kernels' other widths, which have no uint8 grid kind): the caller decodes through fp_load + the layer-wise route instead
    if (!cp_default(d)) return NIC_E_UNSUPPORTED;
    if (mlp && depth_unsupported(mlp)) return NIC_E_UNSUPPORTED;      // the fused kernels exist for 3 and 5 Linear layers
    int rc = check_geometry(d);
    if (rc) return rc;
    if (!g0_u8 || !g1_u8 || !origins || !mlp_ok(mlp) || (!y && !y_u8)) return NIC_E_NULL;
    if (d->noise_mode != NIC_NOISE_NONE) return NIC_E_ARG;            // decoding never adds noise (image_compression.py:307-346)
    if (d->num_bits < 1 || d->num_bits > 8) return NIC_E_ARG;
    const bool deep = mlp_depth(mlp) != 3;                            // 5-layer decoders: the depth-generic kernel, 2D (grid kind 3 = the codec)
    if (deep && layout != 1 && layout != 2) return NIC_E_UNSUPPORTED;
    const FusedInfo fi = deep ? info_mlpn(mlp_depth(mlp)) : info_of(layout);
    FusedParams p = zero_params();
    fill_encode(p, d, fi, reinterpret_cast<const float*>(g0_u8), reinterpret_cast<const float*>(g1_u8), origins, nullptr, !deep);
    fill_mlp(p, mlp);
    p.dq_sub = (float)((1 << (d->num_bits - 1)) - 1);
    p.dq_den = (float)((1 << d->num_bits) - 1);
    p.dq_rcp = 1.0f / p.dq_den;
    p.y = y; p.y_u8 = y_u8;
    if (deep) {
        p.grid_kind = 3;
        balance_units(p, 1, 4);
        if ((p.n_tiles << p.rg_log2) >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;
        return launch_mlpn(layout, p.n_linear, MODE_INFER, p, grid_for(p.n_tiles << p.rg_log2, 1, 4, d->max_workgroups), (hipStream_t)stream);
    }
    p.grid_u8 = 1;
    balance_units(p, 2);
    if ((p.n_tiles << p.rg_log2) >= ((int64_t)1 << 30)) return NIC_E_UNSUPPORTED;
    return launch(layout, SRC_ENCODE, MODE_INFER, p, grid_for(p.n_tiles << p.rg_log2, 2, 4, d->max_workgroups), (hipStream_t)stream);
}

int nic_fused_forward_backward(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, const nic_mlp* mlp,
                               const float* noise, const float* target, float* y, float* loss, float* g0_grad, float* g1_grad,
                               const nic_mlp_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (!target || !loss) return NIC_E_NULL;
    return fused_train(d, g0, g1, origins, mlp, noise, target, nullptr, y, loss, g0_grad, g1_grad, grads, workspace, workspace_bytes, stream);
}

int nic_fused_forward_backward_img(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, const nic_mlp* mlp,
                                   const float* noise, const nic_target_image* image, float* y, float* loss, float* g0_grad,
                                   float* g1_grad, const nic_mlp_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (!image || !loss) return NIC_E_NULL;
    return fused_train(d, g0, g1, origins, mlp, noise, nullptr, nullptr, y, loss, g0_grad, g1_grad, grads, workspace, workspace_bytes, stream,
                       image);
}

int nic_fused_forward_backward_img_dev(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, const nic_mlp* mlp,
                                       const nic_target_image* image, float* loss, float* g0_grad, float* g1_grad, const nic_mlp_grads* grads,
                                       const int64_t* step_dev, void* workspace, size_t workspace_bytes, void* stream) {
    if (!image || !loss || !step_dev) return NIC_E_NULL;
    if (d && d->noise_mode == NIC_NOISE_TENSOR) return NIC_E_ARG;                      // a captured step draws its noise in the kernel (or none)
    return fused_train(d, g0, g1, origins, mlp, nullptr, nullptr, nullptr, nullptr, loss, g0_grad, g1_grad, grads, workspace, workspace_bytes, stream,
                       image, step_dev);
}

int nic_fused_backward_dy(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, const nic_mlp* mlp,
                          const float* noise, const float* dy, float* g0_grad, float* g1_grad, const nic_mlp_grads* grads,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy) return NIC_E_NULL;
    return fused_train(d, g0, g1, origins, mlp, noise, nullptr, dy, nullptr, nullptr, g0_grad, g1_grad, grads, workspace, workspace_bytes, stream);
}

int nic_decoder_forward(const nic_mlp* mlp, const float* x, int64_t n, int cin, int hidden, float* y, void* stream) {
    if (!mlp_ok(mlp) || !x || !y) return NIC_E_NULL;
    if (hidden != kH || mlp_depth(mlp) != 3) return NIC_E_UNSUPPORTED;
    const int layout = layout_of_cin(cin);
    if (layout < 0) return layout;
    if (n < 0) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    FusedParams p = zero_params();
    fill_mlp(p, mlp);
    p.x = x; p.y = y; p.n_total = n; p.n_tiles = (n + 31) / 32;
    return launch(layout, SRC_MEMORY, MODE_INFER, p, grid_for(p.n_tiles, 2), (hipStream_t)stream);
}

int nic_decoder_backward(const nic_mlp* mlp, const float* x, const float* dy, int64_t n, int cin, int hidden, float* dx,
                         const nic_mlp_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    if (!mlp_ok(mlp) || !x || !dy || !grads || !workspace) return NIC_E_NULL;
    if (hidden != kH || mlp_depth(mlp) != 3) return NIC_E_UNSUPPORTED;
    const int layout = layout_of_cin(cin);
    if (layout < 0) return layout;
    if (n <= 0) return NIC_E_ARG;
    const FusedInfo fi = info_of(layout);
    FusedParams p = zero_params();
    fill_mlp(p, mlp);
    p.x = x; p.dy = dy; p.dx = dx; p.n_total = n; p.n_tiles = (n + 31) / 32;
    p.partials = (float*)workspace;
    const int grid = grid_for(p.n_tiles, 1);
    const int n_rec = grid * 4 / fi.waves_per_rec;
    if (workspace_bytes < (size_t)n_rec * fi.rec * sizeof(float)) return NIC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int rc = launch(layout, SRC_MEMORY, MODE_TRAIN_DY, p, grid, s);
    if (rc) return rc;
    return reduce(layout, p.partials, n_rec, *grads, nullptr, 0.f, s);
}

}  // extern "C"
