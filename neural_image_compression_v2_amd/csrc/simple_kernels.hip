// Element-wise / gather kernels of the path that are not matrix shaped: the reference-layout encode
// (one thread per sample), positional encodings on explicit coordinates, the quantisers and uint8
// codec, PSNR and the fused Adam + clamp step.  All HBM-bound streaming kernels: coalesced accesses,
// grid-stride loops capped at 2048 blocks.
#include "nic_adam.hpp"
#include <math.h>

namespace nic {

// ---------------------------------------------------------------------------------------------------
// encode, reference channel order.  SPLIT = false: out[n, Cin] rows (create_decoder_input_*,
// image_compression.py:71-167).  SPLIT = true: out[row, n] with the G1 corners kept separate
// (create_g0_g1*, fp_def.py:115-223).
// ---------------------------------------------------------------------------------------------------
struct EncodeParams {
    nic_path_desc d;
    GridView g0, g1;
    const int32_t* origins;
    float* out;
    int cin;
    int64_t n_per_crop, n_total;
};

// memory safety: a corner index never leaves the grid (same clamp as the fused kernels)
__device__ __forceinline__ int cl(int i, int nodes) { return i < 0 ? 0 : (i > nodes - 2 ? nodes - 2 : i); }

template <int DIM, bool SPLIT>
__global__ void __launch_bounds__(256) encode_kernel(EncodeParams p) {
    const nic_path_desc& d = p.d;
    const int C = d.channels, P = d.pe_channels;
    const int K0 = (DIM == 2 || d.method == 4) ? 4 : 8;
    const int K1 = DIM == 2 ? 4 : 8;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < p.n_total; n += (int64_t)gridDim.x * blockDim.x) {
        const int crop = (int)(n / p.n_per_crop);
        int64_t r = n - (int64_t)crop * p.n_per_crop;
        int idx[3] = {0, 0, 0};
        if (DIM == 3) { idx[2] = (int)(r % d.extent[2]); r /= d.extent[2]; }
        idx[1] = (int)(r % d.extent[1]);
        idx[0] = (int)(r / d.extent[1]);
        Axis ax[3];
#pragma unroll
        for (int a = 0; a < DIM; ++a) ax[a] = axis_coords(p.origins[crop * DIM + a] + idx[a], d.log2_step);
        if (DIM == 2) { ax[2].i0 = ax[2].i1 = 0; ax[2].t1 = ax[2].k1 = 0.f; }

        float* o = SPLIT ? p.out + (n - (int64_t)crop * p.n_per_crop) : p.out + n * p.cin;
        const int64_t os = SPLIT ? p.n_per_crop : 1;      // stride between consecutive channels
        int row = 0;
        // --- G0: raw corners, corner-major then channel
        for (int q = 0; q < K0; ++q) {
            int dx, dy, dz;
            if (DIM == 2) { dx = q >> 1; dy = q & 1; dz = 0; }
            else if (d.method == 4) { dx = q >> 1; dy = q & 1; dz = dx ^ dy; }
            else { dx = (q >> 2) & 1; dy = (q >> 1) & 1; dz = q & 1; }
            const float* base = p.g0.p + p.g0.at(cl(ax[0].i0, p.g0.nx) + dx, cl(ax[1].i0, p.g0.ny) + dy, DIM == 3 ? cl(ax[2].i0, p.g0.nz) + dz : 0);
            for (int c = 0; c < C; ++c) o[(row++) * os] = base[c * p.g0.plane];
        }
        // --- G1: weighted corners
        for (int c = 0; c < C; ++c) {
            float sum = 0.f;
            for (int q = 0; q < K1; ++q) {
                int dx, dy, dz;
                if (DIM == 2) { dx = q >> 1; dy = q & 1; dz = 0; }
                else { dx = (q >> 2) & 1; dy = (q >> 1) & 1; dz = q & 1; }
                float v = p.g1.p[c * p.g1.plane + p.g1.at(cl(ax[0].i1, p.g1.nx) + dx, cl(ax[1].i1, p.g1.ny) + dy, DIM == 3 ? cl(ax[2].i1, p.g1.nz) + dz : 0)];
                if (d.g1_weight_mode != NIC_G1_UNWEIGHTED) {
                    int bits = dx | (dy << 1) | (dz << 2);
                    if (DIM == 3 && d.g1_weight_mode == NIC_G1_REFERENCE) bits = g1_ref_weight_bits(q);
                    v = mul_rn(v, (bits & 1) ? ax[0].k1 : 1.0f - ax[0].k1);
                    v = mul_rn(v, (bits & 2) ? ax[1].k1 : 1.0f - ax[1].k1);
                    if (DIM == 3) v = mul_rn(v, (bits & 4) ? ax[2].k1 : 1.0f - ax[2].k1);
                }
                if (SPLIT) o[(int64_t)(K0 * C + q * C + c) * os] = v;
                else sum = q == 0 ? v : add_rn(sum, v);
            }
            if (!SPLIT) o[(int64_t)(K0 * C + c) * os] = sum;
        }
        row = SPLIT ? (K0 + K1) * C : (K0 + 1) * C;
        // --- PE on the G1-cell coordinate t1, dimension-major
        for (int a = 0; a < DIM; ++a)
            for (int r2 = 0; r2 < P; ++r2)
                o[(int64_t)(row++) * os] = d.pe_mode == NIC_PE_TRIANGULAR ? tri_pe_row(ax[a].t1, r2, P) : sin_pe_row(ax[a].t1, r2, d.pe_div);
        // --- LOD
        if (!SPLIT) o[(int64_t)row * os] = d.lod_value;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward of encode_kernel<DIM,false>: scatter-add of dx[n, Cin] into the two grid gradients
struct EncodeBwdParams {
    nic_path_desc d;
    GridView g0, g1;          // .p unused; geometry only
    const int32_t* origins;
    const float* dx;
    float* g0_grad;
    float* g1_grad;
    int cin;
    int64_t n_per_crop, n_total;
};

// Consecutive lanes are consecutive samples along the last axis: with cells of 4 .. 4096 samples (the multi-level extension reads pair l at
// 4^-(l+1)) whole runs of lanes add into the SAME nodes - a coarse pair takes every sample of the launch on a few hundred addresses.  Runs of
// equal cells are therefore summed across lanes first (segmented shuffle reduction: lane i gathers the values of the up to 2^k following lanes
// of its run) and only the head of a run issues the atomic: 8.3 M samples on the [12,5,4] grid went from seconds to the cost of the shuffles.
struct RunMasks {
    bool same[6];     // lane + 2^k is in this lane's run
    bool head;        // first lane of its run
    bool any_shared;  // wave-uniform: some run is longer than one lane
};
__device__ __forceinline__ RunMasks run_masks(int64_t key, int lane) {
    RunMasks m;
    const int plo = __shfl_up((int)(uint32_t)key, 1), phi = __shfl_up((int)(key >> 32), 1);
    m.head = lane == 0 || plo != (int)(uint32_t)key || phi != (int)(key >> 32);
    const unsigned long long hb = __ballot(m.head);
    const int run = __popcll(hb & (~0ull >> (63 - lane)));
    m.any_shared = __popcll(hb) != 64;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int other = __shfl_down(run, 1 << k);
        m.same[k] = lane + (1 << k) < 64 && other == run;
    }
    return m;
}
__device__ __forceinline__ float run_sum(float v, const RunMasks& m) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float o = __shfl_down(v, 1 << k);
        v += m.same[k] ? o : 0.f;
    }
    return v;
}

template <int DIM>
__global__ void __launch_bounds__(256) encode_backward_kernel(EncodeBwdParams p) {
    const nic_path_desc& d = p.d;
    const int C = d.channels;
    const int K0 = (DIM == 2 || d.method == 4) ? 4 : 8;
    const int K1 = DIM == 2 ? 4 : 8;
    const int lane = threadIdx.x & 63;
    // A wave takes a PATCH of samples - 8 x 8 (2D) or 4 x 4 x 4 (3D) - with its lanes ordered cell-major: 2D lane = 16 (2 cx + cy) + 4 ix + iy (sample
    // 4 cx + ix, 4 cy + iy of the patch), 3D lane = 16 ix + 4 iy + iz.  At mip 0 (G0 cells of 4, G1 cells of 8 samples per axis) an aligned patch is
    // four G0 cells of 16 consecutive lanes and ONE G1 cell (3D: one G0 cell, an eighth of a G1 cell): runs of 16 / 64 lanes instead of the 4 / 8 of a
    // row-major walk - 3.75 instead of 18 atomics per sample; unaligned crops and other steps just get the runs the keys say (run_masks is generic).
    constexpr int PS0 = DIM == 2 ? 8 : 4, PS1 = DIM == 2 ? 8 : 4, PS2 = DIM == 2 ? 1 : 4;
    const int np1 = (d.extent[1] + PS1 - 1) / PS1, np2 = DIM == 3 ? (d.extent[2] + PS2 - 1) / PS2 : 1;
    const int64_t per_crop = (int64_t)((d.extent[0] + PS0 - 1) / PS0) * np1 * np2;
    const int64_t n_patches = per_crop * d.num_crops;
    for (int64_t wb = (int64_t)blockIdx.x * 4; wb < n_patches; wb += (int64_t)gridDim.x * 4) {      // block-uniform trip count: the shuffles see whole waves
        const int64_t wv = wb + (threadIdx.x >> 6);
        const int64_t wc = wv < n_patches ? wv : n_patches - 1;
        const int crop = (int)(wc / per_crop);
        int64_t pr = wc - (int64_t)crop * per_crop;
        int pt[3] = {0, 0, 0};
        if (DIM == 3) { pt[2] = (int)(pr % np2); pr /= np2; }
        pt[1] = (int)(pr % np1);
        pt[0] = (int)(pr / np1);
        int idx[3];
        if (DIM == 2) {
            const int c = lane >> 4;
            idx[0] = PS0 * pt[0] + 4 * (c >> 1) + ((lane >> 2) & 3);
            idx[1] = PS1 * pt[1] + 4 * (c & 1) + (lane & 3);
            idx[2] = 0;
        } else {
            idx[0] = PS0 * pt[0] + (lane >> 4);
            idx[1] = PS1 * pt[1] + ((lane >> 2) & 3);
            idx[2] = PS2 * pt[2] + (lane & 3);
        }
        bool live = wv < n_patches;
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            live = live && idx[a] < d.extent[a];
            idx[a] = idx[a] < d.extent[a] ? idx[a] : d.extent[a] - 1;
        }
        const int64_t n = (int64_t)crop * p.n_per_crop + ((int64_t)idx[0] * d.extent[1] + idx[1]) * (DIM == 3 ? d.extent[2] : 1) + (DIM == 3 ? idx[2] : 0);
        Axis ax[3];
#pragma unroll
        for (int a = 0; a < DIM; ++a) ax[a] = axis_coords(p.origins[crop * DIM + a] + idx[a], d.log2_step);
        if (DIM == 2) { ax[2].i0 = ax[2].i1 = 0; ax[2].t1 = ax[2].k1 = 0.f; }
        const float* row = p.dx + n * p.cin;
        const int64_t cell0 = p.g0.at(cl(ax[0].i0, p.g0.nx), cl(ax[1].i0, p.g0.ny), DIM == 3 ? cl(ax[2].i0, p.g0.nz) : 0);
        const int64_t cell1 = p.g1.at(cl(ax[0].i1, p.g1.nx), cl(ax[1].i1, p.g1.ny), DIM == 3 ? cl(ax[2].i1, p.g1.nz) : 0);
        // dead lanes of the last wave: negative keys of their own (they add zeros and never issue)
        const RunMasks m0 = run_masks(live ? cell0 : -1 - (int64_t)lane, lane);
        const RunMasks m1 = run_masks(live ? cell1 : -1 - (int64_t)lane, lane);
        for (int q = 0; q < K0; ++q) {
            int dx, dy, dz;
            if (DIM == 2) { dx = q >> 1; dy = q & 1; dz = 0; }
            else if (d.method == 4) { dx = q >> 1; dy = q & 1; dz = dx ^ dy; }
            else { dx = (q >> 2) & 1; dy = (q >> 1) & 1; dz = q & 1; }
            float* base = p.g0_grad + cell0 + p.g0.at(dx, dy, dz);
            for (int c = 0; c < C; ++c) {
                float v = live ? row[q * C + c] : 0.f;
                if (m0.any_shared) v = run_sum(v, m0);
                if (live && m0.head) atomicAdd(base + c * p.g0.plane, v);
            }
        }
        for (int q = 0; q < K1; ++q) {
            int dx, dy, dz;
            if (DIM == 2) { dx = q >> 1; dy = q & 1; dz = 0; }
            else { dx = (q >> 2) & 1; dy = (q >> 1) & 1; dz = q & 1; }
            float w = 1.0f;
            if (d.g1_weight_mode != NIC_G1_UNWEIGHTED) {
                int bits = dx | (dy << 1) | (dz << 2);
                if (DIM == 3 && d.g1_weight_mode == NIC_G1_REFERENCE) bits = g1_ref_weight_bits(q);
                w = ((bits & 1) ? ax[0].k1 : 1.0f - ax[0].k1) * ((bits & 2) ? ax[1].k1 : 1.0f - ax[1].k1);
                if (DIM == 3) w *= (bits & 4) ? ax[2].k1 : 1.0f - ax[2].k1;
            }
            float* base = p.g1_grad + cell1 + p.g1.at(dx, dy, dz);
            for (int c = 0; c < C; ++c) {
                float v = live ? row[K0 * C + c] * w : 0.f;
                if (m1.any_shared) v = run_sum(v, m1);
                if (live && m1.head) atomicAdd(base + c * p.g1.plane, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// create_g / create_g_3d / create_g_3d_v2 (fp_def.py:81-112): out[k][c][n] = grid[c, z+dz, y+dy, x+dx]
__global__ void __launch_bounds__(256) gather_corners_kernel(GridView g, int C, const int32_t* xi, const int32_t* yi, const int32_t* zi,
                                                             int64_t n, int corner_set, float* out) {
    const int K = corner_set == 1 ? 8 : 4;
    const int64_t total = (int64_t)K * C * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i % n;
        const int c = (int)((i / n) % C);
        const int q = (int)(i / (n * C));
        int dx, dy, dz;
        if (corner_set == 0) { dx = q >> 1; dy = q & 1; dz = 0; }
        else if (corner_set == 1) { dx = (q >> 2) & 1; dy = (q >> 1) & 1; dz = q & 1; }
        else { dx = q >> 1; dy = q & 1; dz = dx ^ dy; }
        int x = xi[s] + dx, y = yi[s] + dy, z = zi ? zi[s] + dz : 0;
        x = x < 0 ? 0 : (x >= g.nx ? g.nx - 1 : x);
        y = y < 0 ? 0 : (y >= g.ny ? g.ny - 1 : y);
        z = z < 0 ? 0 : (z >= g.nz ? g.nz - 1 : z);
        out[i] = g.p[(int64_t)c * g.plane + g.at(x, y, z)];
    }
}

// ---------------------------------------------------------------------------------------------------
struct Div8 {
    float v[8];
};
template <int MODE>
__global__ void __launch_bounds__(256) pe_kernel(const float* coord, int64_t n, int dim, int P, float* out, Div8 dv) {
    __shared__ float div[8];
    if (threadIdx.x < 8) div[threadIdx.x] = dv.v[threadIdx.x];
    __syncthreads();
    const int64_t total = n * dim * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i % n;
        const int row = (int)(i / n);
        const int a = row / P, r = row % P;
        const float c = coord[(int64_t)a * n + s];
        out[i] = MODE == NIC_PE_TRIANGULAR ? tri_pe_row(c, r, P) : sin_pe_row(c, r, div);
    }
}

__global__ void __launch_bounds__(256) lut_gather_kernel(const float* lut, int rows, int seq, const int64_t* coord, int64_t b, int64_t L, float* out) {
    const int64_t total = b * rows * L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = i % L;
        const int r = (int)((i / L) % rows);
        const int64_t bb = i / (L * rows);
        int64_t c = coord[bb * L + l] % seq;
        if (c < 0) c += seq;                                    // python / torch remainder
        out[i] = lut[(int64_t)r * seq + c];
    }
}

// ---------------------------------------------------------------------------------------------------
// quantisers and codec (models.py:29-71).  Division (not reciprocal multiply) like the reference.
// ---------------------------------------------------------------------------------------------------
enum { Q_QUANT = 0, Q_TO_BIT = 1 };
template <int OP>
__global__ void __launch_bounds__(256) quantize_kernel(const float* src, float* dst, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float q = __fdiv_rn(floorf(__fadd_rn(__fmul_rn(src[i], scale), 0.5f)), scale);
        dst[i] = OP == Q_QUANT ? q : __fmul_rn(q, scale);
    }
}
__global__ void __launch_bounds__(256) clamp_kernel(float* x, int64_t n, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = clamp_keep_nan(x[i], lo, hi);
}
__global__ void __launch_bounds__(256) save4fp_kernel(const float* src, uint8_t* dst, int64_t n, float scale, float bias) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = __fadd_rn(floorf(__fadd_rn(__fmul_rn(src[i], scale), 0.5f)), bias);
        dst[i] = (uint8_t)(int)v;                               // torch float -> uint8 cast truncates
    }
}
__global__ void __launch_bounds__(256) load4fp_kernel(const uint8_t* src, float* dst, int64_t n, float scale, float bias) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = __fdiv_rn(__fadd_rn(__fsub_rn((float)src[i], bias), 1.0f), scale);
}

// ---------------------------------------------------------------------------------------------------
// PSNR (utils.py:117-130): two-stage fixed-order reduction of the squared error in fp64.
// ---------------------------------------------------------------------------------------------------
constexpr int kPsnrBlocks = 1024;
__global__ void __launch_bounds__(256) sqerr_partial_kernel(const float* a, const float* b, int64_t n, double* part) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float dd = a[i] - b[i];
        acc += (double)dd * (double)dd;
    }
    __shared__ double sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ void __launch_bounds__(256) psnr_final_kernel(const double* part, int nparts, int64_t n, float peak, float* out2) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = (float)(sm[0] / (double)n);
        out2[0] = mse;
        out2[1] = mse == 0.f ? INFINITY : 10.0f * log10f(peak * peak / mse);
    }
}

// ---------------------------------------------------------------------------------------------------
// torch.optim.Adam single-tensor semantics (default: no weight decay, no amsgrad, eps outside sqrt after
// bias correction) + the fp_quantize_clamp that follows it (image_compression.py:266-269).
// ---------------------------------------------------------------------------------------------------
// The scalars are formed on the host in double and cast once, like torch does with its Python-float hyper-parameters:
// omb1 = (float)(1 - beta1), omb2 = (float)(1 - beta2), step_size = (float)(lr / bias_correction1), bc2_sqrt = (float)sqrt(bias_correction2)
// (1.0f - 0.999f in fp32 is 9.99987e-4, not 0.001f: exp_avg_sq would drift 1.3e-5 low).
__global__ void __launch_bounds__(256) adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float b2,
                                                   float omb1, float omb2, float eps, float bc2_sqrt, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;                    // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + omb2 * gi * gi;                  // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        float x = p[i] - step_size * (mi / denom);
        if (lo <= hi) x = clamp_keep_nan(x, lo, hi);
        p[i] = x;
    }
}

// Multi-tensor form (nic_adam.hpp): one launch for the whole parameter list, a block per 4096-element chunk of one tensor
__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamTable t) { adam_block(t, t.count, (int)blockIdx.x); }

// ---------------------------------------------------------------------------------------------------
// Device-side sampler and resident RGBX targets
// ---------------------------------------------------------------------------------------------------
__global__ void draw_origins_kernel(uint64_t seed, uint64_t step, int num_crops, int dim, uint32_t range, int32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_crops) return;
    const nic::U4 b = nic::sampler_block(seed, step, (uint32_t)i);
    for (int a = 0; a < dim; ++a) out[i * dim + a] = nic::sampler_origin(b, a, range);
}
// Head of a hipGraph-captured training step: takes the step number t from a device counter, publishes it for the kernels that follow
// (counters[1]; the fused step adds it to its noise offset, the optimiser indexes its schedule with it), advances the counter, files the
// previous step's loss (the reduction of step t - 1 wrote `loss_slot`) and draws the crop origins of step t.  One block.
__global__ void step_begin_kernel(uint64_t seed, int64_t* counters, int num_crops, int dim, uint32_t range, int32_t* out, const float* loss_slot,
                                  float* loss_hist, int64_t hist_len) {
    const int64_t t = counters[0];
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[1] = t;
        counters[0] = t + 1;
        if (loss_hist != nullptr && loss_slot != nullptr && t >= 1 && t - 1 < hist_len) loss_hist[t - 1] = *loss_slot;
    }
    for (int i = threadIdx.x; i < num_crops; i += blockDim.x) {
        const nic::U4 b = nic::sampler_block(seed, (uint64_t)t, (uint32_t)i);
        for (int a = 0; a < dim; ++a) out[i * dim + a] = nic::sampler_origin(b, a, range);
    }
}
// planar uint8 [3][n] -> interleaved R | G << 8 | B << 16 (one dword per sample: the fused kernels fetch a target with ONE load)
__global__ void __launch_bounds__(256) rgbx_interleave_kernel(const uint8_t* src, int64_t n, uint32_t* dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (uint32_t)src[i] | ((uint32_t)src[n + i] << 8) | ((uint32_t)src[2 * n + i] << 16);
}
// 2 x 2 box filter, round to nearest (ties up): [s0][s1] RGBX -> [s0/2][s1/2] RGBX
__global__ void __launch_bounds__(256) rgbx_down2_kernel(const uint32_t* src, int s0, int s1, uint32_t* dst) {
    const int d0 = s0 >> 1, d1 = s1 >> 1;
    const int64_t n = (int64_t)d0 * d1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i / d1), y = (int)(i - (int64_t)x * d1);
        const uint32_t a = src[(int64_t)(2 * x) * s1 + 2 * y], b = src[(int64_t)(2 * x) * s1 + 2 * y + 1];
        const uint32_t c = src[(int64_t)(2 * x + 1) * s1 + 2 * y], d = src[(int64_t)(2 * x + 1) * s1 + 2 * y + 1];
        uint32_t o = 0;
        for (int k = 0; k < 3; ++k) {
            const uint32_t sum = ((a >> (8 * k)) & 255u) + ((b >> (8 * k)) & 255u) + ((c >> (8 * k)) & 255u) + ((d >> (8 * k)) & 255u);
            o |= ((sum + 2u) >> 2) << (8 * k);
        }
        dst[i] = o;
    }
}

// One pass of Pillow's two-pass resize (src/libImaging/Resample.c: ImagingResampleHorizontal_8bpc / Vertical_8bpc) on an RGBX image - what
// torchvision's transforms.Resize runs on the PIL image the reference builds its mip chain from (image_compression.py:434-440): out = clip8((2^21 +
// sum over the taps of pixel * k) >> 22) per channel with the caller's fixed-point coefficients (sampler.resize_coeffs restates precompute_coeffs +
// normalize_coeffs_8bpc).  axis 1: along the contiguous axis, dst [s0][out]; axis 0: across rows, dst [out][s1].
__global__ void __launch_bounds__(256) rgbx_resample_kernel(const uint32_t* src, int s0, int s1, int axis, int out_size, const int32_t* bounds, const int32_t* kk,
                                                            int ksize, uint32_t* dst) {
    const int64_t n = axis == 1 ? (int64_t)s0 * out_size : (int64_t)out_size * s1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = axis == 1 ? out_size : s1;
        const int r = (int)(i / w), c = (int)(i - (int64_t)r * w);
        const int o = axis == 1 ? c : r;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int32_t* k = kk + (int64_t)o * ksize;
        int ss0 = 1 << 21, ss1 = 1 << 21, ss2 = 1 << 21;
        for (int t = 0; t < cnt; ++t) {
            const uint32_t px = axis == 1 ? src[(int64_t)r * s1 + lo + t] : src[(int64_t)(lo + t) * s1 + c];
            ss0 += (int)(px & 255u) * k[t];
            ss1 += (int)((px >> 8) & 255u) * k[t];
            ss2 += (int)((px >> 16) & 255u) * k[t];
        }
        auto clip8 = [](int v) { v >>= 22; return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        dst[i] = clip8(ss0) | (clip8(ss1) << 8) | (clip8(ss2) << 16);
    }
}

static inline int blocks_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace nic

using namespace nic;

static int check_desc_common(const nic_path_desc* d) {
    if (!d) return NIC_E_NULL;
    if (!((d->dim == 2 && d->method == 1) || (d->dim == 3 && (d->method == 3 || d->method == 4)))) return NIC_E_UNSUPPORTED;
    if (d->channels < 1 || d->pe_channels < 2 || (d->pe_channels & 1) || d->pe_channels > 16) return NIC_E_UNSUPPORTED;
    if (d->num_crops < 1) return NIC_E_SHAPE;
    for (int a = 0; a < d->dim; ++a)
        if (d->extent[a] < 1 || d->g0_nodes[a] < 2 || d->g1_nodes[a] < 2) return NIC_E_SHAPE;
    // the element-wise kernels shift: any cell size up to 2^16 samples (the reference's own steps are 1/4 .. 2; the multi-level extension reads pair l at 4^-(l+1))
    if (d->log2_step < -16 || d->log2_step > 8) return NIC_E_ARG;
    if (d->passes < 0 || d->passes > 1) return NIC_E_ARG;             // passes: the fused training entry points only
    if (d->flags & NIC_FLAG_ORIGINS_HOST) return NIC_E_ARG;           // the layer-wise kernels read their origins from device memory
    return NIC_OK;
}

static void make_views(const nic_path_desc* d, const float* g0, const float* g1, GridView& v0, GridView& v1) {
    v0.p = g0; v0.nx = d->g0_nodes[0]; v0.ny = d->g0_nodes[1]; v0.nz = d->dim == 3 ? d->g0_nodes[2] : 1;
    v0.plane = (int64_t)v0.nx * v0.ny * v0.nz;
    v1.p = g1; v1.nx = d->g1_nodes[0]; v1.ny = d->g1_nodes[1]; v1.nz = d->dim == 3 ? d->g1_nodes[2] : 1;
    v1.plane = (int64_t)v1.nx * v1.ny * v1.nz;
}

template <bool SPLIT>
static int encode_impl(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, float* out, void* stream) {
    int rc = check_desc_common(d);
    if (rc) return rc;
    if (!g0 || !g1 || !origins || !out) return NIC_E_NULL;
    if (SPLIT && d->num_crops != 1) return NIC_E_ARG;
    EncodeParams p;
    p.d = *d;
    make_views(d, g0, g1, p.g0, p.g1);
    p.origins = origins;
    p.out = out;
    p.cin = nic_decoder_input_channels(d->dim, d->method, d->channels, d->pe_channels);
    p.n_per_crop = (int64_t)d->extent[0] * d->extent[1] * (d->dim == 3 ? d->extent[2] : 1);
    p.n_total = p.n_per_crop * d->num_crops;
    hipStream_t s = (hipStream_t)stream;
    const int nb = blocks_for(p.n_total);
    if (d->dim == 2) hipLaunchKernelGGL((encode_kernel<2, SPLIT>), dim3(nb), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((encode_kernel<3, SPLIT>), dim3(nb), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

// stripe exchange (include/nicv2_hip.h: nic_stripe_pack / _unpack): one thread per float of the exchange buffer
struct RowSets {
    nic_row_set s[2];
    int n;
    int64_t n_small, count[2];
};
template <bool PACK>
__global__ void __launch_bounds__(256) stripe_rows_kernel(float* small_buf, RowSets rs, float* buf) {
    const int64_t total = rs.n_small + rs.count[0] + rs.count[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float* elem;
        if (i < rs.n_small) {
            elem = small_buf + i;
        } else {
            int64_t j = i - rs.n_small;
            const int k = (rs.n > 1 && j >= rs.count[0]) ? 1 : 0;
            if (k) j -= rs.count[0];
            const nic_row_set& q = rs.s[k];
            const int64_t per_c = (int64_t)q.nrows * q.row_elems;
            const int c = (int)(j / per_c);
            const int64_t r = j - (int64_t)c * per_c;
            const int b = (int)(r / q.row_elems), x = (int)(r - (int64_t)b * q.row_elems);
            elem = q.base + (int64_t)c * q.plane + (int64_t)q.rows[b] * q.row_elems + x;
        }
        if (PACK) buf[i] = *elem;
        else *elem = buf[i];
    }
}
static int stripe_rows(bool pack, float* small_buf, int64_t n_small, const nic_row_set* sets, int nsets, float* buf, void* stream) {
    if (!buf || (n_small > 0 && !small_buf) || (nsets > 0 && !sets)) return NIC_E_NULL;
    if (n_small < 0 || nsets < 0 || nsets > 2) return NIC_E_ARG;
    RowSets rs;
    rs.n = nsets; rs.n_small = n_small; rs.count[0] = rs.count[1] = 0;
    for (int k = 0; k < nsets; ++k) {
        if (!sets[k].base) return NIC_E_NULL;
        if (sets[k].nrows < 0 || sets[k].nrows > NIC_STRIPE_MAX_ROWS || sets[k].row_elems < 1 || sets[k].channels < 1) return NIC_E_ARG;
        for (int b = 0; b < sets[k].nrows; ++b)
            if (sets[k].rows[b] < 0 || (int64_t)(sets[k].rows[b] + 1) * sets[k].row_elems > sets[k].plane) return NIC_E_SHAPE;
        rs.s[k] = sets[k];
        rs.count[k] = (int64_t)sets[k].channels * sets[k].nrows * sets[k].row_elems;
    }
    const int64_t total = n_small + rs.count[0] + rs.count[1];
    if (total == 0) return NIC_OK;
    const dim3 g(blocks_for(total)), b(256);
    if (pack) hipLaunchKernelGGL(stripe_rows_kernel<true>, g, b, 0, (hipStream_t)stream, small_buf, rs, buf);
    else hipLaunchKernelGGL(stripe_rows_kernel<false>, g, b, 0, (hipStream_t)stream, small_buf, rs, buf);
    return (int)hipGetLastError();
}

extern "C" {

int nic_stripe_pack(const float* small_buf, int64_t n_small, const nic_row_set* sets, int nsets, float* buf, void* stream) {
    return stripe_rows(true, const_cast<float*>(small_buf), n_small, sets, nsets, buf, stream);
}
int nic_stripe_unpack(float* small_buf, int64_t n_small, const nic_row_set* sets, int nsets, const float* buf, void* stream) {
    return stripe_rows(false, small_buf, n_small, sets, nsets, const_cast<float*>(buf), stream);
}

int nic_abi_version(void) { return NIC_ABI_VERSION; }

const char* nic_error_string(int code) {
    switch (code) {
        case NIC_OK: return "ok";
        case NIC_E_NULL: return "required pointer is null";
        case NIC_E_UNSUPPORTED: return "unsupported dim/method/channels/hidden combination";
        case NIC_E_SHAPE: return "inconsistent extents, node counts or origins";
        case NIC_E_WORKSPACE: return "workspace too small (see nic_workspace_bytes)";
        case NIC_E_ARG: return "invalid argument";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int nic_decoder_input_channels(int dim, int method, int channels, int pe_channels) {
    const int k0 = (dim == 2 || method == 4) ? 4 : 8;          // var2.py:114-118
    return channels * (k0 + 1) + pe_channels * dim + 1;
}

int nic_encode(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, float* out, void* stream) {
    return encode_impl<false>(d, g0, g1, origins, out, stream);
}
int nic_encode_split(const nic_path_desc* d, const float* g0, const float* g1, const int32_t* origins, float* out, void* stream) {
    return encode_impl<true>(d, g0, g1, origins, out, stream);
}

int nic_encode_backward(const nic_path_desc* d, const int32_t* origins, const float* dx, float* g0_grad, float* g1_grad, void* stream) {
    int rc = check_desc_common(d);
    if (rc) return rc;
    if (!origins || !dx || !g0_grad || !g1_grad) return NIC_E_NULL;
    EncodeBwdParams p;
    p.d = *d;
    make_views(d, nullptr, nullptr, p.g0, p.g1);
    p.origins = origins; p.dx = dx; p.g0_grad = g0_grad; p.g1_grad = g1_grad;
    p.cin = nic_decoder_input_channels(d->dim, d->method, d->channels, d->pe_channels);
    p.n_per_crop = (int64_t)d->extent[0] * d->extent[1] * (d->dim == 3 ? d->extent[2] : 1);
    p.n_total = p.n_per_crop * d->num_crops;
    // one wave per patch of 8 x 8 (2D) / 4 x 4 x 4 (3D) samples, four waves per block
    int64_t patches = d->num_crops;
    for (int a = 0; a < d->dim; ++a) patches *= (d->extent[a] + (d->dim == 2 ? 8 : 4) - 1) / (d->dim == 2 ? 8 : 4);
    const int nb = blocks_for(patches * 64);
    if (d->dim == 2) hipLaunchKernelGGL((encode_backward_kernel<2>), dim3(nb), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((encode_backward_kernel<3>), dim3(nb), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int nic_gather_corners(const float* grid, int channels, int nx, int ny, int nz, const int32_t* xi, const int32_t* yi, const int32_t* zi,
                       int64_t n, int corner_set, float* out, void* stream) {
    if (!grid || !xi || !yi || !out) return NIC_E_NULL;
    if (channels < 1 || nx < 1 || ny < 1 || nz < 1 || n < 0 || corner_set < 0 || corner_set > 2) return NIC_E_ARG;
    if ((corner_set != 0) != (zi != nullptr)) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    GridView g;
    g.p = grid; g.nx = nx; g.ny = ny; g.nz = nz; g.plane = (int64_t)nx * ny * nz;
    const int K = corner_set == 1 ? 8 : 4;
    hipLaunchKernelGGL(gather_corners_kernel, dim3(blocks_for((int64_t)K * channels * n)), dim3(256), 0, (hipStream_t)stream, g, channels, xi,
                       yi, zi, n, corner_set, out);
    return (int)hipGetLastError();
}

int nic_positional_encoding(const float* coord, int64_t n, int dim, int P, int pe_mode, const float* pe_div_host, float* out, void* stream) {
    if (!coord || !out) return NIC_E_NULL;
    if (n < 0 || dim < 1 || dim > 3 || P < 2 || (P & 1) || P > 16) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nb = blocks_for(n * dim * P);
    Div8 dv;
    for (int i = 0; i < 8; ++i) dv.v[i] = 0.f;
    if (pe_mode == NIC_PE_TRIANGULAR) {
        hipLaunchKernelGGL((pe_kernel<NIC_PE_TRIANGULAR>), dim3(nb), dim3(256), 0, s, coord, n, dim, P, out, dv);
    } else if (pe_mode == NIC_PE_SINUSOIDAL) {
        if (!pe_div_host) return NIC_E_NULL;
        for (int i = 0; i < P / 2; ++i) dv.v[i] = pe_div_host[i];      // travels as a kernel argument: no allocation
        hipLaunchKernelGGL((pe_kernel<NIC_PE_SINUSOIDAL>), dim3(nb), dim3(256), 0, s, coord, n, dim, P, out, dv);
    } else {
        return NIC_E_ARG;
    }
    return (int)hipGetLastError();
}

int nic_lut_gather(const float* lut, int rows, int seq_len, const int64_t* coord, int64_t b, int64_t L, float* out, void* stream) {
    if (!lut || !coord || !out) return NIC_E_NULL;
    if (rows < 1 || seq_len < 1 || b < 0 || L < 0) return NIC_E_ARG;
    if (b * L == 0) return NIC_OK;
    hipLaunchKernelGGL(lut_gather_kernel, dim3(blocks_for(b * rows * L)), dim3(256), 0, (hipStream_t)stream, lut, rows, seq_len, coord, b, L, out);
    return (int)hipGetLastError();
}

int nic_quantize(const float* src, float* dst, int64_t n, int num_bits, void* stream) {
    if (!src || !dst) return NIC_E_NULL;
    if (num_bits < 1 || num_bits > 16 || n < 0) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    hipLaunchKernelGGL((quantize_kernel<Q_QUANT>), dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n, (float)((1 << num_bits) - 1));
    return (int)hipGetLastError();
}
int nic_quantize_to_bit(const float* src, float* dst, int64_t n, int num_bits, void* stream) {
    if (!src || !dst) return NIC_E_NULL;
    if (num_bits < 1 || num_bits > 16 || n < 0) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    hipLaunchKernelGGL((quantize_kernel<Q_TO_BIT>), dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n, (float)((1 << num_bits) - 1));
    return (int)hipGetLastError();
}
int nic_clamp(float* x, int64_t n, float lo, float hi, void* stream) {
    if (!x) return NIC_E_NULL;
    if (n <= 0) return n == 0 ? NIC_OK : NIC_E_ARG;
    hipLaunchKernelGGL(clamp_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, lo, hi);
    return (int)hipGetLastError();
}
int nic_save4fp_u8(const float* src, uint8_t* dst, int64_t n, int num_bits, void* stream) {
    if (!src || !dst) return NIC_E_NULL;
    if (num_bits < 1 || num_bits > 8 || n < 0) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    hipLaunchKernelGGL(save4fp_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n, (float)((1 << num_bits) - 1),
                       (float)((1 << (num_bits - 1)) - 1));
    return (int)hipGetLastError();
}
int nic_load4fp_u8(const uint8_t* src, float* dst, int64_t n, int num_bits, void* stream) {
    if (!src || !dst) return NIC_E_NULL;
    if (num_bits < 1 || num_bits > 8 || n < 0) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    hipLaunchKernelGGL(load4fp_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n, (float)((1 << num_bits) - 1),
                       (float)(1 << (num_bits - 1)));
    return (int)hipGetLastError();
}

int nic_psnr(const float* a, const float* b, int64_t n, int num_bits, float* out2, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !b || !out2 || !workspace) return NIC_E_NULL;
    if (n <= 0 || num_bits < 1 || num_bits > 24) return NIC_E_ARG;
    if (workspace_bytes < sizeof(double) * kPsnrBlocks) return NIC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int nb = blocks_for(n);
    if (nb > kPsnrBlocks) nb = kPsnrBlocks;
    hipLaunchKernelGGL(sqerr_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, n, (double*)workspace);
    hipLaunchKernelGGL(psnr_final_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, nb, n, (float)(1 << num_bits), out2);
    return (int)hipGetLastError();
}

int nic_sampler_lod_host(uint64_t seed, uint64_t step, int uniform, int max_mip) {
    if (max_mip < 0) return NIC_E_ARG;
    return nic::sampler_lod(seed, step, uniform, max_mip);
}
int nic_sampler_origins_host(uint64_t seed, uint64_t step, int num_crops, int dim, int32_t range, int32_t* origins_host) {
    if (!origins_host) return NIC_E_NULL;
    if (num_crops < 0 || dim < 1 || dim > 3 || range < 1) return NIC_E_ARG;
    for (int i = 0; i < num_crops; ++i) {
        const nic::U4 b = nic::sampler_block(seed, step, (uint32_t)i);
        for (int a = 0; a < dim; ++a) origins_host[i * dim + a] = nic::sampler_origin(b, a, (uint32_t)range);
    }
    return NIC_OK;
}
int nic_sampler_draw_origins(uint64_t seed, uint64_t step, int num_crops, int dim, int32_t range, int32_t* origins, void* stream) {
    if (!origins) return NIC_E_NULL;
    if (num_crops < 1 || dim < 1 || dim > 3 || range < 1) return NIC_E_ARG;
    hipLaunchKernelGGL(draw_origins_kernel, dim3((num_crops + 63) / 64), dim3(64), 0, (hipStream_t)stream, seed, step, num_crops, dim, (uint32_t)range, origins);
    return (int)hipGetLastError();
}
int nic_rgbx_interleave(const uint8_t* planar, int64_t n, uint32_t* rgbx, void* stream) {
    if (!planar || !rgbx) return NIC_E_NULL;
    if (n < 0) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    hipLaunchKernelGGL(rgbx_interleave_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, planar, n, rgbx);
    return (int)hipGetLastError();
}
int nic_rgbx_downsample2(const uint32_t* src, int s0, int s1, uint32_t* dst, void* stream) {
    if (!src || !dst) return NIC_E_NULL;
    if (s0 < 2 || s1 < 2) return NIC_E_ARG;
    hipLaunchKernelGGL(rgbx_down2_kernel, dim3(blocks_for((int64_t)(s0 >> 1) * (s1 >> 1))), dim3(256), 0, (hipStream_t)stream, src, s0, s1, dst);
    return (int)hipGetLastError();
}

int nic_rgbx_resample_axis(const uint32_t* src, int s0, int s1, int axis, int out_size, const int32_t* bounds, const int32_t* kk, int ksize, uint32_t* dst,
                           void* stream) {
    if (!src || !dst || !bounds || !kk) return NIC_E_NULL;
    if (s0 < 1 || s1 < 1 || out_size < 1 || ksize < 1 || (axis != 0 && axis != 1)) return NIC_E_ARG;
    const int64_t n = axis == 1 ? (int64_t)s0 * out_size : (int64_t)out_size * s1;
    hipLaunchKernelGGL(rgbx_resample_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, s0, s1, axis, out_size, bounds, kk, ksize, dst);
    return (int)hipGetLastError();
}

int nic_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                  double eps, int64_t step, float clamp_lo, float clamp_hi, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return NIC_E_NULL;
    if (n <= 0 || step < 1) return n == 0 ? NIC_OK : NIC_E_ARG;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       (float)(lr / bc1), (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2), clamp_lo, clamp_hi);
    return (int)hipGetLastError();
}

static int adam_multi_impl(const nic_adam_tensor* tensors, int count, double beta1, double beta2, double eps, const float* sched, int64_t sched_rows,
                           const int64_t* step_dev, void* stream);

int nic_adam_multi(const nic_adam_tensor* tensors, int count, double beta1, double beta2, double eps, void* stream) {
    return adam_multi_impl(tensors, count, beta1, beta2, eps, nullptr, 0, nullptr, stream);
}

int nic_adam_multi_dev(const nic_adam_tensor* tensors, int count, double beta1, double beta2, double eps, const float* sched, int64_t sched_rows,
                       const int64_t* step_dev, void* stream) {
    if (!sched || !step_dev) return NIC_E_NULL;
    if (sched_rows < 1) return NIC_E_ARG;
    return adam_multi_impl(tensors, count, beta1, beta2, eps, sched, sched_rows, step_dev, stream);
}

int nic_sampler_step_begin(uint64_t seed, int64_t* counters, int num_crops, int dim, int32_t range, int32_t* origins, const float* loss_slot,
                           float* loss_hist, int64_t hist_len, void* stream) {
    if (!counters || !origins) return NIC_E_NULL;
    if (num_crops < 1 || dim < 2 || dim > 3 || range < 1 || hist_len < 0) return NIC_E_ARG;
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seed, counters, num_crops, dim, (uint32_t)range, origins, loss_slot,
                       loss_hist, hist_len);
    return (int)hipGetLastError();
}

static int adam_multi_impl(const nic_adam_tensor* tensors, int count, double beta1, double beta2, double eps, const float* sched, int64_t sched_rows,
                           const int64_t* step_dev, void* stream) {
    if (count == 0) return NIC_OK;
    if (!tensors) return NIC_E_NULL;
    if (count < 0 || count > NIC_ADAM_MAX_TENSORS) return NIC_E_ARG;
    AdamTable t;
    int n_stream = 0;
    int64_t blocks = 0;
    const int rc = adam_build_table(tensors, count, count, beta1, beta2, eps, sched, sched_rows, step_dev, t, n_stream, blocks);
    if (rc) return rc;
    if (t.count == 0) return NIC_OK;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    return (int)hipGetLastError();
}

}  // extern "C"
