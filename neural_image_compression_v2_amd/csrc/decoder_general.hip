// ColorDecoder of ANY width and depth on explicit [n, Cin] inputs (image_compression.py:54-68: DECODER_INPUT_CHANNELS and
// HIDDEN_LAYER_CHANNELS are reference flags, var2.py:72,114-118) - the layer-wise fp32 path behind nic_decoder_general_forward /
// _backward.  The fused kernels specialise the reference's defaults (H = 64, the listed channel counts, 3 or 5 layers); every other
// configuration of the reference's flags runs here: one LDS-tiled fp32 MFMA (v_mfma_f32_32x32x2_f32) product kernel per layer and direction, activations kept in a
// caller-owned workspace the way autograd keeps them, the sample axis walked in chunks sized to stay in L2 / Infinity Cache.
//
//   forward   A_k = gelu(A_{k-1} W_k^T + b_k)  (D_k = gelu'(.) kept when training),  y = sigmoid(A_last W_out^T + b_out)
//   backward  dZ_out = dy y (1 - y);  dA_{k} = dZ_{k+1} W_{k+1};  dZ_k = dA_k D_k;  dW_k = dZ_k^T A_{k-1};  db_k = column sums of dZ_k
// Weight gradients: every workgroup sums a slice of the chunk's rows into ITS slot of the workspace (slots accumulate over the chunks,
// launches of one stream are ordered), one fixed-order reduction at the end: results are bit-stable run to run.
// Arithmetic: fp32 MFMA = exact fmaf chains along the reduction index, GELU / sigmoid of nic_device.hpp (the fused kernels' own).
#include "nic_device.hpp"

namespace nic {
namespace general {

constexpr int TM = 64, TN = 64, TK = 32, LDT = TM + 4;
typedef float f32x16v __attribute__((ext_vector_type(16)));

// A 64 x 64 output tile per 256-thread workgroup on the fp32 matrix pipe: wave w owns the 32 x 32 sub-tile (w >> 1, w & 1) and runs
// v_mfma_f32_32x32x2_f32 over k-steps of 32 staged through two LDS tiles ([32][68] floats, k-major; the next step's global loads are issued before
// this step's products).  The fp32 MFMA is an exact fmaf chain along k
// (MI355X_MICROARCH: bitwise), so the results are those of the scalar loop - at the pipe's 256 FLOP / cycle / CU instead of the VALU's LDS-bound ~ 20 %.
//   acc += sum over the reduction range [k0, k1) of a(row, k) * b(col, k);  la / lb return 0 outside their operand.
//   register r of lane l of a wave: row 32 (w >> 1) + (r & 3) + 8 (r >> 2) + 4 (l >> 5), column 32 (w & 1) + (l & 31)   (tile_row / tile_col)
// AFAST / BFAST: consecutive lanes of the LOADS walk the reduction index (the operand is contiguous along it) instead of the row / column index.
// side_sum (nullable): += every A operand this lane feeds the pipe (the weight kernel's bias gradient: column sums of dZ).
__device__ __forceinline__ int tile_row(int r) { return 32 * ((threadIdx.x >> 6) >> 1) + (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x & 63) >> 5); }
__device__ __forceinline__ int tile_col() { return 32 * ((threadIdx.x >> 6) & 1) + (threadIdx.x & 31); }

template <bool AFAST, bool BFAST, class LoadA, class LoadB>
__device__ __forceinline__ void tile_product(int k0, int k1, LoadA&& la, LoadB&& lb, f32x16v& acc, float* side_sum) {
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
    constexpr int NE = TK * TM / 256;                     // elements of a tile per thread and operand
    constexpr int KSH = TK == 32 ? 5 : 4;                 // log2(TK)
    static_assert(TK == (1 << KSH) && TM == 64 && TN == 64, "tile shape");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int arow = 32 * (wave >> 1) + (lane & 31), bcol = 32 * (wave & 1) + (lane & 31), half = lane >> 5;
    float av[NE], bv[NE];
    auto fetch = [&](int kb) {                            // global -> registers (the next tile's loads are in flight while this one multiplies)
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int e = tid + 256 * r;
            const int ak = AFAST ? (e & (TK - 1)) : (e >> 6), ai = AFAST ? (e >> KSH) : (e & 63);
            const int bk = BFAST ? (e & (TK - 1)) : (e >> 6), bj = BFAST ? (e >> KSH) : (e & 63);
            av[r] = kb + ak < k1 ? la(ai, kb + ak) : 0.f;
            bv[r] = kb + bk < k1 ? lb(bj, kb + bk) : 0.f;
        }
    };
    if (k0 < k1) fetch(k0);
    for (int kb = k0; kb < k1; kb += TK) {
        __syncthreads();                                  // the previous tile has been consumed
#pragma unroll
        for (int r = 0; r < NE; ++r) {
            const int e = tid + 256 * r;
            As[AFAST ? (e & (TK - 1)) : (e >> 6)][AFAST ? (e >> KSH) : (e & 63)] = av[r];
            Bs[BFAST ? (e & (TK - 1)) : (e >> 6)][BFAST ? (e >> KSH) : (e & 63)] = bv[r];
        }
        __syncthreads();
        if (kb + TK < k1) fetch(kb + TK);
#pragma unroll
        for (int kk = 0; kk < TK / 2; ++kk) {             // A[i][k] in lane (i, k half), B[k][j] in lane (j, k half): two k per instruction
            const float a = As[2 * kk + half][arow], b = Bs[2 * kk + half][bcol];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (side_sum) *side_sum += a;
        }
    }
}

enum { ACT_GELU = 0, ACT_SIGMOID = 1, ACT_SIGMOID_BWD = 2 };

// out[i][o] = act(sum_k in[i][k] W[o][k] + bias[o]):  in [m][K], W [N][K], out_a / out_d [m][N]
template <int ACT>
__global__ void __launch_bounds__(256) linear_forward_kernel(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
                                                             int64_t m, int N, int K, float* __restrict__ out_a, float* __restrict__ out_d,
                                                             const float* __restrict__ dy) {
    const int64_t i0 = (int64_t)blockIdx.x * TM;
    const int j0 = blockIdx.y * TN;
    f32x16v acc = {};
    tile_product<true, true>(0, K,
        [&](int i, int k) { return i0 + i < m ? in[(i0 + i) * K + k] : 0.f; },
        [&](int j, int k) { return j0 + j < N ? W[(int64_t)(j0 + j) * K + k] : 0.f; }, acc, nullptr);
    const int j = j0 + tile_col();
    if (j >= N) return;
    const float bj = bias[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t i = i0 + tile_row(r);
        if (i >= m) continue;
        const float z = acc[r] + bj;
        if (ACT == ACT_GELU) {
            float a, d;
            gelu_and_grad(z, a, d);
            out_a[i * N + j] = a;
            if (out_d) out_d[i * N + j] = d;
        } else {
            const float y = sigmoid_f(z);
            if (ACT == ACT_SIGMOID) out_a[i * N + j] = y;
            else out_a[i * N + j] = dy[i * N + j] * y * (1.0f - y);         // dZ_out
        }
    }
}

// out[i][k] = (sum_o dz[i][o] W[o][k]) * (D ? D[i][k] : 1):  dz [m][N], W [N][K], D / out [m][K]
__global__ void __launch_bounds__(256) linear_backward_input_kernel(const float* __restrict__ dz, const float* __restrict__ W, const float* __restrict__ D,
                                                                    int64_t m, int N, int K, float* __restrict__ out) {
    const int64_t i0 = (int64_t)blockIdx.x * TM;
    const int j0 = blockIdx.y * TN;
    f32x16v acc = {};
    tile_product<true, false>(0, N,
        [&](int i, int o) { return i0 + i < m ? dz[(i0 + i) * N + o] : 0.f; },
        [&](int j, int o) { return j0 + j < K ? W[(int64_t)o * K + j0 + j] : 0.f; }, acc, nullptr);
    const int j = j0 + tile_col();
    if (j >= K) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t i = i0 + tile_row(r);
        if (i >= m) continue;
        out[i * K + j] = D ? acc[r] * D[i * K + j] : acc[r];
    }
}

// slot[s][o][k] (+)= sum over the rows of slice s of dz[i][o] a[i][k];  slot[s][N K + o] (+)= sum of dz[i][o]:  dz [m][N], a [m][K]
__global__ void __launch_bounds__(256) linear_backward_weight_kernel(const float* __restrict__ dz, const float* __restrict__ a, int64_t m, int N, int K,
                                                                     int rows_per_slice, float* __restrict__ slots, int64_t slot_stride, int accumulate) {
    const int s = blockIdx.x;
    const int o0 = blockIdx.y * TM, k0 = blockIdx.z * TN;
    const int64_t r0 = (int64_t)s * rows_per_slice;
    const int64_t r1 = r0 + rows_per_slice < m ? r0 + rows_per_slice : m;
    f32x16v acc = {};
    float dbacc = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool do_db = blockIdx.z == 0 && (wave & 1) == 0;          // the waves of column block 0: each A operand (a dZ value) counted once
    if (r0 < r1)
        tile_product<false, false>(0, (int)(r1 - r0),
            [&](int i, int r) { return o0 + i < N ? dz[(r0 + r) * N + o0 + i] : 0.f; },
            [&](int j, int r) { return k0 + j < K ? a[(r0 + r) * K + k0 + j] : 0.f; }, acc, do_db ? &dbacc : nullptr);
    float* slot = slots + (int64_t)s * slot_stride;
    const int k = k0 + tile_col();
    if (k < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + tile_row(r);
            if (o >= N) continue;
            float* d = slot + (int64_t)o * K + k;
            *d = accumulate ? *d + acc[r] : acc[r];
        }
    }
    if (do_db) {                                                    // lane (i, half) summed the rows of k parity `half`: add the two halves
        const float tot = dbacc + __shfl_down(dbacc, 32);
        const int o = o0 + 32 * (wave >> 1) + lane;
        if (lane < 32 && o < N) {
            float* d = slot + (int64_t)N * K + o;
            *d = accumulate ? *d + tot : tot;
        }
    }
}

// dW[o][k] = sum_s slot[s][o K + k], db[o] = sum_s slot[s][N K + o], fixed order
__global__ void __launch_bounds__(256) reduce_slots_kernel(const float* __restrict__ slots, int n_slots, int64_t slot_stride, int N, int K,
                                                           float* __restrict__ dW, float* __restrict__ db) {
    const int64_t total = (int64_t)N * K + N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= n_slots; s += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] += slots[(int64_t)(s + j) * slot_stride + e];
        }
        for (; s < n_slots; ++s) part[0] += slots[(int64_t)s * slot_stride + e];
        const float v = (part[0] + part[1]) + (part[2] + part[3]);
        if (e < (int64_t)N * K) { if (dW) dW[e] = v; }
        else if (db) db[e - (int64_t)N * K] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side: the chunk plan and the launches
// ---------------------------------------------------------------------------------------------------
struct Plan {
    int64_t rows;                 // samples per chunk
    int n_slots, rows_per_slice;  // weight-gradient slices of a chunk
    int64_t act_floats;           // one [rows][H] activation buffer
    int64_t off_a, off_d, off_dz, off_slots, slot_stride;
    int64_t slot_layer[NIC_MAX_LINEAR];
    int64_t total_floats;
};

inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }

inline Plan make_plan(int64_t n, int cin, int H, int NL, bool training) {
    Plan p = {};
    const int64_t per_row = (int64_t)(training ? 2 * (NL - 1) + 2 : 2) * (H > 3 ? H : 3);
    int64_t rows = ((int64_t)64 << 20) / per_row;                     // <= 256 MiB of activations per chunk: L2 / Infinity Cache sized
    rows = rows / 1024 * 1024;
    if (rows < 1024) rows = 1024;
    if (rows > 131072) rows = 131072;
    if (rows > round_up(n > 0 ? n : 1, 64)) rows = round_up(n > 0 ? n : 1, 64);
    p.rows = rows;
    p.act_floats = rows * (H > 3 ? H : 3);
    p.off_a = 0;
    p.off_d = p.off_a + (training ? NL - 1 : 2) * p.act_floats;       // inference: two ping-pong buffers, no derivatives
    p.off_dz = p.off_d + (training ? NL - 1 : 0) * p.act_floats;
    p.off_slots = p.off_dz + (training ? 2 : 0) * p.act_floats;
    p.n_slots = (int)((rows + 255) / 256);                            // slices of >= 256 rows: enough workgroups in flight for the narrow layers (H = 64: one tile per slice)
    if (p.n_slots > 512) p.n_slots = 512;
    p.rows_per_slice = (int)round_up((rows + p.n_slots - 1) / p.n_slots, TK);
    int64_t off = 0;
    for (int l = 0; l < NL; ++l) {
        const int K = l == 0 ? cin : H, N = l == NL - 1 ? 3 : H;
        p.slot_layer[l] = off;
        off += round_up((int64_t)N * K + N, 4);
    }
    p.slot_stride = off;
    p.total_floats = p.off_slots + (training ? (int64_t)p.n_slots * p.slot_stride : 0);
    return p;
}

inline bool mlp_general_ok(const nic_mlp* m, int& NL) {
    if (!m) return false;
    NL = m->n_linear == 0 ? 3 : m->n_linear;
    if (NL < 2 || NL > NIC_MAX_LINEAR) return false;
    for (int i = 0; i < NL; ++i)
        if (!m->w[i] || !m->b[i]) return false;
    return true;
}

// forward pass of one chunk; training: A_k / D_k of every hidden activation stay in the workspace and the output layer leaves dZ_out in dzout
inline int forward_chunk(const Plan& p, const nic_mlp* mlp, int NL, const float* x, int64_t m, int cin, int H, float* ws, bool training, float* y,
                         const float* dy, float* dzout, hipStream_t s) {
    const dim3 blk(256);
    const float* in = x;
    int K = cin;
    for (int l = 0; l < NL - 1; ++l) {
        float* a = ws + p.off_a + (training ? l : (l & 1)) * p.act_floats;
        float* d = training ? ws + p.off_d + l * p.act_floats : nullptr;
        const dim3 grid((unsigned)((m + TM - 1) / TM), (unsigned)((H + TN - 1) / TN));
        hipLaunchKernelGGL(linear_forward_kernel<ACT_GELU>, grid, blk, 0, s, in, mlp->w[l], mlp->b[l], m, H, K, a, d, (const float*)nullptr);
        in = a;
        K = H;
    }
    const dim3 grid((unsigned)((m + TM - 1) / TM), 1u);
    if (dzout) hipLaunchKernelGGL(linear_forward_kernel<ACT_SIGMOID_BWD>, grid, blk, 0, s, in, mlp->w[NL - 1], mlp->b[NL - 1], m, 3, K, dzout, (float*)nullptr, dy);
    else hipLaunchKernelGGL(linear_forward_kernel<ACT_SIGMOID>, grid, blk, 0, s, in, mlp->w[NL - 1], mlp->b[NL - 1], m, 3, K, y, (float*)nullptr, (const float*)nullptr);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Hidden widths below 64 on the FUSED kernels: a decoder with H hidden units is exactly the decoder with HP > H units whose extra units have
// zero weights and biases - their pre-activations are 0, GELU(0) = 0 feeds nothing, and with zero outgoing weights their gradients are
// exact zeros.  pad: [H, K] -> [HP, K'] (zero-filled); unpad: the real block of every padded gradient.  One launch each.
// ---------------------------------------------------------------------------------------------------
struct PadTable {
    const float* src[2 * NIC_MAX_LINEAR];   // W_0, b_0, W_1, b_1, ..
    float* dst[2 * NIC_MAX_LINEAR];
    int rows_s[2 * NIC_MAX_LINEAR], cols_s[2 * NIC_MAX_LINEAR];   // the SMALL shape (biases: cols 1)
    int rows_l[2 * NIC_MAX_LINEAR], cols_l[2 * NIC_MAX_LINEAR];   // the LARGE (padded) shape
    int first[2 * NIC_MAX_LINEAR + 1];      // first thread index of tensor t (over the shape the launch walks)
    int count;
};
// GROW: walk the large tensors, read the small ones (zero outside); else walk the small tensors, read the large ones
template <bool GROW>
__global__ void __launch_bounds__(256) pad_kernel(const PadTable t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.first[t.count]) return;
    int k = 0;
    while (k + 1 < t.count && i >= t.first[k + 1]) ++k;
    const int e = i - t.first[k];
    const int cw = GROW ? t.cols_l[k] : t.cols_s[k];
    const int r = e / cw, c = e - r * cw;
    if (GROW) t.dst[k][e] = (r < t.rows_s[k] && c < t.cols_s[k]) ? t.src[k][r * t.cols_s[k] + c] : 0.f;
    else t.dst[k][e] = t.src[k][r * t.cols_l[k] + c];
}

inline int pad_launch(bool grow, const float* const* small_w, const float* const* small_b, float* const* small_w_out, float* const* small_b_out,
                      const float* const* large_w, const float* const* large_b, float* const* large_w_out, float* const* large_b_out, int NL, int cin, int H,
                      int HP, hipStream_t s) {
    PadTable t = {};
    int n = 0, total = 0;
    for (int l = 0; l < NL; ++l) {
        const int Ks = l == 0 ? cin : H, Kl = l == 0 ? cin : HP, Ns = l == NL - 1 ? 3 : H, Nl = l == NL - 1 ? 3 : HP;
        for (int b = 0; b < 2; ++b) {
            const float* src = grow ? (b ? small_b[l] : small_w[l]) : (b ? large_b[l] : large_w[l]);
            float* dst = grow ? (b ? large_b_out[l] : large_w_out[l]) : (b ? small_b_out[l] : small_w_out[l]);
            if (!dst) continue;                                   // a gradient the caller does not want
            if (!src) return NIC_E_NULL;
            t.src[n] = src; t.dst[n] = dst;
            t.rows_s[n] = Ns; t.cols_s[n] = b ? 1 : Ks;
            t.rows_l[n] = Nl; t.cols_l[n] = b ? 1 : Kl;
            t.first[n] = total;
            total += grow ? t.rows_l[n] * t.cols_l[n] : t.rows_s[n] * t.cols_s[n];
            ++n;
        }
    }
    t.first[n] = total;
    t.count = n;
    if (n == 0) return NIC_OK;
    if (grow) hipLaunchKernelGGL(pad_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, s, t);
    else hipLaunchKernelGGL(pad_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, s, t);
    return (int)hipGetLastError();
}

}  // namespace general
}  // namespace nic

using namespace nic;
using namespace nic::general;

extern "C" {

size_t nic_decoder_general_workspace_bytes(int64_t n, int cin, int hidden, int n_linear, int training) {
    if (n < 0 || cin < 1 || hidden < 1 || n_linear < 2 || n_linear > NIC_MAX_LINEAR) return 0;
    return (size_t)make_plan(n, cin, hidden, n_linear, training != 0).total_floats * sizeof(float);
}

int nic_decoder_general_forward(const nic_mlp* mlp, const float* x, int64_t n, int cin, int hidden, float* y, void* workspace,
                                size_t workspace_bytes, void* stream) {
    int NL = 0;
    if (!mlp_general_ok(mlp, NL) || !x || !y || !workspace) return NIC_E_NULL;
    if (n < 0 || cin < 1 || hidden < 1) return NIC_E_ARG;
    if (n == 0) return NIC_OK;
    const Plan p = make_plan(n, cin, hidden, NL, false);
    if (workspace_bytes < (size_t)p.total_floats * sizeof(float)) return NIC_E_WORKSPACE;
    for (int64_t r0 = 0; r0 < n; r0 += p.rows) {
        const int64_t m = n - r0 < p.rows ? n - r0 : p.rows;
        const int rc = forward_chunk(p, mlp, NL, x + r0 * cin, m, cin, hidden, (float*)workspace, false, y + r0 * 3, nullptr, nullptr, (hipStream_t)stream);
        if (rc) return rc;
    }
    return NIC_OK;
}

int nic_decoder_general_backward(const nic_mlp* mlp, const float* x, const float* dy, int64_t n, int cin, int hidden, float* dx,
                                 const nic_mlp_grads* grads, void* workspace, size_t workspace_bytes, void* stream) {
    int NL = 0;
    if (!mlp_general_ok(mlp, NL) || !x || !dy || !grads || !workspace) return NIC_E_NULL;
    if (n <= 0 || cin < 1 || hidden < 1) return NIC_E_ARG;
    const int H = hidden;
    const Plan p = make_plan(n, cin, H, NL, true);
    if (workspace_bytes < (size_t)p.total_floats * sizeof(float)) return NIC_E_WORKSPACE;
    float* ws = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const dim3 blk(256);
    float* slots = ws + p.off_slots;
    bool first = true;
    for (int64_t r0 = 0; r0 < n; r0 += p.rows, first = false) {
        const int64_t m = n - r0 < p.rows ? n - r0 : p.rows;
        float* dz_cur = ws + p.off_dz;                     // dZ of the layer at hand: [m][3] for the output layer, [m][H] below
        float* dz_next = ws + p.off_dz + p.act_floats;
        int rc = forward_chunk(p, mlp, NL, x + r0 * cin, m, cin, H, ws, true, nullptr, dy + r0 * 3, dz_cur, s);
        if (rc) return rc;
        const int n_slices = (int)((m + p.rows_per_slice - 1) / p.rows_per_slice);
        for (int l = NL - 1; l >= 0; --l) {
            const int K = l == 0 ? cin : H, N = l == NL - 1 ? 3 : H;
            const float* a_in = l == 0 ? x + r0 * cin : ws + p.off_a + (l - 1) * p.act_floats;
            // a chunk with fewer slices than the plan's leaves the other slots untouched: they were written (or zeroed) by the first chunk
            {
                const dim3 grid((unsigned)(first ? p.n_slots : n_slices), (unsigned)((N + TM - 1) / TM), (unsigned)((K + TN - 1) / TN));
                hipLaunchKernelGGL(linear_backward_weight_kernel, grid, blk, 0, s, (const float*)dz_cur, a_in, m, N, K, p.rows_per_slice,
                                   slots + p.slot_layer[l], p.slot_stride, first ? 0 : 1);
            }
            if (l > 0 || dx) {
                const float* D = l > 0 ? ws + p.off_d + (l - 1) * p.act_floats : nullptr;
                float* out = l > 0 ? dz_next : dx + r0 * cin;
                const dim3 grid((unsigned)((m + TM - 1) / TM), (unsigned)((K + TN - 1) / TN));
                hipLaunchKernelGGL(linear_backward_input_kernel, grid, blk, 0, s, (const float*)dz_cur, mlp->w[l], D, m, N, K, out);
            }
            float* t = dz_cur; dz_cur = dz_next; dz_next = t;
        }
        { const int e = (int)hipGetLastError(); if (e) return e; }
    }
    for (int l = 0; l < NL; ++l) {
        const int K = l == 0 ? cin : H, N = l == NL - 1 ? 3 : H;
        if (!grads->w[l] && !grads->b[l]) continue;
        const int64_t total = (int64_t)N * K + N;
        const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
        hipLaunchKernelGGL(reduce_slots_kernel, dim3(grid), blk, 0, s, (const float*)(slots + p.slot_layer[l]), p.n_slots, p.slot_stride, N, K, grads->w[l],
                           grads->b[l]);
    }
    return (int)hipGetLastError();
}

int nic_decoder_pad(const nic_mlp* src, int cin, int hidden, int hidden_padded, const nic_mlp_grads* dst, void* stream) {
    int NL = 0;
    if (!mlp_general_ok(src, NL) || !dst) return NIC_E_NULL;
    if (cin < 1 || hidden < 1 || hidden_padded < hidden) return NIC_E_ARG;
    for (int l = 0; l < NL; ++l)
        if (!dst->w[l] || !dst->b[l]) return NIC_E_NULL;
    return pad_launch(true, src->w, src->b, nullptr, nullptr, nullptr, nullptr, dst->w, dst->b, NL, cin, hidden, hidden_padded, (hipStream_t)stream);
}

int nic_decoder_unpad(const nic_mlp_grads* src_padded, int n_linear, int cin, int hidden, int hidden_padded, const nic_mlp_grads* dst, void* stream) {
    if (!src_padded || !dst) return NIC_E_NULL;
    if (n_linear < 2 || n_linear > NIC_MAX_LINEAR || cin < 1 || hidden < 1 || hidden_padded < hidden) return NIC_E_ARG;
    return pad_launch(false, nullptr, nullptr, dst->w, dst->b, src_padded->w, src_padded->b, nullptr, nullptr, n_linear, cin, hidden, hidden_padded,
                      (hipStream_t)stream);
}

}  // extern "C"
