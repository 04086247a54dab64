// torch.optim.Adam single-tensor semantics (default: no weight decay, no amsgrad, eps outside sqrt after bias correction) + the
// fp_quantize_clamp that follows it (image_compression.py:266-269), as device code shared by the optimiser launch (simple_kernels.hip:
// nic_adam_multi) and by the TAIL of a fused training step (nic_path_desc.tail): the reduction of the per-workgroup decoder-gradient
// records and the Adam update of every parameter of the step in ONE launch - reduce blocks first (latency-bound: a few hundred blocks
// walking the record list), Adam blocks over the grids behind them (HBM-bound); the decoder's parameters are updated by the very threads
// that finish their gradient.
#pragma once
#include "nic_device.hpp"
#include <math.h>

namespace nic {

// torch.clamp_ (fp_quantize_clamp, fp_def.py:227-232) propagates NaN; fminf / fmaxf would map a diverged parameter to `lo`
__device__ __forceinline__ float clamp_keep_nan(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }

// Multi-tensor form: one launch for the whole parameter list.  A block owns a contiguous 4096-element chunk of one tensor
// (found by walking the short prefix table), float4 where the chunk is 16-byte aligned in all four arrays.
struct AdamEntry {
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float step_size, bc2_sqrt, lo, hi;
    int first_block;
    uint16_t* p16;       // optional 16-bit mirror of p (NIC_FLAG_GRID_BF16 / _FP16 storage), rewritten with the rounded new value
    int p16_kind;
    int zero_g;          // NIC_ADAM_ZERO_GRAD: the gradient is zeroed once read (an atomically accumulated bucket is clean for the next step)
    int rep_blocks;      // blocks per run; runs of n elements rep_pg (p, g, p16) / rep_mv (m, v) elements apart (nic_adam_tensor.reps)
    int64_t rep_pg, rep_mv;
};
__device__ __forceinline__ uint16_t to_store16(float x, int kind) {
    if (kind == 1) return __builtin_bit_cast(uint16_t, (__bf16)x);               // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(uint16_t, (_Float16)x);
}
struct AdamTable {
    AdamEntry e[NIC_ADAM_MAX_TENSORS];
    int count;
    float b2, omb1, omb2, eps;
    // hipGraph-captured loops (nic_adam_multi_dev): row *step_dev of a device table [rows][4] = {step_size of column 0, step_size of column 1,
    // sqrt(bias_correction2), -} replaces the per-launch scalars of entry k (column sched_col[k])
    const float* sched;
    const int64_t* step_dev;
    int64_t sched_rows;
    unsigned char sched_col[NIC_ADAM_MAX_TENSORS];
};
// The scalars are formed on the host in double and cast once, like torch does with its Python-float hyper-parameters:
// omb1 = (float)(1 - beta1), omb2 = (float)(1 - beta2), step_size = (float)(lr / bias_correction1), bc2_sqrt = (float)sqrt(bias_correction2)
// (1.0f - 0.999f in fp32 is 9.99987e-4, not 0.001f: exp_avg_sq would drift 1.3e-5 low).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamTable& t, float step_size, float bc2_sqrt,
                                         float lo, float hi) {
    // no FMA contraction: torch's eager Adam rounds every product and sum on its own, and the update must not depend on the kernel this function is
    // inlined into (the optimiser launch, its float4 path, the tail of a reduction: tests/test_gpu_tail.py holds them to the same bits)
#pragma clang fp contract(off)
    m = m + (g - m) * t.omb1;                                          // exp_avg.lerp_(grad, 1 - beta1)
    v = v * t.b2 + t.omb2 * g * g;                                     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(v) / bc2_sqrt + t.eps;
    float x = p - step_size * (m / denom);
    if (lo <= hi) x = clamp_keep_nan(x, lo, hi);
    p = x;
}
// entry k with its per-step scalars (the device schedule's row when there is one)
__device__ __forceinline__ AdamEntry adam_entry(const AdamTable& t, int k) {
    AdamEntry e = t.e[k];
    if (t.sched != nullptr) {
        int64_t row = *t.step_dev;
        row = row < 0 ? 0 : (row >= t.sched_rows ? t.sched_rows - 1 : row);
        e.step_size = t.sched[4 * row + t.sched_col[k]];
        e.bc2_sqrt = t.sched[4 * row + 2];
    }
    return e;
}
constexpr int kAdamChunk = 4096;
// block `blk` (256 threads) of the launch over the first `count` entries of the table
__device__ __forceinline__ void adam_block(const AdamTable& t, int count, int blk) {
    int k = 0;
    while (k + 1 < count && blk >= t.e[k + 1].first_block) ++k;
    const AdamEntry e = adam_entry(t, k);
    const int rb = blk - e.first_block, rep = rb / e.rep_blocks;
    const int64_t base = (int64_t)(rb - rep * e.rep_blocks) * kAdamChunk;
    const int64_t left = e.n - base;
    const int cnt = left < kAdamChunk ? (int)left : kAdamChunk;
    const int64_t base_pg = base + rep * e.rep_pg, base_mv = base + rep * e.rep_mv;
    float* p = e.p + base_pg; const float* g = e.g + base_pg; float* m = e.m + base_mv; float* v = e.v + base_mv;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (vec) {
        const int n4 = cnt >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            adam_one(pp.x, gg.x, mm.x, vv.x, t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            adam_one(pp.y, gg.y, mm.y, vv.y, t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            adam_one(pp.z, gg.z, mm.z, vv.z, t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            adam_one(pp.w, gg.w, mm.w, vv.w, t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
            if (e.zero_g) reinterpret_cast<float4*>(const_cast<float*>(g))[i] = float4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = 4 * n4 + threadIdx.x; i < cnt; i += 256) {
            adam_one(p[i], g[i], m[i], v[i], t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            if (e.zero_g) const_cast<float*>(g)[i] = 0.f;
        }
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) {
            adam_one(p[i], g[i], m[i], v[i], t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            if (e.zero_g) const_cast<float*>(g)[i] = 0.f;
        }
    }
    if (e.p16 != nullptr) {                                       // the block re-reads its own chunk of the master (its own stores: visible to it)
        __syncthreads();
        uint16_t* q = e.p16 + base_pg;
        for (int i = threadIdx.x; i < cnt; i += 256) q[i] = to_store16(p[i], e.p16_kind);
    }
}

// ---- the tail of a fused training step (nic_path_desc.tail).  Entries [0, n_stream) of the table are streamed by the blocks behind the
// reduction's own (`reduce_blocks`); entries [n_stream, count) are the decoder's tensors: their .g are the buffers the reduction writes, and
// the thread that stores element i of one of them updates parameter i right there.
struct StepTail {
    AdamTable t;
    int n_stream;
    int reduce_blocks;         // 0x7fffffff when there is no tail: every block is a reduce block
};
__device__ __forceinline__ bool tail_block(const StepTail& tl) {
    if ((int)blockIdx.x < tl.reduce_blocks) return false;
    adam_block(tl.t, tl.n_stream, (int)blockIdx.x - tl.reduce_blocks);
    return true;
}
// *dst = g (a finished decoder gradient) + the Adam update of the parameter it belongs to
__device__ __forceinline__ void tail_store(const StepTail& tl, float* dst, float g) {
    *dst = g;
    for (int k = tl.n_stream; k < tl.t.count; ++k) {
        const float* base = tl.t.e[k].g;
        if (dst >= base && dst < base + tl.t.e[k].n) {
            const AdamEntry e = adam_entry(tl.t, k);
            const int64_t i = dst - base;
            adam_one(e.p[i], g, e.m[i], e.v[i], tl.t, e.step_size, e.bc2_sqrt, e.lo, e.hi);
            if (e.p16 != nullptr) e.p16[i] = to_store16(e.p[i], e.p16_kind);
            return;
        }
    }
}

// the launch shape of a reduction with the tail of the call in progress attached: `reduce_blocks` blocks of the reduction + the streaming blocks.
// tail_for() is defined in fused_capi.hip: the training entry point parks its StepTail in a thread-local slot around its reduce dispatch, so the
// dozen layout-specific launchers in between need no extra parameter (no tail parked: a plain reduction).
struct TailLaunch {
    StepTail tl;
    unsigned blocks;
};
TailLaunch tail_for(int reduce_blocks);

// ---- host side: nic_adam_tensor list -> AdamTable.  Entries [0, n_stream) get chunk blocks (returned in `blocks`), the rest none.
inline int adam_build_table(const nic_adam_tensor* tensors, int count, int n_stream, double beta1, double beta2, double eps, const float* sched,
                            int64_t sched_rows, const int64_t* step_dev, AdamTable& t, int& n_stream_out, int64_t& blocks) {
    t.b2 = (float)beta2; t.omb1 = (float)(1.0 - beta1); t.omb2 = (float)(1.0 - beta2); t.eps = (float)eps;
    int nt = 0;
    blocks = 0;
    n_stream_out = 0;
    for (int i = 0; i < count; ++i) {
        const nic_adam_tensor& a = tensors[i];
        if (a.n == 0) continue;
        if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq) return NIC_E_NULL;
        if (a.n < 0 || (a.step < 1 && !sched)) return NIC_E_ARG;
        const double bc1 = 1.0 - pow(beta1, (double)a.step);
        const double bc2 = 1.0 - pow(beta2, (double)a.step);
        AdamEntry& e = t.e[nt++];
        e.p = a.param; e.g = a.grad; e.m = a.exp_avg; e.v = a.exp_avg_sq; e.n = a.n;
        e.step_size = (float)(a.lr / bc1);        // formed in double like torch's Python-float step_size, cast once
        e.bc2_sqrt = (float)sqrt(bc2);
        e.lo = a.clamp_lo; e.hi = a.clamp_hi;
        e.p16 = (uint16_t*)a.param16; e.p16_kind = a.param16_kind;
        if (a.flags & ~(NIC_ADAM_ZERO_GRAD | NIC_ADAM_SCHED_COL1)) return NIC_E_ARG;
        e.zero_g = (a.flags & NIC_ADAM_ZERO_GRAD) ? 1 : 0;
        t.sched_col[nt - 1] = (a.flags & NIC_ADAM_SCHED_COL1) ? 1 : 0;
        if (e.p16 != nullptr && e.p16_kind != 1 && e.p16_kind != 2) return NIC_E_ARG;
        const int64_t reps = a.reps > 1 ? a.reps : 1;
        if (a.reps < 0 || (reps > 1 && (a.rep_stride < a.n || a.state_rep_stride < a.n))) return NIC_E_ARG;     // runs do not overlap
        const int64_t rep_blocks = (a.n + kAdamChunk - 1) / kAdamChunk;
        e.rep_blocks = (int)rep_blocks;
        e.rep_pg = reps > 1 ? a.rep_stride : 0;
        e.rep_mv = reps > 1 ? a.state_rep_stride : 0;
        if (i < n_stream) {
            e.first_block = (int)blocks;
            blocks += rep_blocks * reps;
            if (rep_blocks > 0x3fffffff || blocks > 0x3fffffff) return NIC_E_ARG;
            n_stream_out = nt;
        } else {
            e.first_block = 0x7fffffff;
            if (e.zero_g || reps > 1) return NIC_E_ARG;       // a decoder gradient is one contiguous tensor, overwritten by the next reduction
        }
    }
    t.count = nt;
    t.sched = sched; t.step_dev = step_dev; t.sched_rows = sched_rows;
    return NIC_OK;
}

}  // namespace nic
