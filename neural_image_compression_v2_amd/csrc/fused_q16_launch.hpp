// Explicit instantiations of fused_q16_kernel, one translation unit per layout (fused_q1.hip .. fused_q4.hip) so they compile in parallel.
#pragma once
#include "fused_q16.hpp"
#include "fused_t16.hpp"

namespace nic {

template <class Q, int NL, bool F16 = false>
static int launch_q16_nl(int mode, const FusedParams& p, int grid, hipStream_t s) {
    const dim3 g(grid), b(512);
    if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_q16_kernel<Q, MODE_TRAIN_MSE, NL, F16>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_IMG) hipLaunchKernelGGL((fused_q16_kernel<Q, MODE_TRAIN_IMG, NL, F16>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_DY) hipLaunchKernelGGL((fused_q16_kernel<Q, MODE_TRAIN_DY, NL, F16>), g, b, 0, s, p);
    else if (mode == MODE_INFER) hipLaunchKernelGGL((fused_q16_kernel<Q, MODE_INFER, NL, F16>), g, b, 0, s, p);
    else return NIC_E_UNSUPPORTED;
    return (int)hipGetLastError();
}
template <class Q, int NL>
static int reduce_q16_nl(const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
    constexpr int n_out = reduce_q16_outputs<Q, NL>();
    static_assert(256 % NIC_RQ_SLICES == 0, "256 threads per block (the tail's streaming blocks too: nic_adam.hpp)");
    constexpr int outs = 256 / NIC_RQ_SLICES;
    const TailLaunch t = tail_for((n_out + outs - 1) / outs);
    hipLaunchKernelGGL((reduce_q16_kernel<Q, NL>), dim3(t.blocks), dim3(256), 0, s, partials, n_rec, g, loss, loss_scale, t.tl);
    return (int)hipGetLastError();
}

#define NIC_INSTANTIATE_Q16(METHOD)                                                                                              \
    template <>                                                                                                                  \
    int launch_q16<METHOD>(int n_linear, int mode, const FusedParams& p, int grid, hipStream_t s) {                              \
        if (p.f16) return n_linear == 5 ? launch_q16_nl<QL<METHOD>, 5, true>(mode, p, grid, s) : launch_q16_nl<QL<METHOD>, 3, true>(mode, p, grid, s); \
        return n_linear == 5 ? launch_q16_nl<QL<METHOD>, 5>(mode, p, grid, s) : launch_q16_nl<QL<METHOD>, 3>(mode, p, grid, s);  \
    }                                                                                                                            \
    template <>                                                                                                                  \
    int reduce_q16<METHOD>(int n_linear, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) { \
        return n_linear == 5 ? reduce_q16_nl<QL<METHOD>, 5>(partials, n_rec, g, loss, loss_scale, s)                             \
                             : reduce_q16_nl<QL<METHOD>, 3>(partials, n_rec, g, loss, loss_scale, s);                            \
    }                                                                                                                            \
    template <>                                                                                                                  \
    int q16_record_floats<METHOD>(int n_linear) { return n_linear == 5 ? LdsQ<QL<METHOD>, 5>::REC : LdsQ<QL<METHOD>, 3>::REC; }

// non-default channel counts (FEATURE_PYRAMID_CHANNELS / PE_CHANNELS, var2.py:68-69): 3-layer decoder only; one translation unit per (method, C, P)
#define NIC_INSTANTIATE_Q16_CP(METHOD, C, P)                                                                                     \
    template <>                                                                                                                  \
    int launch_q16_cp<METHOD, C, P>(int mode, const FusedParams& p, int grid, hipStream_t s) {                                   \
        return launch_q16_nl<QL<METHOD, C, P>, 3>(mode, p, grid, s);                                                             \
    }                                                                                                                            \
    template <>                                                                                                                  \
    int reduce_q16_cp<METHOD, C, P>(const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) { \
        return reduce_q16_nl<QL<METHOD, C, P>, 3>(partials, n_rec, g, loss, loss_scale, s);                                      \
    }                                                                                                                            \
    template <>                                                                                                                  \
    int q16_record_floats_cp<METHOD, C, P>() { return LdsQ<QL<METHOD, C, P>, 3>::REC; }

// multi-level layouts (QML<LV, C, 6, pe>): the training step on a target tensor and the forward pass, both positional encodings; one translation unit per
// (LV, C, NL)
#define NIC_INSTANTIATE_ML(LV, C, NL)                                                                                            \
    template <>                                                                                                                  \
    int launch_ml<LV, C, NL>(int pe_mode, int mode, const FusedParams& p, int grid, hipStream_t s) {                             \
        const dim3 g(grid), b(512);                                                                                              \
        if (pe_mode == NIC_PE_TRIANGULAR) {                                                                                      \
            if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_q16_kernel<QML<LV, C, 6, NIC_PE_TRIANGULAR>, MODE_TRAIN_MSE, NL>), g, b, 0, s, p); \
            else if (mode == MODE_INFER) hipLaunchKernelGGL((fused_q16_kernel<QML<LV, C, 6, NIC_PE_TRIANGULAR>, MODE_INFER, NL>), g, b, 0, s, p);    \
            else return NIC_E_UNSUPPORTED;                                                                                       \
        } else {                                                                                                                 \
            if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_q16_kernel<QML<LV, C, 6, NIC_PE_SINUSOIDAL>, MODE_TRAIN_MSE, NL>), g, b, 0, s, p); \
            else if (mode == MODE_INFER) hipLaunchKernelGGL((fused_q16_kernel<QML<LV, C, 6, NIC_PE_SINUSOIDAL>, MODE_INFER, NL>), g, b, 0, s, p);    \
            else return NIC_E_UNSUPPORTED;                                                                                       \
        }                                                                                                                        \
        return (int)hipGetLastError();                                                                                           \
    }                                                                                                                            \
    template <>                                                                                                                  \
    int reduce_ml<LV, C, NL>(int pe_mode, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) { \
        return pe_mode == NIC_PE_TRIANGULAR ? reduce_q16_nl<QML<LV, C, 6, NIC_PE_TRIANGULAR>, NL>(partials, n_rec, g, loss, loss_scale, s)     \
                                            : reduce_q16_nl<QML<LV, C, 6, NIC_PE_SINUSOIDAL>, NL>(partials, n_rec, g, loss, loss_scale, s);    \
    }                                                                                                                            \
    template <>                                                                                                                  \
    int ml_record_floats<LV, C, NL>() { return LdsQ<QML<LV, C, 6, NIC_PE_TRIANGULAR>, NL>::REC; }

}  // namespace nic
