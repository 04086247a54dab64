// host interface of the 8-wave / 16-sample 2D training kernels (fused_t16.hip)
#pragma once
#include "fused_kernel.hpp"

namespace nic {
int launch_train16(int layout, int mode, const FusedParams& p, int grid, hipStream_t s);
int launch_reduce16(int layout, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s);
int train16_record_floats();
// depth-generic kernels (fused_mlpn.hip)
int launch_mlpn(int layout, int n_linear, int mode, const FusedParams& p, int grid, hipStream_t s);
int launch_reducen(int layout, int n_linear, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s);
int mlpn_record_floats(int n_linear);
// plain-bf16 kernels, every layout (fused_q1.hip .. fused_q4.hip; layout id = QL<> method: 1, 2: 2D triangular / sinusoidal PE, 3, 4: the 3D methods)
template <int METHOD>
int launch_q16(int n_linear, int mode, const FusedParams& p, int grid, hipStream_t s);
template <int METHOD>
int reduce_q16(int n_linear, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s);
template <int METHOD>
int q16_record_floats(int n_linear);
// .. multi-level layouts QML<LV, C, 6, pe> (fused_ml_*.hip): LV level pairs per sample, n_linear NL
template <int LV, int C, int NL>
int launch_ml(int pe_mode, int mode, const FusedParams& p, int grid, hipStream_t s);
template <int LV, int C, int NL>
int reduce_ml(int pe_mode, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s);
template <int LV, int C, int NL>
int ml_record_floats();
// .. with non-default channel counts C = FEATURE_PYRAMID_CHANNELS, P = PE_CHANNELS (3 Linear layers; fused_qc_*.hip)
template <int METHOD, int C, int P>
int launch_q16_cp(int mode, const FusedParams& p, int grid, hipStream_t s);
template <int METHOD, int C, int P>
int reduce_q16_cp(const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s);
template <int METHOD, int C, int P>
int q16_record_floats_cp();
}  // namespace nic
