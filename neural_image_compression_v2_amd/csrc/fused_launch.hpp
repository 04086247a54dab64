// Host-side launchers of the fused kernels; one translation unit per layout (fused_m1/m3/m4.hip) so the
// three big template instantiation sets compile in parallel.
#pragma once
#include "fused_kernel.hpp"

namespace nic {

struct FusedInfo {
    int nacc, rec, tx, ty, tz, cin, waves_per_rec;
};

template <int METHOD>
FusedInfo fused_info();
template <int METHOD>
int launch_fused(int src, int mode, const FusedParams& p, int grid, hipStream_t s);
template <int METHOD>
int launch_reduce(const float* partials, int n_waves, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s);

#define NIC_INSTANTIATE_LAYOUT(METHOD)                                                                                   \
    template <>                                                                                                          \
    FusedInfo fused_info<METHOD>() {                                                                                     \
        using L = Layout<METHOD>;                                                                                        \
        return FusedInfo{Lds<L>::NACC, Lds<L>::REC, L::TX, L::TY, L::TZ, L::CIN, 4};                                        \
    }                                                                                                                    \
    template <>                                                                                                          \
    int launch_fused<METHOD>(int src, int mode, const FusedParams& p, int grid, hipStream_t s) {                         \
        using L = Layout<METHOD>;                                                                                        \
        const dim3 g(grid), b(256);                                                                                      \
        if (src == SRC_ENCODE) {                                                                                         \
            if (mode == MODE_INFER && (p.d.flags & NIC_FLAG_SPLIT_BF16) && L::NSLOT % 8 == 0) {   /* 2D and 3D method 3 */ \
                if constexpr (L::NSLOT % 8 == 0) {                                                                       \
                    if (p.grid_u8) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_INFER, uint8_t, PREC_SPLIT>), g, b, 0, s, p); \
                    else hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_INFER, float, PREC_SPLIT>), g, b, 0, s, p); \
                }                                                                                                        \
            }                                                                                                            \
            else if (mode == MODE_INFER && p.grid_u8) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_INFER, uint8_t>), g, b, 0, s, p); \
            else if (p.grid_u8) return NIC_E_UNSUPPORTED;                                                               \
            else if (mode == MODE_INFER) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_INFER>), g, b, 0, s, p);   \
            else if (p.d.flags & NIC_FLAG_SPLIT_BF16) {                          /* training kernels, 2D layouts */      \
                if constexpr (L::NSLOT % 8 == 0 && L::DIM == 2) {                                                        \
                    if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_MSE, float, PREC_SPLIT>), g, b, 0, s, p); \
                    else if (mode == MODE_TRAIN_IMG) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_IMG, float, PREC_SPLIT>), g, b, 0, s, p); \
                    else hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_DY, float, PREC_SPLIT>), g, b, 0, s, p); \
                } else if constexpr (L::NSLOT % 8 == 0) {     /* 3D: chained products only (PREC_CHAIN) */              \
                    if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_MSE, float, PREC_CHAIN>), g, b, 0, s, p); \
                    else if (mode == MODE_TRAIN_IMG) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_IMG, float, PREC_CHAIN>), g, b, 0, s, p); \
                    else hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_DY, float, PREC_CHAIN>), g, b, 0, s, p); \
                } else return NIC_E_UNSUPPORTED;                                                                        \
            }                                                                                                            \
            else if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_MSE>), g, b, 0, s, p); \
            else if (mode == MODE_TRAIN_IMG) hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_IMG>), g, b, 0, s, p); \
            else hipLaunchKernelGGL((fused_kernel<L, SRC_ENCODE, MODE_TRAIN_DY>), g, b, 0, s, p);                        \
        } else {                                                                                                         \
            if (mode == MODE_INFER) hipLaunchKernelGGL((fused_kernel<L, SRC_MEMORY, MODE_INFER>), g, b, 0, s, p);        \
            else if (mode == MODE_TRAIN_DY) hipLaunchKernelGGL((fused_kernel<L, SRC_MEMORY, MODE_TRAIN_DY>), g, b, 0, s, p); \
            else return NIC_E_UNSUPPORTED;                                                                               \
        }                                                                                                                \
        return (int)hipGetLastError();                                                                                   \
    }                                                                                                                    \
    template <>                                                                                                          \
    int launch_reduce<METHOD>(const float* partials, int n_waves, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) { \
        using L = Layout<METHOD>;                                                                                        \
        const int n = Lds<L>::NACC * 1024 + Lds<L>::TAIL;                                                                          \
        const TailLaunch t = tail_for((n + 31) / 32);                                                                   \
        hipLaunchKernelGGL((reduce_partials_kernel<L>), dim3(t.blocks), dim3(256), 0, s, partials, n_waves, g, loss, loss_scale, t.tl); \
        return (int)hipGetLastError();                                                                                   \
    }

}  // namespace nic
