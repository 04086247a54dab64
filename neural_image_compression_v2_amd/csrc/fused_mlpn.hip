// depth-generic 2D fused kernels (fused_mlpn.hpp): NL = 5 (the "4 x 64" decoder) for training and decode; NL = 3 is instantiated too,
// as the cross-check of the layer loop against the dedicated 3-layer kernels (NIC_FLAG_MLPN).
#include "fused_mlpn.hpp"
#include "fused_t16.hpp"

namespace nic {

template <class L, int NL>
static int launch_n(int mode, const FusedParams& p, int grid, hipStream_t s) {
    const dim3 g(grid), b(256);
    if (mode == MODE_INFER) hipLaunchKernelGGL((fused_mlpn_kernel<L, MODE_INFER, NL>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_mlpn_kernel<L, MODE_TRAIN_MSE, NL>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_IMG) hipLaunchKernelGGL((fused_mlpn_kernel<L, MODE_TRAIN_IMG, NL>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_DY) hipLaunchKernelGGL((fused_mlpn_kernel<L, MODE_TRAIN_DY, NL>), g, b, 0, s, p);
    else return NIC_E_UNSUPPORTED;
    return (int)hipGetLastError();
}

int launch_mlpn(int layout, int n_linear, int mode, const FusedParams& p, int grid, hipStream_t s) {
    if (n_linear == 5) return layout == 1 ? launch_n<Layout<1>, 5>(mode, p, grid, s) : launch_n<Layout<2>, 5>(mode, p, grid, s);
    if (n_linear == 3) return layout == 1 ? launch_n<Layout<1>, 3>(mode, p, grid, s) : launch_n<Layout<2>, 3>(mode, p, grid, s);
    return NIC_E_UNSUPPORTED;
}

template <int NL>
static int reduce_n(int layout, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
    constexpr int n_out = kH * 73 + kH + (NL - 2) * (kH * kH + kH) + 3 * kH + 3 + 1;
    const TailLaunch t = tail_for((n_out + 31) / 32);
    const dim3 grid(t.blocks), block(256);
    if (layout == 1) hipLaunchKernelGGL((reducen_kernel<Layout<1>, NL>), grid, block, 0, s, partials, n_rec, g, loss, loss_scale, t.tl);
    else hipLaunchKernelGGL((reducen_kernel<Layout<2>, NL>), grid, block, 0, s, partials, n_rec, g, loss, loss_scale, t.tl);
    return (int)hipGetLastError();
}
int launch_reducen(int layout, int n_linear, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
    return n_linear == 5 ? reduce_n<5>(layout, partials, n_rec, g, loss, loss_scale, s) : reduce_n<3>(layout, partials, n_rec, g, loss, loss_scale, s);
}
int mlpn_record_floats(int n_linear) { return n_linear == 5 ? LdsN<5>::REC : LdsN<3>::REC; }

}  // namespace nic
