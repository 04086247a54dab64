// the 8-wave / 16-sample 2D training kernels (fused_train16.hpp): Layout<1> (triangular PE) and Layout<2> (sinusoidal PE)
#include "fused_train16.hpp"
#include "fused_t16.hpp"

namespace nic {

template <class L>
static int launch_t16(int mode, const FusedParams& p, int grid, hipStream_t s) {
    const dim3 g(grid), b(512);
    if (mode == MODE_TRAIN_MSE) hipLaunchKernelGGL((fused_train16_kernel<L, MODE_TRAIN_MSE>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_IMG && p.timg_u8 == 2) hipLaunchKernelGGL((fused_train16_kernel<L, MODE_TRAIN_RGBX>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_IMG) hipLaunchKernelGGL((fused_train16_kernel<L, MODE_TRAIN_IMG>), g, b, 0, s, p);
    else if (mode == MODE_TRAIN_DY) hipLaunchKernelGGL((fused_train16_kernel<L, MODE_TRAIN_DY>), g, b, 0, s, p);
    else return NIC_E_UNSUPPORTED;
    return (int)hipGetLastError();
}

int launch_train16(int layout, int mode, const FusedParams& p, int grid, hipStream_t s) {
    return layout == 1 ? launch_t16<Layout<1>>(mode, p, grid, s) : launch_t16<Layout<2>>(mode, p, grid, s);
}

int launch_reduce16(int layout, const float* partials, int n_rec, nic_mlp_grads g, float* loss, float loss_scale, hipStream_t s) {
    constexpr int outs = 256 / NIC_R16_SLICES;
    const TailLaunch t = tail_for((kR16_SRC + outs - 1) / outs);
    const dim3 grid(t.blocks), block(256);
    if (layout == 1) hipLaunchKernelGGL((reduce16_kernel<Layout<1>>), grid, block, 0, s, partials, n_rec, g, loss, loss_scale, t.tl);
    else hipLaunchKernelGGL((reduce16_kernel<Layout<2>>), grid, block, 0, s, partials, n_rec, g, loss, loss_scale, t.tl);
    return (int)hipGetLastError();
}

int train16_record_floats() { return Lds16::REC; }

}  // namespace nic
