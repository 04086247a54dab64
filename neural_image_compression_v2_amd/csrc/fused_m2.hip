// fused kernels for Layout<2> (2D, sinusoidal PE; see nic_device.hpp)
#include "fused_launch.hpp"
namespace nic {
NIC_INSTANTIATE_LAYOUT(2)
}
