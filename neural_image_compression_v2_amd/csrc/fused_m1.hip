// fused kernels for Layout<1> (2D, triangular PE; see nic_device.hpp)
#include "fused_launch.hpp"
namespace nic {
NIC_INSTANTIATE_LAYOUT(1)
}
