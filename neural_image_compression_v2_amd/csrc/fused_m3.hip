// fused kernels for Layout<3> (see nic_device.hpp)
#include "fused_launch.hpp"
namespace nic {
NIC_INSTANTIATE_LAYOUT(3)
}
