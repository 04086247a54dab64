// plain-bf16 fused kernels (fused_q16.hpp) for quarter layout QL<2> with FEATURE_PYRAMID_CHANNELS = 12, PE_CHANNELS = 8
#include "fused_q16_launch.hpp"
namespace nic {
NIC_INSTANTIATE_Q16_CP(2, 12, 8)
}
