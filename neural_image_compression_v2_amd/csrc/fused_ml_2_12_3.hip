// multi-level plain-bf16 fused kernels (fused_q16.hpp::QML): 2 level pairs, FEATURE_PYRAMID_CHANNELS = 12, 3 Linear layers
#include "fused_q16_launch.hpp"
namespace nic {
NIC_INSTANTIATE_ML(2, 12, 3)
}
