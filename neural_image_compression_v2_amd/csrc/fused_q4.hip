// plain-bf16 fused training kernels (fused_q16.hpp) for quarter layout QL<4>
#include "fused_q16_launch.hpp"
namespace nic {
NIC_INSTANTIATE_Q16(4)
}
