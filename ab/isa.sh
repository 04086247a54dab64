#!/bin/bash
# device ISA of one layout's fused kernels: register / scratch summary per kernel; asm left in /tmp/isa/fused_m$1.s
M=${1:-1}; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S "$@" \
  neural_image_compression_v2_amd/csrc/fused_m$M.hip -o /tmp/isa/fused_m$M.s || exit 1
grep -E "^_ZN3nic12fused_kernel.*:$|\.sgpr_count|\.vgpr_count|\.agpr_count|vgpr_spill|private_segment_fixed_size|^ +\.name: +_ZN3nic" /tmp/isa/fused_m$M.s | \
  awk '/\.name:/{n=$2} /agpr_count/{a=$2} /private_segment_fixed_size/{p=$2} /\.vgpr_count/{v=$2} /vgpr_spill/{print n, "vgpr", v, "agpr", a, "scratch", p, "spill", $2}'
