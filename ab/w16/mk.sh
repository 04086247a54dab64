#!/bin/bash
# builds ab/lib<NAME>.so from the working tree with extra -D flags for the 16-sample training kernel: ab/w16/mk.sh NAME [-DFOO ...]
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
C=neural_image_compression_v2_amd/csrc
python -m neural_image_compression_v2_amd._build >/dev/null
T=/tmp/ab_$NAME; mkdir -p $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c $C/fused_t16.hip -o $T/fused_t16.o 2>&1 | grep -A12 "kernelINS_6LayoutILi1EEELi1EEE" | grep -E "VGPRs:|Scratch|Spill" | sed 's/.*remark: *//' | tr '\n' ' '; echo
B=$C/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$NAME.so $B/simple_kernels.o $B/fused_capi.o $B/fused_m1.o $B/fused_m2.o $B/fused_m3.o $B/fused_m4.o $B/fused_mlpn.o $T/fused_t16.o
echo built ab/lib$NAME.so
