#!/bin/bash
# builds ab/lib<NAME>.so from the working tree with extra -D flags for the fused kernels: ab/mk.sh NAME [-DFOO ...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
C=neural_image_compression_v2_amd/csrc
python -m neural_image_compression_v2_amd._build >/dev/null      # simple_kernels.o / fused_capi.o
T=/tmp/ab_$NAME; mkdir -p $T
for m in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function "$@" -c $C/fused_m$m.hip -o $T/fused_m$m.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$NAME.so $C/build/simple_kernels.o $C/build/fused_capi.o $T/fused_m1.o $T/fused_m2.o $T/fused_m3.o $T/fused_m4.o $C/build/fused_t16.o $C/build/fused_mlpn.o
echo built ab/lib$NAME.so
