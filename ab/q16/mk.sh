#!/bin/bash
# builds ab/lib<NAME>.so from the working tree with extra -D flags for the q16 kernels: ab/q16/mk.sh NAME [-DFOO ...]
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
C=neural_image_compression_v2_amd/csrc
python -m neural_image_compression_v2_amd._build >/dev/null
T=/tmp/abq_$NAME; mkdir -p $T
for m in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -Wno-unused-function "$@" -c $C/fused_q$m.hip -o $T/fused_q$m.o &
done
wait
B=$C/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$NAME.so $B/simple_kernels.o $B/decoder_general.o $B/fused_capi.o $B/fused_m1.o $B/fused_m2.o $B/fused_m3.o $B/fused_m4.o $B/fused_t16.o $B/fused_mlpn.o $B/fused_qc_*.o $B/fused_ml_*.o $T/fused_q1.o $T/fused_q2.o $T/fused_q3.o $T/fused_q4.o
echo built ab/lib$NAME.so
