#!/bin/bash
# builds ab/lib<NAME>.so from the working tree with extra -D flags for EVERY translation unit (host-side launch heuristics included): ab/q16/mkall.sh NAME [-DFOO ...]
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
C=neural_image_compression_v2_amd/csrc
T=/tmp/abq_all_$NAME; mkdir -p $T
N=0
for f in $C/*.hip; do
  b=$(basename $f .hip)
  X=""; [ "$b" = "simple_kernels" ] && X="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function $X "$@" -c $f -o $T/$b.o &
  N=$((N+1)); if [ $((N % 8)) -eq 0 ]; then wait; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$NAME.so $T/*.o
echo built ab/lib$NAME.so
