#!/bin/bash
# builds ab/lib<NAME>.so from the working tree with extra -D flags for the listed translation units only: ab/mkv.sh NAME "fused_t16 fused_m1" [-DFOO ...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNITS=$2; shift; shift
C=neural_image_compression_v2_amd/csrc
python -m neural_image_compression_v2_amd._build >/dev/null
T=/tmp/abv_$NAME; mkdir -p $T
for u in $UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function "$@" -c $C/$u.hip -o $T/$u.o &
done
wait
OBJS=""
for o in $C/build/*.o; do b=$(basename $o .o); if [ -f $T/$b.o ]; then OBJS="$OBJS $T/$b.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$NAME.so $OBJS
echo built ab/lib$NAME.so
