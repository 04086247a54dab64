#!/bin/bash
# extract kernel (layout $1, SRC_ENCODE, MODE $2) from /tmp/isa/fused_m$1.s into /tmp/isa/k.s and print the big loops
M=${1:-1}; MODE=${2:-1}; PREC=${3:-0}
awk -v k="_ZN3nic12fused_kernelINS_6LayoutILi${M}EEELi0ELi${MODE}EfLi${PREC:-0}EEEvNS_11FusedParamsE:" 'index($0,k)==1{f=1} f{print} /s_endpgm/{if(f){exit}}' /tmp/isa/fused_m$M.s > /tmp/isa/k.s
python "$(dirname "$0")/isa_mix.py" /tmp/isa/k.s | grep -E "span [0-9]{4}$"
