#!/bin/bash
# A/B of split-bf16 kernel variants (ab/split_check.py prints fp32 and split timings of the 4K pass)
for r in 1 2; do for v in "$@"; do echo "variant $v round $r: $(NIC_LIB_PATH=$PWD/ab/lib$v.so timeout -k 10 200 python ab/split_check.py 2>/dev/null | grep split)"; done; done
