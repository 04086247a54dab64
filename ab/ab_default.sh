#!/bin/bash
for r in 1 2; do for v in "$@"; do echo "variant $v round $r: $(NIC_LIB_PATH=$PWD/ab/lib$v.so timeout -k 10 200 python ab/bench_default.py 2>/dev/null | tr '\n' '|')"; done; done
