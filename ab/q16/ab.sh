#!/bin/bash
# interleaved A/B of q16 library variants on ONE box: ab/q16/ab.sh "cases" lib1 lib2 ...   (cases: comma list for time_q16.py)
CASES=$1; shift
for r in 1 2; do
  for v in "$@"; do
    echo "== variant $v round $r"
    NIC_LIB_PATH=$PWD/ab/lib$v.so timeout -k 10 300 python ab/q16/time_q16.py 30 $CASES 2>/dev/null | grep -v "^{" | cut -c1-120
  done
done
