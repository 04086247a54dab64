#!/bin/bash
# A/B timing of 16-sample kernel variants on ONE box: interleaved rounds, kernel ms from the bench's HIP events
# usage: ab/w16/ab.sh ROUNDS name1 name2 ...   (name "old" = the 32-sample kernel of the default library)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = "old" ]; then export NIC_T16=0; unset NIC_LIB_PATH; else export NIC_T16=1; export NIC_LIB_PATH=$PWD/ab/lib$v.so; fi
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --stat-launches 30 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); s = d['roofline']['stats']['stat_leg']; print('variant $v round $r: kernel_ms', d['roofline']['kernel_ms'], 'median', s['median'], 'min', s['min'], 'Mpix/s', d['value'], 'loss', d['config']['final_loss'])"
  done
done
