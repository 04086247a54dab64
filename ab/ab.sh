#!/bin/bash
# A/B timing of kernel variants on ONE box: interleaved rounds, kernel ms from the bench's HIP events
for r in 1 2 3; do
  for v in "$@"; do
    NIC_LIB_PATH=$PWD/ab/lib$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('variant $v round $r: kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'], 'Mpix/s', d['value'])"
  done
done
