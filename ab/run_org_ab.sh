cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --stat-launches 0"; O=gpurun_out/orgab; mkdir -p $O; rm -f $O/*
timeout -k 10 600 python -m pytest tests/test_gpu_tail.py -q -x > $O/tests.txt 2>&1
for k in 1 2; do
for v in host dev; do
  if [ $v = dev ]; then export NIC_NO_HOST_ORIGINS=1; else unset NIC_NO_HOST_ORIGINS; fi
  $B --workload default3d --precision bf16 --steps 300 --warmup 30 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$v default3d', d['ms_per_step'])" >> $O/out.txt
  $B --workload default --precision bf16 --steps 300 --warmup 30 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$v default2d', d['ms_per_step'])" >> $O/out.txt
done
done
