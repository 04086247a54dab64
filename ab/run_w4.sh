cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --stat-launches 0"; mkdir -p gpurun_out/w4; rm -f gpurun_out/w4/*
for k in 1 2 3 4; do
$B --workload vol128 --precision bf16 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('plan vol128', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/w4/out.txt
NIC_NO_PLAN=1 $B --workload vol128 --precision bf16 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('noplan vol128', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/w4/out.txt
done
for k in 1 2 3; do
$B --virtual-world 8 --scaling strong --steps 200 --warmup 20 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('plan v8', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/w4/out.txt
NIC_NO_PLAN=1 $B --virtual-world 8 --scaling strong --steps 200 --warmup 20 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('noplan v8', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/w4/out.txt
done
