cd "$(dirname "$0")/.."
O=gpurun_out/check; mkdir -p $O; rm -f $O/*
timeout -k 10 1000 python -m pytest tests -q -x -m gpu -k "not long_fit" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout -k 10 250 python ab/q16/time_ml.py 10 5,4,3 3,4,5 2,12,3 > $O/ml_t.txt 2>&1; grep "^ml" $O/ml_t.txt
timeout -k 10 250 python ab/q16/time_q16.py 30 split_nl3,bf16_nl3,bf16_nl5_g16 2>/dev/null | grep -v "^{" | cut -c1-100
B="python3 bench.py --no-cpu-baseline --stat-launches 0"
$B --workload default3d --precision bf16 --steps 300 --warmup 30 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('default3d bf16', d['ms_per_step'])"
$B --workload default --precision bf16 --steps 300 --warmup 30 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('default2d bf16', d['ms_per_step'])"
$B --steps 40 --warmup 10 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('4k', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
$B --virtual-world 8 --scaling strong --steps 100 --warmup 10 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('v8', d['ms_per_step'], d['roofline']['kernel_ms'])"
