cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --stat-launches 0"; mkdir -p gpurun_out/w3
$B --workload vol128 --precision bf16 --steps 40 --warmup 5 > gpurun_out/w3/vol128_a.json 2>/dev/null
$B --workload vol128 --precision bf16 --steps 40 --warmup 5 > gpurun_out/w3/vol128_b.json 2>/dev/null
NIC_NO_PLAN=1 $B --workload vol128 --precision bf16 --steps 40 --warmup 5 > gpurun_out/w3/vol128_noplan.json 2>/dev/null
$B --virtual-world 8 --scaling strong --steps 100 --warmup 10 > gpurun_out/w3/v8_a.json 2>/dev/null
NIC_NO_PLAN=1 $B --virtual-world 8 --scaling strong --steps 100 --warmup 10 > gpurun_out/w3/v8_noplan.json 2>/dev/null
$B --virtual-world 8 --scaling strong --steps 100 --warmup 10 > gpurun_out/w3/v8_b.json 2>/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_tail.py -q -x > gpurun_out/w3/tail.txt 2>&1
