set -e
cd "$(dirname "$0")/.."
O=gpurun_out/w2; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --stat-launches 0"
$B --workload lut33 --precision bf16 --steps 300 --warmup 30 > $O/lut33_bf16.json 2>> $O/err.txt
$B --workload lut33 --precision split --steps 300 --warmup 30 > $O/lut33_split.json 2>> $O/err.txt
$B --workload vol64 --precision bf16 --steps 100 --warmup 10 > $O/vol64_bf16.json 2>> $O/err.txt
$B --workload vol128 --precision bf16 --steps 40 --warmup 5 > $O/vol128_bf16.json 2>> $O/err.txt
$B --steps 40 --warmup 10 > $O/4k.json 2>> $O/err.txt
$B --virtual-world 8 --scaling strong --steps 100 --warmup 10 > $O/v8.json 2>> $O/err.txt
$B --workload fits8 --precision split --steps 20 --warmup 5 > $O/fits8.json 2>> $O/err.txt
$B --workload multilevel --steps 20 --warmup 5 > $O/ml.json 2>> $O/err.txt
python3 bench.py --steps 20 --warmup 5 > $O/driver.json 2>> $O/err.txt
echo done
