set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tailab
for k in 1 2; do
python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/4k_tail_$k.json 2>> gpurun_out/tailab/err.txt
NIC_NO_TAIL=1 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/4k_sep_$k.json 2>> gpurun_out/tailab/err.txt
done
python3 bench.py --workload default3d --precision bf16 --steps 300 --warmup 30 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/d3b_tail.json 2>> gpurun_out/tailab/err.txt
NIC_NO_TAIL=1 python3 bench.py --workload default3d --precision bf16 --steps 300 --warmup 30 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/d3b_sep.json 2>> gpurun_out/tailab/err.txt
python3 bench.py --workload default --steps 300 --warmup 30 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/d2_tail.json 2>> gpurun_out/tailab/err.txt
NIC_NO_TAIL=1 python3 bench.py --workload default --steps 300 --warmup 30 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/d2_sep.json 2>> gpurun_out/tailab/err.txt
python3 bench.py --workload lut33 --steps 300 --warmup 30 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/lut_tail.json 2>> gpurun_out/tailab/err.txt
NIC_NO_TAIL=1 python3 bench.py --workload lut33 --steps 300 --warmup 30 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/lut_sep.json 2>> gpurun_out/tailab/err.txt
python3 bench.py --virtual-world 8 --scaling strong --steps 100 --warmup 10 --no-cpu-baseline --stat-launches 0 > gpurun_out/tailab/v8.json 2>> gpurun_out/tailab/err.txt
