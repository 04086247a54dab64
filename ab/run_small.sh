set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/small
R=$PWD
export TMPDIR=/tmp
python3 bench.py --workload default3d --steps 200 --warmup 20 --no-cpu-baseline --stat-launches 0 > gpurun_out/small/d3.json 2> gpurun_out/small/d3.err
python3 bench.py --workload default3d --precision bf16 --steps 200 --warmup 20 --no-cpu-baseline --stat-launches 0 > gpurun_out/small/d3b.json 2>> gpurun_out/small/d3.err
python3 bench.py --workload default --steps 200 --warmup 20 --no-cpu-baseline --stat-launches 0 > gpurun_out/small/d2.json 2>> gpurun_out/small/d3.err
python3 bench.py --virtual-world 8 --scaling strong --steps 100 --warmup 10 --no-cpu-baseline --stat-launches 0 > gpurun_out/small/v8.json 2>> gpurun_out/small/d3.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small/st_d3 -- python3 $R/bench.py --workload default3d --steps 200 --warmup 20 --no-cpu-baseline --stat-launches 0 > $R/gpurun_out/small/st_d3.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small/st_v8 -- python3 $R/bench.py --virtual-world 8 --scaling strong --steps 100 --warmup 10 --no-cpu-baseline --stat-launches 0 > $R/gpurun_out/small/st_v8.log 2>&1 )
