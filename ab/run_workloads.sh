# the round's workload records (run on the GPU box): bash ab/run_workloads.sh TAG  ->  gpurun_out/w_TAG/*.json (copy into profiles/TAG_w_*.json)
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
O=gpurun_out/w_$TAG; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --stat-launches 0"
for p in split bf16 fp16; do
  $B --workload default --precision $p --steps 300 --warmup 30 > $O/default_$p.json 2>> $O/err.txt
  $B --workload default3d --precision $p --steps 300 --warmup 30 > $O/default3d_$p.json 2>> $O/err.txt
  $B --workload lut33 --precision $p --steps 300 --warmup 30 > $O/lut33_$p.json 2>> $O/err.txt
  $B --workload vol64 --precision $p --steps 100 --warmup 10 > $O/vol64_$p.json 2>> $O/err.txt
  $B --workload vol128 --precision $p --steps 40 --warmup 5 > $O/vol128_$p.json 2>> $O/err.txt
done
for p in split bf16; do
  $B --workload slab --precision $p --steps 20 --warmup 5 > $O/slab_$p.json 2>> $O/err.txt
  $B --workload video --precision $p --steps 10 --warmup 3 > $O/video_$p.json 2>> $O/err.txt
  $B --workload fits8 --precision $p --steps 20 --warmup 5 > $O/fits8_$p.json 2>> $O/err.txt
done
$B --workload fits64 --steps 5 --warmup 2 > $O/fits64_split.json 2>> $O/err.txt
$B --workload default3d --precision bf16 --graph 8 --steps 300 --warmup 30 > $O/graph.json 2>> $O/err.txt
for n in 2 4 8; do
  $B --virtual-world $n --scaling strong --steps 100 --warmup 10 > $O/v${n}_strong.json 2>> $O/err.txt
done
NIC_NO_TAIL=1 $B --workload default3d --precision bf16 --steps 300 --warmup 30 > $O/default3d_bf16_notail.json 2>> $O/err.txt
NIC_NO_TAIL=1 $B --workload lut33 --precision bf16 --steps 300 --warmup 30 > $O/lut33_bf16_notail.json 2>> $O/err.txt
NIC_NO_TAIL=1 $B --steps 40 --warmup 10 > $O/4k_notail.json 2>> $O/err.txt
echo done
