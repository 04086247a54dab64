#!/bin/bash
# rocprofv3 stats + PMC passes of a bench configuration (run on the GPU box): bash ab/profile.sh TAG [bench.py args ...]
# outputs under gpurun_out/prof_$TAG; ab/collect_profiles.py TAG KERNEL_SUBSTRING turns them into profiles/$TAG_*.{csv,json}
TAG=${1:-x}; shift
cd "$(dirname "$0")/.."
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "$@" > $OUT/args.txt
python3 bench.py --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --stat-launches 0 "$@" > $OUT/stats.log 2>&1 ) || exit 1
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "TCC_EA0_ATOMIC_sum WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_WAVES"; do
  i=$((i+1))
  echo "$set" > $OUT/pmc$i.set
  ( cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --stat-launches 0 "$@" > $OUT/pmc$i.log 2>&1 ) || exit 1
  echo "pass $i done"
done
