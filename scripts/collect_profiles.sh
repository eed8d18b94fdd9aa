#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers (run on the GPU box from the repo root):
#   bash scripts/collect_profiles.sh <tag>
# 1. kernel trace + stats of the very bench.py command; 2..n separate --pmc passes (counters never share a pass
# with the trace domains other than --kernel-trace) of a 5-iteration run of the same workload.
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
  python $ROOT/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  (cd $ROOT && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python scripts/profile_iter.py > $D.log 2>&1)
done
cd $ROOT
python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
