#!/bin/bash
# rocprofv3 evidence at the north-star size (1e6 Pose3 states, run on the GPU box from the repo root):
#   bash scripts/collect_profiles_1e6.sh <tag>
# kernel trace + stats of scripts/profile_iter.py 1000000, then separate --pmc passes (counters never share a pass with
# trace domains other than --kernel-trace).
set -u
TAG=${1:-r3_1e6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/scripts/profile_iter.py 1000000 > $OUT/trace.log 2>&1
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  (cd $ROOT && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python scripts/profile_iter.py 1000000 > $D.log 2>&1)
done
cd $ROOT
