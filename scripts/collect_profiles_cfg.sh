#!/bin/bash
# rocprofv3 evidence for any profiling target (run on the GPU box from the repo root):
#   bash scripts/collect_profiles_cfg.sh <tag> <command ...>        e.g.  ... r4_c4 python scripts/profile_cfg.py c4 1000000
# kernel trace + stats of the command, then separate --pmc passes (counters never share a pass with trace domains other
# than --kernel-trace).  Kernels that overlap on two streams in the plain trace run one after the other in the counter
# passes (counter collection serialises the dispatches): the per-kernel durations of a counter pass are stand-alone times.
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd $ROOT && timeout ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1)
grep "ms/iter" $OUT/trace.log | tail -1
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  (cd $ROOT && timeout ${PROF_TIMEOUT:-240} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- "$@" > $D.log 2>&1)
done
# the raw csv files are large: keep what scripts/summarise_profiles.py reads
find $OUT -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $OUT | tail -1
