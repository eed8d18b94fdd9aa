#!/bin/bash
# The round's rocprofv3 evidence in one GPU call (run from the repo root on the GPU box): the bench.py command (config 3, 1e5 states),
# the north-star size, config 5's SE(3) mix and config 4 -- kernel trace + stats, then the counter passes, each in its own run.
#   bash scripts/collect_round6.sh [tag]
set -u
T=${1:-r6p}
timeout 700 bash scripts/collect_profiles.sh ${T} 2>&1 | tail -1 | cut -c1-300
timeout 600 bash scripts/collect_profiles_1e6.sh ${T}_1e6 2>&1 | tail -2 | cut -c1-300
PROF_TIMEOUT=200 timeout 900 bash scripts/collect_profiles_cfg.sh ${T}_c5b python scripts/profile_cfg.py c5b 1000000 2>&1 | tail -2
PROF_TIMEOUT=200 timeout 900 bash scripts/collect_profiles_cfg.sh ${T}_c4 python scripts/profile_cfg.py c4 1000000 2>&1 | tail -2
# MFMA utilisation of config 4's Schur-complement kernel: its own counter pass
D=gpurun_out/${T}_c4/pmc_MFMA
TMPDIR=/tmp timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $D -o p -- python scripts/profile_cfg.py c4 1000000 > $D.log 2>&1
find gpurun_out/${T}_c4 -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh gpurun_out/${T}* | tail -5
