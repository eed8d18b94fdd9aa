#!/bin/bash
# config 4: the persistent tail of the fat blocks' cyclic reduction against one launch per level (GPSLAM_PLAN_FS_LEVEL_LAUNCHES = 256)
mkdir -p gpurun_out/r6k
for rep in 1 2 3; do
  echo "tail   $(timeout 300 python scripts/profile_cfg.py c4 1000000 2>&1 | tail -1)"
  echo "levels $(GPSLAM_PY_DEFAULT_PLAN=256 timeout 300 python scripts/profile_cfg.py c4 1000000 2>&1 | tail -1)"
done 2>&1 | tee gpurun_out/r6k/ab_c4_tail.txt
timeout 600 bash scripts/kernel_times.sh c4tail python scripts/profile_cfg.py c4 1000000 2>&1 | tee gpurun_out/r6k/c4_tail_kernels.txt
