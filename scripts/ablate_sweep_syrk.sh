#!/bin/bash
# timing ablations of k_fs_sweep_syrk (library built with -DGPS_FSY_DBG)
timeout 900 python -m pytest tests/test_gpu_segmented.py -x -q 2>&1 | tail -2
for d in 0 1 3 7; do
  echo "dbg=$d"; GPSLAM_FSY_DBG=$d timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1
done
GPSLAM_PY_DEFAULT_PLAN=8 timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1
