#!/bin/bash
# fused sweep + Schur complement (k_fs_sweep_syrk): parity tests, then A/B at config 4
mkdir -p gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_split.py -x -q > gpurun_out/r3u/seg.log 2>&1; echo "seg rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3u/seg.log | tail -15
for i in 1 2; do
  timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1
  GPSLAM_FS_FUSED=0 timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1
done
