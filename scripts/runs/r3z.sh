#!/bin/bash
mkdir -p gpurun_out/r3z
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_split.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1; done
bash scripts/prof_c4.sh r3z_c4 > gpurun_out/r3z/prof.log 2>&1; head -40 gpurun_out/r3z_c4/kernel_stats.md
