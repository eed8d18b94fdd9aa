#!/bin/bash
for i in 1 2 3; do
for v in 0 1; do
GPSLAM_FUSED_REV=$v python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rev=$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms_per_iter_1gpu'].items()}, round(d['roofline']['avg_launch_ms'],4))"
done
done
GPSLAM_FUSED_REV=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_upper.py -x -q 2>&1 | tail -2
