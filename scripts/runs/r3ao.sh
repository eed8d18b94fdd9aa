#!/bin/bash
for i in 1 2 3; do
for v in nosplit split; do
GPSLAM_LIB=$(pwd)/gpslam_amd/lib/libgpslam_hip_$v.so python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms_per_iter_1gpu'].items()}, round(d['k1_batched_jacobian']['standalone_ms'],4))"
done
done
GPSLAM_LIB=$(pwd)/gpslam_amd/lib/libgpslam_hip_split.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp32.py tests/test_gpu_upper.py -x -q 2>&1 | tail -2
