#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_upper.py tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phase_ms_per_iter_1gpu'].items()})"
done
GPSLAM_UPPER_PROBE=1 python scripts/profile_iter.py 100000 2>&1 | grep "upper probe" | tail -4
