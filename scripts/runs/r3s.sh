#!/bin/bash
# deferred reductions A/B + GPU suite
mkdir -p gpurun_out/r3s
python -m pytest tests -m gpu -x -q > gpurun_out/r3s/tall.log 2>&1; echo "tall rc=$?"; tail -3 gpurun_out/r3s/tall.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('defer   ', d['ms_per_step'], d['phase_ms_per_iter_1gpu'])"
GPSLAM_DEFER_REDUCE=0 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('no defer', d['ms_per_step'], d['phase_ms_per_iter_1gpu'])"
done
