#!/bin/bash
mkdir -p gpurun_out/r3x
python -m pytest tests -m gpu -x -q > gpurun_out/r3x/tall.log 2>&1; echo "tall rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3x/tall.log | tail -5
timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('headline', d['ms_per_step'], d['phase_ms_per_iter_1gpu'])"
