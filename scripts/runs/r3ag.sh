#!/bin/bash
mkdir -p gpurun_out/r3ag
python -m pytest tests -m gpu -x -q > gpurun_out/r3ag/tall.log 2>&1; echo "tall rc=$?"; grep -E "passed|failed" gpurun_out/r3ag/tall.log | tail -2
python bench.py > gpurun_out/r3ag/bench.json 2> gpurun_out/r3ag/bench.err; tail -c 300 gpurun_out/r3ag/bench.json
bash scripts/prof_c4.sh r3ag_c4 > gpurun_out/r3ag/prof.log 2>&1
