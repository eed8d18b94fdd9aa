#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_split.py -x -q 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python scripts/bench_c4.py 1000000 2>&1 | tail -1; done
