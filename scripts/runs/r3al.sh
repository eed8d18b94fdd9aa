#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -3
for n in 100000 300000 1000000; do python scripts/sweep_chunk.py $n 0 2>&1 | tail -1; done
python scripts/bench_configs.py 1000000 2>&1 | tail -4
python scripts/bench_configs.py 100000 2>&1 | tail -4
