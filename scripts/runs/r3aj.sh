#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -4
python scripts/bench_configs.py 1000000 2>&1 | tail -4 | head -3
