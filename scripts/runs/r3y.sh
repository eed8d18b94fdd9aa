#!/bin/bash
mkdir -p gpurun_out/r3y
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3y/tall.log 2>&1; echo "tall rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r3y/tall.log | tail -8
