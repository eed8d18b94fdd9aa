#!/bin/bash
# kernel trace of a 5-iteration run and the idle time in front of selected kernels (run on the GPU box from the repo root)
ROOT=$(pwd); export TMPDIR=/tmp; rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python scripts/profile_iter.py > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "t_kernel_trace.csv" | head -1)
echo "trace: $f"
for k in "k_gp<double, 3, 0" "k_assemble" "k_simple<double, 3, 2, true" "k_chunk_forward" "k_retract"; do python scripts/gap_before.py "$f" "$k"; done
