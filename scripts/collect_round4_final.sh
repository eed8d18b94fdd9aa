#!/bin/bash
# the end-of-round evidence in one GPU call: tests, the bench.py profile (trace + counter passes), the 1e6 profile, the configs
set -u
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 bash scripts/collect_profiles.sh ${1:-r4v4} 2>&1 | tail -1 | cut -c1-400
timeout 600 bash scripts/collect_profiles_1e6.sh ${1:-r4v4}_1e6 2>&1 | tail -2 | cut -c1-300
for c in "c2 100000" "c2 1000000" "c5 100000" "c5 1000000" "c4 1000000"; do timeout 120 python scripts/profile_cfg.py $c 2>&1 | tail -1; done
for i in 1 2; do timeout 100 python scripts/phase_times.py 1000000 10 | tail -1; done
