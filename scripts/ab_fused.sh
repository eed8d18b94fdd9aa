#!/bin/bash
# A/B of k_fused_level0 builds under build_ab/ (GPSLAM_LIB): the launch alone (time_fused.py) and inside the iteration (bench.py)
#   bash scripts/ab_fused.sh lib_a.so lib_b.so ...
for rep in 1 2; do
  for L in "$@"; do
    for N in 100000 1000000; do
      echo "$L N=$N $(GPSLAM_LIB=$PWD/build_ab/$L python scripts/time_fused.py $N 2>&1 | tail -1)"
    done
    echo "$L bench: $(GPSLAM_LIB=$PWD/build_ab/$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("ms_per_step", round(d["ms_per_step"],4), "l0_in_iter_ms", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3))')"
  done
done
