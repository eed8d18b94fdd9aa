#!/bin/bash
# whole-iteration time of bench.py for every library under build_ab/ (A/B builds made beforehand, loaded through GPSLAM_LIB)
#   bash scripts/ab_libs.sh <states> <repetitions>
N=${1:-100000}; R=${2:-2}
for L in build_ab/lib_*.so; do
  for rep in $(seq $R); do
    echo "$(basename $L) N=$N: $(GPSLAM_LIB=$PWD/$L python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --states $N 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["ms_per_step"], d["roofline"]["achieved"])')"
  done
done
