#!/bin/bash
# the whole stress net with seeds of the caller's choosing (the suite runs it with fixed ones): bash scripts/stress_all.sh <count> <seed0>
C=${1:-30}; S0=${2:-100}; mkdir -p gpurun_out/stress; k=0
for s in stress_sizes.py stress_mixes.py stress_segmented.py stress_pending.py stress_closures.py; do
  k=$((k+1))
  timeout 1200 python scripts/$s $C $((S0+k)) > gpurun_out/stress/${s%.py}_$((S0+k)).txt 2>&1; echo "$s seed $((S0+k)) rc=$? $(tail -1 gpurun_out/stress/${s%.py}_$((S0+k)).txt | cut -c1-200)"
done
