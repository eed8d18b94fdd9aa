#!/bin/bash
# config 4 (1e6 SE(2) states, 5e4 locally visible landmarks) for every library under build_ab/ (GPSLAM_LIB), three repetitions
mkdir -p gpurun_out/r6k
for rep in 1 2 3; do
  for L in build_ab/lib_*.so; do
    echo "$(basename $L) $(GPSLAM_LIB=$PWD/$L timeout 300 python scripts/profile_cfg.py c4 1000000 2>&1 | tail -1)"
  done
done 2>&1 | tee gpurun_out/r6k/ab_c4.txt
