#!/bin/bash
# time k_fused_level0 for every library variant under build_ablate/ (built locally with -DGPS_ABLATE_* flags)
for L in build_ablate/lib_*.so; do
  echo "$(basename $L): $(GPSLAM_LIB=$PWD/$L python scripts/time_fused.py 2>&1 | tail -1)"
done
