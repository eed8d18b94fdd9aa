#!/bin/bash
# timing ablations of k_fused_level0 (results are wrong on purpose): which of its two waves bounds a block step
for F in "" "-DGPS_ABLATE_ASM" "-DGPS_ABLATE_ELIM" "-DGPS_ABLATE_ASM -DGPS_ABLATE_ELIM"; do
  GPSLAM_HIPCC_FLAGS="$F" python gpslam_amd/build.py --force > /dev/null 2>&1
  echo "flags [$F]: $(python scripts/time_fused.py 2>&1 | tail -1)"
done
python gpslam_amd/build.py --force > /dev/null 2>&1
