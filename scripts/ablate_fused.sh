#!/bin/bash
# timing ablations of k_fused_level0 (results are wrong on purpose): which of its two waves bounds a block step.
# Variant builds are keyed by their flags (gpslam_amd/build.py): they never replace the product library, and the timing
# process finds its variant through the same environment variable.
for F in "" "-DGPS_ABLATE_ASM" "-DGPS_ABLATE_ELIM" "-DGPS_ABLATE_ASM -DGPS_ABLATE_ELIM"; do
  GPSLAM_HIPCC_FLAGS="$F" python gpslam_amd/build.py > /dev/null 2>&1
  echo "flags [$F]: $(GPSLAM_HIPCC_FLAGS="$F" python scripts/time_fused.py 2>&1 | tail -1)"
done
