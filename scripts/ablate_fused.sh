#!/bin/bash
# timing ablations of k_fused_level0 (results are wrong on purpose): which of its waves bounds a block step, and what the
# column reconstruction from the structured records costs.  Variant builds are keyed by their flags (gpslam_amd/build.py): they
# never replace the product library, and the timing process finds its variant through the same environment variable.
# Build them where there is no GPU clock running:  for F in ...; do GPSLAM_HIPCC_FLAGS="$F" python gpslam_amd/build.py; done
for F in "" "-DGPS_ABLATE_REC" "-DGPS_ABLATE_ASM -DGPS_ABLATE_REC" "-DGPS_ABLATE_ELIM" "-DGPS_ABLATE_ASM -DGPS_ABLATE_REC -DGPS_ABLATE_ELIM"; do
  echo "flags [$F]: $(GPSLAM_HIPCC_FLAGS="$F" python scripts/time_fused.py ${1:-100000} 2>&1 | tail -1)"
done
