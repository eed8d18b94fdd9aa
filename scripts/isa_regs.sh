#!/bin/bash
# Registers / scratch / occupancy of chosen kernel instantiations without building the library (hipcc cross-compiles the device
# side of a throw-away translation unit in ~20 s):   bash scripts/isa_regs.sh 'gps::k_fused_level0<4, double, 12, true>(gps::FusedArgs<double, double>)' ...
# The ISA is left in /tmp/isa_regs.s.
cd "$(dirname "$0")/../gpslam_amd/csrc" || exit 1
T=_isa_regs_$$.hip
echo '#include "api_common.hpp"' > $T
for sig in "$@"; do echo "template __global__ void $sig;" >> $T; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast --cuda-device-only -S -o /tmp/isa_regs.s $T 2>&1 | grep -v "warning\|^ *[0-9]* *|\|^ *|\|generated\.$" | head -30
rm -f $T
awk '/^_ZN3gps.*:/{name=$1} /; NumVgprs:|; ScratchSize:|; Occupancy:/{printf "%s %s\n", name, $0}' /tmp/isa_regs.s | c++filt | sed 's/(gps::[A-Za-z]*Args<[^)]*)//' | paste - - - | sed 's/[^ ]*ScratchSize/ScratchSize/; s/[^ ]* ; Occupancy/Occupancy/' | grep -F -f <(for sig in "$@"; do echo "$sig" | sed 's/(.*//; s/gps:://'; done) || true
