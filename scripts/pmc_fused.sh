#!/bin/bash
# stall picture of k_fused_level0 (GPU box, repo root): SQ counters in three passes over scripts/profile_iter.py
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
  echo "== $C"
  bash scripts/pmc_kernel.sh k_fused_level0 "$C" python scripts/profile_iter.py ${1:-100000} 2>&1 | tail -2
done
