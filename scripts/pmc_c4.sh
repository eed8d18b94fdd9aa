#!/bin/bash
# stall / LDS picture of config 4's solve kernels (GPU box, repo root)
for K in k_fs_sweep_syrk k_fs_factor_rows6 k_fs_solve1 k_assemble_ghost; do
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"; do
  PMC_TIMEOUT=200 bash scripts/pmc_kernel.sh $K "$C" python scripts/profile_cfg.py c4 ${1:-1000000} 2>&1 | tail -1 | cut -c1-600
done
done
