cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_segmented.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 600 python scripts/bench_c4.py 1000000 > $O/c4_new.log 2>&1
tail -2 $O/t1.log; tail -1 $O/c4_new.log
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"; do
  PMC_TIMEOUT=200 bash scripts/pmc_kernel.sh k_fs_sy "$C" python scripts/bench_c4.py 1000000 2>&1 | tail -2
  PMC_TIMEOUT=200 bash scripts/pmc_kernel.sh k_fs_sweep "$C" python scripts/bench_c4.py 1000000 2>&1 | tail -2
done > $O/pmc.log 2>&1
cat $O/pmc.log
