cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_split.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
tail -2 $O/t1.log
for t in 0 112 144; do GPSLAM_SYRK_MIN=$t timeout 600 python scripts/bench_c4.py 1000000 2>&1 | grep C4; done > $O/c4.log
cat $O/c4.log
timeout 600 bash scripts/prof_c4.sh c4d > $O/prof_c4.log 2>&1
grep -E "k_fs_factor|k_fs_syrk|k_fs_sweep|k_fat_elim" $O/prof_c4.log
