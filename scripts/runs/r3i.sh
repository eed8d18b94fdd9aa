cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_split.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 600 python scripts/bench_c4.py 1000000 > $O/c4_new.log 2>&1
timeout 600 python scripts/bench_c4.py 1000000 > $O/c4_old.log 2>&1
timeout 600 bash scripts/prof_c4.sh c4b > $O/prof_c4.log 2>&1
tail -3 $O/t1.log; tail -1 $O/c4_new.log; tail -1 $O/c4_old.log; grep -E "k_fs_factor|k_fs_syrk|k_fs_sweep" $O/prof_c4.log
