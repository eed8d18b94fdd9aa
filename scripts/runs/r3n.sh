cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 600 python scripts/bench_c4.py 1000000 > $O/c4_new.log 2>&1
timeout 600 python scripts/bench_c4.py 1000000 >> $O/c4_new.log 2>&1
timeout 600 bash scripts/prof_c4.sh c4c > $O/prof_c4.log 2>&1
cat $O/c4_new.log | grep C4; grep -E "k_fs_factor|k_fs_syrk|k_fs_sweep|k_fat_elim" $O/prof_c4.log
