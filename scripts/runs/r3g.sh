cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_split.py tests/test_cpp_host.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 600 python scripts/bench_c4.py 1000000 > $O/c4.log 2>&1
timeout 600 bash scripts/prof_c4.sh > $O/prof_c4.log 2>&1
tail -3 $O/t1.log; cat $O/c4.log | tail -3; tail -40 $O/prof_c4.log
