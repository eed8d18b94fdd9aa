cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_upper.py tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new.log 2>&1
GPSLAM_UPPER=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_old.log 2>&1
timeout 300 python scripts/bench_configs.py 100000 > $O/cfg_new.log 2>&1
GPSLAM_UPPER=0 timeout 300 python scripts/bench_configs.py 100000 > $O/cfg_old.log 2>&1
timeout 300 bash scripts/trace_iter.sh 100000 > $O/trace_new.log 2>&1
tail -3 $O/t1.log; tail -c 600 $O/bench_new.log; echo; tail -c 600 $O/bench_old.log; echo; cat $O/cfg_new.log $O/cfg_old.log; cat $O/trace_new.log
