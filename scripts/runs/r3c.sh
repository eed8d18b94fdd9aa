cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_upper.py tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py tests/test_gpu_configs.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_tail.log 2>&1
GPSLAM_TAIL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_notail.log 2>&1
GPSLAM_UPPER=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_old.log 2>&1
GPSLAM_UPPER_PROBE=1 timeout 300 python scripts/profile_iter.py 100000 > $O/probe.log 2>&1
timeout 300 bash scripts/trace_iter.sh 100000 > $O/trace_new.log 2>&1
timeout 300 python scripts/bench_configs.py 100000 > $O/cfg_new.log 2>&1
tail -3 $O/t1.log
for f in bench_tail bench_notail bench_old; do python - $O/$f.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['ms_per_step'], d['phase_ms_per_iter_1gpu'], d['kernel_ms'])
PY
done
grep "upper probe" $O/probe.log | tail -3; cat $O/trace_new.log; cat $O/cfg_new.log
