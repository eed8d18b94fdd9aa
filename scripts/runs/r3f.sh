cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fp32.py tests/test_gpu_measurements.py -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
tail -40 $O/t1.log
