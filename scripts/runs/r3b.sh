cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
GPSLAM_UPPER_PROBE=1 timeout 300 python scripts/profile_iter.py 100000 > $O/probe.log 2>&1
grep "upper probe" $O/probe.log | tail -6
