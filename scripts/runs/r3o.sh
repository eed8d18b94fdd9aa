cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
for t in 12 7 4; do GPSLAM_SYRK_TPW=$t timeout 600 python scripts/bench_c4.py 1000000 2>&1 | grep C4; done > $O/c4.log
cat $O/c4.log
