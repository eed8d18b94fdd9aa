cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tall.log 2>&1; echo "tall rc=$?" >> $O/tall.log
tail -30 $O/tall.log
