cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tall.log 2>&1; echo "tall rc=$?" >> $O/tall.log
timeout 300 python scripts/time_sharded_1rank.py > $O/sharded1.log 2>&1
tail -5 $O/tall.log; cat $O/sharded1.log | tail -5
