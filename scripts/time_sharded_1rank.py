#!/usr/bin/env python3
"""What the sharded code path costs on ONE rank (forced sharding, the collective replaced by a device copy): the per-rank
work of an N-GPU weak-scaling run without its all-gather.  python scripts/time_sharded_1rank.py [states]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpslam_amd
from gpslam_amd import sharded, synthetic as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = S.pose3_chain(N)
res = {}
for forced in (False, True):
    s = gpslam_amd.ChainSolver(gpslam_amd.POSE3, force_sharded=forced)
    if forced:
        s.set_stream(torch.cuda.current_stream().cuda_stream)
        lp = sharded.local_problem(p, 0, 1)
        sharded.apply_local(lp, s)
        send, recv = sharded.device_tensors(s)
        sv = sharded.ShardedSolver(s, send, recv, 0, 1, dist=None)
        run = lambda k: [sv.iterate(want_stats=False) for _ in range(k)]
    else:
        S.apply(p, s)
        run = lambda k: s.run_gn(k)
    run(5)
    s.set_states(p["pose"], p["vel"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(50)
    torch.cuda.synchronize()
    res[forced] = (time.perf_counter() - t0) / 50 * 1e3
    print("forced sharded path" if forced else "unsharded", "%.4f ms per iteration" % res[forced], s.plan_info())
print("overhead of the sharded path on one rank: %.1f %%" % (100 * (res[True] / res[False] - 1)))
