"""Per-iteration time of the SHARDED code path on one GPU (one rank that owns the whole chain, the interface exchange is a
local copy): what a rank of the multi-GPU bench does apart from the RCCL all-gather."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import gpslam_amd
from gpslam_amd import synthetic as S, sharded
N = 100000
p = S.pose3_chain(N)
lp = sharded.local_problem(p, 0, 1)
solver = gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=0, rank=0, nranks=1, force_sharded=True) if 'force_sharded' in gpslam_amd.ChainSolver.__init__.__code__.co_varnames else None
if solver is None:
    print('no force_sharded ctor arg'); sys.exit(0)
solver.set_stream(torch.cuda.current_stream().cuda_stream)
sharded.apply_local(lp, solver)
send, recv = sharded.device_tensors(solver)
sv = sharded.ShardedSolver(solver, send, recv, 0, 1, dist=None)
for _ in range(3): sv.iterate(want_stats=False)
solver.set_states(lp["pose"], lp["vel"])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): sv.iterate(want_stats=False)
torch.cuda.synchronize(); print('sharded path, 1 rank: %.4f ms per iteration' % ((time.perf_counter() - t0) / 20 * 1e3))
