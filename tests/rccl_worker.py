"""Worker of tests/test_gpu_sharded.py::test_two_process_rccl_when_two_gpus_are_visible: one process per GPU over RCCL
(torch.distributed backend "nccl").  python rccl_worker.py <rank> <world> <port> <N>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, N = (int(a) for a in sys.argv[1:5])
    import numpy as np
    import torch
    import torch.distributed as dist
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    problem = S.pose3_chain(N)
    lp = sharded.local_problem(problem, rank, world)
    s = gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=rank, rank=rank, nranks=world, force_sharded=(world == 1))
    sharded.apply_local(lp, s)
    send, recv = sharded.device_tensors(s)
    sv = sharded.ShardedSolver(s, send, recv, rank, world, dist=dist)      # moves the handle onto torch's stream itself
    hist = [sv.iterate() for _ in range(5)]
    pose, vel = s.get_states()
    # every rank also solves the whole chain unsharded on its own GPU: the segment must agree with it
    ref = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=rank))
    for _ in range(5):
        _, st = ref.iterate_gn()
    p0, v0 = ref.get_states()
    bounds = sharded.partition(N, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    ok = (np.abs(pose - p0[lo:hi]).max() <= 1e-9 * max(1.0, np.abs(p0).max()) and np.abs(vel - v0[lo:hi]).max() <= 1e-9 * max(1.0, np.abs(v0).max())
          and abs(hist[-1]["error_after"] - st.error_after) <= 1e-9 * max(1.0, st.error_after))
    # (round 5) the optimiser LOOPS of the C ABI across the ranks: gpslam_hip_iterate_lm / gpslam_hip_optimize on every rank's handle,
    # the collectives behind gpslam_hip_set_collectives -- the lambda schedule and the iteration count of the unsharded handle
    def restart():
        s.set_states(lp["pose"], lp["vel"])
        if "halo_pose" in lp:
            s.set_halo_state(lp["halo_pose"], lp["halo_vel"])
        ref.set_states(problem["pose"], problem["vel"])
    restart()
    lam_s = lam_r = 1e-5
    for _ in range(3):
        st_s, lam_s = sv.iterate_lm(lam_s)
        _, st_r, lam_r = ref.iterate_lm(lam_r)[:3]
        ok = ok and lam_s == lam_r and st_s["accepted"] == bool(st_r.accepted) and abs(st_s["error_after"] - st_r.error_after) <= 1e-9 * max(1.0, st_r.error_after)
    # ... and LevenbergMarquardtOptimizer::iterate through the C ABI itself (gpslam_hip_iterate_lm on this rank's handle, the collectives
    # enqueued by torch_collectives on the stream the library names)
    restart()
    lam_s = lam_r = 1e-5
    for _ in range(3):
        _, st_c, lam_s = s.iterate_lm(lam_s)[:3]
        _, st_r, lam_r = ref.iterate_lm(lam_r)[:3]
        ok = ok and lam_s == lam_r and bool(st_c.accepted) == bool(st_r.accepted) and int(st_c.trials) == int(st_r.trials) \
            and abs(st_c.error_after - st_r.error_after) <= 1e-9 * max(1.0, st_r.error_after)
    restart()
    _, so = sv.optimize()
    _, so_r = ref.optimize()
    ok = ok and so.iterations == so_r.iterations and abs(so.error_after - so_r.error_after) <= 1e-9 * max(1.0, so_r.error_after)
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_WORKERS_OK" if flag.item() == 1.0 else "RCCL_WORKERS_MISMATCH")
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
