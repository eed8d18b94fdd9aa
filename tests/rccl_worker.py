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
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_WORKERS_OK" if flag.item() == 1.0 else "RCCL_WORKERS_MISMATCH")
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
