"""Worker of tests/test_gpu_split.py::test_split_chain_one_process_per_gpu_over_rccl: BASELINE config 4's graph split into one
piece per GPU (gpslam_amd/sharded.py: SplitSolver; the all-gather of the interface records runs over RCCL).
python rccl_split_worker.py <rank> <world> <port> <N>"""
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
    kw = dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, device=rank)
    problem = S.pose2_local_landmarks_chain(N, L=N // 20, window=200)
    lp = sharded.split_local_problem(problem, rank, world)
    s = gpslam_amd.ChainSolver(gpslam_amd.POSE2, **kw)
    sharded.apply_split(lp, s, rank, world)
    sv = sharded.SplitSolver(s, rank, world, dist=dist)            # agrees on the record size with one all-reduce (max)
    hist = [sv.iterate() for _ in range(6)]
    pose, vel = s.get_states()
    lmk = s.get_landmarks()
    # every rank also solves the whole chain unsplit on its own GPU: its piece must agree with it
    ref = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, force_segmented=True, **kw))
    for _ in range(6):
        _, st = ref.iterate_gn()
    p0, v0 = ref.get_states()
    l0 = ref.get_landmarks()[lp["lm_global"]]
    lo, hi = lp["lo"], lp["hi"]
    ok = (np.abs(pose - p0[lo:hi + 1]).max() <= 1e-9 * max(1.0, np.abs(p0).max())
          and np.abs(vel - v0[lo:hi + 1]).max() <= 1e-8 * max(1.0, np.abs(v0).max())
          and np.abs(lmk - l0).max() <= 1e-8 * max(1.0, np.abs(l0).max())
          and abs(hist[-1]["error_after"] - st.error_after) <= 1e-7 * max(1.0, st.error_after))
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_SPLIT_OK" if flag.item() == 1.0 else "RCCL_SPLIT_MISMATCH")
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
