"""world_size-2 gloo test of the segment-sharding host logic (gpslam_amd/sharded.py) on CPU.

The two phases are played by tests/segment_model.py (numpy + the oracle); what is under test is the partitioning,
the halo bookkeeping, the single all-gather of interface records and the algebra of the reduced interface system:
the sharded iteration must reproduce the unsharded oracle iteration on the same problem.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, kind, N, out, lm=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpslam_amd import sharded, synthetic as S
    from segment_model import SegmentModel
    problem = S.pose3_chain(N) if kind == S.POSE3 else S.linear_chain(N)
    lp = sharded.local_problem(problem, rank, world)
    backend = sharded.apply_local(lp, SegmentModel(kind, rank, world))
    sv = sharded.ShardedSolver(backend, backend.send, backend.recv, rank, world, dist=dist)
    hist = []
    if lm:          # LevenbergMarquardtOptimizer::iterate across the ranks: the caller-owned loop around gpslam_hip_lm_decide
        lam = 1e-5
        for _ in range(6):
            st, lam = sv.iterate_lm(lam)
            hist.append(dict(st, lam=lam))
    else:
        for _ in range(5):
            hist.append(sv.iterate())
    pose, vel = backend.get_states()
    torch.save(dict(pose=pose, vel=vel, hist=hist, lo=lp["lo"], hi=lp["hi"]), "%s.%d" % (out, rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind_name,N", [("pose3", 41), ("linear3", 57)])
def test_sharded_iteration_matches_unsharded_oracle(tmp_path, kind_name, N):
    sys.path.insert(0, ROOT)
    from gpslam_amd import synthetic as S
    from oracle import oracle as O
    kind = S.POSE3 if kind_name == "pose3" else S.LINEAR3
    out = str(tmp_path / "res")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, kind, N, out), nprocs=2, join=True)
    parts = [torch.load("%s.%d" % (out, r), weights_only=False) for r in range(2)]
    pose = np.vstack([p["pose"] for p in parts])
    vel = np.vstack([p["vel"] for p in parts])
    problem = S.pose3_chain(N) if kind == S.POSE3 else S.linear_chain(N)
    ref = S.apply(problem, O.Chain(kind))
    ref_hist = []
    for _ in range(5):
        rc, st = ref.iterate_gn()
        ref_hist.append(st)
    p0, v0 = ref.get_states()
    assert np.abs(pose - p0).max() <= 1e-9 * max(1.0, np.abs(p0).max())
    assert np.abs(vel - v0).max() <= 1e-9 * max(1.0, np.abs(v0).max())
    for a, b in zip(parts[0]["hist"], ref_hist):          # reduced scalars equal the unsharded ones
        assert abs(a["error_before"] - b.error_before) <= 1e-8 * max(1.0, b.error_before)
        assert abs(a["error_after"] - b.error_after) <= 1e-8 * max(1.0, b.error_after)
        assert abs(a["delta_inf_norm"] - b.delta_inf_norm) <= 1e-8 * max(1.0, b.delta_inf_norm) + 1e-12
    assert parts[0]["hist"] == parts[1]["hist"]            # every rank sees the same reduced statistics


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_levenberg_marquardt_matches_unsharded_oracle(tmp_path, world):
    """ShardedSolver.iterate_lm over gloo at world size 2 and 3: every rank takes gpslam_hip_lm_decide's branch
    on identical all-gathered scalars -- the oracle's lambda schedule, accept flags and trial counts while the cost moves, the
    rule of tests/lm_lockstep.py once it does not (one trial, lambda kept or divided once)."""
    sys.path.insert(0, ROOT)
    from gpslam_amd import synthetic as S
    from oracle import oracle as O
    N = 47
    out = str(tmp_path / "res")
    port = 26500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, S.POSE3, N, out, True), nprocs=world, join=True)
    parts = [torch.load("%s.%d" % (out, r), weights_only=False) for r in range(world)]
    for p in parts[1:]:
        assert p["hist"] == parts[0]["hist"]               # identical decisions and statistics on every rank
    ref = S.apply(S.pose3_chain(N), O.Chain(S.POSE3))
    lam, compared = 1e-5, 0
    for h in parts[0]["hist"]:
        _, st, new = ref.iterate_lm(lam)[:3]
        moved = abs(st.error_before - st.last_trial_error)
        assert abs(h["error_before"] - st.error_before) <= 1e-8 * max(1.0, st.error_before)
        if moved > 1e-10 * max(1.0, st.error_before) and abs(h["error_before"] - h["last_trial_error"]) > 1e-10 * max(1.0, st.error_before):
            assert (h["lam"], h["accepted"], h["trials"]) == (new, bool(st.accepted), st.trials), (h, new, st.accepted, st.trials)
            assert abs(h["error_after"] - st.error_after) <= 1e-8 * max(1.0, st.error_after)
            compared += 1
        else:                                               # converged: decided by rounding on either side
            assert h["trials"] == 1 and h["lam"] in (lam, lam / 10.0)
        lam = new if h["lam"] == new else h["lam"]
        if h["lam"] != new:                                 # the two parted on a rounding-level decision: nothing more to compare in lock step
            break
    assert compared >= 2
    pose = np.vstack([p["pose"] for p in parts])
    p0, _v0 = ref.get_states()
    assert np.abs(pose - p0).max() <= 1e-6 * max(1.0, np.abs(p0).max())


def test_partition_and_local_problem_cover_every_factor_once():
    sys.path.insert(0, ROOT)
    from gpslam_amd import sharded, synthetic as S
    p = S.pose3_chain(103)
    for P in (1, 2, 3, 8):
        b = sharded.partition(103, P)
        assert b[0] == 0 and b[-1] == 103 and all(b[i] < b[i + 1] for i in range(P))
        n_gp = n_btw = n_pri = 0
        for r in range(P):
            lp = sharded.local_problem(p, r, P)
            n_gp += len(lp["gp_left"])
            n_btw += len(lp["between_left"])
            n_pri += len(lp["prior_idx"])
            assert ("halo_pose" in lp) == (r < P - 1)
            if len(lp["gp_left"]):
                assert lp["gp_left"].max() <= lp["N"] - 1 and (r < P - 1 or lp["gp_left"].max() <= lp["N"] - 2)
        assert (n_gp, n_btw, n_pri) == (102, 102, 1)
