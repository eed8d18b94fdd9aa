"""BASELINE config 4 across GPUs, host side, on CPU: gpslam_amd/sharded.py's SplitSolver over gloo with world_size 2 and 3.
The two phases are played by tests/split_model.py (the oracle's normal equations of each piece + dense numpy, records in the
HIP library's layout); under test are the cut of the graph into overlapping pieces (who owns which factor, landmark and
prior), the agreement on the record size, the single all-gather, and the algebra of the shared-separator system: the split
iteration must reproduce the unsplit oracle iteration."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(N, anchor=256):
    from gpslam_amd import synthetic as S
    return S.pose2_local_landmarks_chain(N, L=N // 20, window=100, anchor=anchor)


def _worker(rank, world, port, N, out, lm=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpslam_amd import sharded
    from oracle import oracle as O
    from split_model import SplitPieceModel
    lp = sharded.split_local_problem(_problem(N, 0 if lm else 256), rank, world)   # LM: open-loop dead reckoning, a real descent
    backend = sharded.apply_split(lp, SplitPieceModel(O.POSE2, O.CHART_FIRST_ORDER, 2), rank, world)
    sv = sharded.SplitSolver(backend, rank, world, dist=dist, device="cpu")
    if lm:
        hist, lam = [], 1e-5
        for _ in range(4):
            st, lam = sv.iterate_lm(lam)
            hist.append(dict(st, lam=lam))
    else:
        hist = [sv.iterate() for _ in range(4)]
    torch.save(dict(lp=lp, states=backend.get_states(), lmk=backend.get_landmarks(), hist=hist), "%s.%d" % (out, rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 300), (3, 420)])
def test_split_iteration_matches_unsplit_oracle(tmp_path, world, N):
    sys.path.insert(0, ROOT)
    from gpslam_amd import sharded, synthetic as S
    from oracle import oracle as O
    out = str(tmp_path / "res")
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, N, out), nprocs=world, join=True)
    parts = [torch.load("%s.%d" % (out, r), weights_only=False) for r in range(world)]
    problem = _problem(N)
    pose, vel, lmk = sharded.merge_pieces(problem, [p["lp"] for p in parts], [p["states"] for p in parts], [p["lmk"] for p in parts])
    ref = S.apply(problem, O.Chain(O.POSE2, O.CHART_FIRST_ORDER, landmark_dim=2))
    for k in range(4):
        _, st = ref.iterate_gn()
        for p in parts:                                  # every rank holds the same reduced scalars
            h = p["hist"][k]
            assert abs(h["error_before"] - st.error_before) <= 1e-9 * max(1.0, st.error_before)
            assert abs(h["error_after"] - st.error_after) <= 1e-8 * max(1.0, st.error_after)
            assert abs(h["delta_inf_norm"] - st.delta_inf_norm) <= 1e-7 * max(1.0, st.delta_inf_norm)
    p0, v0 = ref.get_states()
    assert np.abs(pose - p0).max() <= 1e-9 * max(1.0, np.abs(p0).max())
    assert np.abs(vel - v0).max() <= 1e-8 * max(1.0, np.abs(v0).max())
    assert np.abs(lmk - ref.get_landmarks()).max() <= 1e-8 * max(1.0, np.abs(lmk).max())
    for a, b in zip(parts[:-1], parts[1:]):              # both copies of a shared state / landmark end up identical
        assert np.allclose(a["states"][0][-1], b["states"][0][0], rtol=0, atol=1e-12)
        assert np.allclose(a["lmk"][a["lp"]["last_lm"]], b["lmk"][b["lp"]["first_lm"]], rtol=0, atol=1e-12)


def test_the_cut_conserves_factors_landmarks_and_priors():
    sys.path.insert(0, ROOT)
    from gpslam_amd import sharded
    problem = _problem(2000)
    L = len(problem["landmarks"])
    for P in (1, 2, 3, 5):
        lps = [sharded.split_local_problem(problem, r, P) for r in range(P)]
        assert sum(len(lp["range_left"]) for lp in lps) == len(problem["range_left"])
        assert sum(len(lp["gp_left"]) for lp in lps) == len(problem["gp_left"])
        assert sum(len(lp["between_left"]) for lp in lps) == len(problem["between_left"])
        assert sum(len(lp["prior_idx"]) for lp in lps) == len(problem["prior_idx"])
        assert sum(len(lp["lprior_idx"]) for lp in lps) == L                 # every landmark prior exactly once
        assert sum(int(lp["own_lm"].sum()) for lp in lps) == L               # every landmark reported exactly once
        assert lps[0]["lo"] == 0 and lps[-1]["hi"] == problem["N"] - 1
        for a, b in zip(lps[:-1], lps[1:]):
            assert a["hi"] == b["lo"]                                        # the pieces overlap in one state
            assert np.array_equal(a["lm_global"][a["last_lm"]], b["lm_global"][b["first_lm"]])   # same landmarks, same order
        for lp in lps:
            assert len(lp["range_lm"]) == 0 or (lp["range_lm"].min() >= 0 and lp["range_lm"].max() < len(lp["landmarks"]))
    with pytest.raises(ValueError):
        sharded.split_local_problem(_problem(400), 1, 8)                    # pieces shorter than the window of visibility


def test_split_levenberg_marquardt_matches_unsplit_oracle(tmp_path):
    """SplitSolver.iterate_lm over gloo, world size 2: the decisions come from all-gathered scalars, so both ranks follow
    the oracle's lambda schedule (matlab/PlazaPose2.m:217-229 optimises this kind of graph with LM)."""
    sys.path.insert(0, ROOT)
    from gpslam_amd import sharded, synthetic as S
    from oracle import oracle as O
    out = str(tmp_path / "res")
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, 300, out, True), nprocs=2, join=True)
    parts = [torch.load("%s.%d" % (out, r), weights_only=False) for r in range(2)]
    problem = _problem(300, 0)
    ref = S.apply(problem, O.Chain(O.POSE2, O.CHART_FIRST_ORDER, landmark_dim=2))
    lam, compared = 1e-5, 0
    for k in range(4):
        _, st, lam = ref.iterate_lm(lam)[:3]
        for p in parts:
            h = p["hist"][k]
            assert h["lam"] == lam and h["accepted"] == bool(st.accepted)
            assert abs(h["error_after"] - st.error_after) <= 1e-8 * max(1.0, st.error_after)
        compared += 1
        if st.error_before - st.error_after <= 1e-7 * st.error_before:
            break           # converged: accept / reject now hangs on the rounding of err - newErr
    assert compared >= 2
    if compared < 4:
        return
    pose, vel, lmk = sharded.merge_pieces(problem, [p["lp"] for p in parts], [p["states"] for p in parts], [p["lmk"] for p in parts])
    p0, v0 = ref.get_states()
    assert np.abs(pose - p0).max() <= 1e-9 * max(1.0, np.abs(p0).max())
    assert np.abs(lmk - ref.get_landmarks()).max() <= 1e-8 * max(1.0, np.abs(lmk).max())
