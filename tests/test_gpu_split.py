"""BASELINE config 4 across GPUs: the segmented landmark elimination on a chain split into pieces joined at shared cut
states (include/gpslam_hip.h, gpslam_hip_fs_set_split; gpslam_amd/sharded.py, SplitSolver).  P handles on ONE device play
the P ranks (the all-gather is a device copy), so the whole HIP path runs on the single-GPU build farm; the reference is the
unsplit segmented solve of the same graph and, at the small size, the oracle's dense bordered solve."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _pieces(problem, P, segment_length=0, precision=0):
    import gpslam_amd
    from gpslam_amd import sharded
    locals_, pieces = [], []
    for r in range(P):
        lp = sharded.split_local_problem(problem, r, P)
        s = gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, segment_length=segment_length,
                                   precision=precision)
        sharded.apply_split(lp, s, r, P)
        locals_.append(lp)
        pieces.append(sharded.SplitSolver(s, r, P))
    nb_top = max(sv.nb_local for sv in pieces)
    for sv in pieces:
        sv.set_top(nb_top)
    return locals_, pieces


def _merged(problem, locals_, pieces):
    from gpslam_amd import sharded
    return sharded.merge_pieces(problem, locals_, [sv.backend.get_states() for sv in pieces], [sv.backend.get_landmarks() for sv in pieces])


@pytest.mark.parametrize("P,N,L,window,C", [(1, 700, 35, 200, 0), (2, 2000, 100, 200, 0), (3, 3000, 150, 200, 0), (4, 3001, 150, 120, 128),
                                            (5, 6000, 300, 200, 256), (8, 12000, 600, 200, 0), (4, 200000, 10000, 200, 0), (4, 1000000, 50000, 200, 0)])   # the last one: BASELINE config 4 at full size
def test_split_chain_equals_the_unsplit_segmented_solve(P, N, L, window, C):
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose2_local_landmarks_chain(N, L=L, window=window)
    locals_, pieces = _pieces(problem, P, C)
    ref = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True))
    for it in range(6):
        got = sharded.iterate_pieces(pieces)
        _, st = ref.iterate_gn()
        assert abs(got["error_before"] - st.error_before) <= 1e-9 * max(1.0, st.error_before)
        assert abs(got["error_after"] - st.error_after) <= 1e-7 * max(1.0, st.error_after)
        assert abs(got["delta_inf_norm"] - st.delta_inf_norm) <= 1e-6 * max(1.0, st.delta_inf_norm) + 1e-10
    pose, vel, lmk = _merged(problem, locals_, pieces)
    p1, v1 = ref.get_states()
    assert np.abs(pose - p1).max() <= 1e-9 * max(1.0, np.abs(p1).max())
    assert np.abs(vel - v1).max() <= 1e-8 * max(1.0, np.abs(v1).max())
    assert np.abs(lmk - ref.get_landmarks()).max() <= 1e-8 * max(1.0, np.abs(lmk).max())
    # the two copies of every shared state / landmark moved in lock step (same top solve on both sides): bit-identical
    for a, b, la, lb in zip(pieces[:-1], pieces[1:], locals_[:-1], locals_[1:]):
        pa, va = a.backend.get_states()
        pb, vb = b.backend.get_states()
        assert np.array_equal(pa[-1], pb[0]) and np.array_equal(va[-1], vb[0])
        assert np.array_equal(a.backend.get_landmarks()[la["last_lm"]], b.backend.get_landmarks()[lb["first_lm"]])
    for sv in pieces:
        sv.backend.close()
    ref.close()


def test_split_chain_against_the_oracle():
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose2_local_landmarks_chain(1200, L=60, window=120)
    locals_, pieces = _pieces(problem, 3)
    orc = S.apply(problem, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    for it in range(5):
        got = sharded.iterate_pieces(pieces)
        _, s0 = orc.iterate_gn()
        assert abs(got["error_after"] - s0.error_after) <= 1e-7 * max(1.0, s0.error_after)
    pose, vel, lmk = _merged(problem, locals_, pieces)
    p0, v0 = orc.get_states()
    assert np.abs(pose - p0).max() <= 1e-8 * max(1.0, np.abs(p0).max())
    assert np.abs(vel - v0).max() <= 1e-7 * max(1.0, np.abs(v0).max())
    assert np.abs(lmk - orc.get_landmarks()).max() <= 1e-7 * max(1.0, np.abs(lmk).max())
    for sv in pieces:
        sv.backend.close()


@pytest.mark.parametrize("P,N,L,C,precision", [(2, 1300, 225, 256, 0), (3, 2000, 340, 256, 0), (2, 1300, 240, 256, 1)],
                         ids=["2-pieces-NB96", "3-pieces-NB100+", "2-pieces-fp32-rows"])
def test_split_chain_with_fat_blocks_beyond_80_columns(P, N, L, C, precision):
    """Round 5 (kFatMax 128): pieces whose fat blocks -- and whose shared top blocks (gpslam_hip_fs_set_top) -- are wider than 80
    columns run k_fat_elim_wide both in their own cyclic reduction and in the redundant top solve."""
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose2_local_landmarks_chain(N, L=L, window=200)
    locals_, pieces = _pieces(problem, P, C, precision=precision)
    assert max(sv.nb_local for sv in pieces) > 80
    ref = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True,
                                                  segment_length=C, precision=precision))
    assert ref.segment_plan()["NB"] > 80
    tol = 1e-9 if precision == 0 else 1e-6
    for it in range(5):
        got = sharded.iterate_pieces(pieces)
        _, st = ref.iterate_gn()
        assert abs(got["error_after"] - st.error_after) <= max(1e-7, tol) * max(1.0, st.error_after)
    pose, vel, lmk = _merged(problem, locals_, pieces)
    p1, v1 = ref.get_states()
    assert np.abs(pose - p1).max() <= tol * max(1.0, np.abs(p1).max())
    assert np.abs(lmk - ref.get_landmarks()).max() <= 10 * tol * max(1.0, np.abs(lmk).max())
    if precision == 0:          # ... and the oracle's dense bordered solve
        orc = S.apply(problem, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
        for it in range(5):
            orc.iterate_gn()
        po, vo = orc.get_states()
        assert np.abs(pose - po).max() <= 1e-9 * max(1.0, np.abs(po).max())
    for sv in pieces:
        sv.backend.close()
    ref.close()


@pytest.mark.parametrize("P", [1, 3])
def test_split_levenberg_marquardt_follows_the_unsplit_lambda_schedule(P):
    """LevenbergMarquardtOptimizer::iterate as matlab/PlazaPose2.m:217-229 drives it, from an open-loop dead-reckoned start
    (no re-anchoring: the error falls from 1.5e7 to 1.8e3 over the first iterations), until the reference run has converged."""
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose2_local_landmarks_chain(2400, L=120, window=160, anchor=0)
    locals_, pieces = _pieces(problem, P)
    ref = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True))
    import lm_lockstep
    lam, lams = 1e-5, []
    for it in range(10):
        st, _, lam, _ = lm_lockstep.step(ref, lambda lam_: sharded.iterate_pieces_lm(pieces, lam_), lam, err_tol=1e-6, tag=it)
        lams.append(lam)
    assert len(set(lams)) >= 4
    pose, vel, lmk = _merged(problem, locals_, pieces)
    p1, v1 = ref.get_states()
    assert np.abs(pose - p1).max() <= 1e-7 * max(1.0, np.abs(p1).max())
    assert np.abs(lmk - ref.get_landmarks()).max() <= 1e-7 * max(1.0, np.abs(lmk).max())
    if P > 1:
        with pytest.raises(gpslam_amd.GpslamHipError, match="collectives"):
            pieces[0].backend.iterate_lm(1e-5)      # without the host's collectives (gpslam_hip_set_collectives) the loop is the caller's
    for sv in pieces:
        sv.backend.close()
    ref.close()


def test_segmented_and_split_chains_with_fp32_jacobian_rows():
    """GPSLAM_FP32 (fp32 row tables; residual, normal equations, Schur complements and fat solve stay fp64: DESIGN.md section 4b)
    on the segmented landmark path, unsplit and cut in three: the split run follows the unsplit fp32 run (same rows, another
    summation order), and both land within fp32-row accuracy of the fp64 solve."""
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose2_local_landmarks_chain(3000, L=150, window=200)
    kw = dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True)
    ref64 = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, **kw))
    ref32 = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, precision=gpslam_amd.FP32, **kw))
    locals_, pieces = _pieces(problem, 3, precision=gpslam_amd.FP32)
    for it in range(8):
        got = sharded.iterate_pieces(pieces)
        _, s32 = ref32.iterate_gn()
        _, s64 = ref64.iterate_gn()
        assert abs(got["error_after"] - s32.error_after) <= 1e-7 * max(1.0, s32.error_after)
    assert abs(s32.error_after - s64.error_after) <= 1e-5 * max(1.0, s64.error_after)      # at convergence
    pose, vel, lmk = _merged(problem, locals_, pieces)
    p32, v32 = ref32.get_states()
    p64, v64 = ref64.get_states()
    assert np.abs(pose - p32).max() <= 1e-7 * max(1.0, np.abs(p32).max())
    assert np.abs(lmk - ref32.get_landmarks()).max() <= 1e-7 * max(1.0, np.abs(lmk).max())
    assert np.abs(p32 - p64).max() <= 1e-5 * max(1.0, np.abs(p64).max())          # north_star's fp32 tolerance
    assert np.abs(ref32.get_landmarks() - ref64.get_landmarks()).max() <= 1e-5 * max(1.0, np.abs(lmk).max())
    for sv in pieces:
        sv.backend.close()
    ref32.close()
    ref64.close()


@pytest.mark.parametrize("P", [2, 3])
def test_c_abi_levenberg_marquardt_on_split_pieces(P):
    """gpslam_hip_iterate_lm / gpslam_hip_iterate_gn THEMSELVES on the pieces of a split chain (round 5: gpslam_hip_set_collectives; one
    thread per piece plays the ranks, tests/thread_ranks.py): the optimiser matlab/PlazaPose2.m:210-226 runs on this kind of graph,
    from an open-loop dead-reckoned start -- the lambda schedule and the errors of the unsplit segmented solve."""
    import gpslam_amd
    from gpslam_amd import synthetic as S
    from thread_ranks import ThreadRanks
    problem = S.pose2_local_landmarks_chain(2400, L=120, window=160, anchor=0)
    locals_, pieces = _pieces(problem, P)
    ref = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True))
    tr = ThreadRanks(P)
    for r, sv in enumerate(pieces):
        sv.backend.set_collectives(*tr.collectives(r))

    def lm_run(r):
        lam, hist = 1e-5, []
        for _ in range(6):
            _rc, st, lam = pieces[r].backend.iterate_lm(lam)[:3]
            hist.append((lam, int(st.accepted), int(st.trials), st.error_before, st.error_after))
        return hist
    hists = tr.run(lm_run)
    lam = 1e-5
    for it in range(6):
        _rc, st, lam = ref.iterate_lm(lam)[:3]
        for r in range(P):
            h = hists[r][it]
            assert h == hists[0][it]
            assert h[:3] == (lam, int(st.accepted), int(st.trials)), (it, r, h, lam)
            assert abs(h[3] - st.error_before) <= 1e-8 * max(1.0, st.error_before) and abs(h[4] - st.error_after) <= 1e-6 * max(1.0, st.error_after)
    _rc, st_ref = ref.iterate_gn()
    for st in tr.run(lambda r: pieces[r].backend.iterate_gn()[1]):
        assert abs(st.error_after - st_ref.error_after) <= 1e-6 * max(1.0, st_ref.error_after)
    pose, vel, lmk = _merged(problem, locals_, pieces)
    p1, v1 = ref.get_states()
    assert np.abs(pose - p1).max() <= 1e-6 * max(1.0, np.abs(p1).max())
    for sv in pieces:
        sv.backend.close()
    ref.close()


def test_split_handles_refuse_the_whole_chain_entry_points_and_bad_plans():
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose2_local_landmarks_chain(2000, L=100, window=200)
    lp = sharded.split_local_problem(problem, 0, 2)
    s = gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2)
    sharded.apply_split(lp, s, 0, 2)
    with pytest.raises(gpslam_amd.GpslamHipError):
        s.iterate_gn()
    with pytest.raises(gpslam_amd.GpslamHipError):
        s.run_gn(2)
    with pytest.raises(gpslam_amd.GpslamHipError):
        s.optimize()                           # every whole-chain driver ends in the same refusal
    with pytest.raises(gpslam_amd.GpslamHipError):
        s.fs_phase1(0.0)                       # fs_set_top has not been called
    with pytest.raises(gpslam_amd.GpslamHipError):
        s.fs_set_top(4)                        # smaller than this piece's own fat blocks
    s.fs_set_top(s.fs_split_info()["fat_block"])
    s.fs_phase1(0.0)
    s.close()
    # pieces shorter than a landmark's window of visibility: refused when the problem is cut
    with pytest.raises(ValueError):
        sharded.split_local_problem(S.pose2_local_landmarks_chain(600, L=30, window=400), 1, 6)
    # the two ends of the whole chain share nothing
    t = gpslam_amd.ChainSolver(gpslam_amd.POSE2, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2)
    with pytest.raises(gpslam_amd.GpslamHipError):
        t.fs_set_split(0, 2, [1], [])
    t.close()


def test_split_chain_one_process_per_gpu_over_rccl():
    """tests/rccl_split_worker.py, one process per visible GPU (up to eight) over RCCL; on the single-GPU build farm the
    worker runs with world size 1 -- the collectives (the all-reduce of the record size, the all-gather of the records and
    of the statistics) still go through torch.distributed / RCCL."""
    import os
    import subprocess
    import sys
    import torch
    world = max(1, min(torch.cuda.device_count(), 8))
    port = 29300 + os.getpid() % 90
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_split_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(3000 * world)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "RCCL_SPLIT_OK" in outs[0]
