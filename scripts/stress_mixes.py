"""The synthetic factor mixes of the BASELINE configs at random small sizes against the oracle (3 Gauss-Newton iterations, 1e-9, then 5
Levenberg-Marquardt iterations in lock step).
   python scripts/stress_mixes.py [count] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from oracle import oracle as O
import test_gpu_parity as T
import lm_lockstep as LM
import gpslam_amd
from gpslam_amd import synthetic as S
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for t in range(cnt):
    N = int(rng.integers(150, 1400))
    which = t % 4
    kw, okw = {}, {}
    if which == 0:
        p = S.pose3_gps_chain(N, per_interval=int(rng.integers(1, 5)), seed=t, keep_odometry=True)
    elif which == 1:
        p = S.rot3_attitude_chain(N, seed=t)
    elif which == 2:
        p = S.pose2_local_landmarks_chain(N, window=100, seed=t)
        kw = dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2); okw = dict(chart=O.CHART_FIRST_ORDER, landmark_dim=2)
    else:
        p = S.pose3_gps_chain(N, per_interval=2, seed=t, keep_odometry=False)
        fix = np.arange(0, N, 15).astype(np.int32)      # (position fixes alone leave the attitude weakly observable)
        p.update(prior_idx=fix, prior_pose=p["pose"][fix].copy(), prior_sig=np.full((len(fix), 6), 0.05))
    orc = S.apply(p, O.Chain(p["kind"], **okw))
    dev = S.apply(p, gpslam_amd.ChainSolver(p["kind"], **kw))
    # the oracle's twin, started 1e-15 (relative) away: how far the two ORACLES part in three iterations is what rounding alone does to
    # this graph (the landmark mix has soft directions; 200 cases, seed 502: one such graph with velocities 1e-9 apart)
    twin = S.apply(p, O.Chain(p["kind"], **okw))
    twin.set_states(p["pose"] * (1.0 + 1e-15 * np.random.default_rng(t).standard_normal(p["pose"].shape)), p["vel"])
    for it in range(3):
        rc0, s0 = orc.iterate_gn(); rc1, s1 = dev.iterate_gn(); rc2, s2 = twin.iterate_gn()
        assert rc0 == 0 and rc1 == 0 and rc2 == 0, (t, which, N, rc0, rc1)
        # (+ 1e-11 error_before: a step that takes the cost from 8.6e6 to 2e3 leaves error_after with the rounding of the larger number --
        #  the oracle alone moves by 1e-9 of error_after there when its input states are perturbed by 1e-15)
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, abs(s0.error_after)) + 1e-11 * abs(s0.error_before) + 10 * abs(s0.error_after - s2.error_after), (t, which, N, it, s0.error_after, s1.error_after)
    (x0, v0), (x2, v2) = orc.get_states(), twin.get_states()
    noise = max(np.abs(x0 - x2).max() / max(1.0, np.abs(x0).max()), np.abs(v0 - v2).max() / max(1.0, np.abs(v0).max()))
    T.states_close(p["kind"], *orc.get_states(), *dev.get_states(), 1e-9 + 10 * noise)
    if "landmarks" in p:
        l0, l1, l2 = orc.get_landmarks(), dev.get_landmarks(), twin.get_landmarks()
        assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max()) + 10 * np.abs(l0 - l2).max()
    # Levenberg-Marquardt from the initial values, in lock step (tests/lm_lockstep.py: the lambda schedule exactly while the cost
    # moves, its rule past convergence)
    for s_ in (orc, dev):
        s_.set_states(p["pose"], p["vel"])
        if "landmarks" in p:
            s_.set_landmarks(p["landmarks"])
    lam, n_noise, slack = LM.run(orc, dev, 1e-3, 5, tag=(which, N))
    assert slack <= 1e-3, (t, which, N, slack)     # (range-only landmarks leave flat directions: a 1e-4 step there moves the cost by 1e-10 of itself)
    T.states_close(p["kind"], *orc.get_states(), *dev.get_states(), 1e-9 + 2 * slack)
    print("ok mix %d N %d plan %s (LM: lambda %.1e, %d of 5 calls at rounding level)" % (which, N, dev.plan_info(), lam, n_noise))
print("all %d mixes agree with the oracle" % cnt)
