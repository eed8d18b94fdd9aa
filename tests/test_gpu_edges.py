"""Edge sizes through the C ABI: chains of 1, 2, 3 ... states (single chunk, chunk boundaries, one block per level) and
landmark borders from R = 3 to the limit R = 28 (FAST and non-FAST solver paths), HIP vs oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from gpslam_amd import synthetic as S
import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [O.POSE3, O.POSE2, O.LINEAR2], ids=["pose3", "pose2", "linear2"])
def test_chain_lengths(kind):
    for N in (1, 2, 3, 16, 17, 26, 51, 401):
        orc, dev, c = T.build_pair(kind, N, seed=N)
        for _ in range(4):
            rc0, s0 = orc.iterate_gn()
            rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0, N
        (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
        T.states_close(kind, x0, v0, x1, v1, 1e-9)


def test_pose2_landmark_border_widths():
    gp = T.gpu()
    for L in (1, 4, 5, 13):                    # R = 3, 9, 11 (last FAST width for b = 6), 27
        p = S.pose2_range_chain(400, L=L, rate=0.8)
        orc = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
        dev = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))
        for _ in range(6):
            rc0, s0 = orc.iterate_gn()
            rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0
        assert np.abs(orc.get_states()[0] - dev.get_states()[0]).max() < 1e-7
        assert np.abs(orc.get_landmarks() - dev.get_landmarks()).max() < 1e-7
    # 14 landmarks: 29 right-hand-side columns do not fit the dense border -> the segmented elimination takes over (all 14
    # are seen from the whole 100-state chain: one segment between two fat separators of 7 landmarks each)
    p = S.pose2_range_chain(100, L=14, rate=0.8)
    orc = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    dev = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))
    assert dev.segment_plan()["active"] == 1
    for _ in range(6):
        orc.iterate_gn()
        dev.iterate_gn()
    assert np.abs(orc.get_states()[0] - dev.get_states()[0]).max() < 1e-7
    assert np.abs(orc.get_landmarks() - dev.get_landmarks()).max() < 1e-7


def test_pose3_landmark_border_widths():
    gp = T.gpu()
    for L in (2, 3, 9):                        # R = 7 (FAST), 10 (first non-FAST width for b = 12), 28 (the limit)
        c = T.random_chain(O.POSE3, 60, 3)
        rng = np.random.default_rng(L)
        lm = c["truth_pose"][rng.integers(0, 60, L), 9:12] + rng.uniform(3, 6, (L, 3))
        left = np.repeat(np.arange(59), 2).astype(np.int32)
        lmi = rng.integers(0, L, len(left)).astype(np.int32)
        tau = c["dt"][left] * rng.random(len(left))
        z = np.linalg.norm(lm[lmi] - c["truth_pose"][left, 9:12], axis=1) + 0.05 * rng.standard_normal(len(left))
        sol = []
        for mk in (lambda: O.Chain(O.POSE3, landmark_dim=3), lambda: gp.ChainSolver(O.POSE3, landmark_dim=3)):
            s = mk()
            s.set_qc(0.02 * np.eye(6))
            s.set_states(c["pose"], c["vel"])
            s.set_landmarks(lm + 0.1)
            s.add_gp_priors(np.arange(59), c["dt"])
            fix = np.arange(0, 60, 10)
            s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), 6), 0.01))
            s.add_vel_priors([0, 59], c["truth_vel"][[0, 59]], np.full((2, 6), 0.05))
            s.add_landmark_priors(np.arange(L), lm, np.full((L, 3), 0.5))
            s.add_interp_range(left, lmi, z, np.full(len(left), 0.1), c["dt"][left], tau)
            s.compile()
            sol.append(s)
        for _ in range(5):
            rc0, s0 = sol[0].iterate_gn()
            rc1, s1 = sol[1].iterate_gn()
            assert rc0 == 0 and rc1 == 0
        T.states_close(O.POSE3, *sol[0].get_states(), *sol[1].get_states(), 1e-8)
