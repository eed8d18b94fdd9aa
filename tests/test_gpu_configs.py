"""The synthetic factor mixes of BASELINE configs 4 and 5 (gpslam_amd/synthetic.py) at sizes the oracle solves in
seconds: HIP path vs oracle, Gauss-Newton (C4') and Levenberg-Marquardt (C5, whose yaw is only weakly observable)."""
import numpy as np
import pytest

from oracle import oracle as O
from gpslam_amd import synthetic as S
from test_gpu_parity import gpu, states_close

pytestmark = pytest.mark.gpu


def test_c4_pose2_ranges_gauss_newton():
    p = S.pose2_range_chain(700, L=6)
    orc = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    dev = S.apply(p, gpu().ChainSolver(O.POSE2, chart=gpu().CHART_FIRST_ORDER, landmark_dim=2))
    assert len(p["range_left"]) > 200
    for it in range(7):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-6 * max(1.0, s0.error_after), it
    assert s1.delta_inf_norm < 1e-3
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    assert np.abs(x0 - x1).max() <= 1e-7 and np.abs(v0 - v1).max() <= 1e-7
    assert np.abs(orc.get_landmarks() - dev.get_landmarks()).max() <= 1e-7
    # the estimate is a real improvement over dead reckoning
    err = lambda x: np.linalg.norm(x[:, :2] - p["truth"][:, :2], axis=1).mean()
    assert err(x1) < 0.5 * err(p["pose"])


def test_c5_rot3_attitude_levenberg_marquardt():
    p = S.rot3_attitude_chain(300)
    orc = S.apply(p, O.Chain(O.ROT3))
    dev = S.apply(p, gpu().ChainSolver(O.ROT3))
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    lam0 = lam1 = 1e-5
    for it in range(4):
        rc0, s0, lam0 = orc.iterate_lm(lam0)[:3]
        rc1, s1, lam1 = dev.iterate_lm(lam1)[:3]
        assert rc0 == 0 and rc1 == 0 and lam0 == lam1, it
        assert abs(s0.error_after - s1.error_after) <= 1e-6 * max(1.0, s0.error_after), it
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    states_close(O.ROT3, x0, v0, x1, v1, 1e-7)
