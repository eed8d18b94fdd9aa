"""Landmark elimination at scale (BASELINE config 4): the segmented path with fat separators (gpslam_amd/csrc/fatsep.hpp)
against the CPU oracle's dense bordered solve, through the C ABI."""
import numpy as np
import pytest

from oracle import oracle as O
from gpslam_amd import synthetic as S
from test_gpu_parity import gpu, states_close

pytestmark = pytest.mark.gpu


def _pair(p, **kw):
    orc = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    dev = S.apply(p, gpu().ChainSolver(O.POSE2, chart=gpu().CHART_FIRST_ORDER, landmark_dim=2, **kw))
    return orc, dev


@pytest.mark.parametrize("N,seglen", [(2000, 0), (2000, 500), (700, 250), (333, 128)])
def test_c4_local_landmarks_gauss_newton_matches_oracle(N, seglen):
    """N / 20 landmarks with a 200-state visibility window (100 at N = 2000: R = 201 columns, far beyond the dense border):
    Gauss-Newton iterations in lock step with the oracle, 1e-9 relative."""
    p = S.pose2_local_landmarks_chain(N, window=100 if N < 1000 else 200)
    orc, dev = _pair(p, segment_length=seglen)
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    for it in range(4):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (it, s0.error_after, s1.error_after)
        assert abs(s0.delta_inf_norm - s1.delta_inf_norm) <= 1e-8 * max(1.0, s0.delta_inf_norm)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)
    l0, l1 = orc.get_landmarks(), dev.get_landmarks()
    assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max())


def test_segmented_path_equals_dense_border_on_a_small_graph():
    """The same small graph (8 landmarks seen from everywhere would not segment; 6 local ones do) through both landmark
    paths of the library."""
    p = S.pose2_local_landmarks_chain(400, L=6, window=120)
    gp = gpu()
    a = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))
    b = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True, segment_length=150))
    for _ in range(3):
        _, sa = a.iterate_gn()
        _, sb = b.iterate_gn()
        assert abs(sa.error_after - sb.error_after) <= 1e-9 * max(1.0, sa.error_after)
    states_close(O.POSE2, *a.get_states(), *b.get_states(), rel=1e-9)
    assert np.abs(a.get_landmarks() - b.get_landmarks()).max() <= 1e-9 * 100


def test_c4_levenberg_marquardt_matches_oracle():
    p = S.pose2_local_landmarks_chain(1200, window=200)
    orc, dev = _pair(p)
    lam0 = lam1 = 1e-5
    for it in range(5):
        rc0, s0, lam0 = orc.iterate_lm(lam0)
        rc1, s1, lam1 = dev.iterate_lm(lam1)
        assert rc0 == 0 and rc1 == 0
        assert lam0 == lam1 and s0.accepted == s1.accepted, (it, lam0, lam1)
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)


def test_landmark_seen_from_everywhere_is_rejected_with_a_message():
    p = S.pose2_range_chain(600, L=80)          # 80 landmarks each seen from the whole chain: no cut holds them
    gp = gpu()
    with pytest.raises(gp.GpslamHipError, match="more than two segments|too many landmarks per cut"):
        S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))
