"""BASELINE config 1: the Plaza2 range-only SLAM log through the reference's graph recipe (matlab/PlazaPose2.m).

CPU part: fixture integrity, the range bias fit, and the oracle solving the real dataset.  GPU part: the HIP library
against the oracle on the same graph -- LM iteration count, error after every iteration, final states, metrics.
"""
import os

import numpy as np
import pytest

from gpslam_amd import plaza, synthetic as S
from oracle import oracle as O

FIX = os.path.join(os.path.dirname(__file__), "golden", "plaza2.npz")


@pytest.fixture(scope="module")
def data():
    return plaza.load(FIX)


def test_fixture_shapes(data):
    assert data["GT"].shape == (4091, 4) and data["DR"].shape == (4090, 3)
    assert data["TD"].shape == (1816, 4) and data["TL"].shape == (4, 3)
    assert np.all(np.diff(data["GT"][:, 0]) > 0) and np.all(np.diff(data["TD"][:, 0]) >= 0)
    assert sorted(data["TL"][:, 0].astype(int)) == [0, 1, 5, 6]


def test_range_fit(data):
    trans, outlier = plaza.range_measure_fit(data["GT"], data["TL"], data["TD"])
    # the fit is a pure least-squares problem on the log: independent re-derivation with the normal equations
    T, t = data["GT"][:, 0], data["TD"][:, 0]
    hi = np.searchsorted(T, t)
    idx = np.where(np.abs(T[hi - 1] - t) <= np.abs(T[hi] - t), hi - 1, hi)
    lm = {int(i): k for k, i in enumerate(data["TL"][:, 0])}
    true = np.array([np.hypot(*(data["GT"][i, 1:3] - data["TL"][lm[int(l)], 1:3])) for i, l in zip(idx, data["TD"][:, 2])])
    m = data["TD"][:, 3][~outlier]
    A = np.stack([m, np.ones_like(m)], 1)
    ref = np.linalg.solve(A.T @ A, A.T @ true[~outlier])
    assert np.allclose(trans, ref, rtol=1e-9, atol=1e-12)
    assert 0.9 < trans[0] < 1.0 and abs(trans[1]) < 0.5
    assert np.all(np.abs(trans[0] * data["TD"][:, 3] + trans[1] - true)[~outlier] <= 2.0)


def test_graph_recipe(data):
    p = plaza.build_problem(data)
    N = 4091
    assert p["kind"] == S.POSE2 and len(p["gp_left"]) == N - 1 and len(p["between_left"]) == N - 1
    assert np.allclose(p["gp_dt"], np.diff(data["GT"][:, 0]))
    assert np.all(p["range_tau"] >= 0) and np.all(p["range_tau"] <= p["range_dt"] + 1e-12)
    # every measurement lies inside the interval it is attached to
    t = data["GT"][p["range_left"], 0] + p["range_tau"]
    assert np.all(t <= data["GT"][p["range_left"] + 1, 0] + 1e-9)
    # dead reckoning = composition of the odometry increments from the first ground-truth pose
    assert np.allclose(p["pose"][0], [data["GT"][0, 1], data["GT"][0, 2], data["GT"][0, 3] + data["init_heading_offset"]])
    k = 1234
    c, s = np.cos(p["pose"][k, 2]), np.sin(p["pose"][k, 2])
    step = data["DR"][k, 1]
    assert np.allclose(p["pose"][k + 1], [p["pose"][k, 0] + c * step, p["pose"][k, 1] + s * step, p["pose"][k, 2] + data["DR"][k, 2]])


def test_oracle_solves_plaza2(data):
    p = plaza.build_problem(data)
    before = plaza.metrics(p, p["pose"], p["landmarks"])
    ch = plaza.apply(p, O.Chain(S.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    errs = plaza.optimize(ch)
    pose, _ = ch.get_states()
    after = plaza.metrics(p, pose, ch.get_landmarks())
    assert before["position_m"] > 20.0                       # dead reckoning drifts by tens of metres
    assert after["position_m"] < 0.25 and after["rotation_deg"] < 1.5 and after["landmark_m"] < 0.1
    assert 4 <= len(errs) - 1 <= 12 and all(b <= a for a, b in zip(errs, errs[1:]))


def _pair(data, linear):
    import gpslam_amd
    p = plaza.build_problem(data, linear=linear)
    kind = S.LINEAR3 if linear else S.POSE2
    orc = plaza.apply(p, O.Chain(kind, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    dev = plaza.apply(p, gpslam_amd.ChainSolver(kind, chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2))
    return p, orc, dev


@pytest.mark.gpu
def test_gpu_plaza2_pose2_matches_oracle(data):
    p, orc, dev = _pair(data, linear=False)
    e0 = plaza.optimize(orc)
    e1 = plaza.optimize(dev)
    assert len(e0) == len(e1)
    assert np.allclose(e0, e1, rtol=1e-7)
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    assert np.abs(x0 - x1).max() <= 1e-7 and np.abs(v0 - v1).max() <= 1e-7
    assert np.abs(orc.get_landmarks() - dev.get_landmarks()).max() <= 1e-7
    m0, m1 = plaza.metrics(p, x0, orc.get_landmarks()), plaza.metrics(p, x1, dev.get_landmarks())
    for k in m0:
        assert abs(m0[k] - m1[k]) <= 1e-7
    assert m1["position_m"] < 0.25


@pytest.mark.gpu
def test_gpu_plaza2_linear_first_iterations_match_oracle(data):
    """useLinearPose2 = true (PlazaPose2.m:28): OdometryFactor2DLinear + GaussianProcessPriorLinear<3> +
    GPInterpolatedRangeFactor2DLinear.  LM crawls here (79 iterations); the first ones are compared in lock step."""
    p, orc, dev = _pair(data, linear=True)
    import lm_lockstep
    assert abs(orc.error() - dev.error()) <= 1e-9 * orc.error()
    lm_lockstep.run(orc, dev, 1e-5, 6, err_tol=1e-7)
    assert np.abs(orc.get_states()[0] - dev.get_states()[0]).max() <= 1e-6
