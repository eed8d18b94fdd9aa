"""GPInterpolatedProjectionFactorPose3<Cal3_S2> on the HIP path vs the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import gpu, random_chain, states_close
from test_oracle_golden import build_projection_problem, check_projection_result

pytestmark = pytest.mark.gpu
K2 = [50.0, 50.0, 0.7, 40.0, 30.0]          # a skewed calibration exercises the s term
KDS2 = K2 + [0.08, -0.03, 0.004, -0.006]    # gtsam::Cal3DS2: + radial k1, k2 and tangential p1, p2 (round 4)


def build_pair(N=24, seed=4, sensor=True, behind=False, K2=K2):
    c = random_chain(O.POSE3, N, seed, motion=0.2, noise=0.02)
    rng = np.random.default_rng(seed + 9)
    Qc = np.diag(0.01 + 0.02 * rng.random(6))
    body_T_sensor = O.pose3((0.1, -0.2, 0.15), (0.3, 0.6, -0.7)) if sensor else None
    # landmarks a few metres in front of the cameras along the trajectory
    L = 5
    anchors = rng.integers(0, N, L)
    lm = np.zeros((L, 3))
    for k, i in enumerate(anchors):
        cam = c["truth_pose"][i]
        if sensor:
            out = np.zeros(12)
            O.call("orc_pose3_compose", cam, body_T_sensor, out, None, None)
            cam = out
        R, t = cam[:9].reshape(3, 3), cam[9:]
        lm[k] = t + R @ np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(6, 12)])
    left, lmi, meas, dt, tau = [], [], [], [], []
    for i in range(N - 1):
        for k in range(L):
            cam = c["truth_pose"][i]
            if sensor:
                out = np.zeros(12)
                O.call("orc_pose3_compose", cam, body_T_sensor, out, None, None)
                cam = out
            try:
                uv = O.pinhole_project(cam, K2, lm[k])
            except ValueError:
                continue
            if abs(uv[0]) > 400 or abs(uv[1]) > 400 or rng.random() < 0.5:
                continue
            left.append(i); lmi.append(k); meas.append(uv + 0.05 * rng.standard_normal(2))
            dt.append(c["dt"][i]); tau.append(c["dt"][i] * rng.uniform(0.1, 0.9))
    lm_init = lm + 0.1 * rng.standard_normal(lm.shape)
    if behind:
        lm_init[0] = c["pose"][left[0], 9:12] - 5.0 * c["pose"][left[0], :9].reshape(3, 3)[:, 2]   # behind the first camera
    solvers = []
    for make in (lambda: O.Chain(O.POSE3, landmark_dim=3), lambda: gpu().ChainSolver(O.POSE3, landmark_dim=3)):
        s = make()
        s.set_qc(Qc)
        s.set_states(c["pose"], c["vel"])
        s.set_landmarks(lm_init)
        s.add_gp_priors(np.arange(N - 1), c["dt"])
        fix = np.arange(0, N, 8)
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), 6), 0.01))
        s.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, 6), 0.05))
        s.add_landmark_priors(np.arange(L), lm, np.full((L, 3), 0.5))
        s.add_interp_projection(left, lmi, np.array(meas), np.full((len(left), 2), 0.1), dt, tau, K2, body_T_sensor)
        s.compile()
        solvers.append(s)
    return solvers[0], solvers[1], len(left)


@pytest.mark.parametrize("sensor", [False, True], ids=["no-sensor", "body_P_sensor"])
def test_projection_normal_equations_and_gauss_newton(sensor):
    orc, dev, n = build_pair(sensor=sensor)
    assert n > 20
    assert abs(orc.error() - dev.error()) <= 1e-10 * max(1.0, orc.error())
    D0, O0, g0, B0, _, _ = orc.normal_equations()
    D1, O1, g1, B1 = dev.normal_equations()
    for a, b in ((D0, D1), (O0, O1), (g0, g1), (B0, B1)):
        assert np.abs(a - b).max() <= 1e-8 * max(1.0, np.abs(a).max())
    for _ in range(5):
        rc0, st0 = orc.iterate_gn()
        rc1, st1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(st0.error_after - st1.error_after) <= 1e-6 * max(1.0, st0.error_after)
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    states_close(O.POSE3, x0, v0, x1, v1, 1e-8)
    assert np.abs(orc.get_landmarks() - dev.get_landmarks()).max() <= 1e-8


def test_projection_with_a_distorting_calibration():
    """GPInterpolatedProjectionFactorPose3<Cal3DS2>: the reference class is a template over CALIBRATION
    (GPInterpolatedProjectionFactorPose3.h:29); radial + tangential distortion on the HIP path vs the oracle, and zero
    coefficients through the Cal3DS2 entry point against the Cal3_S2 one, bit for bit."""
    orc, dev, n = build_pair(sensor=True, K2=KDS2)
    assert n > 20
    assert abs(orc.error() - dev.error()) <= 1e-10 * max(1.0, orc.error())
    D0, O0, g0, B0, _, _ = orc.normal_equations()
    D1, O1, g1, B1 = dev.normal_equations()
    for a, b in ((D0, D1), (O0, O1), (g0, g1), (B0, B1)):
        assert np.abs(a - b).max() <= 1e-8 * max(1.0, np.abs(a).max())
    for _ in range(5):
        rc0, st0 = orc.iterate_gn()
        rc1, st1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(st0.error_after - st1.error_after) <= 1e-6 * max(1.0, st0.error_after)
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    states_close(O.POSE3, x0, v0, x1, v1, 1e-8)
    # the distortion matters at image coordinates like these (otherwise the test would not notice a dropped term)
    cam = O.pose3((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    assert np.abs(O.pinhole_project(cam, K2, [-3.0, 2.5, 8.0]) - O.pinhole_project(cam, KDS2, [-3.0, 2.5, 8.0])).min() > 0.05
    _, plain, _ = build_pair(sensor=True, K2=K2)
    _, zero, _ = build_pair(sensor=True, K2=K2 + [0.0, 0.0, 0.0, 0.0])
    assert zero.error() == plain.error()
    assert np.array_equal(zero.normal_equations()[0], plain.normal_equations()[0])


def test_projection_cheirality_masked_like_the_reference():
    orc, dev, _ = build_pair(behind=True)
    e0, e1 = orc.error(), dev.error()
    assert e0 > 0.5 * (2 * 50.0 / 0.1) ** 2            # at least one masked factor: error 2 fx per component
    assert abs(e0 - e1) <= 1e-10 * e0
    g0, g1 = orc.normal_equations()[2], dev.normal_equations()[2]
    assert np.abs(g0 - g1).max() <= 1e-8 * max(1.0, np.abs(g0).max())


def test_projection_reference_optimisation(golden):
    """testGPInterpolatedProjectionFactorPose3.cpp:180-262 through the C ABI."""
    c = golden["projection_optimization"]
    dev = build_projection_problem(c, gpu().ChainSolver(O.POSE3, landmark_dim=3))
    rc, st = dev.optimize()
    assert rc == 0
    check_projection_result(c, dev)
    orc = build_projection_problem(c, O.Chain(O.POSE3, landmark_dim=3))
    rc0, st0 = orc.optimize()
    assert st0.iterations == st.iterations


def test_gps_reference_optimisation(golden):
    """testGPInterpolatedGPSFactorPose3.cpp:193-262 through the C ABI (two of the three fixes extrapolate)."""
    from test_oracle_golden import build_gps_problem, check_gps_result
    c = golden["gps_optimization"]
    dev = build_gps_problem(c, gpu().ChainSolver(O.POSE3))
    rc, st = dev.optimize()
    assert rc == 0
    check_gps_result(c, dev)
    orc = build_gps_problem(c, O.Chain(O.POSE3))
    rc0, st0 = orc.optimize()
    assert st0.iterations == st.iterations
