"""The world-frame (v, w) velocity parameterisation -- GaussianProcessPriorPose3VW, GaussianProcessInterpolatorPose3VW,
GPInterpolatedGPSFactorPose3VW -- HIP path against the oracle, through the C ABI.
The oracle side restates the reference formulas term by term (oracle/orc_gp.c) and is pinned by the reference's own VW
test vectors (tests/test_oracle_golden.py); the HIP side obtains the same Jacobians by the chain rule through
convertVWtoVb applied to the body-velocity rows, so the two are independent derivations."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import gpu, random_chain, states_close

pytestmark = pytest.mark.gpu


def world_velocities(c):
    """Turn the body-frame velocities of a random Pose3 chain into world-frame [v; w] of the same motion."""
    out = {}
    for key_p, key_v in (("truth_pose", "truth_vel"), ("pose", "vel")):
        R = c[key_p][:, :9].reshape(-1, 3, 3)
        wb, vb = c[key_v][:, :3], c[key_v][:, 3:]
        out[key_v] = np.concatenate([np.einsum("nij,nj->ni", R, vb), np.einsum("nij,nj->ni", R, wb)], axis=1)
    return out


def build_vw_pair(N, seed, gps=False):
    c = random_chain(O.POSE3, N, seed)
    c.update(world_velocities(c))
    rng = np.random.default_rng(seed + 5)
    Qc = np.diag(0.01 + 0.02 * rng.random(6))
    Qc[0, 1] = Qc[1, 0] = 0.003
    left = np.arange(0, N - 1, 3)
    tau = c["dt"][left] * (0.2 + 0.6 * rng.random(len(left)))
    sensor = O.pose3((1.4, 4.4, -0.5), (0.3, 0.6, -0.7))   # testGPInterpolatedGPSFactorPose3VW.cpp:45
    meas = c["truth_pose"][left, 9:12] + 0.05 * rng.standard_normal((len(left), 3))
    solvers = []
    for make in (lambda: O.Chain(O.POSE3, velocity_world=True), lambda: gpu().ChainSolver(O.POSE3, velocity_world=True)):
        s = make()
        s.set_qc(Qc)
        s.set_states(c["pose"], c["vel"])
        s.add_gp_priors(np.arange(N - 1), c["dt"])
        fix = np.arange(0, N, 20)
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), 6), 0.01))
        s.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, 6), 0.05))
        if gps:
            s.add_interp_gps(left, meas, np.full((len(left), 3), 0.1), c["dt"][left], tau, sensor)
        s.compile()
        solvers.append(s)
    return solvers[0], solvers[1], c, Qc


def test_vw_linearize_matches_oracle():
    orc, dev, _, _ = build_vw_pair(130, 3)
    e0, H0 = orc.linearize_gp()
    e1, H1 = dev.linearize_gp()
    assert np.abs(e0 - e1).max() <= 1e-10 * max(1.0, np.abs(e0).max())
    # the finite-difference (h = 1e-6) rows carry 1/h-amplified rounding, as in the body-velocity factor
    assert np.abs(H0 - H1).max() <= 1e-7 * max(1.0, np.abs(H0).max())


@pytest.mark.parametrize("gps", [False, True], ids=["prior-only", "with-gps"])
def test_vw_normal_equations_and_gauss_newton(gps):
    orc, dev, _, _ = build_vw_pair(90, 8, gps=gps)
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    D0, O0, g0 = orc.normal_equations()[:3]
    D1, O1, g1 = dev.normal_equations()[:3]
    for a, b in ((D0, D1), (O0, O1), (g0, g1)):
        assert np.abs(a - b).max() <= 1e-8 * max(1.0, np.abs(a).max())
    for _ in range(6):
        rc0, st0 = orc.iterate_gn()
        rc1, st1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(st0.error_after - st1.error_after) <= 1e-6 * max(1.0, st0.error_after)
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    states_close(O.POSE3, x0, v0, x1, v1, 1e-9)


def test_vw_reference_optimisation_case():
    """testGaussianProcessPriorPose3VW.cpp:142-188: two poses fixed by priors one metre apart, dt = 1, world velocity
    (1, 0, 0): already the optimum, Gauss-Newton must leave it there (the reference only checks that it runs)."""
    p1, p2 = O.pose3((0, 0, 0), (0, 0, 0)), O.pose3((0, 0, 0), (1, 0, 0))
    s = np.array([[1.0, 0, 0, 0, 0, 0], [1.0, 0, 0, 0, 0, 0]])
    outs = []
    for make in (lambda: O.Chain(O.POSE3, velocity_world=True), lambda: gpu().ChainSolver(O.POSE3, velocity_world=True)):
        c = make()
        c.set_qc(0.01 * np.eye(6))
        c.set_states(np.stack([p1, p2]), s)
        c.add_pose_priors([0, 1], np.stack([p1, p2]), np.full((2, 6), 0.001))
        c.add_gp_priors([0], [1.0])
        c.compile()
        rc, st = c.optimize()
        assert rc == 0
        outs.append(c.get_states())
    (x0, v0), (x1, v1) = outs
    assert np.abs(v1 - s).max() <= 1e-9 and np.abs(x1 - np.stack([p1, p2])).max() <= 1e-9
    states_close(O.POSE3, x0, v0, x1, v1, 1e-9)


def test_vw_interpolate_poses_query():
    orc, dev, c, Qc = build_vw_pair(40, 2)
    rng = np.random.default_rng(4)
    left = rng.integers(0, 39, 32).astype(np.int32)
    dt = np.asarray(c["dt"])[left]
    tau = dt * rng.random(32)
    got = dev.interpolate_poses(left, dt, tau)
    pose, vel = dev.get_states()
    for q in range(32):
        Lam, Psi = O.lambda_psi(6, Qc, dt[q], tau[q])
        i = left[q]
        want, _ = O.interpolate_vw(Lam, Psi, pose[i], vel[i, :3], vel[i, 3:], pose[i + 1], vel[i + 1, :3], vel[i + 1, 3:], jac=False)
        assert np.abs(got[q] - want).max() <= 1e-11 * max(1.0, np.abs(want).max())


def test_vw_only_for_pose3():
    with pytest.raises(Exception):
        gpu().ChainSolver(O.POSE2, velocity_world=True)
