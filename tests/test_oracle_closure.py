"""The oracle's side of loop closures (gtsam::BetweenFactor<Pose> between non-adjacent states): orc_chain_add_between_pairs and the
envelope Cholesky that solves chains holding them (oracle/orc_chain.c: skyline_solve).  Nothing here has a reference vector -- the
reference's own tests never build a closure (its factors take arbitrary keys: gpslam/gp/GaussianProcessPriorPose3.h:43-47) -- so the
solver is pinned three ways that do not go through its own assembly:
  * on graphs WITHOUT closures it must reproduce the block-tridiagonal + border elimination (the solver every other test pins);
  * on a LINEAR chain one Gauss-Newton step must land on the least-squares solution of the stacked whitened rows, built here with
    numpy from the factors' own definitions;
  * on SE(2) / SE(3) the point Gauss-Newton converges to must be a stationary point of NonlinearFactorGraph::error, differentiated
    numerically through orc_chain_error alone (no Jacobian, no normal equation, no solver involved).
CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpslam_amd import synthetic  # noqa: E402
from oracle import oracle  # noqa: E402


def _chain(p, **kw):
    c = oracle.Chain(p["kind"], landmark_dim=2 if "landmarks" in p else 0, **kw)
    return synthetic.apply(p, c)


@pytest.mark.parametrize("name", ["pose3", "pose2_landmarks", "linear3"])
def test_envelope_cholesky_reproduces_the_chain_solver_without_closures(name):
    p = {"pose3": lambda: synthetic.pose3_chain(60, seed=1), "pose2_landmarks": lambda: synthetic.pose2_range_chain(120, L=4, seed=2),
         "linear3": lambda: synthetic.linear_chain(90, seed=3)}[name]()
    a, b = _chain(p), _chain(p)
    try:
        for it in range(2):
            oracle.force_envelope_solver(False)
            rca, sa = a.iterate_gn()
            oracle.force_envelope_solver(True)
            rcb, sb = b.iterate_gn()
            assert rca == 0 and rcb == 0
            pa, va = a.get_states()
            pb, vb = b.get_states()
            assert np.max(np.abs(pa - pb)) < 1e-9 * max(1.0, np.max(np.abs(pa))), (it, np.max(np.abs(pa - pb)))
            assert np.max(np.abs(va - vb)) < 1e-9 * max(1.0, np.max(np.abs(va)))
            assert abs(sa.error_after - sb.error_after) <= 1e-9 * abs(sa.error_after)
            if "landmarks" in p:
                assert np.max(np.abs(a.get_landmarks() - b.get_landmarks())) < 1e-9
        # Levenberg-Marquardt takes the same solver switch
        oracle.force_envelope_solver(False)
        _, la, lam_a = a.iterate_lm(1e-3)
        oracle.force_envelope_solver(True)
        _, lb, lam_b = b.iterate_lm(1e-3)
        assert lam_a == lam_b and la.trials == lb.trials and la.accepted == lb.accepted
    finally:
        oracle.force_envelope_solver(False)


def test_linear_chain_with_closures_one_step_is_the_least_squares_solution():
    N, D = 40, 3
    p = synthetic.add_loop_closures(synthetic.linear_chain(N, seed=5), [[2, 31], [37, 9], [0, 20]], seed=1)
    c = _chain(p)
    # stacked whitened rows A z = r over z = [p_0, v_0, p_1, v_1, ...] (the GP prior's rows from the oracle's own linearisation,
    # which the reference's tests pin; everything else written out here)
    b = 2 * D
    rows, rhs = [], []
    e, H = c.linearize_gp()                      # unwhitened e (F x 2D), H (F x 4 x 2D x D)
    Qc, dt = p["qc"], p["gp_dt"][0]
    Q = np.block([[dt ** 3 / 3 * Qc, dt ** 2 / 2 * Qc], [dt ** 2 / 2 * Qc, dt * Qc]])
    W = np.linalg.cholesky(np.linalg.inv(Q)).T   # R with R^T R = Q^-1
    for f, l in enumerate(p["gp_left"]):
        J = np.zeros((b, N * b))
        J[:, l * b:l * b + D], J[:, l * b + D:l * b + b] = H[f, 0], H[f, 1]
        J[:, (l + 1) * b:(l + 1) * b + D], J[:, (l + 1) * b + D:(l + 2) * b] = H[f, 2], H[f, 3]
        rows.append(W @ J)
        rhs.append(-W @ e[f])
    pose, vel = c.get_states()

    def unary(idx, val, sig, off, cur):
        for k, i in enumerate(idx):
            J = np.zeros((D, N * b))
            J[:, i * b + off:i * b + off + D] = np.diag(1.0 / sig[k])
            rows.append(J)
            rhs.append(-(cur[i] - val[k]) / sig[k])
    unary(p["prior_idx"], p["prior_pose"], p["prior_sig"], 0, pose)
    unary(p["vprior_idx"], p["vprior"], p["vprior_sig"], D, vel)
    for k in range(len(p["closure_first"])):
        i, j = p["closure_first"][k], p["closure_second"][k]
        J = np.zeros((D, N * b))
        J[:, i * b:i * b + D] = -np.diag(1.0 / p["closure_sig"][k])
        J[:, j * b:j * b + D] = np.diag(1.0 / p["closure_sig"][k])
        rows.append(J)
        rhs.append(-((pose[j] - pose[i]) - p["closure_meas"][k]) / p["closure_sig"][k])
    Aall, rall = np.vstack(rows), np.concatenate(rhs)
    z = np.linalg.lstsq(Aall, rall, rcond=None)[0].reshape(N, b)
    err0 = c.error()
    assert abs(err0 - 0.5 * rall @ rall) < 1e-9 * err0
    rc, st = c.iterate_gn()
    assert rc == 0
    p1, v1 = c.get_states()
    assert np.max(np.abs(p1 - (pose + z[:, :D]))) < 1e-8
    assert np.max(np.abs(v1 - (vel + z[:, D:]))) < 1e-8
    res = Aall @ z.ravel() - rall
    assert abs(st.error_after - 0.5 * res @ res) < 1e-9 * max(st.error_after, 1e-12)
    rc, st2 = c.iterate_gn()                      # linear: the second step is zero
    assert st2.delta_inf_norm < 1e-9


@pytest.mark.parametrize("kind", ["pose2", "pose3"])
def test_gauss_newton_with_closures_converges_to_a_stationary_point_of_the_error(kind):
    if kind == "pose2":
        base = synthetic.pose2_range_chain(24, L=3, seed=4)
        pairs = [[1, 19], [22, 6]]
    else:
        base = synthetic.pose3_chain(16, seed=6)
        pairs = [[0, 12], [14, 3]]
    p = synthetic.add_loop_closures(base, pairs, seed=2)
    c = _chain(p)
    e_prev = c.error()
    for it in range(40):     # (the range factors make Gauss-Newton converge linearly, a factor ~0.36 per step)
        rc, st = c.iterate_gn()
        assert rc == 0
        assert st.error_after <= e_prev * (1 + 1e-9)
        e_prev = st.error_after
        if st.delta_inf_norm < 1e-10:
            break
    assert st.delta_inf_norm < 1e-8
    # numerical gradient of the total error at the fixed point, through retractions of single coordinates
    pose, vel = c.get_states()
    knd = p["kind"]
    d = vel.shape[1]
    h = 1e-5
    scale = c.error()
    worst = 0.0
    for i in range(len(pose)):
        for k in range(2 * d):
            vals = []
            for sgn in (1.0, -1.0):
                pp, vv = pose.copy(), vel.copy()
                if k < d:
                    dl = np.zeros(d)
                    dl[k] = sgn * h
                    pp[i] = oracle.retract(knd, pose[i], dl)
                else:
                    vv[i, k - d] += sgn * h
                c.set_states(pp, vv)
                vals.append(c.error())
            worst = max(worst, abs(vals[0] - vals[1]) / (2 * h))
    c.set_states(pose, vel)
    # (a unit step along any coordinate changes the cost by ~1e4..1e6 away from the optimum: sigmas of 1e-3)
    assert worst < 1e-4 * max(1.0, scale), worst
