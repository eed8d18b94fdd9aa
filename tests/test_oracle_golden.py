"""Pins the CPU oracle against the reference's own unit-test vectors (tests/golden/reference_tests.json,
transcribed from gpslam/gp/tests/*.cpp and gpslam/slam/tests/*.cpp).  CPU only.

Same four patterns as the reference: known-answer errors, analytic Jacobian ==
numericalDerivative11 of the same error function, 2-state Gauss-Newton fixed points, Lie-utility checks.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from helpers import KIND, dec_pose, numdiff_manifold, numdiff_vector, pose_close


def _is_vec(kind):
    return kind in (O.LINEAR2, O.LINEAR3)


def _nd_pose(kind, f, x, h):
    return numdiff_vector(f, x, h) if _is_vec(kind) else numdiff_manifold(kind, f, x, h)


JAC_STEPS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jac_steps.json")
_JAC_LOG = {}       # case -> difference step at which the analytic Jacobian met the reference's tolerance in this run


def _jac_exceptions():
    with open(JAC_STEPS_FILE) as f:
        return json.load(f)["needs_larger_step"]


def _jac_ok(H, make_num, h_ref, tol, case):
    """analytic H == central difference, at the reference's step h_ref and tolerance tol.

    A few reference cases sit where the central difference itself is rounding-limited in this
    arithmetic (Pose2::Expmap's (v - R v)/w with |w| ~ 1e-8 when h = 1e-6; GTSAM's own rounding
    there differs and cannot be reproduced without GTSAM; a 1e-6 step on a velocity scaled by Lambda_12 = 0.0147
    lands at theta^2 = 2.16e-16, just below Pose3::Expmap's first-order branch at theta^2 <= 2.22e-16, where the
    function drops the omega x v / 2 term that ExpmapDerivative keeps).  The thing being pinned is the analytic
    Jacobian, so for THOSE cases -- listed by name, with the step they need, in tests/golden/jac_steps.json -- the same
    tolerance is retried at larger steps, up to the listed one.  Every other case must pass at the reference's own step:
    a case that passes there today can not quietly fall back to a larger step tomorrow, and
    test_jacobian_step_table_is_current fails when the table and the run disagree.
    """
    allowed = _jac_exceptions().get(case)
    if os.environ.get("GPSLAM_JAC_DISCOVER"):      # tests/golden/make_jac_steps.py regenerates the table with the full ladder
        allowed = 1e-3
    steps = [h_ref] + ([h for h in (1e-5, 1e-4, 1e-3) if h > h_ref and h <= allowed * (1 + 1e-12)] if allowed else [])
    worst = None
    for h in steps:
        err = float(np.abs(H - make_num(h)).max())
        worst = err if worst is None else min(worst, err)
        if err <= tol:
            _JAC_LOG[case] = (h, h_ref)
            return True, err
    return False, worst


def test_gp_prior_cases(golden):
    for c in golden["gp_prior"]:
        kind = KIND[c["kind"]]
        p1, p2 = dec_pose(kind, c["p1"]), dec_pose(kind, c["p2"])
        v1, v2 = O.A(c["v1"]), O.A(c["v2"])
        e, H = O.gp_prior(kind, p1, v1, p2, v2, c["dt"])
        if c["expect"] is not None:
            assert np.abs(e - np.array(c["expect"])).max() <= c["tol_e"], c["src"]
        f = lambda a, b, cc, dd: O.gp_prior(kind, a, b, cc, dd, c["dt"], jac=False)[0]
        num = [lambda h: _nd_pose(kind, lambda x: f(x, v1, p2, v2), p1, h),
               lambda h: numdiff_vector(lambda x: f(p1, x, p2, v2), v1, h),
               lambda h: _nd_pose(kind, lambda x: f(p1, v1, x, v2), p2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, p2, x), v2, h)]
        for k in range(4):
            ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
            assert ok, (c["src"], k, err)


def test_interpolator_cases(golden):
    for c in golden["interpolator"]:
        kind = KIND[c["kind"]]
        d = O.TANGENT_DIM[kind]
        Lam, Psi = O.lambda_psi(d, c["qc"] * np.eye(d), c["dt"], c["tau"])
        p1, p2 = dec_pose(kind, c["p1"]), dec_pose(kind, c["p2"])
        v1, v2 = O.A(c["v1"]), O.A(c["v2"])
        out, H = O.interpolate(kind, Lam, Psi, p1, v1, p2, v2)
        if c["expect"] is not None:
            exp = dec_pose(kind, c["expect"])
            assert pose_close(kind, exp, out, c["tol_e"]), c["src"]

        # numericalDerivative11 of a manifold-valued function: local coordinates of the output around f(x)
        def g(a, b, cc, dd):
            y = O.interpolate(kind, Lam, Psi, a, b, cc, dd, jac=False)[0]
            return O.local(kind, out, y)
        num = [lambda h: _nd_pose(kind, lambda x: g(x, v1, p2, v2), p1, h),
               lambda h: numdiff_vector(lambda x: g(p1, x, p2, v2), v1, h),
               lambda h: _nd_pose(kind, lambda x: g(p1, v1, x, v2), p2, h),
               lambda h: numdiff_vector(lambda x: g(p1, v1, p2, x), v2, h)]
        for k in range(4):
            ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
            assert ok, (c["src"], k, err)


def _interp_range(kind, Lam, Psi, meas, sensor, p1, v1, p2, v2, land, jac):
    d = O.TANGENT_DIM[kind]
    ld = len(land)
    H = [np.zeros((1, d)) for _ in range(4)] + [np.zeros((1, ld))] if jac else [None] * 5
    if kind == O.LINEAR3:
        e = O.call("orc_interp_range_2dlinear", O.A(Lam), O.A(Psi), float(meas), O.A(p1), O.A(v1), O.A(p2), O.A(v2),
                   O.A(land), *H)
    else:
        name = "orc_interp_range_pose2" if kind == O.POSE2 else "orc_interp_range_pose3"
        e = O.call(name, O.A(Lam), O.A(Psi), float(meas), None if sensor is None else O.A(sensor), O.A(p1), O.A(v1),
                   O.A(p2), O.A(v2), O.A(land), *H)
    return e, H


def test_interp_range_cases(golden):
    for c in golden["interp_range"]:
        kind = KIND[c["kind"]]
        d = O.TANGENT_DIM[kind]
        Lam, Psi = O.lambda_psi(d, c["qc"] * np.eye(d), c["dt"], c["tau"])
        p1, p2 = dec_pose(kind, c["p1"]), dec_pose(kind, c["p2"])
        v1, v2, land = O.A(c["v1"]), O.A(c["v2"]), O.A(c["land"])
        sensor = None if c["sensor"] is None else dec_pose(kind, c["sensor"])
        meas = c["meas"]
        if isinstance(meas, dict):  # meas = (true_pose * body_T_sensor).range(land)
            tp = dec_pose(kind, meas["true_pose"])
            sp = np.zeros(12)
            O.call("orc_pose3_compose", tp, sensor, sp, None, None)
            O.lib().orc_pose3_range.restype = __import__("ctypes").c_double
            meas = O.call("orc_pose3_range", sp, land, None, None)
        e, H = _interp_range(kind, Lam, Psi, meas, sensor, p1, v1, p2, v2, land, True)
        if c["expect"] is not None:
            assert abs(e - c["expect"]) <= c["tol_e"], c["src"]
        f = lambda a, b, cc, dd, l: _interp_range(kind, Lam, Psi, meas, sensor, a, b, cc, dd, l, False)[0]
        num = [lambda h: _nd_pose(kind, lambda x: f(x, v1, p2, v2, land), p1, h),
               lambda h: numdiff_vector(lambda x: f(p1, x, p2, v2, land), v1, h),
               lambda h: _nd_pose(kind, lambda x: f(p1, v1, x, v2, land), p2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, p2, x, land), v2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, p2, v2, x), land, h)]
        for k in range(5):
            ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
            assert ok, (c["src"], k, err)


def test_2dlinear_factor_cases(golden):
    for c in golden["range2d"]:
        H1, H2 = np.zeros((1, 3)), np.zeros((1, 2))
        pose, land = O.A(c["pose"]), O.A(c["land"])
        e = O.call("orc_range_2dlinear", float(c["meas"]), pose, land, H1, H2) if c["check_H"] else \
            O.call("orc_range_2dlinear", float(c["meas"]), pose, land, None, None)
        assert abs(e - c["expect"]) <= 1e-6, c["src"]
        if c["check_H"]:
            f = lambda p, l: O.call("orc_range_2dlinear", float(c["meas"]), O.A(p), O.A(l), None, None)
            assert np.abs(H1 - numdiff_vector(lambda x: f(x, land), pose, 1e-6)).max() <= 1e-6
            assert np.abs(H2 - numdiff_vector(lambda x: f(pose, x), land, 1e-6)).max() <= 1e-6
    for c in golden["bearing_range2d"]:
        pose, land = O.A(c["pose"]), O.A(c["land"])
        e, H1, H2 = np.zeros(2), np.zeros((2, 3)), np.zeros((2, 2))
        O.call("orc_range_bearing_2dlinear", float(c["bearing"]), float(c["range"]), pose, land, e, H1, H2)
        assert np.abs(e - np.array(c["expect"])).max() <= 1e-6, c["src"]

        def f(p, l):
            out = np.zeros(2)
            O.call("orc_range_bearing_2dlinear", float(c["bearing"]), float(c["range"]), O.A(p), O.A(l), out, None, None)
            return out
        assert np.abs(H1 - numdiff_vector(lambda x: f(x, land), pose, 1e-6)).max() <= 1e-6, c["src"]
        assert np.abs(H2 - numdiff_vector(lambda x: f(pose, x), land, 1e-6)).max() <= 1e-6, c["src"]
    for c in golden["odometry2d"]:
        p1, p2, m = O.A(c["pose1"]), O.A(c["pose2"]), O.A(c["meas"])
        e, H1, H2 = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 3))
        O.call("orc_odometry_2dlinear", m, p1, p2, e, H1, H2)
        assert np.abs(e - np.array(c["expect"])).max() <= 1e-6, c["src"]
        if c["check_H"]:
            def f(a, b):
                out = np.zeros(3)
                O.call("orc_odometry_2dlinear", m, O.A(a), O.A(b), out, None, None)
                return out
            assert np.abs(H1 - numdiff_vector(lambda x: f(x, p2), p1, 1e-6)).max() <= 1e-6, c["src"]
            assert np.abs(H2 - numdiff_vector(lambda x: f(p1, x), p2, 1e-6)).max() <= 1e-6, c["src"]


def test_body_centric_velocity(golden):
    for c in golden["body_centric_velocity"]:
        p1, p2 = dec_pose("pose3", c["p1"]), dec_pose("pose3", c["p2"])
        vb, vs = np.zeros(6), np.zeros(6)
        O.call("orc_getBodyCentricVb", p1, p2, 0.1, vb)
        O.call("orc_getBodyCentricVs", p1, p2, 0.1, vs)
        assert np.abs(vb - np.array(c["vb"])).max() <= 1e-6, c["src"]
        assert np.abs(vs - np.array(c["vs"])).max() <= 1e-6, c["src"]


def _log(group, x):
    if group == "rot3":
        w = np.zeros(3)
        O.call("orc_rot3_logmap", x, w, None)
        return w
    xi = np.zeros(6)
    O.call("orc_pose3_logmap", x, xi, None)
    return xi


def _exp(group, v):
    if group == "rot3":
        R = np.zeros(9)
        O.call("orc_rot3_expmap", O.A(v), R, None)
        return R
    T = np.zeros(12)
    O.call("orc_pose3_expmap", O.A(v), T, None)
    return T


def test_lie_right_jacobians(golden):
    """numericalLieRightJacobian, gpslam/gp/tests/testPose3Utils.cpp:44-56"""
    for c in golden["lie_jacobians"]:
        g = c["group"]
        kind = O.ROT3 if g == "rot3" else O.POSE3
        x = dec_pose(kind, c["x"])
        om = _log(g, x)
        dim = len(om)
        dt = 1e-6
        J_expect = np.zeros((dim, dim))
        for i in range(dim):
            dl = np.zeros(dim)
            dl[i] = dt
            r = _exp(g, om + dl)
            J_expect[:, i] = O.local(kind, x, r) / dt     # Logmap(lie^-1 * r) / dt
        J = np.zeros((dim, dim))
        Jinv = np.zeros((dim, dim))
        if g == "rot3":
            O.call("orc_rightJacobianRot3", om, J)
            O.call("orc_rightJacobianRot3inv", om, Jinv)
        else:
            O.call("orc_rightJacobianPose3", om, J)
            O.call("orc_rightJacobianPose3inv", om, Jinv)
        assert np.abs(J - J_expect).max() <= c["tol"], c["src"]
        assert np.abs(Jinv - np.linalg.inv(J_expect)).max() <= c["tol_inv"], c["src"]


def test_se3_velocity_identity(golden):
    """Anderson15iros eq. (8): Vb = Jr(log T) dlog / dt, gpslam/gp/tests/testPose3Utils.cpp:289-328"""
    for c in golden["se3_velocity"]:
        base = dec_pose("pose3", c["base"])
        dlog = np.array(c["dlog"])
        xi = _log("pose3", base)
        add = _exp("pose3", xi + dlog)
        vb = np.zeros(6)
        O.call("orc_getBodyCentricVb", base, add, 0.01, vb)
        J = np.zeros((6, 6))
        O.call("orc_rightJacobianPose3", xi, J)
        assert np.abs(vb - J @ dlog / 0.01).max() <= c["tol"], c["src"]


def build_opt_problem(c, make_chain):
    """Build one of the reference's 2-state optimisation problems on any chain implementation."""
    kind = KIND[c["kind"]]
    d = O.TANGENT_DIM[kind]
    ld = c["landmark_dim"]
    ch = make_chain(kind, O.CHART_EXPMAP, ld)
    ch.set_qc(c["qc"] * np.eye(d))
    pose = np.stack([dec_pose(kind, p) for p in c["init"]["pose"]])
    vel = np.array(c["init"]["vel"], dtype=np.float64)
    ch.set_states(pose, vel)
    if ld:
        ch.set_landmarks(np.array(c["init"]["land"], dtype=np.float64))
    ch.add_gp_priors([0], [c["dt"]])
    for pp in c.get("pose_priors", []):
        ch.add_pose_priors([pp["idx"]], dec_pose(kind, pp["prior"])[None], np.full((1, d), pp["sigma"]))
    for vp in c.get("vel_priors", []):
        ch.add_vel_priors([vp["idx"]], np.array([vp["prior"]], dtype=np.float64), np.full((1, d), vp["sigma"]))
    for lp in c.get("land_priors", []):
        ch.add_landmark_priors([lp["idx"]], np.array([lp["prior"]], dtype=np.float64), np.full((1, ld), lp["sigma"]))
    land_true = np.array(c["expect"]["land"][0]) if ld else None
    for r in c.get("ranges", []):
        cam = dec_pose(kind, r["cam"])
        if kind == O.POSE3:
            z = O.call("orc_pose3_range", cam, O.A(land_true), None, None)
        else:
            z = float(np.hypot(land_true[0] - cam[0], land_true[1] - cam[1]))
        ch.add_interp_range([0], [0], [z], [c["range_sigma"]], [c["dt"]], [r["tau"]])
    ch.compile()
    return ch, kind


def check_opt_result(c, ch, kind):
    pose, vel = ch.get_states()
    for i, p in enumerate(c["expect"]["pose"]):
        assert pose_close(kind, dec_pose(kind, p), pose[i], c["tol"]), (c["src"], "pose", i)
    assert np.abs(vel - np.array(c["expect"]["vel"])).max() <= c["tol"], (c["src"], "vel")
    if c["landmark_dim"]:
        assert np.abs(ch.get_landmarks() - np.array(c["expect"]["land"])).max() <= c["tol"], (c["src"], "land")
    assert abs(ch.error()) <= c["tol_err"], (c["src"], "error", ch.error())


def test_two_state_optimisation_fixed_points(golden):
    """GaussNewtonOptimizer(graph, init).optimize() with default params recovers the ground truth."""
    for c in golden["optimization"]:
        ch, kind = build_opt_problem(c, O.Chain)
        rc, st = ch.optimize()
        assert rc == 0 and st.status == 0, c["src"]
        assert st.iterations < 100, c["src"]
        check_opt_result(c, ch, kind)


def test_gp_prior_vw_cases(golden):
    """GaussianProcessPriorPose3VW (testGaussianProcessPriorPose3VW.cpp): zero-error configurations and all six
    analytic Jacobians against numericalDerivative11 of the same error function."""
    for c in golden["gp_prior_vw"]:
        p1, p2 = dec_pose(O.POSE3, c["p1"]), dec_pose(O.POSE3, c["p2"])
        v1, w1, v2, w2 = O.A(c["v1"]), O.A(c["w1"]), O.A(c["v2"]), O.A(c["w2"])
        e, H = O.gp_prior_vw(p1, v1, w1, p2, v2, w2, c["dt"])
        if c["expect"] is not None:
            assert np.abs(e - np.array(c["expect"])).max() <= c["tol_e"], c["src"]
        f = lambda a, b, cc, dd, ee, ff: O.gp_prior_vw(a, b, cc, dd, ee, ff, c["dt"], jac=False)[0]
        num = [lambda h: numdiff_manifold(O.POSE3, lambda x: f(x, v1, w1, p2, v2, w2), p1, h),
               lambda h: numdiff_vector(lambda x: f(p1, x, w1, p2, v2, w2), v1, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, x, p2, v2, w2), w1, h),
               lambda h: numdiff_manifold(O.POSE3, lambda x: f(p1, v1, w1, x, v2, w2), p2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, w1, p2, x, w2), v2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, w1, p2, v2, x), w2, h)]
        for k in range(6):
            ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
            assert ok, (c["src"], k, err)


def test_interpolator_vw_cases(golden):
    """GaussianProcessInterpolatorPose3VW (testGaussianProcessInterpolatorPose3VW.cpp)."""
    for c in golden["interpolator_vw"]:
        Lam, Psi = O.lambda_psi(6, c["qc"] * np.eye(6), c["dt"], c["tau"])
        p1, p2 = dec_pose(O.POSE3, c["p1"]), dec_pose(O.POSE3, c["p2"])
        v1, w1, v2, w2 = O.A(c["v1"]), O.A(c["w1"]), O.A(c["v2"]), O.A(c["w2"])
        out, H = O.interpolate_vw(Lam, Psi, p1, v1, w1, p2, v2, w2)
        if c["expect"] is not None:
            assert pose_close(O.POSE3, dec_pose(O.POSE3, c["expect"]), out, c["tol_e"]), c["src"]

        def g(a, b, cc, dd, ee, ff):
            return O.local(O.POSE3, out, O.interpolate_vw(Lam, Psi, a, b, cc, dd, ee, ff, jac=False)[0])
        num = [lambda h: numdiff_manifold(O.POSE3, lambda x: g(x, v1, w1, p2, v2, w2), p1, h),
               lambda h: numdiff_vector(lambda x: g(p1, x, w1, p2, v2, w2), v1, h),
               lambda h: numdiff_vector(lambda x: g(p1, v1, x, p2, v2, w2), w1, h),
               lambda h: numdiff_manifold(O.POSE3, lambda x: g(p1, v1, w1, x, v2, w2), p2, h),
               lambda h: numdiff_vector(lambda x: g(p1, v1, w1, p2, x, w2), v2, h),
               lambda h: numdiff_vector(lambda x: g(p1, v1, w1, p2, v2, x), w2, h)]
        for k in range(6):
            ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
            assert ok, (c["src"], k, err)


def test_vw_equals_body_velocity_factor_under_change_of_variables():
    """convertVWtoVb (Pose3utils.cpp:47-64) ties the two families together: the VW prior at (T, v, w) is the
    body-velocity prior at (T, [R^T w; R^T v])."""
    rng = np.random.default_rng(3)
    p1 = O.pose3(rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3))
    p2 = O.retract(O.POSE3, p1, 0.2 * rng.standard_normal(6))
    s1, s2 = rng.standard_normal(6), rng.standard_normal(6)
    vb1, vb2 = np.zeros(6), np.zeros(6)
    O.call("orc_convertVWtoVb", O.A(s1[:3]), O.A(s1[3:]), p1, vb1, None, None, None)
    O.call("orc_convertVWtoVb", O.A(s2[:3]), O.A(s2[3:]), p2, vb2, None, None, None)
    e_vw, _ = O.gp_prior_vw(p1, s1[:3], s1[3:], p2, s2[:3], s2[3:], 0.1, jac=False)
    e_b, _ = O.gp_prior(O.POSE3, p1, vb1, p2, vb2, 0.1, jac=False)
    assert np.abs(e_vw - e_b).max() <= 1e-14
    R = p1[:9].reshape(3, 3)
    assert np.allclose(vb1, np.concatenate([R.T @ s1[3:], R.T @ s1[:3]]), atol=1e-15)


def _proj_meas(c):
    if isinstance(c["meas"], dict):       # cam.project(land) with cam at true_pose * body_T_sensor
        cam = dec_pose(O.POSE3, c["meas"]["true_pose"])
        if c["sensor"] is not None:
            out = np.zeros(12)
            O.call("orc_pose3_compose", cam, dec_pose(O.POSE3, c["sensor"]), out, None, None)
            cam = out
        return O.pinhole_project(cam, c["K"], c["land"])
    return O.A(c["meas"])


def test_interp_projection_cases(golden):
    """GPInterpolatedProjectionFactorPose3<Cal3_S2> (testGPInterpolatedProjectionFactorPose3.cpp:37-177)."""
    for c in golden["interp_projection"]:
        Lam, Psi = O.lambda_psi(6, c["qc"] * np.eye(6), c["dt"], c["tau"])
        p1, p2 = dec_pose(O.POSE3, c["p1"]), dec_pose(O.POSE3, c["p2"])
        v1, v2, land = O.A(c["v1"]), O.A(c["v2"]), O.A(c["land"])
        sensor = None if c["sensor"] is None else dec_pose(O.POSE3, c["sensor"])
        meas = _proj_meas(c)
        e, H, behind = O.interp_projection(Lam, Psi, meas, c["K"], sensor, p1, v1, p2, v2, land)
        assert not behind and np.abs(e - np.array(c["expect"])).max() <= c["tol_e"], c["src"]
        f = lambda a, b, cc, dd, ll: O.interp_projection(Lam, Psi, meas, c["K"], sensor, a, b, cc, dd, ll, jac=False)[0]
        num = [lambda h: numdiff_manifold(O.POSE3, lambda x: f(x, v1, p2, v2, land), p1, h),
               lambda h: numdiff_vector(lambda x: f(p1, x, p2, v2, land), v1, h),
               lambda h: numdiff_manifold(O.POSE3, lambda x: f(p1, v1, x, v2, land), p2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, p2, x, land), v2, h),
               lambda h: numdiff_vector(lambda x: f(p1, v1, p2, v2, x), land, h)]
        for k in range(5):
            ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
            assert ok, (c["src"], k, err)


def test_projection_cheirality_is_masked_not_thrown():
    """GPInterpolatedProjectionFactorPose3.h:122-138 with throwCheirality = false: error = 2 fx, zero Jacobians."""
    Lam, Psi = O.lambda_psi(6, 0.001 * np.eye(6), 0.1, 0.04)
    p = O.pose3((0, 0, 0), (0, 0, 0))
    e, H, behind = O.interp_projection(Lam, Psi, [0, 0], [50, 50, 0, 40, 30], None, p, np.zeros(6), p, np.zeros(6), [0, 0, -5])
    assert behind and np.all(e == 100.0) and all(np.all(h == 0) for h in H)


def build_projection_problem(c, chain):
    """testGPInterpolatedProjectionFactorPose3.cpp:180-262 on a ChainSolver-like object (landmark_dim = 3)."""
    p1, p2 = dec_pose(O.POSE3, c["p1"]), dec_pose(O.POSE3, c["p2"])
    meas = np.stack([O.pinhole_project(dec_pose(O.POSE3, cp), c["K"], c["land"]) for cp in c["cam_poses"]])
    chain.set_qc(c["qc"] * np.eye(6))
    chain.set_states(np.stack([dec_pose(O.POSE3, c["p1_init"]), dec_pose(O.POSE3, c["p2_init"])]),
                     np.stack([O.A(c["v1_init"]), O.A(c["v2_init"])]))
    chain.set_landmarks(np.array([c["land_init"]], dtype=float))
    chain.add_pose_priors([0, 1], np.stack([p1, p2]), np.full((2, 6), c["prior_sigma"]))
    chain.add_gp_priors([0], [c["dt"]])
    n = len(c["taus"])
    chain.add_interp_projection([0] * n, [0] * n, meas, np.full((n, 2), c["cam_sigma"]), [c["dt"]] * n, c["taus"], c["K"])
    chain.compile()
    return chain


def check_projection_result(c, chain):
    pose, vel = chain.get_states()
    assert pose_close(O.POSE3, dec_pose(O.POSE3, c["p1"]), pose[0], c["tol"])
    assert pose_close(O.POSE3, dec_pose(O.POSE3, c["p2"]), pose[1], c["tol"])
    assert np.abs(vel[0] - np.array(c["v1"])).max() <= c["tol"] and np.abs(vel[1] - np.array(c["v2"])).max() <= c["tol"]
    assert np.abs(chain.get_landmarks()[0] - np.array(c["land"])).max() <= c["tol"]
    assert chain.error() <= c["tol"]


def test_projection_optimisation_fixed_point(golden):
    c = golden["projection_optimization"]
    chain = build_projection_problem(c, O.Chain(O.POSE3, landmark_dim=3))
    rc, st = chain.optimize()
    assert rc == 0
    check_projection_result(c, chain)


def _gps_meas(c):
    if isinstance(c["meas"], dict):       # (true_pose * body_T_sensor).translation()
        out = np.zeros(12)
        O.call("orc_pose3_compose", dec_pose(O.POSE3, c["meas"]["true_pose"]), dec_pose(O.POSE3, c["sensor"]), out, None, None)
        return out[9:12].copy()
    return O.A(c["meas"])


def _interp_gps(Lam, Psi, meas, sensor, p1, s1, p2, s2, vw, jac=True):
    e = np.zeros(3)
    H = [np.zeros((3, 6)) for _ in range(4)] if jac else [None] * 4
    O.call("orc_interp_gps_pose3vw" if vw else "orc_interp_gps_pose3", O.A(Lam), O.A(Psi), O.A(meas),
           None if sensor is None else O.A(sensor), O.A(p1), O.A(s1), O.A(p2), O.A(s2), e, *H)
    return e, H


def test_interp_gps_cases(golden):
    """GPInterpolatedGPSFactorPose3 and ...Pose3VW (testGPInterpolatedGPSFactorPose3.cpp, ...Pose3VW.cpp): zero-error
    configurations, the literal (1.6, 0.2, 0) of the rotation-only sensor case, Jacobians vs central differences."""
    for key, vw in (("interp_gps", False), ("interp_gps_vw", True)):
        for c in golden[key]:
            Lam, Psi = O.lambda_psi(6, c["qc"] * np.eye(6), c["dt"], c["tau"])
            p1, p2 = dec_pose(O.POSE3, c["p1"]), dec_pose(O.POSE3, c["p2"])
            s1 = O.A(c["v1"] + c["w1"]) if vw else O.A(c["v1"])
            s2 = O.A(c["v2"] + c["w2"]) if vw else O.A(c["v2"])
            sensor = None if c["sensor"] is None else dec_pose(O.POSE3, c["sensor"])
            meas = _gps_meas(c)
            e, H = _interp_gps(Lam, Psi, meas, sensor, p1, s1, p2, s2, vw)
            if c["expect"] is not None:
                assert np.abs(e - np.array(c["expect"])).max() <= c["tol_e"], c["src"]
            f = lambda a, b, cc, dd: _interp_gps(Lam, Psi, meas, sensor, a, b, cc, dd, vw, jac=False)[0]
            num = [lambda h: numdiff_manifold(O.POSE3, lambda x: f(x, s1, p2, s2), p1, h),
                   lambda h: numdiff_vector(lambda x: f(p1, x, p2, s2), s1, h),
                   lambda h: numdiff_manifold(O.POSE3, lambda x: f(p1, s1, x, s2), p2, h),
                   lambda h: numdiff_vector(lambda x: f(p1, s1, p2, x), s2, h)]
            for k in range(4):      # VW: [H_v | H_w] packed side by side = the reference's H2|H3 and H5|H6
                ok, err = _jac_ok(H[k], num[k], c["fd"], c["tol_H"][k], "%s#H%d" % (c["src"], k + 1))
                assert ok, (c["src"], k, err)


def build_gps_problem(c, chain):
    """testGPInterpolatedGPSFactorPose3.cpp:193-262 on a ChainSolver-like object."""
    p1 = dec_pose(O.POSE3, c["p1"])
    chain.set_qc(c["qc"] * np.eye(6))
    chain.set_states(np.stack([dec_pose(O.POSE3, c["p1_init"]), dec_pose(O.POSE3, c["p2_init"])]),
                     np.stack([O.A(c["v1_init"]), O.A(c["v2_init"])]))
    chain.add_pose_priors([0], p1[None, :], np.full((1, 6), c["loose_sigma"]))
    chain.add_vel_priors([0, 1], np.stack([O.A(c["v1"]), O.A(c["v2"])]), np.full((2, 6), c["prior_sigma"]))
    chain.add_gp_priors([0], [c["dt"]])
    n = len(c["taus"])
    chain.add_interp_gps([0] * n, np.array(c["meas"], dtype=float), np.full((n, 3), c["gps_sigma"]), [c["dt"]] * n, c["taus"])
    chain.compile()
    return chain


def check_gps_result(c, chain):
    pose, vel = chain.get_states()
    assert pose_close(O.POSE3, dec_pose(O.POSE3, c["p1"]), pose[0], c["tol"])
    assert pose_close(O.POSE3, dec_pose(O.POSE3, c["p2"]), pose[1], c["tol"])
    assert np.abs(vel[0] - np.array(c["v1"])).max() <= c["tol"] and np.abs(vel[1] - np.array(c["v2"])).max() <= c["tol"]
    assert chain.error() <= c["tol"]


def test_gps_optimisation_fixed_point(golden):
    c = golden["gps_optimization"]
    chain = build_gps_problem(c, O.Chain(O.POSE3))
    rc, st = chain.optimize()
    assert rc == 0
    check_gps_result(c, chain)


def test_vw_conversion_known_answers(golden):
    """convertVWtoVb (testPose3Utils.cpp:345-420): values to 1e-9, the three Jacobians against central differences."""
    for c in golden["vw_conversion"]:
        pose, v, w = dec_pose(O.POSE3, c["pose"]), O.A(c["v"]), O.A(c["w"])
        v6, Hv, Hw, Hp = np.zeros(6), np.zeros((6, 3)), np.zeros((6, 3)), np.zeros((6, 6))
        O.call("orc_convertVWtoVb", v, w, pose, v6, Hv, Hw, Hp)
        assert np.abs(v6 - np.array(c["v6"])).max() <= 1e-9, c["src"]

        def f(a, b, p):
            out = np.zeros(6)
            O.call("orc_convertVWtoVb", O.A(a), O.A(b), O.A(p), out, None, None, None)
            return out
        assert np.abs(Hv - numdiff_vector(lambda x: f(x, w, pose), v, 1e-6)).max() <= 1e-6
        assert np.abs(Hw - numdiff_vector(lambda x: f(v, x, pose), w, 1e-6)).max() <= 1e-6
        assert np.abs(Hp - numdiff_manifold(O.POSE3, lambda x: f(v, w, x), pose, 1e-6)).max() <= 1e-6


def _fd_simple(kind, chart, fn, xs, which, h=1e-6):
    """Central difference of a PriorFactor / BetweenFactor error w.r.t. argument `which` under the chart's retract
    (numericalDerivative11 with traits::Retract, as the reference's Jacobian tests do)."""
    d = O.TANGENT_DIM[kind]
    cols = []
    for k in range(d):
        dl = np.zeros(d)
        dl[k] = h
        xp = list(xs); xm = list(xs)
        xp[which] = O.retract(kind, xs[which], dl, chart)
        xm[which] = O.retract(kind, xs[which], -dl, chart)
        cols.append((fn(*xp) - fn(*xm)) / (2 * h))
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("kind,chart", [(O.POSE2, O.CHART_FIRST_ORDER), (O.POSE2, O.CHART_EXPMAP),
                                        (O.POSE3, O.CHART_EXPMAP), (O.ROT3, O.CHART_EXPMAP), (O.LINEAR3, O.CHART_EXPMAP)])
def test_prior_between_jacobians_at_large_residual(kind, chart):
    """PriorFactor<T> / BetweenFactor<T> Jacobians vs central differences under the SAME chart at a residual heading far
    from zero (the Pose2 first-order chart had dLocal = R^T instead of R; nothing in the reference pins it)."""
    rng = np.random.default_rng(11)
    d, pd = O.TANGENT_DIM[kind], O.POSE_DIM[kind]

    def rnd():
        if kind == O.POSE2:
            return np.array([rng.normal(), rng.normal(), rng.uniform(-2.5, 2.5)])
        if kind == O.LINEAR3:
            return rng.normal(size=3)
        ident = np.concatenate([np.eye(3).ravel(), np.zeros(3)])[:pd]
        return O.retract(kind, ident, rng.normal(size=d) * 0.9)

    def prior_err(pr, x):
        e = np.zeros(d)
        O.call("orc_prior_factor", kind, chart, O.A(pr), O.A(x), e, None)
        return e

    def btw_err(m, x1, x2):
        e = np.zeros(d)
        O.call("orc_between_factor", kind, chart, O.A(m), O.A(x1), O.A(x2), e, None, None)
        return e

    for _ in range(5):
        pr, x, x2, m = rnd(), rnd(), rnd(), rnd()
        e, H = np.zeros(d), np.zeros((d, d))
        O.call("orc_prior_factor", kind, chart, pr, x, e, H)
        if kind == O.POSE2:
            assert abs(e[2]) > 1e-3            # the residual heading is not ~0: R and R^T differ
        np.testing.assert_allclose(H, _fd_simple(kind, chart, prior_err, [pr, x], 1), atol=2e-7)
        H1, H2 = np.zeros((d, d)), np.zeros((d, d))
        O.call("orc_between_factor", kind, chart, m, x, x2, e, H1, H2)
        np.testing.assert_allclose(H1, _fd_simple(kind, chart, btw_err, [m, x, x2], 1), atol=2e-7)
        np.testing.assert_allclose(H2, _fd_simple(kind, chart, btw_err, [m, x, x2], 2), atol=2e-7)


def test_jacobian_step_table_is_current():
    """runs last in this file: every case listed in tests/golden/jac_steps.json really needed a step larger than the reference's
    (a case that now passes at the reference step must leave the table), and needed no larger one than listed"""
    exc = _jac_exceptions()
    if os.environ.get("GPSLAM_JAC_DISCOVER"):      # discovery run: write what this run needed
        need = {k: h for k, (h, h_ref) in sorted(_JAC_LOG.items()) if h > h_ref * (1 + 1e-12)}
        with open(os.environ["GPSLAM_JAC_DISCOVER"], "w") as f:
            json.dump({"needs_larger_step": need}, f, indent=1, sort_keys=True)
        return
    if not _JAC_LOG:
        pytest.skip("the Jacobian cases did not run in this session")
    for case, allowed in exc.items():
        if case not in _JAC_LOG:
            continue
        h, h_ref = _JAC_LOG[case]
        assert h > h_ref * (1 + 1e-12), "%s now passes at the reference's own step %g: remove it from tests/golden/jac_steps.json" % (case, h_ref)
        assert abs(h - allowed) <= 1e-12 * h, "%s passed at step %g, the table says %g: update it" % (case, h, allowed)


def test_cal3ds2_projection_against_central_differences_and_cal3_s2():
    """GPInterpolatedProjectionFactorPose3<Cal3DS2> in the oracle (no reference test uses a distorting calibration: parity
    unpinned): H5 and the pose Jacobian of PinholeCamera<Cal3DS2>::project against central differences, Cal3_S2 at zero distortion."""
    rng = np.random.default_rng(77)
    cam = O.pose3((0.2, -0.1, 0.3), (0.5, -0.4, 0.2))
    K9 = [60.0, 55.0, 0.4, 32.0, 24.0, 0.09, -0.04, 0.005, -0.007]
    for _ in range(5):
        pt = cam[9:] + cam[:9].reshape(3, 3) @ np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(4, 9)])
        uv = np.zeros(2); Dpose = np.zeros((2, 6)); Dpoint = np.zeros((2, 3))
        assert O.call("orc_pinhole_project_ds2", cam, np.array(K9), pt, uv, Dpose, Dpoint) == 0
        h = 1e-6
        num = np.zeros((2, 3))
        for j in range(3):
            d = np.zeros(3); d[j] = h
            num[:, j] = (O.pinhole_project(cam, K9, pt + d) - O.pinhole_project(cam, K9, pt - d)) / (2 * h)
        assert np.abs(num - Dpoint).max() <= 1e-6 * max(1.0, np.abs(Dpoint).max())
        nump = np.zeros((2, 6))
        for j in range(6):
            d = np.zeros(6); d[j] = h
            nump[:, j] = (O.pinhole_project(O.retract(O.POSE3, cam, d), K9, pt) - O.pinhole_project(O.retract(O.POSE3, cam, -d), K9, pt)) / (2 * h)
        assert np.abs(nump - Dpose).max() <= 1e-5 * max(1.0, np.abs(Dpose).max())
        assert np.array_equal(O.pinhole_project(cam, K9[:5], pt), O.pinhole_project(cam, K9[:5] + [0, 0, 0, 0], pt))
        assert np.abs(O.pinhole_project(cam, K9[:5], pt) - uv).max() > 1e-3      # the distortion is not a no-op here
