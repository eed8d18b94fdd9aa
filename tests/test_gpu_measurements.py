"""GPU parity for the measurement factors, the landmark border and Levenberg-Marquardt.

Every problem is built twice from the same seeded arrays: on the CPU oracle and on the GPU through the C ABI.
"""
import numpy as np
import pytest

from helpers import KIND, dec_pose, pose_close
from oracle import oracle as O
from test_gpu_parity import gpu, random_chain, states_close, NAMES

pytestmark = pytest.mark.gpu

LD = {O.POSE2: 2, O.POSE3: 3, O.LINEAR3: 2, O.ROT3: 0}


def ident(kind):
    return {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0),
            O.LINEAR3: np.zeros(3)}[kind]


def interp_truth(kind, Qc, c, i, tau):
    """pose at time tau inside interval i of the truth trajectory (oracle interpolator)"""
    d = O.TANGENT_DIM[kind]
    Lam, Psi = O.lambda_psi(d, Qc, c["dt"][i], tau)
    return O.interpolate(kind, Lam, Psi, c["truth_pose"][i], c["truth_vel"][i], c["truth_pose"][i + 1],
                         c["truth_vel"][i + 1], jac=False)[0]


def true_range(kind, pose, land, sensor=None):
    if kind == O.POSE3:
        sp = pose
        if sensor is not None:
            sp = np.zeros(12)
            O.call("orc_pose3_compose", O.A(pose), O.A(sensor), sp, None, None)
        return O.call("orc_pose3_range", O.A(sp), O.A(land), None, None)
    if kind == O.POSE2 and sensor is not None:
        sp = np.zeros(3)
        O.call("orc_pose2_compose", O.A(pose), O.A(sensor), sp, None, None)
        pose = sp
    return float(np.hypot(land[0] - pose[0], land[1] - pose[1]))


def build_meas_pair(kind, N=48, seed=3, sensor=False, chart=None, extra_makers=(), cov_hook=None):
    """cov_hook(solver, MEAS kind, count): called after each batch of measurement factors (e.g. to give them full covariances)"""
    hook = cov_hook if cov_hook is not None else (lambda s, k, n: None)
    rng = np.random.default_rng(seed + 1000)
    d, ld = O.TANGENT_DIM[kind], LD[kind]
    if chart is None:
        chart = O.CHART_FIRST_ORDER if kind == O.POSE2 else O.CHART_EXPMAP
    c = random_chain(kind, N, seed, noise=0.03)
    Qc = np.diag(0.01 + 0.02 * rng.random(d))
    L = 3 if ld else 0
    lands_true = rng.uniform(-6, 6, (L, ld)) if L else None
    lands_init = lands_true + 0.1 * rng.standard_normal((L, ld)) if L else None
    S = None
    if sensor and kind == O.POSE3:
        S = O.pose3((0.3, -0.2, 0.1), (0.2, -0.1, 0.3))
    if sensor and kind == O.POSE2:
        S = np.array([0.2, -0.1, 0.3])
    fix = np.arange(0, N, 16)
    # interpolated measurements: one or two per interval, tau mostly inside [0, dt], some extrapolating
    left = np.sort(rng.integers(0, N - 1, size=2 * N)).astype(np.int32)
    tau = np.array([c["dt"][i] * rng.uniform(-0.2, 1.2) for i in left])
    specs = dict(left=left, tau=tau, dts=c["dt"][left])
    if ld:
        lm = rng.integers(0, L, size=len(left)).astype(np.int32)
        z = np.array([true_range(kind, interp_truth(kind, Qc, c, i, t), lands_true[l], S) for i, t, l in zip(left, tau, lm)])
        z = z + 0.01 * rng.standard_normal(len(z))
        specs.update(lm=lm, z=z)
        uidx = rng.integers(0, N, size=N // 2).astype(np.int32)
        ulm = rng.integers(0, L, size=len(uidx)).astype(np.int32)
        uz = np.array([true_range(kind, c["truth_pose"][i], lands_true[l]) for i, l in zip(uidx, ulm)])
        specs.update(uidx=uidx, ulm=ulm, uz=uz + 0.01 * rng.standard_normal(len(uz)))
    solvers = []
    for make in (lambda: O.Chain(kind, chart, ld), lambda: gpu().ChainSolver(kind, chart, ld)) + tuple(extra_makers):
        s = make()
        s.set_qc(Qc)
        s.set_states(c["pose"], c["vel"])
        if L:
            s.set_landmarks(lands_init)
        s.add_gp_priors(np.arange(N - 1), c["dt"])
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.02))
        s.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, d), 0.05))
        if L:
            s.add_landmark_priors(np.arange(L), lands_true + 0.05, np.full((L, ld), 0.5))
            s.add_interp_range(left, specs["lm"], specs["z"], np.full(len(left), 0.05), specs["dts"], tau, S)
            hook(s, 0, len(left))
            s.add_range(specs["uidx"], specs["ulm"], specs["uz"], np.full(len(specs["uidx"]), 0.05))
            hook(s, 1, len(specs["uidx"]))
        if kind == O.ROT3:
            nZ = np.tile([0.0, 0.0, 1.0], (len(left), 1))
            bref = []
            for i, t in zip(left, tau):
                R = interp_truth(kind, Qc, c, i, t).reshape(3, 3)
                bref.append(R.T @ np.array([0.0, 0.0, 1.0]) + 0.01 * rng.standard_normal(3))
            specs.setdefault("bref", np.array(bref))
            s.add_interp_attitude(left, nZ, specs["bref"], np.full((len(left), 2), 0.05), specs["dts"], tau)
            hook(s, 2, len(left))
        if kind == O.POSE3:
            gl = left[::3]
            gt = tau[::3]
            if "gps" not in specs:
                gm = []
                for i, t in zip(gl, gt):
                    p = interp_truth(kind, Qc, c, i, t)
                    gm.append(p[9:12] + 0.01 * rng.standard_normal(3))
                specs["gps"] = np.array(gm)
            s.add_interp_gps(gl, specs["gps"], np.full((len(gl), 3), 0.05), c["dt"][gl], gt)
            hook(s, 3, len(gl))
        if kind == O.LINEAR3:
            if "odo" not in specs:
                odo = []
                for i in range(N - 1):
                    a, b2 = c["truth_pose"][i], c["truth_pose"][i + 1]
                    cs, sn = np.cos(a[2]), np.sin(a[2])
                    dx, dy = b2[0] - a[0], b2[1] - a[1]
                    odo.append([cs * dx + sn * dy, -sn * dx + cs * dy, b2[2] - a[2]])
                specs["odo"] = np.array(odo) + 0.005 * rng.standard_normal((N - 1, 3))
                bidx = rng.integers(0, N, size=N // 2).astype(np.int32)
                blm = rng.integers(0, L, size=len(bidx)).astype(np.int32)
                bear, brng = [], []
                for i, l in zip(bidx, blm):
                    p = c["truth_pose"][i]
                    dx, dy = lands_true[l][0] - p[0], lands_true[l][1] - p[1]
                    cs, sn = np.cos(p[2]), np.sin(p[2])
                    bear.append(np.arctan2(-sn * dx + cs * dy, cs * dx + sn * dy))
                    brng.append(np.hypot(dx, dy))
                specs.update(bidx=bidx, blm=blm, bear=np.array(bear) + 0.01 * rng.standard_normal(len(bear)),
                             brng=np.array(brng) + 0.01 * rng.standard_normal(len(brng)))
            s.add_odometry2d(np.arange(N - 1), specs["odo"], np.full((N - 1, 3), 0.02))
            hook(s, 4, N - 1)
            s.add_bearing_range(specs["bidx"], specs["blm"], specs["bear"], specs["brng"], np.full((len(specs["bidx"]), 2), 0.05))
            hook(s, 5, len(specs["bidx"]))
        s.compile()
        solvers.append(s)
    if extra_makers:
        return solvers[0], solvers[1], c, solvers[2:]
    return solvers[0], solvers[1], c


CASES = [(O.POSE2, False), (O.POSE2, True), (O.POSE3, False), (O.POSE3, True), (O.LINEAR3, False), (O.ROT3, False)]
IDS = ["pose2", "pose2+sensor", "pose3", "pose3+sensor", "linear3", "rot3-attitude"]


@pytest.mark.parametrize("kind,sensor", CASES, ids=IDS)
def test_measurement_normal_equations_match_oracle(kind, sensor):
    """Rows of every measurement factor (interpolated range / attitude / GPS, range, odometry, bearing-range)
    through the assembled normal equations, including the landmark border B."""
    orc, dev, _ = build_meas_pair(kind, sensor=sensor)
    assert abs(orc.error() - dev.error()) <= 1e-10 * max(1.0, orc.error())
    D0, O0, g0, B0, _, _ = orc.normal_equations()
    D1, O1, g1, B1 = dev.normal_equations()
    for a, b in ((D0, D1), (O0, O1), (g0, g1)):
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max())
    if B0 is not None:
        assert np.abs(B0 - B1).max() <= 1e-9 * max(1.0, np.abs(B0).max())


@pytest.mark.parametrize("kind,sensor", CASES, ids=IDS)
def test_gauss_newton_with_measurements_matches_oracle(kind, sensor):
    orc, dev, _ = build_meas_pair(kind, sensor=sensor)
    for it in range(8):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-6 * max(1.0, s0.error_after)
        assert abs(s0.delta_inf_norm - s1.delta_inf_norm) <= 1e-6 * max(1.0, s0.delta_inf_norm) + 1e-10
    assert s1.delta_inf_norm < 1e-7
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(kind, p0, v0, p1, v1, 1e-9)
    if LD[kind]:
        assert np.abs(orc.get_landmarks() - dev.get_landmarks()).max() <= 1e-9 * 10


@pytest.mark.parametrize("kind", [O.POSE2, O.POSE3, O.LINEAR3, O.ROT3], ids=["pose2", "pose3", "linear3", "rot3"])
def test_levenberg_marquardt_matches_oracle(kind):
    """LevenbergMarquardtOptimizer::iterate step by step while the decisions are well separated from rounding
    (the first iterations), then optimize() to convergence on both sides."""
    orc, dev, _ = build_meas_pair(kind, seed=9)
    import lm_lockstep
    lam = 1e-5
    for it in range(3):
        s0, s1, lam, noise = lm_lockstep.step(orc, dev, lam, err_tol=1e-6, tag=it)
        assert not noise and s0["accepted"] == s1["accepted"] == 1
    rc0, s0 = orc.optimize(O.default_params(use_lm=1))
    rc1, s1 = dev.optimize(dev.default_params(use_lm=1))
    assert rc0 == 0 and rc1 == 0
    assert abs(s0.error_after - s1.error_after) <= 1e-8 * max(1.0, s0.error_after)
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(kind, p0, v0, p1, v1, 1e-6)   # LM stops on the error decrease, not at |delta| = 0


def test_levenberg_marquardt_rejects_and_recovers():
    """A start far from the optimum forces rejected steps: lambda climbs 1e-8 -> 1 inside one iterate() and comes
    back down, identically on both sides."""
    orc, dev, c = build_meas_pair(O.POSE2, seed=21)
    rng = np.random.default_rng(5)
    bad_pose = np.stack([O.retract(O.POSE2, p, 0.5 * rng.standard_normal(3), O.CHART_FIRST_ORDER) for p in c["pose"]])
    for s in (orc, dev):
        s.set_states(bad_pose, c["vel"] * 0.0)
    import lm_lockstep
    lam = 1e-5
    lams = []
    for it in range(8):
        _, _, lam, _ = lm_lockstep.step(orc, dev, lam, err_tol=1e-6, tag=it)
        lams.append(lam)
    assert max(lams) >= 1e-2 and min(lams) <= 1e-7     # the schedule really went up and down


def test_reference_two_state_optimisations_with_landmarks_on_gpu(golden):
    """The reference's own interpolated-range end-to-end tests (tau outside [0, dt] included)."""
    from test_oracle_golden import build_opt_problem, check_opt_result
    n = 0
    for c in golden["optimization"]:
        if not c["landmark_dim"]:
            continue
        ch, kind = build_opt_problem(c, lambda k, chart, ld: gpu().ChainSolver(k, chart, ld))
        rc, st = ch.optimize()
        assert rc == 0 and st.iterations < 100, c["src"]
        check_opt_result(c, ch, kind)
        n += 1
    assert n == 3


def test_reference_interp_range_cases_on_gpu(golden):
    """Known-answer interpolated-range cases of the reference evaluated by the HIP kernel (error = 0.5 (e/sigma)^2)."""
    for c in golden["interp_range"]:
        kind = KIND[c["kind"]]
        d, ld = O.TANGENT_DIM[kind], len(c["land"])
        sensor = None if c["sensor"] is None else dec_pose(kind, c["sensor"])
        meas = c["meas"]
        if isinstance(meas, dict):
            meas = true_range(kind, dec_pose(kind, meas["true_pose"]), np.array(c["land"], dtype=np.float64), sensor)
        s = gpu().ChainSolver(kind, O.CHART_EXPMAP, ld)
        s.set_qc(c["qc"] * np.eye(d))
        s.set_states(np.stack([dec_pose(kind, c["p1"]), dec_pose(kind, c["p2"])]), np.array([c["v1"], c["v2"]], dtype=np.float64))
        s.set_landmarks(np.array([c["land"]], dtype=np.float64))
        s.add_interp_range([0], [0], [meas], [0.1], [c["dt"]], [c["tau"]], sensor)
        s.compile()
        err = s.error()
        if c["expect"] is not None:
            assert abs(np.sqrt(2 * err) * 0.1 - abs(c["expect"])) <= c["tol_e"], c["src"]
        # same factor through the oracle chain
        o = O.Chain(kind, O.CHART_EXPMAP, ld)
        o.set_qc(c["qc"] * np.eye(d))
        o.set_states(np.stack([dec_pose(kind, c["p1"]), dec_pose(kind, c["p2"])]), np.array([c["v1"], c["v2"]], dtype=np.float64))
        o.set_landmarks(np.array([c["land"]], dtype=np.float64))
        o.add_interp_range([0], [0], [meas], [0.1], [c["dt"]], [c["tau"]], sensor)
        assert abs(o.error() - err) <= 1e-9 * max(1.0, err), c["src"]


@pytest.mark.parametrize("kind,sensor", CASES, ids=IDS)
def test_measurement_factors_per_factor_error_and_jacobians(kind, sensor):
    """evaluateError + H1..H5 of every measurement factor, FACTOR BY FACTOR (gpslam_hip_linearize_meas), against the
    oracle's per-factor evaluation: unwhitened errors 1e-11, analytic Jacobians 1e-10; the SE(3) interpolator's rows that
    pass through the reference's h = 1e-6 finite difference (GaussianProcessInterpolatorPose3.h:84-85) 1e-7."""
    orc, dev, c = build_meas_pair(kind, N=40, seed=5, sensor=sensor)
    counts = {0: 80 if LD[kind] else 0, 1: 20 if LD[kind] else 0, 2: 80 if kind == O.ROT3 else 0,
              3: 27 if kind == O.POSE3 else 0, 4: 39 if kind == O.LINEAR3 else 0, 5: 20 if kind == O.LINEAR3 else 0}
    checked = 0
    for mk, n in counts.items():
        if n == 0:
            continue
        e0, J0 = orc.linearize_meas(mk, n)
        e1, J1 = dev.linearize_meas(mk, n)
        assert np.abs(e0 - e1).max() <= 1e-11 * max(1.0, np.abs(e0).max()), (mk, np.abs(e0 - e1).max())
        tol = 1e-7 if kind == O.POSE3 and mk in (0, 3) else 1e-10
        assert np.abs(J0 - J1).max() <= tol * max(1.0, np.abs(J0).max()), (mk, np.abs(J0 - J1).max())
        assert np.abs(J0).max() > 0.1
        checked += n
    assert checked > 0
