"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

fp64 tolerances: factor errors and analytic Jacobians 1e-11 relative; the finite-difference rows of the Pose3
Jacobian (h = 1e-6 inside the reference's own evaluateError, GaussianProcessPriorPose3.h:81) amplify last-bit
differences between the CPU and GPU libm by 1/h, so those rows are compared at 1e-7 absolute; final states of an
optimisation 1e-9 relative (north-star tolerance); iteration counts exact.
"""
import numpy as np
import pytest

from helpers import KIND, dec_pose, pose_close
from oracle import oracle as O

pytestmark = pytest.mark.gpu

KINDS = [O.LINEAR2, O.LINEAR3, O.POSE2, O.POSE3, O.ROT3]
NAMES = {O.LINEAR2: "linear2", O.LINEAR3: "linear3", O.POSE2: "pose2", O.POSE3: "pose3", O.ROT3: "rot3"}


def gpu():
    import gpslam_amd
    return gpslam_amd


def random_chain(kind, N, seed, motion=0.3, noise=0.05):
    """A random but smooth trajectory (truth) plus noisy initial values, through the oracle's retract."""
    rng = np.random.default_rng(seed)
    d, pd = O.TANGENT_DIM[kind], O.POSE_DIM[kind]
    pose = np.zeros((N, pd))
    if kind in (O.POSE3,):
        pose[0] = O.pose3((0.3, -0.2, 0.1), (1.0, -2.0, 0.5))
    elif kind == O.ROT3:
        pose[0] = O.rot3_ypr(0.3, -0.2, 0.1)
    elif kind == O.POSE2:
        pose[0] = [1.0, -2.0, 0.4]
    vel = np.zeros((N, d))
    v = motion * rng.standard_normal(d)
    dt = 0.05 + 0.1 * rng.random(N - 1)
    for i in range(N - 1):
        v = v + 0.05 * rng.standard_normal(d)
        vel[i] = v
        pose[i + 1] = O.retract(kind, pose[i], dt[i] * v)
    vel[N - 1] = v
    init_pose = np.stack([O.retract(kind, pose[i], noise * rng.standard_normal(d)) for i in range(N)])
    init_vel = vel + noise * rng.standard_normal((N, d))
    return dict(truth_pose=pose, truth_vel=vel, pose=init_pose, vel=init_vel, dt=dt)


def build_pair(kind, N, seed, chart=None, with_between=True, chunk=0, vel_priors=True):
    """The same problem on the oracle and on the GPU.  chart=None: GTSAM's default chart of the manifold
    (first-order for Pose2, Expmap otherwise)."""
    if chart is None:
        chart = O.CHART_FIRST_ORDER if kind == O.POSE2 else O.CHART_EXPMAP
    rng = np.random.default_rng(seed + 77)
    d = O.TANGENT_DIM[kind]
    c = random_chain(kind, N, seed)
    Qc = np.diag(0.01 + 0.02 * rng.random(d))
    if d > 1:
        Qc[0, 1] = Qc[1, 0] = 0.003   # a non-diagonal Qc exercises the general whitening path
    solvers = []
    for make in (lambda: O.Chain(kind, chart), lambda: gpu().ChainSolver(kind, chart, chunk=chunk)):
        s = make()
        s.set_qc(Qc)
        s.set_states(c["pose"], c["vel"])
        s.add_gp_priors(np.arange(N - 1), c["dt"])
        # absolute fixes on the first and every 20th state keep the problem well conditioned (cond(H) ~ 1e6),
        # so the 1e-9 fixed-point tolerance is above cond * eps
        fix = np.arange(0, N, 20)
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
        if vel_priors:
            s.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, d), 0.05))
        if with_between and N > 1:
            meas = []
            for i in range(N - 1):
                if kind in (O.LINEAR2, O.LINEAR3):
                    meas.append(c["truth_pose"][i + 1] - c["truth_pose"][i])
                else:
                    # measured = x_i^-1 x_{i+1}: retract identity by the local coordinates
                    ident = {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0)}[kind]
                    meas.append(O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][i + 1])))
            s.add_between(np.arange(N - 1), np.stack(meas), np.full((N - 1, d), 0.02))
        s.compile()
        solvers.append(s)
    return solvers[0], solvers[1], c


def states_close(kind, a_pose, a_vel, b_pose, b_vel, rel):
    scale = max(1.0, float(np.abs(a_vel).max()))
    assert np.abs(a_vel - b_vel).max() <= rel * scale
    if kind in (O.LINEAR2, O.LINEAR3):
        assert np.abs(a_pose - b_pose).max() <= rel * max(1.0, float(np.abs(a_pose).max()))
    else:
        worst = max(float(np.abs(O.local(kind, a_pose[i], b_pose[i])).max()) for i in range(0, len(a_pose), max(1, len(a_pose) // 400)))
        pscale = max(1.0, float(np.abs(a_pose).max()))
        assert worst <= rel * pscale, worst


@pytest.mark.parametrize("kind", KINDS, ids=[NAMES[k] for k in KINDS])
def test_linearize_gp_matches_oracle(kind):
    orc, dev, _ = build_pair(kind, 257, seed=11 + kind, with_between=False)
    e0, H0 = orc.linearize_gp()
    e1, H1 = dev.linearize_gp()
    assert e0.shape == e1.shape and H0.shape == H1.shape
    assert np.abs(e0 - e1).max() <= 1e-11 * max(1.0, np.abs(e0).max())
    d = O.TANGENT_DIM[kind]
    if kind == O.POSE3:
        # top rows of H1/H3 and all of H2/H4 are analytic; bottom rows of H1/H3 carry the h = 1e-6 central difference
        assert np.abs(H0[:, :, :d, :] - H1[:, :, :d, :]).max() <= 1e-10
        assert np.abs(H0[:, [1, 3]] - H1[:, [1, 3]]).max() <= 1e-10
        assert np.abs(H0[:, [0, 2], d:, :] - H1[:, [0, 2], d:, :]).max() <= 1e-7
    else:
        assert np.abs(H0 - H1).max() <= 1e-10


def test_golden_gp_prior_cases_on_gpu(golden):
    """The reference's own known-answer cases (zero-error configurations) evaluated by the HIP kernel."""
    for c in golden["gp_prior"]:
        kind = KIND[c["kind"]]
        d = O.TANGENT_DIM[kind]
        s = gpu().ChainSolver(kind)
        s.set_qc(0.01 * np.eye(d))
        pose = np.stack([dec_pose(kind, c["p1"]), dec_pose(kind, c["p2"])])
        vel = np.array([c["v1"], c["v2"]], dtype=np.float64)
        s.set_states(pose, vel)
        s.add_gp_priors([0], [c["dt"]])
        s.compile()
        e, H = s.linearize_gp()
        if c["expect"] is not None:
            assert np.abs(e[0] - np.array(c["expect"])).max() <= c["tol_e"], c["src"]
        eo, Ho = O.gp_prior(kind, pose[0], vel[0], pose[1], vel[1], c["dt"])
        assert np.abs(e[0] - eo).max() <= 1e-10 * max(1.0, np.abs(eo).max()), c["src"]
        for k in range(4):
            assert np.abs(H[0, k] - Ho[k]).max() <= 2e-6, (c["src"], k)   # the reference's own Jacobian tolerance band


@pytest.mark.parametrize("kind", KINDS, ids=[NAMES[k] for k in KINDS])
def test_normal_equations_match_oracle(kind):
    orc, dev, _ = build_pair(kind, 130, seed=23 + kind)
    D0, O0, g0, _, _, _ = orc.normal_equations()
    D1, O1, g1, _ = dev.normal_equations()
    for a, b in ((D0, D1), (O0, O1), (g0, g1)):
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max())


def ident_states(kind, N):
    ident = {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0),
             O.LINEAR2: np.zeros(2), O.LINEAR3: np.zeros(3)}[kind]
    return np.tile(ident, (N, 1)), np.zeros((N, O.TANGENT_DIM[kind]))


def random_block_tridiag(N, b, seed):
    rng = np.random.default_rng(seed)
    D = np.zeros((N, b, b))
    Ocp = np.zeros((N, b, b))
    for i in range(N):
        A = rng.standard_normal((b, b))
        D[i] = A @ A.T + b * np.eye(b)
    for i in range(N - 1):
        Ocp[i] = 0.4 * rng.standard_normal((b, b))
    g = rng.standard_normal((N, b))
    return D, Ocp, g


@pytest.mark.parametrize("kind,N", [(O.POSE3, n) for n in (1, 2, 3, 16, 17, 31, 32, 33, 100, 257, 513, 4099)] +
                         [(O.POSE2, n) for n in (1, 5, 33, 600, 2049)] + [(O.LINEAR2, n) for n in (2, 40, 1000)])
def test_block_tridiag_solve_matches_oracle(kind, N):
    """Exercises every level count and the ragged / single-block chunk edges of the partitioned elimination."""
    b = 2 * O.TANGENT_DIM[kind]
    D, Ocp, g = random_block_tridiag(N, b, seed=N + b)
    s = gpu().ChainSolver(kind)
    s.set_states(*ident_states(kind, N))
    s.compile()
    x1 = s.block_tridiag_solve(D, Ocp, g)
    x0 = O.block_tridiag_solve(D, Ocp, g)
    assert np.abs(x0 - x1).max() <= 1e-10 * max(1.0, np.abs(x0).max())
    # residual of the original system (size-independent property)
    r = np.einsum("nij,nj->ni", D, x1) - g
    r[:-1] += np.einsum("nji,nj->ni", Ocp[:-1], x1[1:])
    r[1:] += np.einsum("nij,nj->ni", Ocp[:-1], x1[:-1])
    assert np.abs(r).max() <= 1e-9 * max(1.0, np.abs(g).max())


@pytest.mark.parametrize("chunk", [2, 3, 7, 16, 64])
def test_solver_chunk_lengths(chunk):
    N, kind = 333, O.POSE3
    D, Ocp, g = random_block_tridiag(N, 12, seed=chunk)
    s = gpu().ChainSolver(kind, chunk=chunk)
    s.set_states(*ident_states(kind, N))
    s.compile()
    x1 = s.block_tridiag_solve(D, Ocp, g)
    x0 = O.block_tridiag_solve(D, Ocp, g)
    assert np.abs(x0 - x1).max() <= 1e-10 * max(1.0, np.abs(x0).max())


@pytest.mark.parametrize("kind", KINDS, ids=[NAMES[k] for k in KINDS])
def test_gauss_newton_iterations_match_oracle(kind):
    """Iterate both sides in lock step.  Far from the optimum the error is steep in delta, so per-iteration
    scalars are compared at 1e-6 relative (last-bit differences in delta times the gradient); once converged
    the fixed point is compared at the north-star tolerance 1e-9."""
    orc, dev, _ = build_pair(kind, 300, seed=31 + kind)
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    for it in range(8):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_before - s1.error_before) <= 1e-6 * max(1.0, s0.error_before)
        assert abs(s0.error_after - s1.error_after) <= 1e-6 * max(1.0, s0.error_after)
        assert abs(s0.delta_inf_norm - s1.delta_inf_norm) <= 1e-6 * max(1.0, s0.delta_inf_norm) + 1e-10
    assert s0.delta_inf_norm < 1e-8 and s1.delta_inf_norm < 1e-8
    assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after)
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(kind, p0, v0, p1, v1, 1e-9)


@pytest.mark.parametrize("kind", KINDS, ids=[NAMES[k] for k in KINDS])
def test_levenberg_marquardt_past_convergence_matches_oracle(kind):
    """LevenbergMarquardtOptimizer::iterate() in lock step INTO and PAST convergence on every manifold (round 4's red case was a
    linear chain on its third call): the lambda schedule, accept flags and trial counts are identical while the cost moves; once
    a call moves it by rounding only, GTSAM's small-cost-change stop ends the call after one trial on both sides and lambda
    stays or is divided once (tests/lm_lockstep.py).  Nine calls: the linear chains converge in two."""
    import lm_lockstep
    orc, dev, _ = build_pair(kind, 300, seed=61 + kind)
    lam, n_noise, slack = lm_lockstep.run(orc, dev, 1e-2, 9, tag=NAMES[kind])
    assert n_noise >= 2, (NAMES[kind], n_noise)        # the run really went past convergence
    # lambda never climbed on noise (it did before the stop rule: 1e-5 -> 1e4).  slack: on a flat cost (SO(3): the velocities are
    # weakly held) a step that moves the cost by 1e-10 relative is still ~1e-6 long -- the values of two optimisers that disagree on
    # keeping it differ by that much, their errors do not
    assert lam <= 1e-2 and slack <= 1e-4, (lam, slack)
    assert abs(orc.error() - dev.error()) <= 1e-9 * max(1.0, orc.error())
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(kind, p0, v0, p1, v1, 1e-9 + 2 * slack)


@pytest.mark.parametrize("chart", [O.CHART_EXPMAP, O.CHART_FIRST_ORDER])
def test_pose2_chart_option_matches_oracle(chart):
    """Pose2 with both charts (Expmap = GTSAM's SLOW_BUT_CORRECT_EXPMAP, first-order = GTSAM's default)."""
    orc, dev, _ = build_pair(O.POSE2, 120, seed=5, chart=chart)
    for it in range(9):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
    assert s0.delta_inf_norm < 1e-10 and s1.delta_inf_norm < 1e-10
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(O.POSE2, p0, v0, p1, v1, 1e-9)


def test_optimize_iteration_count_and_fixed_point_match_oracle():
    for kind in KINDS:
        orc, dev, _ = build_pair(kind, 200, seed=41 + kind)
        rc0, s0 = orc.optimize()
        rc1, s1 = dev.optimize()
        assert rc0 == 0 and rc1 == 0
        assert s0.iterations == s1.iterations, (NAMES[kind], s0.iterations, s1.iterations)
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after)
        p0, v0 = orc.get_states()
        p1, v1 = dev.get_states()
        states_close(kind, p0, v0, p1, v1, 1e-9)


def test_reference_two_state_optimisations_on_gpu(golden):
    """The reference's own end-to-end tests (GaussNewtonOptimizer on 2 states), landmark-free ones."""
    from test_oracle_golden import build_opt_problem, check_opt_result
    for c in golden["optimization"]:
        if c["landmark_dim"]:
            continue
        ch, kind = build_opt_problem(c, lambda k, chart, ld: gpu().ChainSolver(k, chart, ld))
        rc, st = ch.optimize()
        assert rc == 0 and st.iterations < 100, c["src"]
        check_opt_result(c, ch, kind)


def test_full_size_pose3_chain_converges_and_matches_oracle():
    """BASELINE config 3 at full size (1e5 Pose3 states): size-independent properties + final-state parity."""
    from gpslam_amd import synthetic as S
    N = 100000
    p = S.pose3_chain(N)
    dev = S.apply(p, gpu().ChainSolver(O.POSE3))
    errs, iters = [dev.error()], 0
    while iters < 20:
        rc, st = dev.iterate_gn()
        assert rc == 0
        iters += 1
        errs.append(st.error_after)
        if st.delta_inf_norm < 1e-6:
            break
    assert st.delta_inf_norm < 1e-6 and iters <= 8, (iters, st.delta_inf_norm)
    assert all(errs[i + 1] <= errs[i] * (1 + 1e-12) for i in range(len(errs) - 1))      # monotone descent
    rc, st2 = dev.iterate_gn()                                                            # idempotence at the fixed point
    assert st2.delta_inf_norm < 1e-7 and abs(st2.error_after - errs[-1]) <= 1e-9 * errs[-1]
    # gradient of the normal equations vanishes at the fixed point
    D, Ocp, g, _ = dev.normal_equations()
    assert np.abs(g).max() <= 1e-5 * max(1.0, np.abs(D).max())
    # final state against the oracle run on the identical problem with the identical number of iterations
    orc = S.apply(p, O.Chain(O.POSE3))
    for _ in range(iters + 1):
        orc.iterate_gn()
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(O.POSE3, p0, v0, p1, v1, 1e-9)


def test_full_size_linear_chain_one_step():
    """BASELINE config 2 (1e5 Linear<3> states): a linear problem is solved by one Gauss-Newton step."""
    from gpslam_amd import synthetic as S
    p = S.linear_chain(100000)
    dev = S.apply(p, gpu().ChainSolver(O.LINEAR3))
    rc, st1 = dev.iterate_gn()
    rc, st2 = dev.iterate_gn()
    assert st2.delta_inf_norm <= 1e-8 and abs(st2.error_after - st1.error_after) <= 1e-9 * st1.error_after
    orc = S.apply(p, O.Chain(O.LINEAR3))
    orc.iterate_gn()
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    states_close(O.LINEAR3, p0, v0, p1, v1, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [O.LINEAR3, O.POSE2, O.ROT3, O.POSE3])
def test_interpolate_poses_query(kind):
    """gpslam_hip_interpolate_poses = batched GaussianProcessInterpolator*::interpolatePose (gpslam.h:57-86)
    against the oracle's per-call interpolators, on a perturbed chain."""
    orc, dev, c = build_pair(kind, 40, seed=5)
    rng = np.random.default_rng(11)
    Q = 64
    left = rng.integers(0, 39, Q).astype(np.int32)
    dt = np.full(Q, 0.1)
    tau = rng.uniform(0.0, 0.1, Q)
    tau[:2] = [0.0, 0.1]                      # the interval's end points reproduce the states themselves
    pose, vel = dev.get_states()
    d = O.TANGENT_DIM[kind]
    Qc = np.diag(0.01 + 0.02 * np.random.default_rng(5 + 77).random(d))   # build_pair's Qc
    Qc[0, 1] = Qc[1, 0] = 0.003
    dt = np.asarray(c["dt"])[left]
    tau = tau / 0.1 * dt
    got = dev.interpolate_poses(left, dt, tau)
    for q in range(Q):
        Lam, Psi = O.lambda_psi(d, Qc, dt[q], tau[q])
        want, _ = O.interpolate(kind, Lam, Psi, pose[left[q]], vel[left[q]], pose[left[q] + 1], vel[left[q] + 1], jac=False)
        assert np.abs(got[q] - want).max() <= 1e-11 * max(1.0, np.abs(want).max())
    assert np.abs(got[0] - pose[left[0]]).max() <= 1e-12
    if kind != O.POSE2:
        assert np.abs(got[1] - pose[left[1] + 1]).max() <= 1e-10
    with pytest.raises(Exception):
        dev.interpolate_poses([39], [0.1], [0.05])


@pytest.mark.parametrize("kind", KINDS, ids=[NAMES[k] for k in KINDS])
def test_interpolate_poses_jacobians(kind):
    """H1..H4 of GaussianProcessInterpolator*::interpolatePose (gpslam.h:57-86, e.g. GaussianProcessInterpolatorPose3.h:82-98)
    query by query against the oracle's interpolators; the SE(3) rows that pass through the h = 1e-6 difference 1e-7."""
    orc, dev, c = build_pair(kind, 30, seed=9)
    rng = np.random.default_rng(12)
    Q = 40
    left = rng.integers(0, 29, Q).astype(np.int32)
    d = O.TANGENT_DIM[kind]
    Qc = np.diag(0.01 + 0.02 * np.random.default_rng(9 + 77).random(d))
    if d > 1:
        Qc[0, 1] = Qc[1, 0] = 0.003
    dt = np.asarray(c["dt"])[left]
    tau = rng.uniform(-0.1, 1.1, Q) * dt
    pose, vel = dev.get_states()
    got, H = dev.interpolate_poses_jac(left, dt, tau)
    tol = 1e-7 if kind == O.POSE3 else 1e-10
    for q in range(Q):
        Lam, Psi = O.lambda_psi(d, Qc, dt[q], tau[q])
        want, Hw = O.interpolate(kind, Lam, Psi, pose[left[q]], vel[left[q]], pose[left[q] + 1], vel[left[q] + 1], jac=True)
        assert np.abs(got[q] - want).max() <= 1e-11 * max(1.0, np.abs(want).max())
        for m in range(4):
            assert np.abs(H[q, m] - Hw[m]).max() <= tol * max(1.0, np.abs(Hw[m]).max()), (q, m, np.abs(H[q, m] - Hw[m]).max())
