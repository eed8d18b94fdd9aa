"""The row-layout level-0 elimination (k_chunk_forward_rows: four chunks per wave on 16-lane DPP rows) against the
oracle on the shapes its control flow distinguishes: chunk counts that do not fill a wave, a ragged last chunk, a last
chunk without interior (n mod m == 1), chunks of 2 (one interior block), explicit chunk lengths either side of the
automatic one, and Levenberg-Marquardt damping (lambda added to the staged record images)."""
import numpy as np
import pytest
import torch  # noqa: F401  -- before the HIP library: torch ships its own HIP runtime, and whichever loads first must be it

from oracle import oracle as O
import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chunk", [2, 3, 5, 13, 16])
def test_pose3_chunk_shapes(chunk):
    for N in (chunk + 1, 4 * chunk + 1, 5 * chunk, 7 * chunk + 2, 211):
        orc, dev, c = T.build_pair(O.POSE3, N, seed=100 + N, chunk=chunk)
        for _ in range(3):
            rc0, s0 = orc.iterate_gn()
            rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0, (N, chunk)
            assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, abs(s0.error_after)), (N, chunk)
        (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
        T.states_close(O.POSE3, x0, v0, x1, v1, 1e-9)


@pytest.mark.parametrize("chunk", [2, 3, 5, 13, 16, 0])
def test_pose3_structured_gp_records(chunk):
    """Chains whose only full-width rows are GP priors: K1 hands them to the fused kernel as structured records (the
    velocity columns of the whitened Jacobian are multiples of U and of U Jr^-1, synthesised by the assembly wave) instead of
    rows.  Same shapes as above, against the oracle."""
    for N in ((chunk + 2, 4 * chunk + 1, 5 * chunk + 3, 7 * chunk + 2, 211, 1500) if chunk else (40, 211, 1500, 6000)):
        orc, dev, c = T.build_pair(O.POSE3, N, seed=300 + N, chunk=chunk, vel_priors=False)
        info = dev.plan_info()
        assert info["structured_gp"] == (1 if info["fused"] else 0) and info["rows_full"] == 12 * (N - 1)
        for _ in range(3):
            rc0, s0 = orc.iterate_gn()
            rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0, (N, chunk)
            assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, abs(s0.error_after)), (N, chunk)
        (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
        T.states_close(O.POSE3, x0, v0, x1, v1, 1e-9)
    assert info["fused"] == 1 and info["structured_gp"] == 1      # (N = 1500: more than one level for every chunk length)


def test_structured_records_reproduce_the_row_path():
    """The same chain four ways: structured records alone; with ONE zero-weight velocity prior (a full-width row the
    structured kernel fetches without a ring: k_fused_level0<2>); with nine of them (more than that variant takes: the records
    next to a ring of full-width rows, <3> -- what chains with interpolated measurement factors run); and with the GP priors as
    plain Jacobian rows (GPSLAM_PLAN_GP_ROWS).  An infinite sigma makes a prior
    contribute exactly nothing: all four must agree to rounding (round 4: the record holds Jr^-1, J and the
    finite-difference block, the assembly wave forms the whitened columns -- U (sa J + sb F J) where K1's rows are
    sa U J + sb U (F J); the pure variant also takes the between factors as records and adds their rows before the pose
    priors', the variant with odd rows keeps them as compact rows behind the pose priors': same numbers, different order)."""
    N = 900
    res = []
    import gpslam_amd
    for extra in (0, 1, 9, -1):
        orc, dev, c = T.build_pair(O.POSE3, N, seed=77, vel_priors=False)
        if extra:
            if extra < 0:      # the row path: a handle of its own with the plan bit, same factors as the first one
                Qc = np.diag(0.01 + 0.02 * np.random.default_rng(77 + 77).random(6))
                Qc[0, 1] = Qc[1, 0] = 0.003
                dev = gpslam_amd.ChainSolver(O.POSE3, O.CHART_EXPMAP, plan=gpslam_amd.PLAN_GP_ROWS)
                dev.set_qc(Qc)
                dev.set_states(c["pose"], c["vel"])
            dev.clear_factors()
            d = 6
            dev.add_gp_priors(np.arange(N - 1), c["dt"])
            fix = np.arange(0, N, 20)
            dev.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
            ident = O.pose3((0, 0, 0), (0, 0, 0))
            meas = np.stack([O.retract(O.POSE3, ident, O.local(O.POSE3, c["truth_pose"][i], c["truth_pose"][i + 1])) for i in range(N - 1)])
            dev.add_between(np.arange(N - 1), meas, np.full((N - 1, d), 0.02))
            if extra > 0:
                where = np.arange(5, 5 + 37 * extra, 37)
                dev.add_vel_priors(where, np.zeros((extra, d)), np.full((extra, d), np.inf))
            dev.compile()
        info = dev.plan_info()
        assert info["structured_gp"] == (0 if extra < 0 else 1) and info["rows_full"] == 12 * (N - 1) + 6 * max(extra, 0)
        for _ in range(3):
            dev.iterate_gn()
        res.append(dev.get_states())
    for k in (1, 2, 3):
        assert np.abs(res[0][0] - res[k][0]).max() <= 1e-11 * max(1.0, np.abs(res[k][0]).max())
        assert np.abs(res[0][1] - res[k][1]).max() <= 1e-11 * max(1.0, np.abs(res[k][1]).max())


@pytest.mark.parametrize("N", [700, 1500])
def test_diagonal_qc_takes_the_short_form_of_the_assembly_wave(N):
    """A diagonal Qc (chol_upper(Qc^-1) diagonal: the host looks at the 36 numbers per launch) selects
    k_fused_level0<1, double, 12, true>: row q of the whitened L meets only velocity column 6 + q mod 6, U Z is six products.
    What it skips are products with exact zeros, so it must land on the general form's (GPSLAM_PLAN_GENERIC_QC) numbers exactly,
    and both on the oracle's; a set_qc to a non-diagonal Qc in mid-run must switch forms (test below keeps that honest too)."""
    import gpslam_amd
    rng = np.random.default_rng(5)
    Qc = np.diag(0.01 + 0.02 * rng.random(6))
    c = T.random_chain(O.POSE3, N, 31)
    out = []
    for make in (lambda: O.Chain(O.POSE3, O.CHART_EXPMAP), lambda: gpslam_amd.ChainSolver(O.POSE3, O.CHART_EXPMAP),
                 lambda: gpslam_amd.ChainSolver(O.POSE3, O.CHART_EXPMAP, plan=gpslam_amd.PLAN_GENERIC_QC)):
        s = make()
        s.set_qc(Qc)
        s.set_states(c["pose"], c["vel"])
        s.add_gp_priors(np.arange(N - 1), c["dt"])
        fix = np.arange(0, N, 20)
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), 6), 0.01))
        ident = O.pose3((0, 0, 0), (0, 0, 0))
        meas = np.stack([O.retract(O.POSE3, ident, O.local(O.POSE3, c["truth_pose"][i], c["truth_pose"][i + 1])) for i in range(N - 1)])
        s.add_between(np.arange(N - 1), meas, np.full((N - 1, 6), 0.02))
        s.compile()
        for _ in range(4):
            s.iterate_gn()
        out.append(s.get_states())
    T.states_close(O.POSE3, out[0][0], out[0][1], out[1][0], out[1][1], 1e-9)
    assert np.array_equal(out[1][0], out[2][0]) and np.array_equal(out[1][1], out[2][1])


def test_set_qc_after_compile_reaches_the_structured_path():
    """set_qc after compile(): the row path reads U per launch, the structured records' assembly reads a device copy of U --
    both must see the new Qc (same problem on the oracle)."""
    orc, dev, c = T.build_pair(O.POSE3, 700, seed=21, vel_priors=False)
    assert dev.plan_info()["structured_gp"] == 1
    dev.iterate_gn(); orc.iterate_gn()
    Qc = np.diag([0.05, 0.02, 0.03, 0.04, 0.06, 0.01])
    Qc[0, 2] = Qc[2, 0] = 0.004
    dev.set_qc(Qc); orc.set_qc(Qc)
    for _ in range(3):
        _, s0 = orc.iterate_gn()
        _, s1 = dev.iterate_gn()
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, abs(s0.error_after))
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    T.states_close(O.POSE3, x0, v0, x1, v1, 1e-9)


def test_pose3_levenberg_marquardt_through_rows_kernel():
    orc, dev, c = T.build_pair(O.POSE3, 300, seed=9, chunk=13)
    import lm_lockstep
    _, _, slack = lm_lockstep.run(orc, dev, 1e-3, 7)   # the lambda schedule is decided by the same comparisons; two calls past convergence
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    T.states_close(O.POSE3, x0, v0, x1, v1, 1e-9 + 2 * slack)


def test_rows_kernel_matches_block_tridiag_solve_api():
    """gpslam_hip_block_tridiag_solve drives the same hierarchy on caller-supplied blocks: random SPD block-tridiagonal
    system, 1000 blocks of 12, against a dense numpy solve."""
    gp = T.gpu()
    rng = np.random.default_rng(5)
    N, b = 1000, 12
    orc, dev, c = T.build_pair(O.POSE3, N, seed=3)
    J = rng.standard_normal((N, 2 * b, 2 * b))
    D = np.zeros((N, b, b)); Oo = np.zeros((N, b, b))
    for s in range(N):
        D[s] += J[s, :, :b].T @ J[s, :, :b] + 0.5 * np.eye(b)
        if s + 1 < N:
            D[s + 1] += J[s, :, b:].T @ J[s, :, b:]
            Oo[s] = J[s, :, b:].T @ J[s, :, :b]           # H[s+1, s]
    g = rng.standard_normal((N, b))
    x = dev.block_tridiag_solve(D, Oo, g)
    # residual of the block-tridiagonal system
    r = np.einsum('sij,sj->si', D, x) - g
    r[1:] += np.einsum('sij,sj->si', Oo[:-1], x[:-1])
    r[:-1] += np.einsum('sji,sj->si', Oo[:-1], x[1:])
    assert np.abs(r).max() <= 1e-9 * np.abs(g).max() * 1e3


def test_fused_level0_damped_step_matches_unfused_trial():
    """A damped Gauss-Newton step two ways on a forced-sharded single segment: iterate(lambda) goes through
    k_fused_level0 (the assembly wave adds lambda to the diagonals it forms), an accepted first Levenberg-Marquardt
    trial with the same lambda goes through k_assemble_ghost + k_chunk_forward_rows (lambda added to the staged
    records).  Same states afterwards."""
    import torch
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose3_chain(700)
    lam = 1e-2
    out = []
    for use_lm in (False, True):
        s = gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=0, rank=0, nranks=1, force_sharded=True)
        s.set_stream(torch.cuda.current_stream().cuda_stream)
        sharded.apply_local(sharded.local_problem(problem, 0, 1), s)
        send, recv = sharded.device_tensors(s)
        sv = sharded.ShardedSolver(s, send, recv, 0, 1, dist=None)
        if use_lm:
            st, lam_new = sv.iterate_lm(lam)
            assert st["accepted"] and abs(lam_new - lam / 10.0) <= 1e-15      # the first trial was taken
        else:
            sv.iterate(lam)
        torch.cuda.synchronize()
        out.append(s.get_states())
    (p0, v0), (p1, v1) = out
    assert np.abs(p0 - p1).max() <= 1e-9 * max(1.0, np.abs(p0).max())
    assert np.abs(v0 - v1).max() <= 1e-9 * max(1.0, np.abs(v0).max())


def test_fused_level0_with_measurement_rows_and_ragged_row_counts():
    """Pose3 + GPInterpolatedGPSFactorPose3 at an irregular rate + scattered pose / velocity priors: the states of a
    chunk have different numbers of full-width and compact rows, which the assembly wave of k_fused_level0 has to walk
    in lock-step over four chunks.  Gauss-Newton against the oracle."""
    gp = T.gpu()
    from gpslam_amd import synthetic as S
    N = 420
    p = S.pose3_gps_chain(N, per_interval=3, seed=4)
    rng = np.random.default_rng(11)
    keep = rng.random(len(p["gps_left"])) < 0.6                 # 0 .. 3 GPS factors per interval
    for k in ("gps_left", "gps_meas", "gps_sigma", "gps_dt", "gps_tau"):
        p[k] = p[k][keep]
    fix = np.sort(rng.choice(N, 25, replace=False)).astype(np.int32)
    p.update(prior_idx=fix, prior_pose=p["pose"][fix].copy(), prior_sig=np.full((len(fix), 6), 0.05))
    vfix = np.sort(rng.choice(N, 17, replace=False)).astype(np.int32)
    p.update(vprior_idx=vfix, vprior=p["vel"][vfix].copy(), vprior_sig=np.full((len(vfix), 6), 0.1))
    for chunk in (0, 7, -7):
        q = dict(p)
        if chunk < 0:
            # (round 4) some intervals lose their GP prior but keep their GPS factors: the measurement kernel reads the prior's
            # record where there is one and forms Jr^-1, Jr^-1 Ad and the finite-difference block itself where there is none
            # (odometry on every interval keeps those states determined)
            full = S.pose3_gps_chain(N, per_interval=3, seed=4, keep_odometry=True)
            q.update({k: full[k] for k in ("between_left", "between_meas", "between_sig")})
            gone = (np.arange(len(q["gp_left"])) % 7) == 3      # isolated gaps: every state keeps a GP prior on one side
            q["gp_left"], q["gp_dt"] = q["gp_left"][~gone], q["gp_dt"][~gone]
        orc = S.apply(q, O.Chain(O.POSE3))
        dev = S.apply(q, gp.ChainSolver(O.POSE3, chunk=abs(chunk)))
        if chunk < 0:
            assert dev.plan_info()["structured_gp"] == 1
        for it in range(4):
            rc0, s0 = orc.iterate_gn()
            rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0
            assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, abs(s0.error_after)), (chunk, it)
        (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
        T.states_close(O.POSE3, x0, v0, x1, v1, 1e-9)


@pytest.mark.parametrize("variant", ["sparse-between", "double-between", "dense-priors", "gaps-in-gp"])
def test_structured_path_with_irregular_factor_sets(variant):
    """Round 4: the structured path of k_fused_level0 takes the GP priors as 80-double records and the between factors as
    48-double records, one per left state, with an all-zero record for the states that have none.  Irregular graphs:
    between factors on a third of the states only; two between factors on some states (the records do not apply: compact rows
    through the one-slot ring); pose priors on every state (the ring busy on every block step); GP priors missing on some
    intervals that odometry still bridges.  Each against the oracle, 3 Gauss-Newton iterations."""
    N, kind, d = 700, O.POSE3, 6
    rng = np.random.default_rng(31)
    c = T.random_chain(kind, N, 12)
    Qc = np.diag(0.01 + 0.02 * rng.random(d))
    ident = O.pose3((0, 0, 0), (0, 0, 0))
    meas = np.stack([O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][i + 1])) for i in range(N - 1)])
    left_b = np.arange(N - 1)
    left_gp = np.arange(N - 1)
    fix = np.arange(0, N, 20)
    if variant == "sparse-between":
        left_b = np.sort(rng.choice(N - 1, (N - 1) // 3, replace=False))
    elif variant == "double-between":
        left_b = np.concatenate([np.arange(N - 1), np.arange(5, N - 1, 50)])
    elif variant == "dense-priors":
        fix = np.arange(N)
    elif variant == "gaps-in-gp":
        left_gp = np.array([i for i in range(N - 1) if i % 37 != 5])
    solvers = []
    for make in (lambda: O.Chain(kind, O.CHART_EXPMAP), lambda: T.gpu().ChainSolver(kind)):
        s = make()
        s.set_qc(Qc)
        s.set_states(c["pose"], c["vel"])
        s.add_gp_priors(left_gp, c["dt"][left_gp])
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
        s.add_between(left_b, meas[left_b], np.full((len(left_b), d), 0.02))
        s.compile()
        solvers.append(s)
    orc, dev = solvers
    info = dev.plan_info()
    assert info["fused"] == 1 and info["structured_gp"] == 1
    for _ in range(3):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, abs(s0.error_after)), variant
    (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
    T.states_close(kind, x0, v0, x1, v1, 1e-9)
