"""Interpolated measurement rows of an SE(3) chain as 16-double lines (round 5; kernels.hpp kIRow*, k_meas<..., IROW>,
k_fused_level0<4>): GPInterpolatedGPSFactorPose3 (gpslam/slam/GPInterpolatedGPSFactorPose3.h:66-95) on a chain whose GP priors
travel as structured records.  A row [Lp | mu | e, p11, p12, l12] stands for the 24 whitened columns
[Lp | l12 mu | mu (p11 X + p12 F X) | p12 mu X] (GaussianProcessInterpolatorPose3.h:82-98); the assembly wave forms the right half
from the interval's record.  Checked against the oracle AND against the plain-row plan of the same library (GPSLAM_PLAN_MEAS_ROWS)."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import states_close

pytestmark = pytest.mark.gpu


def gpu():
    import gpslam_amd
    return gpslam_amd


def gps_graph(N, seed, per_interval=4, sensor=None, cov=False, skip_gp=(), tau_edge=True):
    """A callable that feeds the same SE(3) GP chain + odometry + interpolated GPS graph to any solver-like object."""
    from gpslam_amd import synthetic as S
    p = S.pose3_gps_chain(N, per_interval=per_interval, seed=seed, keep_odometry=True)
    rng = np.random.default_rng(1000 + seed)
    tau = np.array(p["gps_tau"])
    dt = float(p["gps_dt"][0])
    if tau_edge:                     # tau = 0 and tau = dt (l12 = 0 there), and outside [0, dt] (the reference's tests extrapolate)
        tau[0], tau[1], tau[2], tau[3] = 0.0, dt, -0.3 * dt, 1.4 * dt
    sig = 0.03 + 0.04 * rng.random((len(tau), 3))
    covs = None
    if cov:
        A = rng.standard_normal((len(tau), 3, 3)) * 0.02
        covs = A @ np.transpose(A, (0, 2, 1)) + 0.002 * np.eye(3)
    gp_left = np.array([i for i in p["gp_left"] if i not in set(skip_gp)], dtype=np.int32)

    def feed(s):
        s.set_qc(p["qc"])
        s.set_states(p["pose"], p["vel"])
        s.add_pose_priors(p["prior_idx"], p["prior_pose"], p["prior_sig"])
        s.add_between(p["between_left"], p["between_meas"], p["between_sig"])
        s.add_gp_priors(gp_left, np.full(len(gp_left), dt))
        s.add_interp_gps(p["gps_left"], p["gps_meas"], sig, p["gps_dt"], tau, sensor)
        if covs is not None:
            s.set_meas_covariance(3, covs)          # GPSLAM_MEAS_INTERP_GPS
        s.compile()
        return s
    return feed, p


def lockstep_gn(solvers, kind, iters, rel):
    for it in range(iters):
        sts = [s.iterate_gn()[1] for s in solvers]
        for st in sts[1:]:
            assert abs(st.error_before - sts[0].error_before) <= 1e-9 * max(1.0, sts[0].error_before), it
            assert abs(st.error_after - sts[0].error_after) <= 1e-7 * max(1.0, sts[0].error_after), it
    ref = solvers[0].get_states()
    for s in solvers[1:]:
        x, v = s.get_states()
        states_close(kind, ref[0], ref[1], x, v, rel)


@pytest.mark.parametrize("N,per,sensor,cov,generic", [(700, 4, False, False, False), (333, 4, True, False, False), (257, 3, False, True, False),
                                                     (300, 4, True, True, True), (90, 13, False, False, False), (41, 1, False, False, True)],
                         ids=["plain", "sensor", "covariance", "sensor+covariance+generic-Qc", "13-per-interval", "1-per-interval"])
def test_line_form_agrees_with_the_oracle_and_with_the_row_form(N, per, sensor, cov, generic):
    gp = gpu()
    sens = O.pose3((0.3, -0.2, 0.1), (0.2, -0.1, 0.3)) if sensor else None
    feed, p = gps_graph(N, seed=N, per_interval=per, sensor=sens, cov=cov)
    plan = gp.PLAN_GENERIC_QC if generic else 0
    orc = feed(O.Chain(O.POSE3))
    line = feed(gp.ChainSolver(gp.POSE3, plan=plan))
    rows = feed(gp.ChainSolver(gp.POSE3, plan=plan | gp.PLAN_MEAS_ROWS))
    assert line.plan_info()["structured_gp"] == 2 and rows.plan_info()["structured_gp"] == 1     # 2: records + interpolated lines
    assert abs(orc.error() - line.error()) <= 1e-10 * orc.error()
    lockstep_gn([orc, line, rows], O.POSE3, 5, 1e-9)
    for s in (line, rows):
        s.close()


def test_line_form_through_levenberg_marquardt_and_optimize():
    """The damped trials re-assemble from the lines of the linearisation point (k_fused_level0<4> with lambda, the gradient in gsave)."""
    import lm_lockstep
    gp = gpu()
    feed, p = gps_graph(500, seed=3)
    orc, line = feed(O.Chain(O.POSE3)), feed(gp.ChainSolver(gp.POSE3))
    assert line.plan_info()["structured_gp"] == 2
    _, _, slack = lm_lockstep.run(orc, line, 1e-3, 6)
    x0, v0 = orc.get_states()
    x1, v1 = line.get_states()
    states_close(O.POSE3, x0, v0, x1, v1, 1e-9 + 2 * slack)
    for s in (orc, line):
        s.set_states(p["pose"], p["vel"])
    rc0, s0 = orc.optimize()
    rc1, s1 = line.optimize()
    assert s0.iterations == s1.iterations and abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after)
    line.close()


def test_an_interval_without_a_gp_prior_keeps_the_row_form():
    """The line form needs the interval's GP record (X, J, F): a GPS factor on an interval whose GP prior is missing sends the whole
    graph back to 24-column rows -- same answers."""
    gp = gpu()
    feed, p = gps_graph(200, seed=5, skip_gp=(77,))
    orc, dev = feed(O.Chain(O.POSE3)), feed(gp.ChainSolver(gp.POSE3))
    assert dev.plan_info()["structured_gp"] == 1
    lockstep_gn([orc, dev], O.POSE3, 4, 1e-9)
    dev.close()


@pytest.mark.parametrize("P", [2, 3])
def test_line_form_on_a_sharded_chain(P):
    """GPS factors on the intervals that straddle the cuts: the right half of their rows belongs to the next rank's first state and
    travels in the addend the last chunk sends upward, lines or rows alike."""
    import torch
    gp = gpu()
    from gpslam_amd import sharded
    feed, p = gps_graph(301, seed=11)
    ref = feed(gp.ChainSolver(gp.POSE3))
    rec = feed(sharded.GraphRecorder())
    stream = torch.cuda.current_stream().cuda_stream
    ranks = []
    for r in range(P):
        s = gp.ChainSolver(gp.POSE3, device=0, rank=r, nranks=P)
        s.set_stream(stream)
        rec.replay(s, r, P)
        assert s.plan_info()["structured_gp"] == 2
        send, recv = sharded.device_tensors(s)
        ranks.append((s, send, recv))
    for it in range(4):
        for s, _, _ in ranks:
            s.iterate_phase1(0.0)
        for s, send, recv in ranks:
            rv = recv.view(P, -1)
            for k in range(P):
                rv[k].copy_(ranks[k][1])
        sts = [s.iterate_phase2(True) for s, _, _ in ranks]
        _, st = ref.iterate_gn()
        assert abs(sum(x.error_after for x in sts) - st.error_after) <= 1e-7 * max(1.0, st.error_after)
    pose = np.vstack([s.get_states()[0] for s, _, _ in ranks])
    vel = np.vstack([s.get_states()[1] for s, _, _ in ranks])
    x0, v0 = ref.get_states()
    states_close(O.POSE3, x0, v0, pose, vel, 1e-9)
    for s, _, _ in ranks:
        s.close()
    ref.close()


def test_line_form_at_scale_properties():
    """1e5 states, 4e5 interpolated GPS factors (the oracle does not go there in test time): the line form and the row form of the
    same library agree to 1e-9, Gauss-Newton converges and the error never increases."""
    gp = gpu()
    feed, p = gps_graph(100000, seed=2, tau_edge=False)
    line, rows = feed(gp.ChainSolver(gp.POSE3)), feed(gp.ChainSolver(gp.POSE3, plan=gp.PLAN_MEAS_ROWS))
    last = None
    for it in range(6):
        _, a = line.iterate_gn()
        _, b = rows.iterate_gn()
        assert abs(a.error_after - b.error_after) <= 1e-9 * max(1.0, b.error_after)
        assert last is None or a.error_after <= last * (1 + 1e-12)
        last = a.error_after
    assert a.delta_inf_norm < 1e-6
    (x0, v0), (x1, v1) = line.get_states(), rows.get_states()
    states_close(O.POSE3, x0, v0, x1, v1, 1e-9)
    line.close(); rows.close()
