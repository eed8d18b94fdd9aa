"""The fp32 mode (GPSLAM_FP32): fp32 Jacobian rows, normal equations and solver, fp64 states and fp64 residual.
north_star: final state vector within 1e-5 relative of the fp64 reference -- here the CPU oracle (fp64)."""
import numpy as np
import pytest

from oracle import oracle as O
from gpslam_amd import synthetic as S
from test_gpu_parity import gpu, build_pair, random_chain

pytestmark = pytest.mark.gpu

KW = {O.POSE2: dict(chart=1)}


def _converge(s, iters):
    hist = []
    for _ in range(iters):
        rc, st = s.iterate_gn()
        assert rc == 0
        hist.append(st.delta_inf_norm)
    return hist


def _rel_state_diff(kind, a, b):
    (xa, va), (xb, vb) = a.get_states(), b.get_states()
    if kind in (O.LINEAR2, O.LINEAR3):
        dp = np.abs(xa - xb).max()
    else:
        step = max(1, len(xa) // 500)
        dp = max(float(np.abs(O.local(kind, xa[i], xb[i])).max()) for i in range(0, len(xa), step))
    scale = max(1.0, float(np.abs(xa).max()), float(np.abs(va).max()))
    return max(dp, float(np.abs(va - vb).max())) / scale


@pytest.mark.parametrize("kind", [O.LINEAR3, O.POSE2, O.ROT3, O.POSE3], ids=["linear3", "pose2", "rot3", "pose3"])
def test_fp32_converges_to_the_fp64_fixed_point(kind):
    """GP priors + pose priors + between factors on a random chain: the fp32 handle's Gauss-Newton iterates converge to
    the oracle's fp64 fixed point within 1e-5 relative (measured: 1e-7 .. 1e-6)."""
    gp = gpu()
    orc, dev64, c = build_pair(kind, 257, seed=41 + kind)
    chart = O.CHART_FIRST_ORDER if kind == O.POSE2 else O.CHART_EXPMAP
    rng = np.random.default_rng(41 + kind + 77)
    d = O.TANGENT_DIM[kind]
    Qc = np.diag(0.01 + 0.02 * rng.random(d))
    if d > 1:
        Qc[0, 1] = Qc[1, 0] = 0.003
    dev32 = gp.ChainSolver(kind, chart, precision=gp.FP32)
    dev32.set_qc(Qc)
    dev32.set_states(c["pose"], c["vel"])
    N = 257
    dev32.add_gp_priors(np.arange(N - 1), c["dt"])
    fix = np.arange(0, N, 20)
    dev32.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
    dev32.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, d), 0.05))
    meas = []
    for i in range(N - 1):
        if kind in (O.LINEAR2, O.LINEAR3):
            meas.append(c["truth_pose"][i + 1] - c["truth_pose"][i])
        else:
            ident = {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0)}[kind]
            meas.append(O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][i + 1])))
    dev32.add_between(np.arange(N - 1), np.stack(meas), np.full((N - 1, d), 0.02))
    dev32.compile()
    assert abs(dev32.error() - orc.error()) <= 1e-10 * orc.error()        # the error of an fp32 handle IS an fp64 error
    for _ in range(12):
        orc.iterate_gn()
    h = _converge(dev32, 12)
    rel = _rel_state_diff(kind, orc, dev32)
    print("fp32 vs fp64 oracle, kind %d: relative state difference %.3e, |delta| history %s" % (kind, rel, ["%.1e" % x for x in h]))
    assert rel <= 1e-5, rel
    assert abs(dev32.error() - orc.error()) <= 1e-7 * max(1.0, orc.error())


def _lm_converge(s, iters):
    lam, hist = 1e-5, []
    for _ in range(iters):
        rc, st, lam = s.iterate_lm(lam)[:3]
        assert rc == 0
        hist.append((st.error_after, st.delta_inf_norm))
    return hist


def test_fp32_c3_mix_matches_the_oracle():
    """BASELINE config 3's factor mix (SE(3) GP priors + BetweenFactor<Pose3> odometry, one prior, dead-reckoned start) at
    a size the oracle solves in seconds.  The chain is anchored by a single prior, so its weakest modes amplify the 6e-8
    relative rounding of the stored fp32 rows: the update does not fall below ~1e-4 (fp64: 1e-11), and the state agrees
    with the fp64 fixed point to 1e-5 of its scale (north_star's fp32 tolerance), not better."""
    gp = gpu()
    p = S.pose3_chain(2000)
    orc = S.apply(p, O.Chain(O.POSE3))
    dev = S.apply(p, gp.ChainSolver(O.POSE3, precision=gp.FP32))
    for _ in range(8):
        orc.iterate_gn()
    h = _converge(dev, 10)
    rel = _rel_state_diff(O.POSE3, orc, dev)
    print("C3 mix fp32: relative state difference %.3e, |delta| %s" % (rel, ["%.1e" % x for x in h]))
    assert rel <= 1e-5, rel
    assert abs(dev.error() - orc.error()) <= 1e-9 * orc.error()


def test_fp32_c5_mix_matches_the_oracle():
    """BASELINE config 5's factor mixes: SO(3) GP priors + GPInterpolatedAttitudeFactorRot3 at 4x the state rate (the
    reference-faithful variant; Gauss-Newton does not converge on it from this start in fp64 either, so both sides run
    Levenberg-Marquardt) and the SE(3) variant with GPInterpolatedGPSFactorPose3 (Levenberg-Marquardt as well: position fixes alone leave the
    attitude weakly observable)."""
    gp = gpu()
    p = S.rot3_attitude_chain(600)
    orc = S.apply(p, O.Chain(O.ROT3))
    dev = S.apply(p, gp.ChainSolver(O.ROT3, precision=gp.FP32))
    h0, h1 = _lm_converge(orc, 25), _lm_converge(dev, 25)
    rel = _rel_state_diff(O.ROT3, orc, dev)
    print("C5 rot3 mix fp32 (LM): relative state difference %.3e, errors %.9e / %.9e, last |delta| %.1e / %.1e" % (rel, h0[-1][0], h1[-1][0], h0[-1][1], h1[-1][1]))
    assert rel <= 1e-5, rel
    p = S.pose3_gps_chain(1200)
    orc = S.apply(p, O.Chain(O.POSE3))
    dev = S.apply(p, gp.ChainSolver(O.POSE3, precision=gp.FP32))
    h0, h1 = _lm_converge(orc, 20), _lm_converge(dev, 20)
    rel = _rel_state_diff(O.POSE3, orc, dev)
    print("C5 pose3+gps mix fp32 (LM): relative state difference %.3e, errors %.9e / %.9e, last |delta| %.1e / %.1e" % (rel, h0[-1][0], h1[-1][0], h0[-1][1], h1[-1][1]))
    assert rel <= 1e-5, rel


def test_fp32_2dlinear_factors_with_nonzero_headings():
    """OdometryFactor2DLinear / RangeBearingFactor2DLinear evaluate their Jacobians AT theta; the fp32 re-centring
    must leave the heading alone (it shifts x, y only).  A LINEAR3 chain with headings far from zero: the fp32 handle's
    Gauss-Newton must reach the oracle's fp64 fixed point."""
    from test_gpu_measurements import build_meas_pair
    gp = gpu()
    orc, dev64, c, (dev32,) = build_meas_pair(O.LINEAR3, extra_makers=(lambda: gp.ChainSolver(O.LINEAR3, O.CHART_EXPMAP, 2, precision=gp.FP32),))
    assert np.abs(c["truth_pose"][:, 2]).max() > 0.05         # the headings really are away from zero
    assert abs(dev32.error() - orc.error()) <= 1e-10 * max(1.0, orc.error())
    for _ in range(12):
        orc.iterate_gn()
    h = _converge(dev32, 12)
    rel = _rel_state_diff(O.LINEAR3, orc, dev32)
    print("fp32 2D-linear factors: relative state difference %.3e, |delta| history %s" % (rel, ["%.1e" % x for x in h]))
    assert rel <= 1e-5, rel
