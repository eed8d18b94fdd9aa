"""The retraction folded into the next iteration's linearisation (round 5; kernels.hpp PendUpd, gpslam_hip_run_gn): inside a
fixed-count Gauss-Newton run every iteration but the last leaves `Values::retract` to the K1 launch of the next one, which reads a
state, applies the update, linearises there and writes the state into the other buffer.  Same arithmetic, same order: the states of
run_gn(K) are BIT-IDENTICAL to K single iterations and to the run with a k_retract launch per iteration
(GPSLAM_PLAN_SEPARATE_RETRACT)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def gpu():
    import gpslam_amd
    return gpslam_amd


@pytest.mark.parametrize("N,chart,vprior", [(64, 0, False), (700, 0, False), (1001, 1, False), (333, 0, True), (2, 0, False), (3, 1, True)],
                         ids=["64", "700", "1001-first-order-chart", "333+velocity-prior", "2-states", "3-states"])
def test_run_gn_is_bit_identical_to_single_iterations(N, chart, vprior):
    gp = gpu()
    from gpslam_amd import synthetic as S
    p = S.pose3_chain(N)
    if vprior:
        p = dict(p, vprior_idx=np.array([0], dtype=np.int32), vprior=p["vel"][:1].copy() + 0.01, vprior_sig=np.full((1, 6), 0.05))
    K = 5
    sols = {}
    for name, plan, single in (("folded", 0, False), ("separate", gp.PLAN_SEPARATE_RETRACT, False), ("single", 0, True)):
        s = S.apply(p, gp.ChainSolver(gp.POSE3, chart=chart, plan=plan))
        if single:
            for _ in range(K):
                _, st = s.iterate_gn()
        else:
            st, _ = s.run_gn(K)
        sols[name] = (s.get_states(), st.error_before, st.error_after, st.delta_inf_norm)
        s.close()
    (x0, v0), eb0, ea0, d0 = sols["single"]
    for name in ("folded", "separate"):
        (x, v), eb, ea, d = sols[name]
        assert np.array_equal(x, x0) and np.array_equal(v, v0), name
        assert (eb, ea, d) == (eb0, ea0, d0), (name, eb, ea, d, eb0, ea0, d0)
    # ... and the answer is the oracle's
    orc = S.apply(p, O.Chain(O.POSE3, chart))
    for _ in range(K):
        orc.iterate_gn()
    xo, vo = orc.get_states()
    assert np.abs(xo - x0).max() <= 1e-9 * max(1.0, np.abs(xo).max()) and np.abs(vo - v0).max() <= 1e-9 * max(1.0, np.abs(vo).max())


def _d3_problem(kind, N):
    """Record chains of the d = 3 manifolds without measurement factors: BASELINE config 2 (3-D linear GP chain with position fixes
    and a velocity prior), an SE(2) chain with odometry, an SO(3) chain with attitude priors on a few states."""
    from gpslam_amd import synthetic as S
    if kind == "rot3+interp-attitude":       # BASELINE config 5's SO(3) mix: k_meas runs beside k_lin and applies the update itself
        return S.rot3_attitude_chain(N)
    if kind == O.LINEAR3:
        return S.linear_chain(N)
    if kind == O.POSE2:
        p = S.pose2_range_chain(N, L=4)
        return {k: v for k, v in p.items() if not k.startswith("range_") and not k.startswith("lprior") and not k.startswith("landmark")}
    p = S.rot3_attitude_chain(N)
    q = {k: v for k, v in p.items() if not k.startswith("att_")}
    idx = np.arange(0, N, 7, dtype=np.int32)
    q.update(prior_idx=idx, prior_pose=p["truth"][idx] if "truth" in p else p["pose"][idx], prior_sig=np.full((len(idx), 3), 0.05))
    return q


@pytest.mark.parametrize("kind,N,chart", [(O.LINEAR3, 1000, 0), (O.LINEAR3, 50, 0), (O.POSE2, 700, 0), (O.POSE2, 333, 1), (O.ROT3, 500, 0),
                                          ("rot3+interp-attitude", 600, 0), ("rot3+interp-attitude", 77, 0)],
                         ids=["config2-1000", "config2-50", "pose2+odometry-700", "pose2-first-order-chart", "rot3+attitude-priors",
                              "config5-rot3+interpolated-attitude-600", "config5-rot3+interpolated-attitude-77"])
def test_run_gn_on_the_d3_record_chains_is_bit_identical_to_single_iterations(kind, N, chart):
    """The d = 3 records (kGp3*) take the pending update as the SE(3) records do: K1's GP path reads both states through
    load_state_upd and the left state's owner writes it into the other buffer."""
    gp = gpu()
    from gpslam_amd import synthetic as S
    p = _d3_problem(kind, N)
    kind = p["kind"]
    K = 4
    sols = {}
    for name, plan, single in (("folded", 0, False), ("separate", gp.PLAN_SEPARATE_RETRACT, False), ("single", 0, True)):
        s = S.apply(p, gp.ChainSolver(kind, chart=chart, plan=plan))
        if single:
            for _ in range(K):
                _, st = s.iterate_gn()
        else:
            st, _ = s.run_gn(K)
        sols[name] = (s.get_states(), st.error_before, st.error_after, st.delta_inf_norm)
        s.close()
    (x0, v0), eb0, ea0, d0 = sols["single"]
    for name in ("folded", "separate"):
        (x, v), eb, ea, d = sols[name]
        assert np.array_equal(x, x0) and np.array_equal(v, v0), name
        assert (eb, ea, d) == (eb0, ea0, d0), (name, eb, ea, d, eb0, ea0, d0)
    orc = S.apply(p, O.Chain(kind, chart))
    for _ in range(K):
        orc.iterate_gn()
    xo, vo = orc.get_states()
    assert np.abs(xo - x0).max() <= 1e-9 * max(1.0, np.abs(xo).max()) and np.abs(vo - v0).max() <= 1e-9 * max(1.0, np.abs(vo).max())


@pytest.mark.parametrize("N,per,plan_bits", [(500, 4, 0), (301, 2, "rows")], ids=["gps-lines", "gps-rows"])
def test_run_gn_with_interpolated_gps_factors_is_bit_identical_to_single_iterations(N, per, plan_bits):
    """SE(3) record chain + odometry + GPInterpolatedGPSFactorPose3: k_meas runs behind k_lin on the same stream (it reads the
    interval's record), i.e. behind the buffer swap -- the folded retraction needs nothing from it."""
    gp = gpu()
    from test_gpu_irows import gps_graph
    feed, p = gps_graph(N, seed=N, per_interval=per)
    extra = gp.PLAN_MEAS_ROWS if plan_bits == "rows" else 0
    K = 4
    sols = {}
    for name, plan, single in (("folded", 0, False), ("separate", gp.PLAN_SEPARATE_RETRACT, False), ("single", 0, True)):
        s = feed(gp.ChainSolver(gp.POSE3, plan=plan | extra))
        if single:
            for _ in range(K):
                _, st = s.iterate_gn()
        else:
            st, _ = s.run_gn(K)
        sols[name] = (s.get_states(), st.error_before, st.error_after, st.delta_inf_norm)
        s.close()
    (x0, v0), eb0, ea0, d0 = sols["single"]
    for name in ("folded", "separate"):
        (x, v), eb, ea, d = sols[name]
        assert np.array_equal(x, x0) and np.array_equal(v, v0), name
        assert (eb, ea, d) == (eb0, ea0, d0), (name, eb, ea, d, eb0, ea0, d0)
    orc = feed(O.Chain(O.POSE3))
    for _ in range(K):
        orc.iterate_gn()
    xo, vo = orc.get_states()
    assert np.abs(xo - x0).max() <= 1e-9 * max(1.0, np.abs(xo).max()) and np.abs(vo - v0).max() <= 1e-9 * max(1.0, np.abs(vo).max())


def test_rows_requested_in_the_middle_of_a_run_do_not_lose_the_update():
    """gpslam_hip_get_rows on a d = 3 chain switches K1 back to plain rows (rows3): an update left pending by run_gn must have
    been applied by then -- run_gn never returns with one pending, and the next call linearises at the retracted states."""
    gp = gpu()
    from gpslam_amd import synthetic as S
    p = S.linear_chain(400)
    a = S.apply(p, gp.ChainSolver(O.LINEAR3))
    b = S.apply(p, gp.ChainSolver(O.LINEAR3, plan=gp.PLAN_SEPARATE_RETRACT))
    a.run_gn(2); b.run_gn(2)
    ra, rb = a.get_rows(), b.get_rows()
    for u, v in zip(ra, rb):
        assert (u is None and v is None) or np.array_equal(u, v)
    a.run_gn(2); b.run_gn(2)
    (xa, va), (xb, vb) = a.get_states(), b.get_states()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb)
    a.close(); b.close()


def test_a_chain_with_a_missing_gp_prior_keeps_the_separate_retraction():
    """A state that is nobody's left state has no owner to write it back: compile() leaves the plan alone and the answers stand."""
    gp = gpu()
    from gpslam_amd import synthetic as S
    p = S.pose3_chain(300)
    keep = np.array([i for i in range(299) if i != 120])
    q = dict(p, gp_left=p["gp_left"][keep], gp_dt=p["gp_dt"][keep])
    a = S.apply(q, gp.ChainSolver(gp.POSE3))
    b = S.apply(q, gp.ChainSolver(gp.POSE3))
    a.run_gn(4)
    for _ in range(4):
        b.iterate_gn()
    (xa, va), (xb, vb) = a.get_states(), b.get_states()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb)
    a.close(); b.close()


def test_timed_run_reports_no_retract_phase_for_the_folded_iterations():
    gp = gpu()
    from gpslam_amd import synthetic as S
    p = S.pose3_chain(20000)
    s = S.apply(p, gp.ChainSolver(gp.POSE3))
    s.run_gn(2)
    _, ph = s.run_gn(6, timed=True)
    t = S.apply(p, gp.ChainSolver(gp.POSE3, plan=gp.PLAN_SEPARATE_RETRACT))
    t.run_gn(2)
    _, ph_t = t.run_gn(6, timed=True)
    assert ph[3] < ph_t[3]            # five of six iterations have nothing between the solve and the next linearisation
    s.close(); t.close()
