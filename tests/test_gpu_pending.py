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
