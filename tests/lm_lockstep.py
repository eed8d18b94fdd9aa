"""Two Levenberg-Marquardt optimisers in lock step -- the comparison rule of every LM parity test (round 5).

`LevenbergMarquardtOptimizer::iterate()` keeps a step when  rho = (err - newErr) / (linErr(0) - linErr(delta))  exceeds
minModelFidelity.  While the cost moves, rho is a number both sides agree on to many digits and the lambda schedule, the
accept flags and the trial counts must be IDENTICAL.  At a converged point both the numerator and the denominator are
rounding errors of sums with ~1e5 terms: the sign of rho then depends on the order of summation, on either side, and is
not a property of the algorithm (round 4's red test: the oracle's cost moved by 9.5e-14 relative).  What IS a property
there is GTSAM's small-cost-change stop: the call ends after ONE trial, lambda is either kept (step not kept) or divided by
the factor (step kept), and the values move by no more than a rounding-sized step.  `step` checks exactly that, then puts
the two lambdas back together so that a test can keep going past convergence.
"""
NOISE = 1e-10          # relative cost change below which the fidelity ratio is noise; relativeErrorTol is 1e-5


def _unpack(out):
    """(stats, lambda) of ChainSolver / oracle.Chain iterate_lm -> (rc, Stats, lambda); ShardedSolver -> (dict, lambda)"""
    if len(out) >= 3:
        assert out[0] == 0, out[0]
        st, lam = out[1], out[2]
        return dict(accepted=int(st.accepted), trials=int(st.trials), error_before=st.error_before, error_after=st.error_after,
                    last_trial_error=st.last_trial_error, delta_inf_norm=st.delta_inf_norm), lam
    st, lam = out
    return dict(accepted=int(bool(st["accepted"])), trials=int(st["trials"]), error_before=st["error_before"],
                error_after=st["error_after"], last_trial_error=st["last_trial_error"], delta_inf_norm=st["delta_inf_norm"]), lam


def step(ref, dev, lam, err_tol=1e-9, lambda_factor=10.0, tag=None, ref_kwargs=None, dev_kwargs=None, carry=0.0):
    """One iterate_lm on both sides from the same lambda.  Returns (stats_ref, stats_dev, lambda to go on with, decided_by_noise).
    carry: error_before of the PREVIOUS call.  A step that takes the cost down by three orders of magnitude (4.0e6 -> 3.9e3 from dead
    reckoning on a range-only landmark graph: scripts/stress_mixes.py 36 102, mix 2 at N = 333) leaves the new cost with the rounding
    of the larger number: the oracle ALONE moves by 5e-10 of it when its input states are perturbed by 1e-15.  This call's
    error_before is that number; it is compared at err_tol of itself + 1e-11 of the cost the previous call started from -- the
    allowance the Gauss-Newton comparisons of the stress scripts have always had."""
    call = lambda x, kw: x.iterate_lm(lam, **(kw or {})) if hasattr(x, "iterate_lm") else x(lam)    # an optimiser, or a function of lambda
    s0, lam0 = _unpack(call(ref, ref_kwargs))
    s1, lam1 = _unpack(call(dev, dev_kwargs))
    scale = max(1.0, abs(s0["error_before"]))
    assert abs(s0["error_before"] - s1["error_before"]) <= err_tol * scale + 1e-11 * abs(carry), (tag, s0, s1, carry)
    moved0 = abs(s0["error_before"] - s0["last_trial_error"])
    moved1 = abs(s1["error_before"] - s1["last_trial_error"])
    noise = min(moved0, moved1) <= NOISE * scale and s0["trials"] == 1 and s1["trials"] == 1
    if not noise:
        assert (s0["accepted"], s0["trials"], lam0) == (s1["accepted"], s1["trials"], lam1), (tag, s0, lam0, s1, lam1)
        assert abs(s0["error_after"] - s1["error_after"]) <= err_tol * scale, (tag, s0, s1)
        return s0, s1, lam0, False
    # converged: one trial each, lambda kept or divided once, the error where it was
    for s, l in ((s0, lam0), (s1, lam1)):
        assert s["trials"] == 1, (tag, s)
        assert l == (lam / lambda_factor if s["accepted"] else lam), (tag, s, lam, l)
        assert abs(s["error_after"] - s0["error_before"]) <= max(err_tol, 10 * NOISE) * scale, (tag, s)
    return s0, s1, lam0, True


def run(ref, dev, lam, iters, **kw):
    """`iters` calls in lock step; returns (final lambda, number of calls whose decision was rounding noise, slack).
    slack = the sum of |delta|_inf over the steps either side KEPT on a noise decision: where one side keeps such a step and the
    other does not, their values part by exactly that much (a Newton step at the optimum, i.e. the solver's own rounding floor),
    so a state comparison after the run is  |a - b| <= tolerance * scale + slack."""
    n_noise, slack, carry = 0, 0.0, 0.0
    for it in range(iters):
        kw_it = dict(kw)
        kw_it["tag"] = (kw.get("tag"), it)
        s0, s1, lam, noise = step(ref, dev, lam, carry=carry, **kw_it)
        carry = s0["error_before"]
        if noise:
            n_noise += 1
            slack += sum(s["delta_inf_norm"] for s in (s0, s1) if s["accepted"])
    return lam, n_noise, slack
