"""CPU-only checks of the host side: the C-ABI library builds/loads and exports every symbol of
include/gpslam_hip.h, refuses to run without a GPU (no CPU fallback), and the synthetic generators are
deterministic and well-posed (checked through the oracle)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import gpslam_amd
    from gpslam_amd import chain
    lib = gpslam_amd.load_library()
    with open(os.path.join(ROOT, "include", "gpslam_hip.h")) as f:
        declared = sorted(set(re.findall(r"\b(gpslam_hip_[a-z0-9_]+)\s*\(", f.read())))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(chain.ABI_SYMBOLS) == declared


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import gpslam_amd
    with pytest.raises(gpslam_amd.GpslamHipError):
        gpslam_amd.ChainSolver(gpslam_amd.POSE3)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gpslam_amd/ or include/ may import, link or call it."""
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|liboracle|gpslam_oracle\.h|\borc_[a-z]", re.M)
    for top in ("gpslam_amd", "include"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, top)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    with open(os.path.join(dirpath, fn)) as f:
                        assert not bad.search(f.read()), (dirpath, fn)


def test_synthetic_generators_are_deterministic_and_well_posed():
    from gpslam_amd import synthetic as S
    from oracle import oracle as O
    a, b = S.pose3_chain(300), S.pose3_chain(300)
    assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["between_meas"], b["between_meas"])
    ch = S.apply(a, O.Chain(O.POSE3))
    e0 = ch.error()
    for _ in range(5):
        rc, st = ch.iterate_gn()
        assert rc == 0
    assert st.error_after < e0 and st.delta_inf_norm < 1e-6
    lin = S.apply(S.linear_chain(500), O.Chain(O.LINEAR3))
    rc, s1 = lin.iterate_gn()
    rc, s2 = lin.iterate_gn()
    assert s2.delta_inf_norm < 1e-9


def test_algorithmic_byte_figures_match_the_survey():
    from gpslam_amd import synthetic as S
    ab = S.algorithmic_bytes_per_state(S.POSE3)
    assert ab["linearize"] == 2552            # 144 + 8 + 96 + 2304
    assert S.algorithmic_bytes_per_state(S.POSE2)["linearize"] == 680


def test_lm_decide_follows_trylambda():
    """gpslam_hip_lm_decide (host arithmetic of the C ABI, no GPU): the three exits of GTSAM 4.0's tryLambda -- keep the step,
    stop on a small cost change with lambda untouched, or raise lambda up to the bound -- in GTSAM's order (increase, THEN test the
    bound).  s6 = (error, trial error, |delta|_inf, delta . g, |delta|^2, indefinite flag)."""
    from gpslam_amd.chain import lm_decide
    # a good step: rho = (100 - 60) / (0.5 * 90 + 0.5 * 1e-3 * 4) > 1e-3
    assert lm_decide([100.0, 60.0, 0.5, 90.0, 4.0, 0.0], 1e-3) == (True, True, 1e-4)
    # cost went UP by a lot: not kept, not done, lambda * 10
    acc, done, lam = lm_decide([100.0, 160.0, 0.5, 90.0, 4.0, 0.0], 1e-3)
    assert (acc, done) == (False, False) and abs(lam - 1e-2) < 1e-18
    # ... and gives up when the RAISED lambda reaches the upper bound (the lambda that was just tried is below it)
    acc, done, lam = lm_decide([100.0, 160.0, 0.5, 90.0, 4.0, 0.0], 1e4)
    assert (acc, done) == (False, True) and lam == 1e5
    # converged: the cost moves by rounding only (|change| < 1e-5 * error).  Negative fidelity: not kept, lambda UNTOUCHED, done
    assert lm_decide([100.0, 100.0 + 1e-11, 1e-9, 1e-13, 1e-18, 0.0], 1e-3) == (False, True, 1e-3)
    # ... positive fidelity on the same noise: kept (GTSAM keeps a successful step whatever its size), lambda / 10
    assert lm_decide([100.0, 100.0 - 1e-11, 1e-9, 1e-13, 1e-18, 0.0], 1e-3) == (True, True, 1e-4)
    # the linear model did not decrease (<= 1e-20): never successful; small change -> stop
    assert lm_decide([100.0, 100.0, 0.0, 0.0, 0.0, 0.0], 1e-3) == (False, True, 1e-3)
    # an indefinite damped system: no step to judge, lambda goes up
    acc, done, lam = lm_decide([100.0, 0.0, 0.0, 0.0, 0.0, 1.0], 1e-3)
    assert (acc, done) == (False, False) and abs(lam - 1e-2) < 1e-18
    # lambdaLowerBound holds
    assert lm_decide([100.0, 60.0, 0.5, 90.0, 4.0, 0.0], 1e-3, lambda_lower_bound=5e-4) == (True, True, 5e-4)


def test_oracle_lm_stops_searching_on_a_small_cost_change():
    """The oracle's LevenbergMarquardtOptimizer::iterate past convergence (round 4: without tryLambda's small-cost-change stop it
    climbed 1e-5 -> 1e4 on rounding noise): one trial per call, lambda kept or divided once, the values where they were."""
    from gpslam_amd import synthetic as S
    from oracle import oracle as O
    ch = S.apply(S.linear_chain(400), O.Chain(O.LINEAR3))
    lam, seen = 1e-5, []
    for it in range(7):
        rc, st, new = ch.iterate_lm(lam)[:3]
        assert rc == 0
        moved = abs(st.error_before - st.last_trial_error)
        if moved <= 1e-10 * max(1.0, st.error_before):
            assert st.trials == 1 and new == (lam / 10.0 if st.accepted else lam), (it, st.trials, lam, new)
            seen.append(it)
        lam = new
    assert len(seen) >= 3 and lam <= 1e-5
    # far from the optimum nothing changes: rejected steps still raise lambda (trial counts above one)
    p = S.pose3_chain(60)
    ch = S.apply(p, O.Chain(O.POSE3))
    rc, st, lam2 = ch.iterate_lm(1e-9)[:3]
    assert rc == 0 and st.accepted == 1 and st.trials >= 1 and st.last_trial_error == st.error_after
