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
