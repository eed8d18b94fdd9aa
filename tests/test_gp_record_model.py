"""The structured GP-prior record (round 4): the lane-by-lane reconstruction of the whitened Jacobian columns that the assembly
wave of k_fused_level0 performs, modelled in numpy with the kernel's own index expressions, against the definition."""
import numpy as np
import pytest

from gp_record_model import lane_columns, make_record, reference_rows


def _blocks(rng):
    def bl(equal_diag):
        A, C, D = rng.standard_normal((3, 3)), rng.standard_normal((3, 3)), rng.standard_normal((3, 3))
        return np.block([[A, np.zeros((3, 3))], [C, A if equal_diag else D]])
    return bl(True), bl(True), bl(False)


@pytest.mark.parametrize("seed", range(4))
def test_lane_columns_equal_the_whitened_jacobian(seed):
    rng = np.random.default_rng(100 + seed)
    X, J, F = _blocks(rng)
    U = np.triu(rng.standard_normal((6, 6))) + 3 * np.eye(6)
    dt = 0.05 + rng.random()
    ew = rng.standard_normal(12)
    L, R, new = lane_columns(make_record(X, J, F, ew, dt), U)
    L0, R0 = reference_rows(X, J, F, U, dt)
    assert np.abs(L - L0).max() <= 1e-12 * np.abs(L0).max()
    assert np.abs(R - R0).max() <= 1e-12 * np.abs(R0).max()
    assert np.array_equal(new, -ew)


def test_the_zero_record_contributes_nothing():
    L, R, new = lane_columns(np.zeros(80), np.triu(np.ones((6, 6))))
    assert not L.any() and not R.any() and not new.any()


def test_between_record_columns_equal_the_weighted_jacobians():
    from gp_record_model import between_lane_columns, make_between_record
    rng = np.random.default_rng(7)
    for _ in range(3):
        def bl():
            A, C = rng.standard_normal((3, 3)), rng.standard_normal((3, 3))
            return np.block([[A, np.zeros((3, 3))], [C, A]])
        H1, H2 = bl(), bl()
        w = 1.0 / (0.01 + rng.random(6))
        L, R = between_lane_columns(make_between_record(H1, H2, w, rng.standard_normal(6)))
        assert np.array_equal(L, w[:, None] * H1) and np.array_equal(R, w[:, None] * H2)
