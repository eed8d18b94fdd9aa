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


@pytest.mark.parametrize("seed", range(3))
def test_diagonal_qc_zero_pattern_the_short_assembly_wave_relies_on(seed):
    """k_fused_level0<1, double, 12, true> (a diagonal chol(Qc^-1)) skips products it knows to be zero: row q of the whitened L
    meets, among the six velocity columns, only column 6 + q mod 6, and the ROTATION rows (q mod 6 < 3) of [L | R] are zero in every
    translation column (3..5 of L; 3..5 and 9..11 of R) because X, J and F are block lower triangular.  The zeros must be EXACT
    zeros of the lane arithmetic (the kernel's claim is "the same values"), and what it keeps must be the rest."""
    rng = np.random.default_rng(300 + seed)
    X, J, F = _blocks(rng)
    U = np.diag(0.5 + rng.random(6))
    dt = 0.05 + rng.random()
    L, R, _ = lane_columns(make_record(X, J, F, rng.standard_normal(12), dt), U)      # L[q][c], R[q][c]: row q, column (lane) c
    L0, R0 = reference_rows(X, J, F, U, dt)
    assert np.abs(L - L0).max() <= 1e-12 * np.abs(L0).max() and np.abs(R - R0).max() <= 1e-12 * np.abs(R0).max()
    keepL = np.zeros((12, 12), bool)
    keepR = np.zeros((12, 12), bool)
    for q in range(12):
        rot = (q % 6) < 3
        keepL[q, :3 if rot else 6] = True                     # pose columns: the rotation columns, and for a translation row all six
        keepL[q, 6 + q % 6] = True                            # its own velocity column
        keepR[q, :] = True
        if rot:
            keepR[q, 3:6] = False
            keepR[q, 9:12] = False
    assert np.all(L[~keepL] == 0.0) and np.all(R[~keepR] == 0.0)
    assert np.all(L[keepL] != 0.0) and np.all(R[keepR] != 0.0)


def test_between_record_rotation_rows_are_zero_in_the_translation_columns():
    """[[A, 0], [C, A]] blocks: rows 0..2 of the whitened [H1 | H2] have nothing in columns 3..5 (the assembly wave gathers three
    lanes for them instead of six)."""
    from gp_record_model import between_lane_columns, make_between_record
    rng = np.random.default_rng(9)
    def bl():
        A, C = rng.standard_normal((3, 3)), rng.standard_normal((3, 3))
        return np.block([[A, np.zeros((3, 3))], [C, A]])
    L, R = between_lane_columns(make_between_record(bl(), bl(), 1.0 / (0.01 + rng.random(6)), rng.standard_normal(6)))
    assert np.all(L[:3, 3:6] == 0.0) and np.all(R[:3, 3:6] == 0.0)
    assert np.all(L[:3, :3] != 0.0) and np.all(L[3:, :6] != 0.0)


@pytest.mark.parametrize("diag", [False, True])
def test_d3_record_rows_equal_the_whitened_jacobian(diag):
    """SE(2), SO(3) and 3-D linear chains (round 4): the 32-double record holds A1 = U sa J1, A3 = U sa J3, the whitened error and four
    coefficients; its consumers rebuild the six whitened rows [L | R] lane by lane.  Against R_w [H1 H2 | H3 H4] with the constants of
    GaussianProcessPriorLinear (h2t = -dt, h2b = -1, h4b = 1) and of the Lie-group priors (J1, J3 dense, the same constants)."""
    from gp_record_model import make_record3, reference_rows3, rows_from_record3
    rng = np.random.default_rng(17)
    for trial in range(4):
        dt = 0.05 + rng.random()
        U = np.diag(0.5 + rng.random(3)) if diag else np.triu(rng.standard_normal((3, 3))) + 2 * np.eye(3)
        if trial % 2:
            J1, J3 = -np.eye(3), np.eye(3)                              # GaussianProcessPriorLinear.h:72-81
        else:
            J1, J3 = rng.standard_normal((3, 3)), rng.standard_normal((3, 3))   # -Jr^-1 Ad, Jr^-1 of the Lie-group priors
        e = rng.standard_normal(6)
        rec = make_record3(J1, J3, -dt, -1.0, 1.0, U, e, dt)
        L, R, ew = rows_from_record3(rec, U)
        L0, R0, e0 = reference_rows3(J1, J3, -dt, -1.0, 1.0, U, e, dt)
        assert np.abs(L - L0).max() <= 1e-12 * np.abs(L0).max()
        assert np.abs(R - R0).max() <= 1e-12 * np.abs(R0).max()
        assert np.abs(ew - e0).max() <= 1e-12 * np.abs(e0).max()
        assert np.all(L[3:, :3] == 0.0) and np.all(R[3:, :3] == 0.0)    # rows 3..5 have no pose part
