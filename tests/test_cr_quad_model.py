"""The four-row form of the cyclic-reduction elimination (cr_quad.hpp) restated in numpy (tests/cr_quad_model.py) against the
definition: U = D_j^-1 O_j^T, V = D_j^-1 F, Y = D_j^-1 g_j and the three Schur complements -- including the shortened
Gauss-Jordan on D (only the blocks of four columns right of the pivot), for the block sizes the solver has."""
import numpy as np
import pytest

import cr_quad_model as M


@pytest.mark.parametrize("B", [12, 6, 4])
def test_four_row_elimination_equals_the_definition(B):
    rng = np.random.default_rng(B)
    for trial in range(5):
        A = rng.standard_normal((B, B + 3))
        Dj = A @ A.T + B * np.eye(B)
        A = rng.standard_normal((B, B + 3))
        Ds = A @ A.T + B * np.eye(B)
        Oj, Os = rng.standard_normal((B, B)), rng.standard_normal((B, B))
        gj, gs = rng.standard_normal(B), rng.standard_normal(B)
        U, V, Y, Dsn, Fn, gsn, Pn, gn = M.quad_elimination(Dj, Oj, gj, Ds, Os, gs)
        F = Os
        np.testing.assert_allclose(U, np.linalg.solve(Dj, Oj.T), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(V, np.linalg.solve(Dj, F), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(Y, np.linalg.solve(Dj, gj), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(Dsn, Ds - F.T @ np.linalg.solve(Dj, F), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(gsn, gs - F.T @ np.linalg.solve(Dj, gj), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(Fn, -Oj @ np.linalg.solve(Dj, F), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(Pn, -Oj @ np.linalg.solve(Dj, Oj.T), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(gn, -Oj @ np.linalg.solve(Dj, gj), rtol=1e-10, atol=1e-11)
        # the eliminated unknown really is gone: with x_n, x_s given, x_j = Y - U x_n - V x_s solves block row j
        xn, xs = rng.standard_normal(B), rng.standard_normal(B)
        xj = Y - U @ xn - V @ xs
        np.testing.assert_allclose(Dj @ xj + Oj.T @ xn + F @ xs, gj, rtol=1e-9, atol=1e-10)
