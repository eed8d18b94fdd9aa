"""CPU: the indexing rules of the LDS-resident cyclic-reduction levels (upper.hip) in numpy against a dense solve."""
import numpy as np
import pytest

import upper_model as M


@pytest.mark.parametrize("n,B,m0", [(9, 4, 3), (40, 6, 3), (200, 6, 4), (700, 4, 5), (1500, 6, 1), (3000, 4, 2), (1030, 12, 1), (33 * 32 + 1, 4, 1)])
@pytest.mark.parametrize("tail", [False, True])
def test_hierarchy_matches_the_dense_solve(n, B, m0, tail):
    D, O, g = M.random_chain(n, B, seed=n + B)
    x = M.solve_chain(D, O, g, m0, tail=tail)
    ref = M.dense_solve(D, O, g) if n * B <= 9000 else None
    if ref is None:
        # too large for a dense factorisation: check the residual of the block-tridiagonal system instead
        r = np.einsum("nij,nj->ni", D, x) - g
        r[:-1] += np.einsum("nji,nj->ni", O[:-1], x[1:])
        r[1:] += np.einsum("nij,nj->ni", O[:-1], x[:-1])
        assert np.abs(r).max() <= 1e-9 * max(1.0, np.abs(g).max())
    else:
        assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_one_group_with_a_block_beyond_it():
    """the sharded form: a level keeps its first block and owes the block beyond it (ext) an addend."""
    n, B = 21, 4
    D, O, g = M.random_chain(n + 1, B, seed=5)
    ref = M.dense_solve(D, O, g)
    # the level = blocks 0 .. n-1; block n lives "on the next rank"
    add = (np.zeros((n + 1, B, B)), np.zeros((n + 1, B)))
    rec, (uD, uO, ug), (aD, ag) = M.multi_forward((D[:n], O[:n], g[:n]), add, top=False, ext=True)
    # reduced 2 x 2 system of block 0 and block n
    H = np.block([[uD[0], uO[0].T], [uO[0], D[n] + aD[1]]])
    xr = np.linalg.solve(H, np.concatenate([ug[0], g[n] + ag[1]]))
    x = M.multi_backward(rec, np.stack([xr[:B], xr[B:]]), n, B, ext=True)
    assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
