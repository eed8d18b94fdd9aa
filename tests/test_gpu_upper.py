"""GPU: the solver levels above level 0 as LDS-resident cyclic reduction (gpslam_amd/csrc/upper.hip) through the C ABI.

gpslam_hip_block_tridiag_solve drives the whole hierarchy on caller-supplied blocks; the level-0 chunk length is chosen so
that the level above it has 1 .. 1250 blocks: one group solved in place (TOP), one reduction launch + TOP, two + TOP,
ragged last groups, a last group of a single block.  Oracle: the CPU restatement's sequential block elimination."""
import numpy as np
import pytest

from oracle import oracle as O
from test_gpu_parity import gpu, ident_states, random_block_tridiag, build_pair

pytestmark = pytest.mark.gpu

CASES = [(O.POSE3, 65, 2), (O.POSE3, 66, 2), (O.POSE3, 67, 2), (O.POSE3, 333, 2), (O.POSE3, 2 * 32 * 32 + 3, 2), (O.POSE3, 2500, 2),
         (O.POSE3, 5000, 0), (O.POSE2, 131, 2), (O.POSE2, 2 * 33 + 1, 2), (O.POSE2, 3 * 1100, 3), (O.LINEAR3, 777, 2),
         (O.LINEAR2, 40, 2), (O.LINEAR2, 2 * 32 * 32 * 2 + 1, 2), (O.ROT3, 4000, 4)]


@pytest.mark.parametrize("kind,N,chunk", CASES)
def test_cyclic_reduction_levels_match_oracle(kind, N, chunk):
    b = 2 * O.TANGENT_DIM[kind]
    D, Ocp, g = random_block_tridiag(N, b, seed=N + b + chunk)
    s = gpu().ChainSolver(kind, chunk=chunk)
    s.set_states(*ident_states(kind, N))
    s.compile()
    x1 = s.block_tridiag_solve(D, Ocp, g)
    x0 = O.block_tridiag_solve(D, Ocp, g)
    assert np.abs(x0 - x1).max() <= 1e-10 * max(1.0, np.abs(x0).max())
    r = np.einsum("nij,nj->ni", D, x1) - g
    r[:-1] += np.einsum("nji,nj->ni", Ocp[:-1], x1[1:])
    r[1:] += np.einsum("nij,nj->ni", Ocp[:-1], x1[:-1])
    assert np.abs(r).max() <= 1e-9 * max(1.0, np.abs(g).max())
    x2 = s.block_tridiag_solve(D, Ocp, g)
    assert np.array_equal(x1, x2)            # fixed summation order: bit-identical from run to run


def test_indefinite_block_in_an_upper_level_is_reported():
    """a non-positive pivot that only appears in a Schur complement above level 0 must raise the NOT_SPD status"""
    kind, N, b = O.POSE2, 400, 6
    D, Ocp, g = random_block_tridiag(N, b, seed=3)
    D[0] = -np.eye(b)                        # state 0 is a separator of every level: it is factored by the top launch
    s = gpu().ChainSolver(kind, chunk=2)
    s.set_states(*ident_states(kind, N))
    s.compile()
    with pytest.raises(gpu().GpslamHipError):
        s.block_tridiag_solve(D, Ocp, g)


@pytest.mark.parametrize("kind", [O.POSE3, O.POSE2, O.LINEAR3])
def test_gauss_newton_through_the_new_levels(kind):
    """a whole optimisation whose hierarchy has a reduction launch and a top launch, in lock step with the oracle"""
    orc, dev, _ = build_pair(kind, 1500, seed=77 + kind)
    for _ in range(4):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-6 * max(1.0, abs(s0.error_after))
    for _ in range(4):
        orc.iterate_gn(); dev.iterate_gn()
    p0, v0 = orc.get_states()
    p1, v1 = dev.get_states()
    assert np.abs(p0 - p1).max() <= 1e-9 * max(1.0, np.abs(p0).max())
    assert np.abs(v0 - v1).max() <= 1e-9 * max(1.0, np.abs(v0).max())
