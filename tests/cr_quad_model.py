"""numpy model of one cyclic-reduction elimination spread over the four DPP rows of a wave (gpslam_amd/csrc/cr_quad.hpp; test
infrastructure).  Pair (s, j) with right neighbour n:  D_j x_j + O_j^T x_n + F x_s = g_j,  F = O_s.  Every "row" (part, ch) of the
model does what the lanes of that DPP row do: Gauss-Jordan on its own copy of D_j by row operations with the pivot rows left
unscaled until the end -- of D only the columns right of the pivot are touched, in blocks of four --, carrying its half of the
columns of O_j^T (part 0) or of F (part 1) and g_j along, then its share of the three Schur products."""
import numpy as np


def gauss_jordan_unscaled(D, cols):
    """Row operations that reduce D to a diagonal (pivot rows unscaled), applied to the column groups in `cols` as well.
    Of D, pivot k only updates the blocks of four columns that still hold a column right of k (what the kernel does for B = 12);
    entries at or left of the pivot inside such a block become garbage that nothing reads.  Returns the reciprocal pivots."""
    B = D.shape[0]
    D = D.copy()
    inv = np.zeros(B)
    for k in range(B):
        piv = D[k, k]
        assert piv > 0.0
        inv[k] = 1.0 / piv
        m = -D[:, k] * inv[k]
        m[k] = 0.0
        lo = (k // 4) * 4 if B == 12 else 0
        blocks = range(lo, B) if (B != 12 or k % 4 != 3) else range(lo + 4, B)     # a block whose last column is the pivot is done
        cols_d = list(blocks)
        D[:, cols_d] += np.outer(m, D[k, cols_d])
        for X in cols:
            X += np.outer(m, X[k])
    return inv


def quad_elimination(Dj, Oj, gj, Ds, Os, gs):
    """-> U, V, Y, Ds', Os' (= -O_j V), gs', Dn_add (= -O_j U), gn_add (= -O_j Y), assembled from the four rows' shares."""
    B = Dj.shape[0]
    H = B // 2
    F = Os
    U = np.zeros((B, B)); V = np.zeros((B, B)); Pn = np.zeros((B, B)); Fn = np.zeros((B, B)); Dsn = Ds.copy()
    Y = gsn = gn = None
    for part in (0, 1):
        for ch in (0, 1):
            c = slice(ch * H, ch * H + H)
            X = (Oj.T[:, c] if part == 0 else F[:, c]).copy()        # row r of O_j^T / of F, the row's columns
            g = gj.copy().reshape(B, 1)
            inv = gauss_jordan_unscaled(Dj, [X, g])
            X *= inv[:, None]; g *= inv[:, None]
            P = -Oj @ X                                              # -O_j U (part 0), -O_j V (part 1): the row's columns
            if part == 0:
                U[:, c], Pn[:, c] = X, P
                if ch == 0:
                    Y, gn = g[:, 0].copy(), -Oj @ g[:, 0]
            else:
                V[:, c], Fn[:, c] = X, P
                Dsn[:, c] = Ds[:, c] - F.T @ X
                if ch == 0:
                    gsn = gs - F.T @ g[:, 0]
    return U, V, Y, Dsn, Fn, gsn, Pn, gn
