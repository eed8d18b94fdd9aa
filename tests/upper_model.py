"""numpy model of the solver hierarchy: level-0 chunks (kernels.hpp) + the LDS-resident cyclic-reduction levels (upper.hip).

Test infrastructure only.  It mirrors the DATA FLOW of the kernels one to one -- records [D | O | g] with O_i = H[i+1, i],
addends [RD | Rg] per block, groups of G blocks, the virtual block beyond a group, the order in which a block receives
its own Schur complement and its left neighbour's share -- so that the indexing rules of upper.hip (which pair eliminates
what, where the addends travel, partial groups) are checked against a dense solve on the CPU.
"""
import numpy as np

G = 32
Q = 5


def level0(D, O, g, m0):
    """chunks of m0 states, the first state of every chunk a separator.  Returns (records, up_blk, up_add)."""
    n, B = g.shape
    nch = (n + m0 - 1) // m0
    rec = [None] * n
    up_D = np.zeros((nch, B, B)); up_O = np.zeros((nch, B, B)); up_g = np.zeros((nch, B))
    add_D = np.zeros((nch + 1, B, B)); add_g = np.zeros((nch + 1, B))
    for c in range(nch):
        s, e = c * m0, min((c + 1) * m0, n)
        right_exists = e < n
        A, ga = D[s].copy(), g[s].copy()
        F = O[s].copy()
        if s + 1 >= e:                      # chunk without interior
            up_D[c], up_g[c] = A, ga
            up_O[c] = F if right_exists else 0.0
            continue
        Dt, gt = D[s + 1].copy(), g[s + 1].copy()
        for j in range(s + 1, e):
            U = np.linalg.solve(Dt, O[j].T); V = np.linalg.solve(Dt, F); Y = np.linalg.solve(Dt, gt)
            rec[j] = (V, U, Y)
            A -= F.T @ V; ga -= F.T @ Y
            Fn, Dn, gn = -O[j] @ V, -O[j] @ U, -O[j] @ Y
            if j + 1 < e:
                Dt, gt, F = D[j + 1] + Dn, g[j + 1] + gn, Fn
            else:
                if right_exists:
                    add_D[c + 1], add_g[c + 1] = Dn, gn
                up_O[c] = Fn if right_exists else 0.0
        up_D[c], up_g[c] = A, ga
    return rec, (up_D, up_O, up_g), (add_D, add_g)


def level0_back(rec, x1, n, m0, B):
    x = np.zeros((n, B))
    nch = (n + m0 - 1) // m0
    for c in range(nch):
        s, e = c * m0, min((c + 1) * m0, n)
        x[s] = x1[c]
        xr = x1[c + 1] if e < n else np.zeros(B)
        for j in range(e - 1, s, -1):
            V, U, Y = rec[j]
            x[j] = Y - U @ xr - V @ x[s]
            xr = x[j]
    return x


def multi_forward(blk, add, top, ext=False, G=G, Q=Q):
    """one launch of k_multi_forward over a level.  blk = (D, O, g) arrays of n blocks, add = (RD, Rg) of n + 1 entries.
    Returns (factor records, up_blk, up_add) or, for TOP, (records, x)."""
    D, O, g = (a.copy() for a in blk)
    aD, ag = add
    n, B = g.shape
    ngroups = (n + G - 1) // G
    assert not top or ngroups == 1
    rec = [None] * n
    up_D = np.zeros((ngroups, B, B)); up_O = np.zeros((ngroups, B, B)); up_g = np.zeros((ngroups, B))
    up_aD = np.zeros((ngroups + 1, B, B)); up_ag = np.zeros((ngroups + 1, B))
    xs_top = None
    for gi in range(ngroups):
        base = gi * G
        cnt = min(G, n - base)
        RD = np.zeros((G + 1, B, B)); RO = np.zeros((G + 1, B, B)); Rg = np.zeros((G + 1, B))
        for i in range(cnt):
            RD[i], RO[i], Rg[i] = D[base + i], O[base + i], g[base + i]
            if i >= 1 or top:
                RD[i] += aD[base + i]; Rg[i] += ag[base + i]
        xi = base + cnt
        if xi < n or ext:
            RD[G], Rg[G] = aD[xi].copy(), ag[xi].copy()
        fac = {}
        for q in range(Q):
            h, npairs = 1 << q, G >> (q + 1)
            pend = []
            newvals = []
            for p in range(npairs):
                s, j = p * 2 * h, p * 2 * h + h
                if j >= cnt:
                    continue
                nn = j + h if j + h < cnt else G
                F = RO[s]
                U = np.linalg.solve(RD[j], RO[j].T); V = np.linalg.solve(RD[j], F); Y = np.linalg.solve(RD[j], Rg[j])
                newvals.append((s, j, RD[s] - F.T @ V, -RO[j] @ V, Rg[s] - F.T @ Y, (V, U, Y)))
                pend.append((nn, -RO[j] @ U, -RO[j] @ Y))
            for s, j, Ds, Os, gs, f in newvals:       # barrier: the pairs' own blocks
                RD[s], RO[s], Rg[s] = Ds, Os, gs
                fac[j] = f
            for nn, Dn, gn in pend:                   # barrier: the right neighbours' shares
                RD[nn] += Dn; Rg[nn] += gn
        for j, f in fac.items():
            rec[base + j] = f
        if not top:
            up_D[gi], up_O[gi], up_g[gi] = RD[0], RO[0], Rg[0]
            up_aD[gi + 1], up_ag[gi + 1] = RD[G], Rg[G]
        else:
            XS = np.zeros((G + 1, B))
            XS[0] = np.linalg.solve(RD[0], Rg[0])
            group_backward(fac, XS, cnt, G, Q)
            xs_top = XS[:cnt].copy()
    if top:
        return rec, xs_top
    return rec, (up_D, up_O, up_g), (up_aD, up_ag)


def group_backward(fac, XS, cnt, G=G, Q=Q):
    for q in range(Q - 1, -1, -1):
        h, npairs = 1 << q, G >> (q + 1)
        for p in range(npairs):
            s, j = p * 2 * h, p * 2 * h + h
            if j >= cnt:
                continue
            nn = j + h if j + h < cnt else G
            V, U, Y = fac[j]
            XS[j] = Y - U @ XS[nn] - V @ XS[s]


def multi_backward(rec, xup, n, B, ext=False, G=G, Q=Q):
    x = np.zeros((n + 1, B))
    ngroups = (n + G - 1) // G
    for gi in range(ngroups):
        base = gi * G
        cnt = min(G, n - base)
        XS = np.zeros((G + 1, B))
        XS[0] = xup[gi]
        if base + cnt < n or ext:
            XS[G] = xup[gi + 1]
        fac = {j: rec[base + j] for j in range(1, cnt)}
        group_backward(fac, XS, cnt, G, Q)
        x[base:base + cnt] = XS[:cnt]
        if ext and base + cnt == n:
            x[n] = XS[G]
    return x


def solve_chain(D, O, g, m0, tail=False):
    """the whole hierarchy: level 0 in chunks of m0, then groups of G until one group is left (TOP).  tail: level 1 in
    groups of four (what k_fused_level0 folds into its own tail) when it has more than G blocks."""
    n, B = g.shape
    rec0, blk, add = level0(D, O, g, m0)
    levels = []
    first = True
    while blk[2].shape[0] > G:
        gq = (4, 2) if (tail and first) else (G, Q)
        first = False
        rec, ublk, uadd = multi_forward(blk, add, top=False, G=gq[0], Q=gq[1])
        levels.append((rec, blk[2].shape[0], gq))
        blk, add = ublk, uadd
    _, x = multi_forward(blk, add, top=True)
    for rec, nl, gq in reversed(levels):
        x = multi_backward(rec, x, nl, B, G=gq[0], Q=gq[1])[:nl]
    return level0_back(rec0, x, n, m0, B)


def dense_solve(D, O, g):
    n, B = g.shape
    H = np.zeros((n * B, n * B))
    for i in range(n):
        H[i * B:(i + 1) * B, i * B:(i + 1) * B] = D[i]
        if i + 1 < n:
            H[(i + 1) * B:(i + 2) * B, i * B:(i + 1) * B] = O[i]
            H[i * B:(i + 1) * B, (i + 1) * B:(i + 2) * B] = O[i].T
    return np.linalg.solve(H, g.reshape(-1)).reshape(n, B)


def random_chain(n, B, seed):
    """an SPD block-tridiagonal system J^T J + I (O of the last block is zero, as the assembly leaves it)."""
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((n, 2 * B, 2 * B))          # rows of "factor i" on (state i, state i + 1)
    D = np.tile(np.eye(B), (n, 1, 1)) * 0.5
    O = np.zeros((n, B, B))
    for i in range(n):
        L, R = J[i][:, :B], J[i][:, B:]
        D[i] += L.T @ L
        if i + 1 < n:
            O[i] = R.T @ L
            D[i + 1] += R.T @ R
    return D, O, rng.standard_normal((n, B))
