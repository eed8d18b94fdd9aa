"""numpy model of the segmented landmark elimination with fat separators (test infrastructure).

The HIP library solves chains with many locally visible landmarks (BASELINE config 4) by nested dissection along the
chain: CUT states every C states, each landmark attached to one cut ("fat separator" = cut state + its landmarks), the
interior of every segment eliminated against the two fat separators at its ends, and the resulting block-tridiagonal
system of fat blocks solved by block cyclic reduction.  This file restates the same plan and the same algebra with dense
numpy so that (a) the plan rules can be tested without a GPU, (b) the fat system the GPU assembles can be compared block
by block, (c) the result can be checked against a dense solve of the whole bordered system.
"""
import bisect

import numpy as np


def make_cuts(N, C):
    cuts = list(range(0, N - 1, C))
    if cuts[-1] != N - 1:
        cuts.append(N - 1)
    if len(cuts) >= 3 and cuts[-1] - cuts[-2] < 2:      # keep at least one interior state in the last segment
        cuts.pop(-2)
    return cuts


def plan(N, L, touch, C):
    """touch[l] = (smin, smax) states touched by landmark l's factors, or None.  Returns (cuts, fat_of, slot_of, counts)
    or None when some landmark spans more than two segments (C too small)."""
    cuts = make_cuts(N, C)
    K = len(cuts)
    counts = [0] * K
    fat_of, slot_of = [0] * L, [0] * L
    for l in range(L):
        if touch[l] is None:
            lo = hi = l % K
        else:
            smin, smax = touch[l]
            k_lo = bisect.bisect_right(cuts, smin) - 1
            k_hi = bisect.bisect_left(cuts, smax)
            if k_hi - k_lo > 2:
                return None
            lo, hi = max(k_hi - 1, 0), min(k_lo + 1, K - 1)
        best = min(range(lo, hi + 1), key=lambda k: (counts[k], k))
        fat_of[l], slot_of[l] = best, counts[best]
        counts[best] += 1
    return cuts, fat_of, slot_of, counts


def border_cost(nb, fat_max=128):
    """What a segmentation costs per state (FatSepPlan::choose, round 4): the Schur complement of a segment is
    (NCP / 16)(NCP / 16 + 1) / 2 MFMA tiles per four rows, the border sweep NC columns; 0.064 ms per tile and 0.025 ms per column
    for 1e6 states on the config-4 graph.  nb = B + ld * (landmarks on the fullest cut)."""
    nbr = min((nb + 3) & ~3, fat_max)
    nc = 2 * nbr + 1
    t = ((nc + 15) & ~15) // 16
    return 0.064 * (t * (t + 1) // 2) + 0.025 * nc


def choose_segment_length(N, L, touch, B, ld, fat_max=128):
    """The search of FatSepPlan::choose with this model's (simpler, greedy) landmark-to-cut assignment: double from 32 until
    every landmark's window fits two segments and the fullest cut fits a fat block, then try the lengths between that and half of
    it in sixteenths of it (at least 16 states), longest first; a shorter length wins only with a strictly lower border cost.
    Returns (C, nb) or None."""
    def attempt(C):
        p = plan(N, L, touch, C)
        if p is None:
            return None
        return B + ld * max(p[3])
    C = 32
    while True:
        nb = attempt(C)
        if nb is not None and nb <= fat_max:
            break
        if len(make_cuts(N, C)) <= 2 or (nb is not None and nb > fat_max):
            return None
        C *= 2
    best, best_nb, best_cost = C, nb, border_cost(nb, fat_max)
    step = max(16, C // 16)
    c = C - step
    while c > C // 2:
        nb = attempt(c)
        if nb is not None and nb <= fat_max and border_cost(nb, fat_max) < best_cost - 1e-12:
            best, best_nb, best_cost = c, nb, border_cost(nb, fat_max)
        c -= step
    return best, best_nb


def fat_system(D, O, g, B, HLL, gL, ld, cuts, fat_of, slot_of, NB, lam=0.0):
    """Dense elimination of every segment interior -> (Dfat K x NB x NB, Ofat (K-1) x NB x NB [H[k+1, k]], gfat K x NB).
    Unused slots get a unit diagonal."""
    N, b = g.shape
    K = len(cuts)
    L = len(fat_of)
    nl = L * ld

    def cols(k):
        """global variable indices of fat block k, padded with -1: state cut_k then its landmarks"""
        idx = [-1] * NB
        for r in range(b):
            idx[r] = cuts[k] * b + r
        for l in range(L):
            if fat_of[l] == k:
                for q in range(ld):
                    idx[b + slot_of[l] * ld + q] = N * b + l * ld + q
        return idx

    # dense full matrix (test sizes only)
    n = N * b + nl
    H = np.zeros((n, n))
    rhs = np.zeros(n)
    for i in range(N):
        H[i * b:(i + 1) * b, i * b:(i + 1) * b] = D[i]
        if i + 1 < N:
            H[(i + 1) * b:(i + 2) * b, i * b:(i + 1) * b] = O[i]
            H[i * b:(i + 1) * b, (i + 1) * b:(i + 2) * b] = O[i].T
        rhs[i * b:(i + 1) * b] = g[i]
    if nl:
        H[:N * b, N * b:] = B.reshape(N * b, nl)
        H[N * b:, :N * b] = B.reshape(N * b, nl).T
        H[N * b:, N * b:] = HLL
        rhs[N * b:] = gL
    H += lam * np.eye(n)
    top = sorted(set(i for k in range(K) for i in cols(k) if i >= 0))
    top_set = set(top)
    inner = [i for i in range(n) if i not in top_set]
    Hti = H[np.ix_(top, inner)]
    Hii = H[np.ix_(inner, inner)]
    S = H[np.ix_(top, top)] - Hti @ np.linalg.solve(Hii, Hti.T)
    sr = rhs[top] - Hti @ np.linalg.solve(Hii, rhs[inner])
    pos = {v: p for p, v in enumerate(top)}
    Dfat = np.zeros((K, NB, NB))
    Ofat = np.zeros((max(K - 1, 0), NB, NB))
    gfat = np.zeros((K, NB))
    for k in range(K):
        ck = cols(k)
        for r, ir in enumerate(ck):
            if ir < 0:
                Dfat[k, r, r] = 1.0
                continue
            gfat[k, r] = sr[pos[ir]]
            for c, ic in enumerate(ck):
                if ic >= 0:
                    Dfat[k, r, c] = S[pos[ir], pos[ic]]
            if k + 1 < K:
                for c2, ic2 in enumerate(cols(k + 1)):
                    if ic2 >= 0:
                        Ofat[k, c2, r] = S[pos[ic2], pos[ir]]
    return Dfat, Ofat, gfat, (H, rhs)


def cyclic_reduction(Dfat, Ofat, gfat):
    """Block cyclic reduction over level sets (the order the HIP fat solver uses): returns x (K x NB)."""
    K, NB = gfat.shape
    Dm = {k: Dfat[k].copy() for k in range(K)}
    gm = {k: gfat[k].copy() for k in range(K)}
    link = {(k, k + 1): Ofat[k].copy() for k in range(K - 1)}    # link[(l, r)] = H[r, l]
    active = list(range(K))
    trail = []
    while len(active) > 1:
        nxt = active[0::2]
        for p in range(1, len(active), 2):
            m, l = active[p], active[p - 1]
            r = active[p + 1] if p + 1 < len(active) else None
            Lc = np.linalg.cholesky(Dm[m])
            P = np.linalg.solve(Lc, link[(l, m)])                 # L^-1 H[m, l]
            z = np.linalg.solve(Lc, gm[m])
            Dm[l] -= P.T @ P
            gm[l] -= P.T @ z
            Q = None
            if r is not None:
                Q = np.linalg.solve(Lc, link[(m, r)].T)           # L^-1 H[m, r]
                Dm[r] -= Q.T @ Q
                gm[r] -= Q.T @ z
                link[(l, r)] = -Q.T @ P
            trail.append((m, l, r, Lc, P, Q, z))
        active = nxt
    x = {}
    k0 = active[0]
    x[k0] = np.linalg.solve(Dm[k0], gm[k0])
    for m, l, r, Lc, P, Q, z in reversed(trail):
        t = z - P @ x[l]
        if r is not None:
            t -= Q @ x[r]
        x[m] = np.linalg.solve(Lc.T, t)
    return np.stack([x[k] for k in range(K)])


def build_levels(K, keep_last):
    """Level sets of the block cyclic reduction as FatSepPlan::build_levels (gpslam_amd/csrc/fatsep.hpp) forms them: every
    level eliminates every other active block; keep_last: the last block is never eliminated (a piece of a split chain
    stops at its two end blocks).  Returns (levels, survivors): levels = list of lists of (m, l, r) with r = None when the
    eliminated block has no right neighbour."""
    active = list(range(K))
    levels = []
    while len(active) > (2 if keep_last else 1):
        n = len(active)
        elim = lambda q: 0 <= q < n and (q & 1) == 1 and not (keep_last and q == n - 1)
        lv = [(active[q], active[q - 1], active[q + 1] if q + 1 < n else None) for q in range(1, n, 2) if elim(q)]
        levels.append(lv)
        active = [active[i] for i in range(n) if not elim(i)]
    return levels, active


def reduce_piece(Dfat, Ofat, gfat, keep_last=True):
    """Forward half of the cyclic reduction over the given level sets.  Returns (ends, D, g, link, trail): what is left on
    the surviving blocks (`ends`: their indices) and the trail for the back-substitution."""
    K, NB = gfat.shape
    Dm = {k: Dfat[k].copy() for k in range(K)}
    gm = {k: gfat[k].copy() for k in range(K)}
    link = {(k, k + 1): Ofat[k].copy() for k in range(K - 1)}    # link[(l, r)] = H[r, l]
    levels, ends = build_levels(K, keep_last)
    trail = []
    for lv in levels:
        for m, l, r in lv:
            Lc = np.linalg.cholesky(Dm[m])
            P = np.linalg.solve(Lc, link[(l, m)])
            z = np.linalg.solve(Lc, gm[m])
            Dm[l] -= P.T @ P
            gm[l] -= P.T @ z
            Q = None
            if r is not None:
                Q = np.linalg.solve(Lc, link[(m, r)].T)
                Dm[r] -= Q.T @ Q
                gm[r] -= Q.T @ z
                link[(l, r)] = -Q.T @ P
            trail.append((m, l, r, Lc, P, Q, z))
    return ends, Dm, gm, link, trail


def back_substitute(trail, x):
    for m, l, r, Lc, P, Q, z in reversed(trail):
        t = z - P @ x[l]
        if r is not None:
            t -= Q @ x[r]
        x[m] = np.linalg.solve(Lc.T, t)
    return x


def split_solve(Dfat, Ofat, gfat, bounds, share=0.5):
    """The fat block-tridiagonal system cut into pieces at the blocks `bounds` (first and last block included): piece r holds
    the blocks bounds[r] .. bounds[r + 1]; a shared block's diagonal / right-hand side is split between its two pieces
    (`share` to the left one -- any split is exact).  Every piece is reduced to its two end blocks (interface record), the
    records are joined into the (P + 1)-block system of the shared separators, solved, and every piece back-substitutes:
    the scheme of gpslam_hip_fs_phase1 / fs_phase2.  Returns x (K x NB)."""
    K, NB = gfat.shape
    P = len(bounds) - 1
    recs, pieces = [], []
    for r in range(P):
        lo, hi = bounds[r], bounds[r + 1]
        Dp, gp = Dfat[lo:hi + 1].copy(), gfat[lo:hi + 1].copy()
        if r > 0:
            Dp[0] *= (1.0 - share); gp[0] *= (1.0 - share)
        if r < P - 1:
            Dp[-1] *= share; gp[-1] *= share
        ends, Dm, gm, link, trail = reduce_piece(Dp, Ofat[lo:hi].copy(), gp, keep_last=True)
        assert ends == [0, hi - lo]
        recs.append((Dm[0], link[(0, hi - lo)], Dm[hi - lo], gm[0], gm[hi - lo]))
        pieces.append((lo, hi, trail))
    T = np.zeros(((P + 1) * NB, (P + 1) * NB))
    tr = np.zeros((P + 1) * NB)
    for j in range(P + 1):
        sl = slice(j * NB, (j + 1) * NB)
        if j > 0:
            T[sl, sl] += recs[j - 1][2]; tr[sl] += recs[j - 1][4]
        if j < P:
            T[sl, sl] += recs[j][0]; tr[sl] += recs[j][3]
            nx = slice((j + 1) * NB, (j + 2) * NB)
            T[nx, sl] = recs[j][1]; T[sl, nx] = recs[j][1].T
    xt = np.linalg.solve(T, tr).reshape(P + 1, NB)
    x = np.zeros((K, NB))
    for r, (lo, hi, trail) in enumerate(pieces):
        xl = {0: xt[r], hi - lo: xt[r + 1]}
        back_substitute(trail, xl)
        for k, v in xl.items():
            x[lo + k] = v
    return x


# ---- round 5: fat blocks beyond 80 columns (fatsep.hpp: k_fat_elim_wide, k_fs_syrk's right-hand-side row and its two-workgroup grid)
def wide_panel_width(NB, elem_bytes=8, fat_max=128):
    """fat_wide_panel(): columns of [H | H | g] per pass -- what 160 KB of LDS hold beside the NB x (NB + 1) block and the 4 x 4 factors"""
    avail = 160 * 1024 - (fat_max // 4) * 10 * elem_bytes - NB * (NB + 1) * elem_bytes - 512
    return max(4, min(avail // (NB * elem_bytes), 2 * NB + 1))


def eliminate_block_in_panels(D, Hl, Hr, g, PW):
    """One block of the fat chain's cyclic reduction the way k_fat_elim_wide walks it: D = L L^T once; X = [Hl | Hr | g] through a panel
    of PW columns at a time, X <- L^-1 X by steps of four pivots (the 4 x 4 diagonal factor inverted, a rank-4 update of the rows below);
    then S1 = P^T P, S2 = Q^T Q, link = -(Q^T P), P^T z, Q^T z."""
    NB = D.shape[0]
    Lc = np.linalg.cholesky(D)
    X = np.concatenate([Hl, Hr, g[:, None]], axis=1)
    out = np.zeros_like(X)
    for c0 in range(0, X.shape[1], PW):
        Xp = X[:, c0:c0 + PW].copy()
        for p in range(0, NB, 4):
            W = np.linalg.inv(Lc[p:p + 4, p:p + 4])
            Xp[p:p + 4] = W @ Xp[p:p + 4]
            Xp[p + 4:] -= Lc[p + 4:, p:p + 4] @ Xp[p:p + 4]
        out[:, c0:c0 + PW] = Xp
    P, Q, z = out[:, :NB], out[:, NB:2 * NB], out[:, 2 * NB]
    return Lc, P, Q, z, P.T @ P, Q.T @ Q, -(Q.T @ P), P.T @ z, Q.T @ z


def syrk_rhs_row_columns(NB):
    """Columns of the right-hand-side row of a segment's Schur complement that k_fs_syrk sums beside the matrix cores (2 NB a multiple
    of 16): wave wv < nrw takes column lane + 64 wv."""
    if (2 * NB) % 16:
        return None
    nrw = min((2 * NB + 63) // 64, 4)
    return sorted({lane + 64 * wv for wv in range(nrw) for lane in range(64) if lane + 64 * wv <= 2 * NB})


def syrk_tiles_of(wv, y, ny, tpw, ntiles):
    """lower-triangle tiles of wave wv in workgroup y of ny: p = (3 - wv) + 4 (q ny + y), q < tpw"""
    return [(3 - wv) + 4 * (q * ny + y) for q in range(tpw) if (3 - wv) + 4 * (q * ny + y) < ntiles]
