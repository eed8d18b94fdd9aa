"""The plan rules and the algebra of the segmented landmark elimination (numpy model, no GPU)."""
import numpy as np

import fatsep_model as FM


def _random_problem(N, L, b, ld, window, seed):
    rng = np.random.default_rng(seed)
    rows = []                                            # (state, two, landmark, JL, JR, m, e)
    centre = np.sort(rng.integers(0, N, L))
    touch = [None] * L
    for l in range(L):
        for _ in range(rng.integers(2, 9)):
            i = int(np.clip(centre[l] + rng.integers(-window // 2, window // 2 + 1), 0, N - 2))
            rows.append((i, l, rng.normal(size=b), rng.normal(size=b), rng.normal(size=ld), rng.normal()))
            lo, hi = (i, i + 1)
            touch[l] = (lo, hi) if touch[l] is None else (min(touch[l][0], lo), max(touch[l][1], hi))
    D = np.zeros((N, b, b)); O = np.zeros((N, b, b)); g = np.zeros((N, b))
    B = np.zeros((N, b, L * ld)); HLL = np.zeros((L * ld, L * ld)); gL = np.zeros(L * ld)
    for i in range(N - 1):                               # chain factors
        J = rng.normal(size=(b + 2, 2 * b))
        e = rng.normal(size=b + 2)
        D[i] += J[:, :b].T @ J[:, :b]; D[i + 1] += J[:, b:].T @ J[:, b:]; O[i] += J[:, b:].T @ J[:, :b]
        g[i] -= J[:, :b].T @ e; g[i + 1] -= J[:, b:].T @ e
    D[0] += np.eye(b)
    for i, l, JL, JR, m, e in rows:
        D[i] += np.outer(JL, JL); D[i + 1] += np.outer(JR, JR); O[i] += np.outer(JR, JL)
        g[i] -= JL * e; g[i + 1] -= JR * e
        sl = slice(l * ld, (l + 1) * ld)
        B[i, :, sl] += np.outer(JL, m); B[i + 1, :, sl] += np.outer(JR, m)
        HLL[sl, sl] += np.outer(m, m); gL[sl] -= m * e
    HLL += 0.1 * np.eye(L * ld)
    return D, O, g, B, HLL, gL, touch


def test_plan_rules():
    assert FM.make_cuts(10, 4) == [0, 4, 9]               # 8 would leave no interior before 9
    assert FM.make_cuts(2, 64) == [0, 1]
    assert FM.make_cuts(130, 64) == [0, 64, 129]          # (128 dropped: it would leave an empty last segment)
    touch = [(3, 4), (60, 70), (0, 140), None]
    assert FM.plan(200, 4, touch, 64) is None            # landmark 2 spans three segments
    cuts, fat_of, slot_of, counts = FM.plan(200, 4, [(3, 4), (60, 70), (100, 130), None], 64)
    assert cuts == [0, 64, 128, 192, 199]
    assert fat_of[1] == 1 and fat_of[2] == 2 and sum(counts) == 4


def test_segment_length_search_finds_the_narrower_border():
    """Config 4's geometry (one landmark per 20 states, each seen from a window of 200): the doubling stops at 256 states per segment
    (the first length whose segments hold every window in two); the search behind it must find a length in (128, 256) with fewer
    landmarks on the fullest cut -- a cheaper border -- and must never return a length that does not fit."""
    N, every, window, B, ld = 40000, 20, 200, 6, 2
    L = N // every
    centre = [min(int((l + 0.5) * every), N - 1) for l in range(L)]
    touch = [(max(c - window // 2, 0), min(c + window // 2, N - 1)) for c in centre]
    C, nb = FM.choose_segment_length(N, L, touch, B, ld)
    nb256 = B + ld * max(FM.plan(N, L, touch, 256)[3])
    assert 128 < C < 256 and FM.plan(N, L, touch, C) is not None
    assert FM.plan(N, L, touch, 128) is None                       # (a 200-state window does not fit two segments of 128)
    assert nb < nb256 and FM.border_cost(nb) < FM.border_cost(nb256)
    # (this model's greedy assignment leaves 12 landmarks on the fullest cut at 208 states where the library's balanced one leaves
    #  11: the library's plan drops a whole 16-column panel there -- NB 36 -> 28, 15 -> 10 tiles -- which is what the cost prices)
    panels = lambda v: (((2 * ((v + 3) & ~3) + 1) + 15) & ~15) // 16
    assert panels(36) == 5 and panels(28) == 4 and FM.border_cost(28) < 0.8 * FM.border_cost(36)
    # a graph whose first fit is already the cheapest keeps it: windows of 40 states fit segments of 32 + 32
    touch2 = [(max(c - 20, 0), min(c + 20, N - 1)) for c in centre]
    C2, nb2 = FM.choose_segment_length(N, L, touch2, B, ld)
    assert FM.plan(N, L, touch2, C2) is not None and C2 <= 64


def test_fat_elimination_and_cyclic_reduction_equal_dense_solve():
    for seed, (N, L, C, window) in enumerate([(40, 6, 8, 6), (97, 20, 16, 12), (33, 5, 64, 30), (64, 9, 7, 5)]):
        b, ld = 4, 2
        D, O, g, B, HLL, gL, touch = _random_problem(N, L, b, ld, window, seed)
        p = FM.plan(N, L, touch, C)
        assert p is not None
        cuts, fat_of, slot_of, counts = p
        NB = b + ld * max(counts)
        for lam in (0.0, 0.3):
            Dfat, Ofat, gfat, (H, rhs) = FM.fat_system(D, O, g, B, HLL, gL, ld, cuts, fat_of, slot_of, NB, lam)
            xfat = FM.cyclic_reduction(Dfat, Ofat, gfat)
            xd = np.linalg.solve(H, rhs)
            for k, c in enumerate(cuts):
                np.testing.assert_allclose(xfat[k, :b], xd[c * b:(c + 1) * b], rtol=1e-8, atol=1e-10)
            for l in range(L):
                got = xfat[fat_of[l], b + slot_of[l] * ld: b + (slot_of[l] + 1) * ld]
                np.testing.assert_allclose(got, xd[N * b + l * ld: N * b + (l + 1) * ld], rtol=1e-8, atol=1e-10)


def test_level_sets_keep_the_last_block_of_a_piece():
    for K in range(2, 40):
        levels, ends = FM.build_levels(K, keep_last=True)
        assert ends == [0, K - 1]
        gone = [m for lv in levels for m, _, _ in lv]
        assert sorted(gone) == list(range(1, K - 1))                   # every interior block exactly once
        for lv in levels:
            for m, l, r in lv:
                assert l < m and r is not None and m < r               # with the last block kept there is always a right neighbour
        levels1, top = FM.build_levels(K, keep_last=False)
        assert top == [0] and sorted(m for lv in levels1 for m, _, _ in lv) == list(range(1, K))
    assert len(FM.build_levels(3908, True)[0]) == 12                   # BASELINE config 4 on one GPU: 12 levels either way


def test_split_fat_chain_equals_the_unsplit_solve():
    """The algebra of gpslam_hip_fs_phase1 / fs_phase2 on CPU: pieces reduced to their end blocks, records joined, top system
    solved, pieces back-substituted -- against the dense solve of the whole bordered system."""
    b, ld = 4, 2
    D, O, g, B, HLL, gL, touch = _random_problem(120, 24, b, ld, 6, 5)
    cuts, fat_of, slot_of, counts = FM.plan(120, 24, touch, 10)
    K, NB = len(cuts), b + ld * max(counts)
    Dfat, Ofat, gfat, (H, rhs) = FM.fat_system(D, O, g, B, HLL, gL, ld, cuts, fat_of, slot_of, NB, 0.2)
    xd = np.linalg.solve(H, rhs)
    for bounds, share in (([0, K - 1], 0.5), ([0, 5, K - 1], 0.5), ([0, 3, 4, 9, K - 1], 0.25), ([0, 1, 2, K - 1], 1.0)):
        x = FM.split_solve(Dfat, Ofat, gfat, bounds, share)
        np.testing.assert_allclose(x, FM.cyclic_reduction(Dfat, Ofat, gfat), rtol=1e-8, atol=1e-9)
        for k, c in enumerate(cuts):
            np.testing.assert_allclose(x[k, :b], xd[c * b:(c + 1) * b], rtol=1e-7, atol=1e-9)


def test_panelled_wide_block_elimination_equals_the_one_pass_algebra():
    """k_fat_elim_wide (round 5): blocks beyond 80 columns keep only D in LDS and pass [H | H | g] through it in panels; whatever the
    panel width, P, Q, z and the five products are those of the one-pass elimination."""
    rng = np.random.default_rng(5)
    for NB in (84, 104, 128):
        A = rng.normal(size=(NB, NB + 7)); D = A @ A.T + NB * np.eye(NB)
        Hl, Hr, g = rng.normal(size=(NB, NB)), rng.normal(size=(NB, NB)), rng.normal(size=NB)
        Lc = np.linalg.cholesky(D)
        P0, Q0, z0 = np.linalg.solve(Lc, Hl), np.linalg.solve(Lc, Hr), np.linalg.solve(Lc, g)
        PW = FM.wide_panel_width(NB)
        assert 4 <= PW <= 2 * NB + 1 and (NB * (NB + 1) + NB * PW) * 8 + 2560 + 512 <= 160 * 1024      # fits beside the block
        for pw in (PW, 4, 2 * NB + 1):
            L1, P, Q, z, S1, S2, lk, pz, qz = FM.eliminate_block_in_panels(D, Hl, Hr, g, pw)
            for a, b in ((P, P0), (Q, Q0), (z, z0), (S1, P0.T @ P0), (S2, Q0.T @ Q0), (lk, -(Q0.T @ P0)), (pz, P0.T @ z0), (qz, Q0.T @ z0)):
                assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    assert FM.wide_panel_width(128) == 28 and FM.wide_panel_width(84) >= 128


def test_every_column_of_the_rhs_row_has_a_lane():
    """Rounds 3-4 summed the row with "waves 0 and 1" whatever NB: at NB = 72 and NB = 80 the columns from 128 on were never written
    (found in round 5 by the NB = 80 graph of test_gpu_segmented.py).  Every NB up to 128 whose 2 NB is a multiple of 16: the 2 NB
    columns k_fs_fat_assemble reads (row 2 NB, columns < 2 NB) are all covered."""
    for NB in range(8, 129, 8):
        cols = FM.syrk_rhs_row_columns(NB)
        assert cols is not None and set(range(2 * NB)) <= set(cols), NB
    assert FM.syrk_rhs_row_columns(28) is None            # config 4: the row is inside the last tile row
    old = lambda NB: {lane + 64 * wv for wv in range(2) for lane in range(64)}
    assert not set(range(2 * 72)) <= old(72) and not set(range(2 * 80)) <= old(80) and set(range(2 * 64)) <= old(64)


def test_two_workgroups_deal_every_tile_of_a_wide_border_once():
    """k_fs_syrk<20, 272> (borders of 177 ... 272 columns): the 16 x 16 tiles of the lower triangle dealt over 2 workgroups x 4 waves x
    20 accumulators; the narrower instantiations keep one workgroup."""
    for tpw, ncm, ny in ((7, 112, 1), (12, 144, 1), (17, 176, 1), (20, 272, 2)):
        t16 = ncm // 16
        ntiles = t16 * (t16 + 1) // 2
        got = sorted(p for y in range(ny) for wv in range(4) for p in FM.syrk_tiles_of(wv, y, ny, tpw, ntiles))
        assert got == list(range(ntiles)), (tpw, ncm, ny)
