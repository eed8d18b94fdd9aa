"""Landmark elimination at scale (BASELINE config 4): the segmented path with fat separators (gpslam_amd/csrc/fatsep.hpp)
against the CPU oracle's dense bordered solve, through the C ABI."""
import numpy as np
import pytest

from oracle import oracle as O
from gpslam_amd import synthetic as S
from test_gpu_parity import gpu, states_close

pytestmark = pytest.mark.gpu


def _pair(p, **kw):
    orc = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    dev = S.apply(p, gpu().ChainSolver(O.POSE2, chart=gpu().CHART_FIRST_ORDER, landmark_dim=2, **kw))
    return orc, dev


@pytest.mark.parametrize("N,seglen", [(2000, 0), (2000, 500), (700, 250), (333, 128)])
def test_c4_local_landmarks_gauss_newton_matches_oracle(N, seglen):
    """N / 20 landmarks with a 200-state visibility window (100 at N = 2000: R = 201 columns, far beyond the dense border):
    Gauss-Newton iterations in lock step with the oracle, 1e-9 relative."""
    p = S.pose2_local_landmarks_chain(N, window=100 if N < 1000 else 200)
    orc, dev = _pair(p, segment_length=seglen)
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    for it in range(4):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (it, s0.error_after, s1.error_after)
        assert abs(s0.delta_inf_norm - s1.delta_inf_norm) <= 1e-8 * max(1.0, s0.delta_inf_norm)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)
    l0, l1 = orc.get_landmarks(), dev.get_landmarks()
    assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max())


def test_twice_config4_landmark_density_matches_oracle():
    """config 4 sat at NB = 40 of 64 and 1.6x its landmark density did not compile.  Round 3: balanced landmark-to-cut
    assignment (config 4: NB = 36) and fat blocks up to 80 columns.  Here L = N / 10 (twice config 4's density, ~28 landmarks
    per cut: NB above the old limit of 64) against the oracle's dense bordered solve."""
    N = 2600
    p = S.pose2_local_landmarks_chain(N, L=N // 10, window=200)
    orc, dev = _pair(p, segment_length=256)      # (the automatic choice would look for a shorter segment with a narrower block)
    plan = dev.segment_plan()
    assert plan["active"] == 1 and plan["NB"] > 48, plan
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    for it in range(4):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (it, s0.error_after, s1.error_after)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)
    l0, l1 = orc.get_landmarks(), dev.get_landmarks()
    assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max())


@pytest.mark.parametrize("L,NB", [(180, 72), (195, 80), (225, 96), (295, 116), (340, 128)],
                         ids=["NB72-rhs-row-beyond-128", "3x-density-NB80", "NB96-wide", "4.5x-density-NB116-wide", "5.2x-density-NB128-the-limit"])
def test_three_times_config4_landmark_density_and_beyond_matches_oracle(L, NB):
    """fat separators wider than 80 columns.  Round 5: kFatMax = 128 -- blocks beyond 80 columns keep the factor in
    LDS and pass [H | H | g] through it in column panels (k_fat_elim_wide), their borders (up to 257 columns) take the five-wave
    sweep and a Schur-complement launch of two workgroups per segment (k_fs_syrk<20, 272>).  N = 1300 states with L landmarks
    (config 4's density is N / 20 = 65): 3x gives NB = 80 exactly, 4.5x NB = 116, 5.2x NB = 128 (257 border columns); L = 345 is
    refused with a message.  NB = 72 and NB = 80 are also the two widths at which the right-hand-side row of the Schur complement
    (summed beside the matrix cores when 2 NB is a multiple of 16) has columns beyond 128, which only two of the three waves it
    needs used to sum: the round-4 library's first Gauss-Newton step is 7-11 % off in error_after on these two graphs."""
    N = 1300
    p = S.pose2_local_landmarks_chain(N, L=L, window=200)
    orc, dev = _pair(p, segment_length=256)
    plan = dev.segment_plan()
    assert plan["active"] == 1 and plan["NB"] == NB, plan
    assert abs(orc.error() - dev.error()) <= 1e-10 * orc.error()
    for it in range(4):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (it, s0.error_after, s1.error_after)
        assert abs(s0.delta_inf_norm - s1.delta_inf_norm) <= 1e-8 * max(1.0, s0.delta_inf_norm)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)
    l0, l1 = orc.get_landmarks(), dev.get_landmarks()
    assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max())


def test_more_landmarks_per_cut_than_128_columns_hold_are_refused_with_a_message():
    gp = gpu()
    p = S.pose2_local_landmarks_chain(1300, L=360, window=200)
    with pytest.raises(gp.GpslamHipError, match="too many landmarks per cut"):
        S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2, segment_length=256))


def test_wide_fat_blocks_through_levenberg_marquardt():
    """The damped trials of the wide path (lambda on the fat blocks' diagonals, the gradient kept for the model): five LM
    iterations in lock step with the oracle from open-loop dead reckoning."""
    import lm_lockstep
    p = S.pose2_local_landmarks_chain(1300, L=240, window=200, anchor=0)
    orc, dev = _pair(p, segment_length=256)
    assert dev.segment_plan()["NB"] > 80
    _, _, slack = lm_lockstep.run(orc, dev, 1e-5, 5)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9 + 2 * slack)


def test_segmented_path_equals_dense_border_on_a_small_graph():
    """The same small graph (8 landmarks seen from everywhere would not segment; 6 local ones do) through both landmark
    paths of the library."""
    p = S.pose2_local_landmarks_chain(400, L=6, window=120)
    gp = gpu()
    a = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))
    b = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2, force_segmented=True, segment_length=150))
    for _ in range(3):
        _, sa = a.iterate_gn()
        _, sb = b.iterate_gn()
        assert abs(sa.error_after - sb.error_after) <= 1e-9 * max(1.0, sa.error_after)
    states_close(O.POSE2, *a.get_states(), *b.get_states(), rel=1e-9)
    assert np.abs(a.get_landmarks() - b.get_landmarks()).max() <= 1e-9 * 100


def test_c4_levenberg_marquardt_matches_oracle():
    # open-loop dead reckoning: five LM iterations stay far from convergence, where accept / reject decisions are not
    # within rounding of the fidelity threshold
    p = S.pose2_local_landmarks_chain(1200, window=200, anchor=0)
    orc, dev = _pair(p)
    import lm_lockstep
    lm_lockstep.run(orc, dev, 1e-5, 5)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)


def test_eighty_landmarks_seen_from_everywhere_fit_three_wide_blocks():
    """80 landmarks each seen from the whole chain: the longest segmentation (two segments) deals them over its three cuts -- fat
    blocks of 84 columns, which exist since round 5.  Until round 4 this graph was refused ("too many landmarks per cut")."""
    p = S.pose2_range_chain(600, L=80)
    orc, dev = _pair(p)
    plan = dev.segment_plan()
    assert plan["active"] == 1 and plan["K"] <= 3 and 80 < plan["NB"] <= 128, plan
    for it in range(3):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (it, s0.error_after, s1.error_after)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)
    l0, l1 = orc.get_landmarks(), dev.get_landmarks()
    assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max())


def test_landmark_seen_from_everywhere_is_rejected_with_a_message():
    p = S.pose2_range_chain(600, L=200)         # 200 landmarks each seen from the whole chain: no cut holds them, nor do the two ends
    gp = gpu()
    with pytest.raises(gp.GpslamHipError, match="more than two segments|too many landmarks per cut"):
        S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))


def test_c4_full_size_properties():
    """BASELINE config 4 at its full size on ONE GPU (1e6 SE(2) states, 5e4 landmarks, 4.4e5 interpolated ranges): no
    oracle can run this, so size-independent properties: Gauss-Newton reaches |delta|_inf < 1e-6, the error never
    increases, a second run is bit-identical, and a different segmentation (another elimination tree, equally exact)
    lands on the same state to 1e-9."""
    gp = gpu()
    N = 1000000
    p = S.pose2_local_landmarks_chain(N)
    assert len(p["landmarks"]) == 50000
    mk = lambda seg: S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2, segment_length=seg))
    a = mk(0)
    plan = a.segment_plan()
    assert plan["active"] == 1 and plan["NB"] <= 64
    hist = []
    for it in range(12):
        rc, st = a.iterate_gn()
        assert rc == 0
        hist.append((st.error_before, st.error_after, st.delta_inf_norm))
        if st.delta_inf_norm < 1e-6:
            break
    assert hist[-1][2] < 1e-6, hist
    assert all(h[1] <= h[0] * (1 + 1e-12) for h in hist), hist
    xa, va = a.get_states()
    la = a.get_landmarks()
    a.set_states(p["pose"], p["vel"])
    a.set_landmarks(p["landmarks"])
    for _ in range(len(hist)):
        a.iterate_gn()
    xb, vb = a.get_states()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb) and np.array_equal(la, a.get_landmarks())
    a.close()
    c = mk(2 * plan["C"] - 64)
    assert c.segment_plan()["K"] != plan["K"]
    for _ in range(len(hist)):
        c.iterate_gn()
    xc, vc = c.get_states()
    # positions reach 1e5 m here, so "1e-9 relative to the state vector" would be 1e-4 m; the two eliminations actually
    # agree to 1e-7 ABSOLUTE in every pose, velocity and landmark component (1e-12 of the vector's scale)
    dx, dv, dl = np.abs(xa - xc).max(), np.abs(va - vc).max(), np.abs(la - c.get_landmarks()).max()
    print("two segmentations: max |dx| %.3e |dv| %.3e |dl| %.3e" % (dx, dv, dl))
    assert dx <= 1e-7 and dv <= 1e-7 and dl <= 1e-7, (dx, dv, dl)


@pytest.mark.parametrize("kind,sensor,N,seg", [(O.POSE3, True, 61, 32), (O.POSE3, False, 130, 64), (O.LINEAR3, False, 61, 32), (O.POSE2, True, 97, 48)],
                         ids=["pose3+sensor", "pose3", "linear3", "pose2+sensor"])
def test_segmented_path_on_every_block_size_and_measurement_mix(kind, sensor, N, seg):
    """The segmented landmark elimination is compiled for the chain block sizes 4 (planar linear), 6 (SE(2)) and 12 (SE(3)) and
    walks generic Jacobian rows: the measurement mixes of tests/test_gpu_measurements.py (interpolated and plain ranges,
    bearing-range, GPS / attitude rows without landmarks, body_P_sensor, velocity priors) recorded once and replayed onto a
    handle that is forced onto the segmented path -- two or three segments, every landmark in the fat separator between them --
    next to the dense-border solve of the same graph."""
    import gpslam_amd
    from gpslam_amd import sharded
    from test_gpu_measurements import build_meas_pair, LD
    orc, ref, c, (rec,) = build_meas_pair(kind, N=N, seed=11, sensor=sensor, extra_makers=(sharded.GraphRecorder,))
    chart = O.CHART_FIRST_ORDER if kind == O.POSE2 else O.CHART_EXPMAP
    s = gpslam_amd.ChainSolver(kind, chart, LD[kind], force_segmented=True, segment_length=seg)
    rec.replay(s)
    assert s.segment_plan()["active"] == 1
    for it in range(5):
        _, a = s.iterate_gn()
        _, b = ref.iterate_gn()
        assert abs(a.error_before - b.error_before) <= 1e-9 * max(1.0, b.error_before), it
        assert abs(a.error_after - b.error_after) <= 1e-7 * max(1.0, b.error_after), it
    pa, va = s.get_states()
    pb, vb = ref.get_states()
    assert np.abs(pa - pb).max() <= 1e-8 * max(1.0, np.abs(pb).max())
    assert np.abs(va - vb).max() <= 1e-8 * max(1.0, np.abs(vb).max())
    assert np.abs(s.get_landmarks() - ref.get_landmarks()).max() <= 1e-8
    s.close()


_FUSED_AB = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import gpslam_amd
from gpslam_amd import synthetic as S
p = S.pose2_local_landmarks_chain(%d, L=%d, window=%d)
s = S.apply(p, gpslam_amd.ChainSolver(p["kind"], chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, segment_length=%d, plan=int(sys.argv[2])))
for _ in range(3):
    s.iterate_gn()
pose, vel = s.get_states()
np.savez(sys.argv[1], pose=pose, vel=vel, lmk=s.get_landmarks(), plan=np.array([s.segment_plan()["NB"], s.segment_plan()["NCP"]]))
"""


@pytest.mark.parametrize("N,seglen,div", [(20000, 0, 20), (3000, 256, 20), (777, 250, 20), (2400, 0, 80), (2400, 0, 40), (2400, 0, 27), (2400, 0, 16), (2400, 0, 13)])
def test_fused_sweep_and_schur_complement_reproduce_the_two_launch_path_bit_for_bit(N, seglen, div, tmp_path):
    """k_fs_sweep_syrk (sweep and MFMA waves sharing an LDS ring, no Y buffer) against k_fs_sweep + k_fs_syrk through the Y
    buffer (GPSLAM_PLAN_FS_TWO_LAUNCHES on the second handle; two child processes): same expressions in the same order -> the
    states after three Gauss-Newton iterations are IDENTICAL, including ragged last chunks and short last segments.
    div: landmarks = N / div -- a quarter to 1.6x config 4's density puts the border into every instantiation of the kernel
    (NCP = 32 .. 112 columns)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for fused in ("1", "0"):
        out = str(tmp_path / ("fused%s.npz" % fused))
        plan = "0" if fused == "1" else "8"
        subprocess.run([sys.executable, "-c", _FUSED_AB % (root, N, max(N // div, 1), 100 if N < 1000 else 200, seglen), out, plan], check=True, timeout=600)
        outs.append(np.load(out))
    a, b = outs
    assert a["plan"][1] <= 112, "the fused kernel serves borders up to 112 columns: this case would not exercise it"
    for k in ("pose", "vel", "lmk"):
        assert np.array_equal(a[k], b[k]), k


def test_segment_length_search_prefers_the_narrower_border():
    """compile() doubles the segment length until every landmark's window fits two segments, then tries the lengths between that
    and half of it (round 4): on config 4's landmark density the first fit is 256 (NB 36: an 80-column border, 15 MFMA tiles per
    chunk), 208 fits too (NB 28: 64 columns, 10 tiles).  The shorter segments must be what compile() picks, and they must solve
    the same problem (oracle, 1e-9)."""
    N = 2600
    p = S.pose2_local_landmarks_chain(N, window=200)
    orc, dev = _pair(p)
    _, ref = _pair(p, segment_length=256)
    plan, plan256 = dev.segment_plan(), ref.segment_plan()
    assert plan["active"] == 1 and plan256["C"] == 256
    assert 128 < plan["C"] <= 256 and plan["NCP"] <= plan256["NCP"] and plan["NB"] <= plan256["NB"], (plan, plan256)
    for it in range(4):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        assert rc0 == 0 and rc1 == 0
        assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (it, plan)
    states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)


def test_register_resident_fat_kernels_at_every_block_width():
    """k_fat_elim_rows<NBP> / k_fat_back_rows<NBP> (round 3: the fat blocks' factorisation and back-substitution in registers)
    are instantiated for NBP = 8 .. 48 in steps of 8; blocks beyond 48 columns keep the LDS kernels.  Landmark densities from a
    quarter to 1.6x config 4's put NB into most of those classes: each against the oracle's dense bordered solve, 1e-9."""
    seen = set()
    for div in (80, 40, 27, 20, 16, 13):
        N = 2400
        p = S.pose2_local_landmarks_chain(N, L=N // div, window=200)
        orc, dev = _pair(p, segment_length=256)  # (fixed: this test is about the block widths, not about the choice of segments)
        plan = dev.segment_plan()
        assert plan["active"] == 1, plan
        seen.add((plan["NB"] + 7) // 8 * 8)
        for it in range(3):
            rc0, s0 = orc.iterate_gn()
            rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0
            assert abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after), (div, plan, it, s0.error_after, s1.error_after)
        states_close(O.POSE2, *orc.get_states(), *dev.get_states(), rel=1e-9)
        l0, l1 = orc.get_landmarks(), dev.get_landmarks()
        assert np.abs(l0 - l1).max() <= 1e-9 * max(1.0, np.abs(l0).max()), (div, plan)
    assert len(seen) >= 4 and min(seen) <= 24 and max(seen) >= 48, seen
