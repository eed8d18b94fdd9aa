"""Random landmark graphs through the segmented landmark elimination (fatsep.hpp) against the oracle's dense bordered solve: random
chain length, landmark density (0.4 ... 5 x config 4's), window of visibility and segment length -- the fat block widths NB, border
widths and level counts the fixed-size tests do not name.  Three Gauss-Newton iterations in lock step: states and landmarks 1e-9
relative at the end, error_after 1e-9 (+ 1e-11 of error_before: the first step from dead reckoning takes the cost down by three to four
orders of magnitude and leaves error_after with the rounding of the larger number) -- plus ten times the distance between the oracle
and its twin started 1e-15 away (80 cases, seed 203: three graphs at 1.3e-9 ... 2.1e-9, the twins of those graphs apart by as much).
   python scripts/stress_segmented.py [count] [seed]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import time
import numpy as np
from oracle import oracle as O
import gpslam_amd as gp
from gpslam_amd import synthetic as S
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
seen, bad, refused = {}, [], 0
for t in range(cnt):
    N = int(rng.integers(300, 1500))
    window = int(rng.choice([60, 100, 150, 200, 250]))
    dens = float(rng.choice([0.4, 1.0, 1.7, 2.5, 3.3, 4.2, 5.0]))
    L = max(int(dens * N / 20), 3)
    seglen = int(rng.choice([0, 0, 128, 192, 256, 300]))
    p = S.pose2_local_landmarks_chain(N, L=L, window=window, seed=t)
    try:
        dev = S.apply(p, gp.ChainSolver(O.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2, segment_length=seglen, force_segmented=True))
    except gp.GpslamHipError as ex:
        refused += 1
        print("refused N %d L %d window %d seglen %d: %s" % (N, L, window, seglen, str(ex)[-70:]), flush=True)
        continue
    plan = dev.segment_plan()
    t0 = time.time()
    orc = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    # the oracle's twin: the same graph from initial poses 1e-15 (relative) away.  How far the two ORACLES part in three iterations is
    # what rounding alone does to this graph (range-only landmarks leave soft directions); the product is held to 1e-9 + ten times that
    twin = S.apply(p, O.Chain(O.POSE2, chart=O.CHART_FIRST_ORDER, landmark_dim=2))
    twin.set_states(p["pose"] * (1.0 + 1e-15 * np.random.default_rng(t).standard_normal(p["pose"].shape)), p["vel"])
    worst = 0.0
    ok = True
    for it in range(3):
        rc0, s0 = orc.iterate_gn()
        rc1, s1 = dev.iterate_gn()
        rc2, s2 = twin.iterate_gn()
        rel = abs(s0.error_after - s1.error_after) / max(1.0, s0.error_after)
        worst = max(worst, rel)
        ok = ok and rc0 == 0 and rc1 == 0 and rc2 == 0 and abs(s0.error_after - s1.error_after) <= 1e-9 * max(1.0, s0.error_after) + 1e-11 * s0.error_before + 10 * abs(s0.error_after - s2.error_after)
    dist = lambda a_, b_: max(np.abs(a_.get_states()[0] - b_.get_states()[0]).max() / max(1.0, np.abs(a_.get_states()[0]).max()),
                              np.abs(a_.get_states()[1] - b_.get_states()[1]).max() / max(1.0, np.abs(a_.get_states()[1]).max()),
                              np.abs(a_.get_landmarks() - b_.get_landmarks()).max() / max(1.0, np.abs(a_.get_landmarks()).max()))
    dx, noise = dist(orc, dev), dist(orc, twin)
    ok = ok and dx <= 1e-9 + 10 * noise
    seen[plan["NB"]] = seen.get(plan["NB"], 0) + 1
    print("%s N %d L %d window %d seglen %d -> C %d K %d NB %d NCP %d levels %d | error rel %.1e states rel %.1e (the oracle's twin: %.1e) (%.1f s)"
          % ("ok " if ok else "BAD", N, L, window, seglen, plan["C"], plan["K"], plan["NB"], plan["NCP"], plan["levels"], worst, dx, noise, time.time() - t0), flush=True)
    if not ok:
        bad.append((N, L, window, seglen, t, plan))
    dev.close()
print("fat block widths seen:", dict(sorted(seen.items())), "refused:", refused)
print("all agree with the oracle" if not bad else "DISAGREE: %s" % bad)
sys.exit(1 if bad else 0)
