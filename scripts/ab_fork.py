import os, sys, time; sys.path.insert(0, os.getcwd())
import gpslam_amd as g
from gpslam_amd import synthetic as S
which, N = sys.argv[1], int(sys.argv[2])
kw = {}
if which == "c5": p = S.rot3_attitude_chain(N)
elif which == "c4":
    p = S.pose2_local_landmarks_chain(N, window=200); kw = dict(chart=g.CHART_FIRST_ORDER, landmark_dim=2)
elif which == "c4p":
    p = S.pose2_range_chain(N, L=8); kw = dict(chart=g.CHART_FIRST_ORDER, landmark_dim=2)
s = S.apply(p, g.ChainSolver(p["kind"], **kw))
s.run_gn(3)
best = 1e9
for rep in range(4):
    s.set_states(p["pose"], p["vel"])
    if "landmarks" in p: s.set_landmarks(p["landmarks"])
    s.run_gn(2)
    t0 = time.perf_counter(); s.run_gn(20); best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
print("%s N %d: %.4f ms per iteration (wall, run_gn(20))" % (which, N, best))
