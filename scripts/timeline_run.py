"""The kernels of ONE steady-state Gauss-Newton iteration inside an untimed run_gn (no host synchronisation between iterations): run under
rocprofv3 --kernel-trace by scripts/timeline_run.sh.   python scripts/timeline_run.py <c2|c3|c5|c5b|c4|c4p> [N]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd as g
from gpslam_amd import synthetic as S
which = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
kw = {}
if which == "c2": p = S.linear_chain(N)
elif which == "c3": p = S.pose3_chain(N)
elif which == "c5": p = S.rot3_attitude_chain(N)
elif which == "c5b": p = S.pose3_gps_chain(N, keep_odometry=True)
elif which == "c4p":
    p = S.pose2_range_chain(N, L=8); kw = dict(chart=g.CHART_FIRST_ORDER, landmark_dim=2)
else:
    p = S.pose2_local_landmarks_chain(N, window=200); kw = dict(chart=g.CHART_FIRST_ORDER, landmark_dim=2)
s = S.apply(p, g.ChainSolver(p["kind"], **kw))
s.run_gn(3)
s.set_states(p["pose"], p["vel"])
if "landmarks" in p: s.set_landmarks(p["landmarks"])
s.run_gn(12)
