"""Profiling target for the BASELINE configs other than 3: a few Gauss-Newton iterations of one factor mix.
   python scripts/profile_cfg.py <c2|c4|c4p|c5|c5b> [N] [fp32]
c2: GaussianProcessPriorLinear<3> chain; c4: SE(2) + odometry + interpolated ranges to N / 20 locally visible landmarks;
c4p: the same with 8 landmarks (dense border); c5: SO(3) + interpolated attitude x4; c5b: SE(3) + odometry + interpolated GPS x4"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd
from gpslam_amd import synthetic as S

which = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
fp32 = len(sys.argv) > 3 and sys.argv[3] == "fp32"
kw = {}
if which == "c2":
    p = S.linear_chain(N)
elif which == "c4":
    p = S.pose2_local_landmarks_chain(N, window=200)
    kw = dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2)
elif which == "c4p":
    p = S.pose2_range_chain(N, L=8)
    kw = dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2)
elif which == "c5":
    p = S.rot3_attitude_chain(N)
elif which == "c5b":
    p = S.pose3_gps_chain(N, keep_odometry=True)
else:
    raise SystemExit("unknown mix " + which)
if fp32:
    kw["precision"] = gpslam_amd.FP32
s = S.apply(p, gpslam_amd.ChainSolver(p["kind"], **kw))
s.run_gn(1)
s.set_states(p["pose"], p["vel"])
if "landmarks" in p:
    s.set_landmarks(p["landmarks"])
st, ph = s.run_gn(3, timed=True)
ph = ph / 3
print("%s N=%d ms/iter: lin %.3f asm %.3f solve %.3f retract+err %.3f total %.3f" % (which, N, ph[0], ph[1], ph[2], ph[3], ph[4]))
