"""Per-iteration device time of the other BASELINE configs' factor mixes on one GPU (not the bench.py contract line):
  C2  GaussianProcessPriorLinear<3> chain            C4' SE(2) + odometry + interpolated ranges, dense landmark border
  C5  SO(3) + interpolated attitude factors at 4x the state rate
python scripts/bench_configs.py [N]"""
import sys; sys.path.insert(0, '.')
import time
import numpy as np
import gpslam_amd
from gpslam_amd import synthetic as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for name, make, kw in (("C2 linear3", lambda: S.linear_chain(N), {}),
                       ("C4' pose2+ranges (L=8)", lambda: S.pose2_range_chain(N, L=8), dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2)),
                       ("C5 rot3+attitude x4", lambda: S.rot3_attitude_chain(N), {}),
                       ("C5b pose3+gps x4", lambda: S.pose3_gps_chain(N), {})):
    t0 = time.time()
    p = make()
    s = S.apply(p, gpslam_amd.ChainSolver(p["kind"], **kw))
    s.run_gn(1)
    s.set_states(p["pose"], p["vel"])
    if "landmarks" in p:
        s.set_landmarks(p["landmarks"])
    st, ph = s.run_gn(3, timed=True)
    ph = ph / 3
    nf = len(p.get("range_left", [])) + len(p.get("att_left", [])) + len(p.get("gps_left", []))
    print("%-26s N=%d meas=%d  ms/iter: lin %.3f asm %.3f solve %.3f retract+err %.3f total %.3f  -> %.3g state-iter/s  (setup %.1fs)"
          % (name, N, nf, ph[0], ph[1], ph[2], ph[3], ph[4], N / (ph[4] * 1e-3), time.time() - t0))
    s.close()
