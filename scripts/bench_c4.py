"""BASELINE config 4 on one GPU: SE(2) chain + odometry + interpolated ranges to N / 20 locally visible landmarks
(segmented landmark elimination, fatsep.hpp).  python scripts/bench_c4.py [N] [window] [segment_length] [density]
density: landmarks per N / 20 states (1 = config 4; 3 and beyond: fat blocks past 80 columns, k_fat_elim_wide)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import gpslam_amd
from gpslam_amd import synthetic as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
window = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seglen = int(sys.argv[3]) if len(sys.argv) > 3 else 0
density = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
t0 = time.time()
p = S.pose2_local_landmarks_chain(N, window=window, L=max(int(density * N / 20), 1))
t1 = time.time()
s = S.apply(p, gpslam_amd.ChainSolver(p["kind"], chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, segment_length=seglen))
t2 = time.time()
s.run_gn(1)
s.set_states(p["pose"], p["vel"])
s.set_landmarks(p["landmarks"])
st, ph = s.run_gn(3, timed=True)
ph = ph / 3
try:
    print("plan:", s.segment_plan())
except Exception as ex:
    print("plan: n/a", ex)
print("C4 N=%d L=%d ranges=%d window=%d  ms/iter: lin %.3f asm %.3f solve %.3f retract+err %.3f total %.3f -> %.3g state-iter/s (gen %.1fs, setup %.1fs)"
      % (N, len(p["landmarks"]), len(p["range_left"]), window, ph[0], ph[1], ph[2], ph[3], ph[4], N / (ph[4] * 1e-3), t1 - t0, t2 - t1))
