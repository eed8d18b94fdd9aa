"""BASELINE config 5's SE(3) mix (GP prior + odometry + interpolated GPS x4) at N states: the interpolated rows as 16-double lines
(default plan, k_fused_level0<4>) against 24-column rows (GPSLAM_PLAN_MEAS_ROWS, <3>): phase times per Gauss-Newton iteration.
   python scripts/ab_c5b.py [N]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = S.pose3_gps_chain(N, keep_odometry=True)
res = {}
for name, plan in (("lines", 0), ("rows", gpslam_amd.PLAN_MEAS_ROWS)):
    s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3, plan=plan))
    s.run_gn(2)
    s.set_states(p["pose"], p["vel"])
    st, ph = s.run_gn(4, timed=True)
    res[name] = (s.get_states(), st.error_after)
    print(name, "plan", s.plan_info()["structured_gp"], "ms/iter %.3f" % (ph[4] / 4), {k: round(float(v) / 4, 3) for k, v in zip(["lin", "asm", "solve", "retract", "total"], ph)},
          "level0 %.3f" % (s.last_level0_ms() / 4), "error %.9e" % st.error_after)
    s.close()
(xa, va), (xb, vb) = res["lines"][0], res["rows"][0]
print("max |difference| pose %.2e vel %.2e" % (np.abs(xa - xb).max(), np.abs(va - vb).max()))
