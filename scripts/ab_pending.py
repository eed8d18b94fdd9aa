"""The retraction folded into the next K1 (default) against a k_retract launch per iteration (GPSLAM_PLAN_SEPARATE_RETRACT):
ms per Gauss-Newton iteration of run_gn(8), same process.   python scripts/ab_pending.py [config2|config3|config5|config5b] [N]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd as gp
from gpslam_amd import synthetic as S
which = sys.argv[1] if len(sys.argv) > 1 else "config2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
p = {"config2": S.linear_chain, "config3": S.pose3_chain, "config5": S.rot3_attitude_chain,
     "config5b": lambda n: S.pose3_gps_chain(n, keep_odometry=True)}[which](N)
for rep in range(2):
    for name, plan in (("folded", 0), ("separate", gp.PLAN_SEPARATE_RETRACT)):
        s = S.apply(p, gp.ChainSolver(p["kind"], plan=plan))
        s.run_gn(2)
        s.set_states(p["pose"], p["vel"])
        st, ph = s.run_gn(8, timed=True)
        print(which, N, name, "ms/iter %.4f" % (ph[4] / 8), {k: round(float(v) / 8, 4) for k, v in zip(["lin", "asm", "solve", "retract"], ph)}, flush=True)
        s.close()
