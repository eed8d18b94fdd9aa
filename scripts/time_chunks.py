"""ms per Gauss-Newton iteration of BASELINE config 3 for a few level-0 chunk lengths / upper chunk lengths / top sizes."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpslam_amd
from gpslam_amd import synthetic as S
p = S.pose3_chain(100000)
for chunk, up, top in [(0, 0, 0), (24, 0, 0), (26, 0, 0), (33, 0, 0), (49, 0, 0), (98, 0, 0), (49, 0, 8), (98, 0, 8), (13, 0, 0)]:
    s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3, chunk=chunk, upper_chunk=up, top_blocks=top))
    s.run_gn(3)
    s.set_states(p["pose"], p["vel"])
    st, ph = s.run_gn(20, timed=True)
    print("chunk %2d upper %d top %2d: %.4f ms/iter (lin %.3f asm %.3f solve %.3f retract %.3f)  plan %s" % (chunk, up, top, ph[4] / 20, ph[0] / 20, ph[1] / 20, ph[2] / 20, ph[3] / 20, list(s.plan_info().values()) if hasattr(s.plan_info(), "values") else s.plan_info()))
    s.close()
