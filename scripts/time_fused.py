import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd
from gpslam_amd import synthetic as S
p = S.pose3_chain(int(sys.argv[1]) if len(sys.argv) > 1 else 100000)
s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
print("k_fused_level0 %.4f ms" % min(s.time_kernel(2, reps=5) for _ in range(3)))
