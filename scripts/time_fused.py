"""k_fused_level0 alone (gpslam_hip_time_kernel) on the BASELINE config-3 chain:  python scripts/time_fused.py [N] [chunk]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd
from gpslam_amd import synthetic as S
p = S.pose3_chain(int(sys.argv[1]) if len(sys.argv) > 1 else 100000)
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3, chunk=chunk))
print("k_fused_level0 %.4f ms (chunk %d)" % (min(s.time_kernel(2, reps=5) for _ in range(3)), s.plan_info()["chunk0"] if hasattr(s, "plan_info") else chunk))
