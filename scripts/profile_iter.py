"""Minimal profiling target: a few Gauss-Newton iterations of BASELINE config 3 (1e5 Pose3 states).
   python scripts/profile_iter.py [N] [upper_chunk]"""
import sys; sys.path.insert(0, '.')
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
uc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = S.pose3_chain(N)
s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3, upper_chunk=uc))
s.run_gn(2)
s.set_states(p['pose'], p['vel'])
st, _ = s.run_gn(5)
print('ok', st.error_after, st.delta_inf_norm)
