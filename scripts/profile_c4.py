"""Profiling target: a few Gauss-Newton iterations of the C4' mix (SE(2) + odometry + interpolated ranges, L = 8)."""
import sys; sys.path.insert(0, '.')
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = S.pose2_range_chain(N, L=8)
s = S.apply(p, gpslam_amd.ChainSolver(p["kind"], chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2))
st, _ = s.run_gn(4)
print('ok', st.error_after, st.delta_inf_norm)
