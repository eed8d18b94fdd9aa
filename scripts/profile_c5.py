"""Profiling target: Gauss-Newton iterations of the C5 mix (SO(3) chain + interpolated attitude factors at 4x the state rate)."""
import sys; sys.path.insert(0, '.')
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = S.rot3_attitude_chain(N)
s = S.apply(p, gpslam_amd.ChainSolver(p["kind"]))
st, _ = s.run_gn(3)
print('ok', st.error_after, st.delta_inf_norm)
