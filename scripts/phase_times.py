"""Phase times of a Pose3 chain's Gauss-Newton iteration, one line per call (run it several times: the question is what moves
from process to process):  python scripts/phase_times.py [N] [iterations]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = S.pose3_chain(N)
s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
s.run_gn(2)
out = []
for rep in range(3):
    s.set_states(p['pose'], p['vel'])
    st, ph = s.run_gn(K, timed=True)
    ph = ph / K
    out.append("lin %.3f solve %.3f retract %.3f total %.3f" % (ph[0], ph[2], ph[3], ph[4]))
print("N=%d | " % N + " | ".join(out))
