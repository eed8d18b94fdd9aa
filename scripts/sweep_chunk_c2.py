"""Config 2 (GaussianProcessPriorLinear<3>, block size 6) as a function of the level-0 chunk length:  python scripts/sweep_chunk_c2.py [N] [chunks...]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
chunks = [int(a) for a in sys.argv[2:]] or [0, 13, 17, 20, 25]
p = S.linear_chain(N)
for m in chunks:
    s = S.apply(p, gpslam_amd.ChainSolver(p["kind"], chunk=m))
    s.run_gn(2)
    s.set_states(p["pose"], p["vel"])
    best = None
    for _ in range(3):
        st, ph = s.run_gn(10, timed=True)
        ph = ph / 10
        if best is None or ph[4] < best[4]:
            best = ph
        s.set_states(p["pose"], p["vel"])
    print("chunk %3d: lin %.3f solve %.3f total %.3f ms   level 0 forward %.4f ms  levels %s" % (m, best[0], best[2], best[4], s.time_kernel(2, reps=5), s.plan_info()))
    s.close()
