#!/usr/bin/env python3
"""Wall time of Levenberg-Marquardt iterations (the reference scripts' loop: one accepted trial per iterate()) next to
Gauss-Newton on the benchmark chain.  python scripts/time_lm.py [states]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = S.pose3_chain(N)
s = S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
lam = 1e-5
s.iterate_lm(lam)
s.set_states(p["pose"], p["vel"])
t0 = time.perf_counter(); n = 0
for _ in range(8):
    rc, st, lam = s.iterate_lm(lam)[:3]; n += 1
t1 = time.perf_counter()
print("LM: %d iterations, %.3f ms each (incl. host decisions), error %.6g, lambda %.3g" % (n, (t1 - t0) / n * 1e3, st.error_after, lam))
s.set_states(p["pose"], p["vel"])
t0 = time.perf_counter()
for _ in range(8):
    rc, st = s.iterate_gn()
t1 = time.perf_counter()
print("GN with statistics: %.3f ms each, error %.6g" % ((t1 - t0) / 8 * 1e3, st.error_after))
