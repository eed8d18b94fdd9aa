"""BASELINE config 1 (Plaza2: 4091 SE(2) states, 1816 interpolated ranges, 4 landmarks) on the GPU: wall clock and device time per
Gauss-Newton / Levenberg-Marquardt iteration -- a graph this small is bound by launches, not by kernels.   python scripts/time_plaza.py"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd as g
from gpslam_amd import plaza
data = plaza.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "plaza2.npz"))
p = plaza.build_problem(data)
s = plaza.apply(p, g.ChainSolver(g.POSE2, chart=g.CHART_FIRST_ORDER, landmark_dim=2))
s.run_gn(2)
for rep in range(3):
    t0 = time.perf_counter(); st, ph = s.run_gn(50, timed=False) if False else (s.run_gn(50), None); t1 = time.perf_counter()
    print("run_gn(50): %.3f ms wall per iteration" % ((t1 - t0) / 50 * 1e3))
st, ph = s.run_gn(20, timed=True)
print("device phases per iteration (ms):", np.round(np.asarray(ph) / 20, 4))
lam = 1e-5
t0 = time.perf_counter()
for it in range(20):
    rc, st, lam = s.iterate_lm(lam)
t1 = time.perf_counter()
print("iterate_lm: %.3f ms wall per call" % ((t1 - t0) / 20 * 1e3))
