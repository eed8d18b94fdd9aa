#!/usr/bin/env python3
"""matlab/GPAHRSexample.m end to end on the GPU (tests/golden/ahrs_imu.npz): prints what the script prints / plots."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpslam_amd as g  # noqa: E402
from gpslam_amd import ahrs  # noqa: E402

data = ahrs.load(os.path.join(ROOT, "tests", "golden", "ahrs_imu.npz"))
t0 = time.perf_counter()
p = ahrs.build_problem(data)
t1 = time.perf_counter()
pose, vel = ahrs.initial_values(p)
s = g.ChainSolver(g.ROT3_BIAS)
s.set_states(pose, vel)
ahrs.apply(p, s, gyro_only=True)
rc, st = ahrs.optimize_default(s, s.default_params(use_lm=1))
gp, _ = s.get_states()
t2 = time.perf_counter()
pose[:, :9] = gp[:, :9]
f = g.ChainSolver(g.ROT3_BIAS)
f.set_states(pose, vel)
ahrs.apply(p, f)
e0 = f.error()
t3 = time.perf_counter()
it, trace = ahrs.iterate_until(f)
t4 = time.perf_counter()
fp, fv = f.get_states()
gt = ahrs.ground_truth_ypr(data, p["state_time"])
wrap = lambda x: np.arctan2(np.sin(x), np.cos(x))
est = np.array([ahrs.rot_ypr(r[:9]) for r in fp])
gyr = np.array([ahrs.rot_ypr(r[:9]) for r in gp])
print("states %d, gyro factors %d, accelerometer factors %d (%d between states)" % (
    p["N"], len(p["ahrs_left"]), len(p["att_left"]), int(np.sum(p["att_tau"] < p["att_dt"] - 1e-12))))
print("graph building (host, numpy) %.3f s; gyro-only LM: %d iterations, error %.3g, %.3f s" % (t1 - t0, st.iterations, st.error_after, t2 - t1))
print("full graph: initial error %.6g, %d LM iterations -> %.6g in %.3f s (%.2f ms per iteration incl. error readback)" % (
    e0, it, trace[-1], t4 - t3, (t4 - t3) / max(it, 1) * 1e3))
print("pitch / roll rms vs motion capture [rad]: estimated %s, gyro-only %s" % (
    np.sqrt(np.mean(wrap(est[:, 1:] - gt[:, 1:]) ** 2, axis=0)), np.sqrt(np.mean(wrap(gyr[:, 1:] - gt[:, 1:]) ** 2, axis=0))))
print("bias at the end %s, max |bias| %.3g" % (fp[-1, 9:], np.abs(fp[:, 9:]).max()))
st, ph = f.run_gn(3, timed=True)
print("device time of one Gauss-Newton iteration at the solution: %.3f ms (linearise %.3f, assemble+eliminate %.3f, solve %.3f, retract %.3f)" % (
    ph[4] / 3, ph[0] / 3, ph[1] / 3, ph[2] / 3, ph[3] / 3))
