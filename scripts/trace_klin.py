"""Where a K1 wave's time goes (a -DGPS_TRACE_KLIN build loaded through GPSLAM_LIB): stamps inside gp_pose3_record of the first 4096
waves of the last k_lin launch (s_memrealtime, 10 ns ticks).   GPSLAM_LIB=build_ab/lib_trace_klin.so python scripts/trace_klin.py [N]"""
import ctypes as C, os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd as gp
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = S.pose3_chain(N)
s = S.apply(p, gp.ChainSolver(gp.POSE3))
s.run_gn(3)
buf = np.zeros((4096, 16), dtype=np.uint64)
rc = s.lib.gpslam_hip_debug_klin_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]
nw = min(len(t), (N - 1 + 63) // 64)
t = t[:nw]
t0 = t[:, 0].min()
names = ["start", "states loaded", "between + Log", "Jinv, error, whitening", "error line stored", "X, J blocks stored", "difference quotient (12 evaluations)", "F blocks stored (end)"]
print("waves %d; launch: first start -> last end %.2f us" % (len(t), (t[:, 7].max() - t0) / 100.0))
for k in range(1, 8):
    d = (t[:, k] - t[:, k - 1]) / 100.0
    print("  %-40s min %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % (names[k], d.min(), np.median(d), np.percentile(d, 90), d.max()))
life = (t[:, 7] - t[:, 0]) / 100.0
print("  %-40s min %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % ("wave life", life.min(), np.median(life), np.percentile(life, 90), life.max()))
st = (t[:, 0] - t0) / 100.0
print("  %-40s min %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % ("wave start after launch start", st.min(), np.median(st), np.percentile(st, 90), st.max()))
