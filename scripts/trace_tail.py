"""Timeline of the persistent tail of the fat blocks' cyclic reduction (a -DGPS_TRACE_TAIL build loaded through GPSLAM_LIB): per task
of every workgroup the stamps fetched / waited / body done / published (s_memrealtime, 10 ns ticks).
   GPSLAM_HIPCC_FLAGS=-DGPS_TRACE_TAIL python scripts/trace_tail.py [N]"""
import ctypes as C, os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd as gp
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = S.pose2_local_landmarks_chain(N, window=200)
s = S.apply(p, gp.ChainSolver(gp.POSE2, chart=gp.CHART_FIRST_ORDER, landmark_dim=2))
s.run_gn(3)
buf = np.zeros((320, 48, 5), dtype=np.uint64)
rc = s.lib.gpslam_hip_debug_tail_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
on = t[:, :, 4] > 0
t0 = t[:, :, 0][on].min()
print("workgroups with tasks: %d, tasks %d, launch span %.1f us" % (on.any(axis=1).sum(), on.sum(), (t[:, :, 3][on].max() - t0) / 100.0))
names = {1: "elim", 2: "update", 3: "top", 4: "back"}
# workgroup 0 runs elimination 0 of every level, the last block and back-substitution 0 of every level: the critical path
for wg in (0, 1):
    print("workgroup %d:" % wg)
    for k in range(48):
        if not on[wg, k]: break
        f, w, d, pb = (t[wg, k, :4] - t0) / 100.0
        print("  task %2d %-6s fetched %7.2f  waited +%5.2f  body +%5.2f  published +%5.2f us" % (k, names[int(t[wg, k, 4])], f, w - f, d - w, pb - d))
for ty in (1, 2, 4):
    sel = on & (t[:, :, 4] == ty)
    if sel.any():
        print("%-6s: wait med %.2f p90 %.2f | body med %.2f p90 %.2f | publish med %.2f p90 %.2f us (%d tasks)" % (
            names[ty], *[f(x) for x in ((t[:, :, 1] - t[:, :, 0])[sel] / 100.0, (t[:, :, 2] - t[:, :, 1])[sel] / 100.0, (t[:, :, 3] - t[:, :, 2])[sel] / 100.0) for f in (np.median, lambda v: np.percentile(v, 90))], sel.sum()))
