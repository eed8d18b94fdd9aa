"""Timeline of the upper solver levels (k_multi_forward, a -DGPS_TRACE_UPPER build loaded through GPSLAM_LIB): stamps of the eight
waves of workgroup 0 in the last group launch and in the TOP launch (s_memrealtime, 10 ns ticks).
   GPSLAM_LIB=build_ab/lib_trace_up.so python scripts/trace_upper.py [N]"""
import ctypes as C, os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd as gp
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = S.pose3_chain(N)
s = S.apply(p, gp.ChainSolver(gp.POSE3))
s.run_gn(3)
buf = np.zeros((2, 8, 64), dtype=np.uint64)
rc = s.lib.gpslam_hip_debug_upper_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
names = {0: "start", 1: "records loaded + written to LDS", 2: "past __syncthreads"}
for q in range(5):
    for k, nm in enumerate(["compute", "barrier 1", "store_own", "barrier 2", "add_right", "barrier 3"]):
        names[3 + 6 * q + k] = "sub-level %d: %s" % (q, nm)
names.update({40: "records stored (end)", 41: "TOP: last block solved", 42: "TOP: group back-substituted", 43: "TOP: solutions stored (end)"})
for top in (0, 1):
    print("=== %s launch, workgroup 0: time since the workgroup's first stamp (us), per wave; then the step's duration for wave 0" % ("TOP" if top else "group"))
    t0 = t[top, :, 0][t[top, :, 0] > 0].min()
    prev = None
    for slot in sorted(names):
        v = t[top, :, slot]
        if (v > 0).sum() == 0:
            continue
        us = (v - t0) / 100.0
        d = "" if prev is None else "  (+%.2f)" % (us[0] - prev)
        prev = us[0]
        print("  %-34s %s%s" % (names[slot], " ".join("%7.2f" % x for x in us), d))
