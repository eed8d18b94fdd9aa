"""Timeline of k_fused_level0's waves (a -DGPS_TRACE_FUSED build of the library, loaded through GPSLAM_LIB):
when each wave starts, passes the hand-over barriers P / Q, each block step, and ends (s_memrealtime, 10 ns ticks), and
where it ran (XCC / SE / CU / SIMD).   GPSLAM_LIB=build_ab/lib_trace.so python scripts/trace_fused.py [N] [out.npz | -] [c3 | c5b]
Build:  python -c "from gpslam_amd import build; print(build.build(extra=['-DGPS_TRACE_FUSED']))"  (copy the result to build_ab/)."""
import ctypes as C, os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd as gp
from gpslam_amd import synthetic as S

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
mix = sys.argv[3] if len(sys.argv) > 3 else "c3"      # c3: BASELINE config 3; c5b: config 5's SE(3) mix (k_fused_level0<4>)
p = S.pose3_gps_chain(N, keep_odometry=True) if mix == "c5b" else S.pose3_chain(N)
s = S.apply(p, gp.ChainSolver(gp.POSE3))
s.run_gn(3)
s.set_states(p["pose"], p["vel"])
s.run_gn(2)                                  # the trace holds the LAST launch (inside a run: K1 in front, upper levels behind)
nw = C.c_int32(0)
cap = 1 << 16
buf = np.zeros((cap, 64), dtype=np.uint64)
rc = s.lib.gpslam_hip_debug_fused_trace(s._h, buf.ctypes.data_as(C.c_void_p), cap, C.byref(nw))
assert rc == 0, rc
n = min(nw.value, cap)
t = buf[:n].astype(np.int64)
elim, asm = t[0::2], t[1::2]
t0 = t[:, 0][t[:, 0] > 0].min()
us = lambda a: (a - t0) / 100.0             # 100 MHz
steps = int(((elim[0, 3:53] > 0).sum()))
print("waves %d, workgroups %d, block steps %d" % (n, n // 2, steps))
def q(name, a):
    a = np.asarray(a, dtype=np.float64)
    print("  %-34s min %8.2f  p10 %8.2f  med %8.2f  p90 %8.2f  max %8.2f us" % (name, a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max()))
q("ELIM start", us(elim[:, 0])); q("ASM start", us(asm[:, 0]))
q("ASM images 0,1 ready (P)", us(asm[:, 1]) - us(asm[:, 0]))
q("ELIM past Q - start", us(elim[:, 2]) - us(elim[:, 0]))
last = 3 + min(steps, 50) - 1
q("ELIM steps total", us(elim[:, last]) - us(elim[:, 2]))
q("ELIM per step", (us(elim[:, last]) - us(elim[:, 3])) / max(steps - 1, 1))
q("ELIM tail (after last step)", us(elim[:, 60]) - us(elim[:, last]))
q("ELIM end", us(elim[:, 60])); q("ASM end", us(asm[:, 60]))
q("ELIM life", us(elim[:, 60]) - us(elim[:, 0])); q("ASM life", us(asm[:, 60]) - us(asm[:, 0]))
hw = elim[:, 61]; xcc = elim[:, 62] & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
hwa = asm[:, 61]; simda = (hwa >> 4) & 3
cuid = (xcc * 8 + se) * 32 + sh * 16 + cu
ids, cnt = np.unique(cuid, return_counts=True)
print("  distinct CUs %d; workgroups per CU: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
print("  workgroups per XCC:", dict(zip(*np.unique(xcc, return_counts=True))))
key = cuid * 4 + simd
pair = {}
for k, kind in list(zip(cuid * 4 + simd, ["E"] * len(simd))) + list(zip(cuid * 4 + simda, ["A"] * len(simda))):
    pair.setdefault(int(k), []).append(kind)
mix = {}
for v in pair.values():
    kk = "".join(sorted(v)); mix[kk] = mix.get(kk, 0) + 1
print("  waves per SIMD by role:", mix)
# end time against the number of workgroups sharing the CU
per_cu = dict(zip(ids, cnt))
for c in sorted(set(cnt)):
    sel = np.array([per_cu[x] == c for x in cuid])
    q("ELIM end, CU holds %d workgroups" % c, us(elim[sel, 60]))
dur = us(elim[:, 60]).max() - min(us(elim[:, 0]).min(), us(asm[:, 0]).min())
print("  first start -> last end: %.2f us" % dur)
# fine stamps of one block step (builds that carry them): assembly wave of image 13, elimination wave of step 10
if (asm[:, 40] > 0).any():
    names = ["shuffles + moves", "reconstruct (waits for the GP record)", "12 GP rows", "6 between rows (waits for its record)", "odd / compact rows", "damping", "open next state"]
    for k, nm in enumerate(names):
        q("ASM image 13: " + nm, us(asm[:, 41 + k]) - us(asm[:, 40 + k]))
    q("ASM image 13: total", us(asm[:, 46]) - us(asm[:, 40]))
if (elim[:, 48] > 0).any():
    names = ["Gauss-Jordan", "scale + record to LDS + wait at the barrier", "products with V_j", "stores + transpose", "products with U_j"]
    for k, nm in enumerate(names):
        q("ELIM step 10: " + nm, us(elim[:, 49 + k]) - us(elim[:, 48 + k]))
    q("ELIM step 10: total", us(elim[:, 53]) - us(elim[:, 48]))
if out:
    np.savez_compressed(out, trace=t)
