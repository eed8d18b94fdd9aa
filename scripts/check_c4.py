"""Config 4 at full size on one GPU (1e6 SE(2) states + 5e4 locally visible range landmarks): convergence of LM / GN,
bit-identical reruns, and agreement between two different segmentations (different elimination trees, both exact)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpslam_amd
from gpslam_amd import synthetic as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = S.pose2_local_landmarks_chain(N, anchor=int(os.environ.get("ANCHOR", "4096")))
mk = lambda seg: S.apply(p, gpslam_amd.ChainSolver(p["kind"], chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2, segment_length=seg))
s = mk(0)
print("plan", s.segment_plan())
t0 = time.time()
lam = 1e-5
for it in range(int(os.environ.get("ITERS", "15"))):
    rc, st, lam = s.iterate_lm(lam)
    print("LM %2d err %.6e -> %.6e |delta| %.3e lambda %.1e accepted %d" % (it, st.error_before, st.error_after, st.delta_inf_norm, lam, st.accepted))
    if st.accepted and st.delta_inf_norm < 1e-6:
        break
print("wall %.2fs" % (time.time() - t0))
x1, v1 = s.get_states(); l1 = s.get_landmarks()
tr = p["truth"]
print("position rmse vs truth %.3f m, landmark rmse %.3f m" % (np.sqrt(np.mean(np.sum((x1[:, :2] - tr[:, :2]) ** 2, 1))), np.sqrt(np.mean(np.sum((l1 - p["landmark_truth"]) ** 2, 1)))))
s.set_states(p["pose"], p["vel"]); s.set_landmarks(p["landmarks"])
for it in range(12):
    rc, st = s.iterate_gn()
    print("GN %2d err %.6e -> %.6e |delta| %.3e" % (it, st.error_before, st.error_after, st.delta_inf_norm))
    if st.delta_inf_norm < 1e-6:
        break
x1, v1 = s.get_states(); l1 = s.get_landmarks()
print("position rmse vs truth %.3f m, landmark rmse %.3f m" % (np.sqrt(np.mean(np.sum((x1[:, :2] - tr[:, :2]) ** 2, 1))), np.sqrt(np.mean(np.sum((l1 - p["landmark_truth"]) ** 2, 1)))))
