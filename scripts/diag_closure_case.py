import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch  # noqa
from oracle import oracle as O
import test_gpu_parity as T
import gpslam_amd
SEED, KIND, NN, KK, MODE = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]) if len(sys.argv) > 5 else (405, 3, 2156, 4, 'gn')
rng = np.random.default_rng(SEED)
for t in range(400):
    kind = [O.POSE2, O.POSE3, O.ROT3, O.LINEAR3, O.POSE2, O.POSE3][t % 6]
    N = int(rng.integers(30, 2500)); d = O.TANGENT_DIM[kind]; cap = 27 // d
    K = int(rng.integers(1, cap + 1))
    chart = O.CHART_FIRST_ORDER if (kind == O.POSE2 and t % 2) else O.CHART_EXPMAP
    c = T.random_chain(kind, N, 300 + t)
    first = rng.integers(0, N, K); second = rng.integers(0, N, K)
    for k in range(K):
        while abs(int(second[k]) - int(first[k])) < 2: second[k] = rng.integers(0, N)
    if K >= 2:
        first[0], second[0] = 0, N - 1; first[1] = first[0]
        if abs(int(second[1]) - int(first[1])) < 2: second[1] = N // 2
    ident = {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0), O.LINEAR3: np.zeros(3)}[kind]
    rel = lambda i, j: O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][j]) + 0.01 * rng.standard_normal(d))
    cmeas = np.stack([rel(int(first[k]), int(second[k])) for k in range(K)])
    csig = 0.01 + 0.05 * rng.random((K, d)); Qc = np.diag(0.01 + 0.02 * rng.random(d))
    if not (kind == KIND and N == NN and K == KK): continue
    print("t", t, "first", first, "second", second)
    sol = []
    for make in (lambda: O.Chain(kind, chart), lambda: gpslam_amd.ChainSolver(kind, chart)):
        s = make(); s.set_qc(Qc); s.set_states(c["pose"], c["vel"]); s.add_gp_priors(np.arange(N - 1), c["dt"])
        fix = np.arange(0, N, 20); s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
        if t % 3 == 0: s.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, d), 0.05))
        if kind != O.LINEAR3 and t % 2 == 0:
            meas = np.stack([O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][i + 1])) for i in range(N - 1)])
            s.add_between(np.arange(N - 1), meas, np.full((N - 1, d), 0.02))
        s.add_between_pairs(first, second, cmeas, csig); s.compile(); sol.append(s)
    orc, dev = sol
    if MODE == 'lm':
        lam0 = lam1 = 1e-2
        for it in range(12):
            rc0, s0, lam0 = orc.iterate_lm(lam0); rc1, s1, lam1 = dev.iterate_lm(lam1)
            (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
            dx = max(float(np.abs(O.local(kind, x0[i], x1[i])).max()) for i in range(0, N, 7))
            print("lm %d: before orc %.12e dev %.12e rel %.2e | after rel %.2e | lambda %.1e %.1e accepted %d %d | states apart %.2e vel %.2e" % (it, s0.error_before, s1.error_before, abs(s0.error_before - s1.error_before) / s0.error_before, abs(s0.error_after - s1.error_after) / s0.error_after, lam0, lam1, s0.accepted, s1.accepted, dx, np.abs(v0 - v1).max()))
        break
    for it in range(8):
        (rc0, s0), (rc1, s1) = orc.iterate_gn(), dev.iterate_gn()
        (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
        dx = max(float(np.abs(O.local(kind, x0[i], x1[i])).max()) for i in range(0, N, 7))
        print("it %d: err before %.6e after orc %.12e dev %.12e rel %.2e | |delta| orc %.3e dev %.3e | states apart %.2e vel %.2e" % (it, s0.error_before, s0.error_after, s1.error_after, abs(s0.error_after - s1.error_after) / s0.error_after, s0.delta_inf_norm, s1.delta_inf_norm, dx, np.abs(v0 - v1).max()))
    break
