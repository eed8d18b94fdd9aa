"""Random chain lengths through the default plan against the oracle (3 Gauss-Newton iterations, 1e-9, then 6 Levenberg-Marquardt
iterations in lock step, the last ones past convergence): the shapes the fixed-size
tests do not name -- ragged last chunks, groups that are not full, chains barely longer than one level.
   python scripts/stress_sizes.py [count] [seed]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
from oracle import oracle as O
import test_gpu_parity as T
import lm_lockstep as L
import gpslam_amd
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for t in range(cnt):
    kind = [O.POSE3, O.POSE3, O.POSE2, O.ROT3, O.LINEAR3][t % 5]
    N = int(rng.integers(40, 6000))
    d = O.TANGENT_DIM[kind]
    chart = O.CHART_FIRST_ORDER if kind == O.POSE2 else O.CHART_EXPMAP
    c = T.random_chain(kind, N, 100 + t)
    Qc = np.diag(0.01 + 0.02 * rng.random(d))
    if t % 3 == 0 and d > 1:
        Qc[0, 1] = Qc[1, 0] = 0.003
    sol, solvers = [], []
    for make in (lambda: O.Chain(kind, chart), lambda: gpslam_amd.ChainSolver(kind, chart)):
        s = make()
        s.set_qc(Qc); s.set_states(c["pose"], c["vel"])
        s.add_gp_priors(np.arange(N - 1), c["dt"])
        fix = np.arange(0, N, 20)
        s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
        if t % 2:
            s.add_vel_priors([0, N - 1], c["truth_vel"][[0, N - 1]], np.full((2, d), 0.05))
        if kind not in (O.LINEAR3,):
            ident = {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0)}[kind]
            meas = np.stack([O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][i + 1])) for i in range(N - 1)])
            s.add_between(np.arange(N - 1), meas, np.full((N - 1, d), 0.02))
        s.compile()
        for _ in range(3):
            s.iterate_gn()
        sol.append(s.get_states())
        solvers.append(s)
    T.states_close(kind, sol[0][0], sol[0][1], sol[1][0], sol[1][1], 1e-9)
    # Levenberg-Marquardt from the initial values, in lock step: the same lambda schedule, the same errors
    # (tests/lm_lockstep.py: exact while the cost moves; past convergence -- the last iterations here, on the linear chains
    # from the third on -- one trial per call and lambda kept or divided once)
    for s_ in solvers:
        s_.set_states(c["pose"], c["vel"])
    lam, n_noise, slack = L.run(solvers[0], solvers[1], 1e-2, 6, tag=(kind, N))
    sol = [s_.get_states() for s_ in solvers]
    # (a step kept on a rounding-level decision by one side only moves its values by |delta|_inf of that step: `slack`)
    assert slack <= 1e-4, (kind, N, slack)
    T.states_close(kind, sol[0][0], sol[0][1], sol[1][0], sol[1][1], 1e-9 + 2 * slack)
    print("ok kind %d N %d (LM: lambda %.1e, %d of 6 calls decided at rounding level, slack %.1e)" % (kind, N, lam, n_noise, slack))
print("all %d sizes agree with the oracle" % cnt)
