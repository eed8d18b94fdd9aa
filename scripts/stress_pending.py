"""run_gn(K) -- every retraction but the last folded into the next K1 where the graph is eligible -- against K single iterations on
random chains of every manifold with random factor sets (missing GP priors, velocity priors anywhere, with and without odometry,
interpolated GPS / attitude factors): the states must be BIT-IDENTICAL whatever compile() decided.
   python scripts/stress_pending.py [count] [seed]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as O
import test_gpu_parity as T
import gpslam_amd as gp
from gpslam_amd import synthetic as S
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for t in range(cnt):
    kind = [O.POSE3, O.POSE2, O.ROT3, O.LINEAR3, O.POSE3, O.ROT3][t % 6]
    N = int(rng.integers(3, 3000))
    d = O.TANGENT_DIM[kind]
    chart = int(rng.integers(0, 2)) if kind in (O.POSE2, O.POSE3) else 0
    meas = None
    if kind == O.POSE3 and t % 6 == 4 and N >= 4:
        p = S.pose3_gps_chain(N, per_interval=int(rng.integers(1, 4)), seed=t, keep_odometry=bool(rng.integers(0, 2)))
        if "prior_idx" not in p or len(p["prior_idx"]) < 2:
            fix = np.arange(0, N, 15).astype(np.int32)
            p.update(prior_idx=fix, prior_pose=p["pose"][fix].copy(), prior_sig=np.full((len(fix), 6), 0.05))
        feed = lambda s, p=p: S.apply(p, s)
    elif kind == O.ROT3 and t % 6 == 5 and N >= 4:
        p = S.rot3_attitude_chain(N, seed=t)
        feed = lambda s, p=p: S.apply(p, s)
    else:
        c = T.random_chain(kind, N, 300 + t)
        gp_left = np.arange(N - 1)
        vp = np.unique(rng.integers(0, N, int(rng.integers(0, 4))))
        btw = rng.random() < 0.6 and kind != O.LINEAR3
        if btw and rng.random() < 0.5 and N > 5:
            gp_left = np.delete(gp_left, int(rng.integers(0, N - 1)))          # a state without an owner: no folding (the odometry keeps the chain connected)
        fix = np.arange(0, N, int(rng.integers(5, 40)))
        Qc = np.diag(0.01 + 0.02 * rng.random(d))
        def feed(s, c=c, gp_left=gp_left, vp=vp, btw=btw, fix=fix, Qc=Qc, kind=kind, N=N, d=d):
            s.set_qc(Qc); s.set_states(c["pose"], c["vel"])
            s.add_gp_priors(gp_left, c["dt"][gp_left])
            s.add_pose_priors(fix, c["truth_pose"][fix], np.full((len(fix), d), 0.01))
            if len(vp):
                s.add_vel_priors(vp, c["truth_vel"][vp], np.full((len(vp), d), 0.05))
            if btw:
                ident = {O.POSE2: np.zeros(3), O.POSE3: O.pose3((0, 0, 0), (0, 0, 0)), O.ROT3: O.rot3_ypr(0, 0, 0)}[kind]
                m = np.stack([O.retract(kind, ident, O.local(kind, c["truth_pose"][i], c["truth_pose"][i + 1])) for i in range(N - 1)])
                s.add_between(np.arange(N - 1), m, np.full((N - 1, d), 0.02))
            s.compile()
            return s
    K = int(rng.integers(2, 6))
    a = feed(gp.ChainSolver(kind, chart=chart))
    b = feed(gp.ChainSolver(kind, chart=chart))
    try:
        sa, _ = a.run_gn(K)
    except gp.GpslamHipError as ex:
        print("run_gn failed:", t, kind, N, K, chart, str(ex)[-60:], flush=True)
        try:
            for _ in range(K):
                _, sb = b.iterate_gn()
            print("  ... but single iterations succeed: BUG", flush=True)
            sys.exit(1)
        except gp.GpslamHipError:
            print("  ... and so do single iterations: an ill-posed graph, skipped", flush=True)
            continue
    for _ in range(K):
        _, sb = b.iterate_gn()
    (xa, va), (xb, vb) = a.get_states(), b.get_states()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb), (t, kind, N, K, np.abs(xa - xb).max(), np.abs(va - vb).max())
    assert (sa.error_before, sa.error_after, sa.delta_inf_norm) == (sb.error_before, sb.error_after, sb.delta_inf_norm), (t, kind, N, K)
    # ... and a second run on the same handle (the two state buffers have changed places an odd or even number of times)
    a.run_gn(2); b.iterate_gn(); b.iterate_gn()
    (xa, va), (xb, vb) = a.get_states(), b.get_states()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb), (t, kind, N, K, "second run")
    print("ok kind %d N %d K %d chart %d" % (kind, N, K, chart), flush=True)
    a.close(); b.close()
print("all %d runs bit-identical to single iterations" % cnt)
