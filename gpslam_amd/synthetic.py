"""Seeded synthetic GP-SLAM chains for the named benchmark configurations.

Pure numpy, no oracle and no GPU: this only PRODUCES problem descriptions (dicts of arrays).  `apply()` feeds a
description to anything with the ChainSolver call surface (gpslam_amd.ChainSolver on the GPU, oracle.Chain in
the tests and in bench.py's cpu_baseline leg), so both sides see bit-identical inputs.

  C2  GaussianProcessPriorLinear<3> chain, N = 1e5, dt = 0.1, Qc = 0.01 I (gpslam/gp/tests/
      testGaussianProcessPriorLinear.cpp:32-33), prior on (x0, v0), position fix on every 10th state
  C3  GaussianProcessPriorPose3 chain, N = 1e5, dt = 0.1, Qc = 0.01 I6 (testGaussianProcessPriorPose3.cpp:29-31),
      BetweenFactor<Pose3> odometry from truth + noise, prior on x0, dead-reckoned initial values with zero
      velocity (the recipe of matlab/PlazaPose2.m:125,:196-202 lifted to SE(3))
"""
import numpy as np

LINEAR2, LINEAR3, POSE2, POSE3, ROT3 = 0, 1, 2, 3, 4
SEED_BASE = 0x6770736C616D  # "gpslam"


def _skew(w):
    z = np.zeros(w.shape[:-1])
    return np.stack([np.stack([z, -w[..., 2], w[..., 1]], -1), np.stack([w[..., 2], z, -w[..., 0]], -1),
                     np.stack([-w[..., 1], w[..., 0], z], -1)], -2)


def so3_exp(w):
    """Rodrigues, batched: (..., 3) -> (..., 3, 3)"""
    th2 = np.sum(w * w, -1)
    th = np.sqrt(np.maximum(th2, 1e-300))
    W = _skew(w)
    a = np.where(th2 > 1e-16, np.sin(th) / th, 1.0 - th2 / 6.0)
    b = np.where(th2 > 1e-16, (1.0 - np.cos(th)) / np.maximum(th2, 1e-300), 0.5 - th2 / 24.0)
    return np.eye(3) + a[..., None, None] * W + b[..., None, None] * (W @ W)


def se3_exp(xi):
    """Pose3::Expmap, batched: (..., 6) [omega, v] -> (R (..., 3, 3), t (..., 3))"""
    w, v = xi[..., :3], xi[..., 3:]
    R = so3_exp(w)
    th2 = np.sum(w * w, -1)
    wxv = np.cross(w, v)
    tpar = w * np.sum(w * v, -1, keepdims=True)
    safe = np.maximum(th2, 1e-300)[..., None]
    t_big = (wxv - np.einsum("...ij,...j->...i", R, wxv) + tpar) / safe
    t = np.where((th2 > 2.220446049250313e-16)[..., None], t_big, v)
    return R, t


def se3_prefix(R, t):
    """Inclusive prefix products T_0, T_0 T_1, ... of a batch of SE(3) elements (N x 3 x 3, N x 3)."""
    R, t = R.copy(), t.copy()
    n, shift = len(R), 1
    while shift < n:
        Ra, ta = R[:-shift], t[:-shift]
        t[shift:] = ta + np.einsum("nij,nj->ni", Ra, t[shift:])
        R[shift:] = Ra @ R[shift:]
        shift *= 2
    return R, t


def flat_pose3(R, t):
    return np.concatenate([R.reshape(R.shape[:-2] + (9,)), t], -1)


def pose3_chain(N, seed=0, dt=0.1, qc=0.01, sigma_odo=1e-3, sigma_prior=1e-3):
    """Config C3 (N = 100000 for the benchmark)."""
    rng = np.random.default_rng(SEED_BASE + 3 + seed)
    i = np.arange(N - 1)
    base = np.array([0.0, 0.0, 0.3, 1.0, 0.0, 0.0])
    pert = np.stack([0.05 * np.sin(0.013 * i), 0.04 * np.cos(0.017 * i), 0.06 * np.sin(0.007 * i + 1.0),
                     0.2 * np.sin(0.011 * i), 0.1 * np.cos(0.019 * i), 0.1 * np.sin(0.005 * i)], -1)
    twist = base + pert                                    # body twist on interval i
    Rs, ts = se3_exp(dt * twist)                           # true relative motions
    nR, nt = se3_exp(sigma_odo * rng.standard_normal((N - 1, 6)))
    mR = Rs @ nR                                           # measured odometry = true * Exp(noise)
    mt = ts + np.einsum("nij,nj->ni", Rs, nt)
    # dead reckoning from the measured odometry (initial values), starting at the prior pose:
    # inclusive prefix product of the measured relative motions (Hillis-Steele doubling, vectorised)
    R0, t0 = np.eye(3), np.zeros(3)
    PR, Pt = se3_prefix(mR, mt)
    pose = np.zeros((N, 12))
    pose[0] = flat_pose3(R0, t0)
    pose[1:] = flat_pose3(PR, Pt)
    return dict(kind=POSE3, name="C3 pose3 GP prior + synthetic odometry", N=N, qc=qc * np.eye(6),
                pose=pose, vel=np.zeros((N, 6)),
                gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.full(N - 1, dt),
                between_left=np.arange(N - 1, dtype=np.int32), between_meas=flat_pose3(mR, mt),
                between_sig=np.full((N - 1, 6), sigma_odo),
                prior_idx=np.array([0], dtype=np.int32), prior_pose=flat_pose3(R0, t0)[None],
                prior_sig=np.full((1, 6), sigma_prior))


def linear_chain(N, D=3, seed=0, dt=0.1, qc=0.01, sigma_fix=0.1, every=10):
    """Config C2: vector-space GP chain; linear, so Gauss-Newton converges in one iteration."""
    kind = LINEAR3 if D == 3 else LINEAR2
    rng = np.random.default_rng(SEED_BASE + 2 + seed)
    acc = 0.3 * np.stack([np.sin(0.01 * np.arange(N) + k) for k in range(D)], -1)
    vel = np.cumsum(acc * dt, 0) + 1.0
    pos = np.cumsum(vel * dt, 0)
    idx = np.arange(0, N, every, dtype=np.int32)
    fix = pos[idx] + sigma_fix * rng.standard_normal((len(idx), D))
    fix[0] = pos[0]
    sig = np.full((len(idx), D), sigma_fix)
    sig[0] = 1e-3                                          # PriorFactor on x0, sigma 1e-3 (test :161-162)
    return dict(kind=kind, name="C2 linear GP chain", N=N, qc=qc * np.eye(D),
                pose=pos + 0.1 * rng.standard_normal((N, D)), vel=vel + 0.1 * rng.standard_normal((N, D)),
                gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.full(N - 1, dt),
                prior_idx=idx, prior_pose=fix, prior_sig=sig,
                vprior_idx=np.array([0], dtype=np.int32), vprior=vel[:1].copy(), vprior_sig=np.full((1, D), 1e-3))


def apply(problem, solver):
    """Feed a problem description to a solver (ChainSolver or oracle.Chain) and compile it."""
    p = problem
    solver.set_qc(p["qc"])
    solver.set_states(p["pose"], p["vel"])
    solver.add_gp_priors(p["gp_left"], p["gp_dt"])
    if "prior_idx" in p:
        solver.add_pose_priors(p["prior_idx"], p["prior_pose"], p["prior_sig"])
    if "vprior_idx" in p:
        solver.add_vel_priors(p["vprior_idx"], p["vprior"], p["vprior_sig"])
    if "between_left" in p:
        solver.add_between(p["between_left"], p["between_meas"], p["between_sig"])
    solver.compile()
    return solver


def algorithmic_bytes_per_state(kind):
    """fp64 algorithmic HBM bytes of one Gauss-Newton iteration per state."""
    d = {LINEAR2: 2, LINEAR3: 3, POSE2: 3, POSE3: 6, ROT3: 3}[kind]
    pdim = {LINEAR2: 2, LINEAR3: 3, POSE2: 3, POSE3: 12, ROT3: 9}[kind]
    state = (pdim + d) * 8
    blocks = (8 * d * d + 2 * d) * 8          # D (2d x 2d) + O (2d x 2d) + g (2d), or e + H1..H4: same count
    linearize = state + 8 + blocks            # read state + dt, write Jacobian rows / blocks
    solve = blocks + 2 * d * 8                # read blocks once, write delta (single-pass lower bound)
    retract = state + 2 * d * 8 + state       # read state + delta, write state
    return dict(linearize=linearize, solve=solve, retract=retract, total=linearize + solve + retract)
