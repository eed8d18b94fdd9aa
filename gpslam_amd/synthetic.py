"""Seeded synthetic GP-SLAM chains for the named benchmark configurations.

Pure numpy, no oracle and no GPU: this only PRODUCES problem descriptions (dicts of arrays).  `apply()` feeds a
description to anything with the ChainSolver call surface (gpslam_amd.ChainSolver on the GPU, oracle.Chain in
the tests and in bench.py's cpu_baseline leg), so both sides see bit-identical inputs.

  C2  GaussianProcessPriorLinear<3> chain, N = 1e5, dt = 0.1, Qc = 0.01 I (gpslam/gp/tests/
      testGaussianProcessPriorLinear.cpp:32-33), prior on (x0, v0), position fix on every 10th state
  C3  GaussianProcessPriorPose3 chain, N = 1e5, dt = 0.1, Qc = 0.01 I6 (testGaussianProcessPriorPose3.cpp:29-31),
      BetweenFactor<Pose3> odometry from truth + noise, prior on x0, dead-reckoned initial values with zero
      velocity (the recipe of matlab/PlazaPose2.m:125,:196-202 lifted to SE(3))
  C4' SE(2) chain with GPInterpolatedRangeFactorPose2 at the Plaza2 rate (0.44 ranges per interval, tau ~ U(0, dt),
      sigma 0.5, Qc sigma 0.1, odometry sigma [1, 1, pi] 1e-3, landmark priors sigma 1: matlab/PlazaPose2.m:40-46)
      to a handful of landmarks -- config 4's factor mix with the dense landmark border this build supports (L <= 13)
  C5  SO(3) chain: GaussianProcessPriorRot3 + GPInterpolatedAttitudeFactorRot3 at 4x the state rate
      (matlab/GPAHRSexample.m:21,:28,:38-39: Qc sigma, accelerometer sigma 0.1), the reference-faithful variant of config 5
"""
import numpy as np

LINEAR2, LINEAR3, POSE2, POSE3, ROT3, ROT3_BIAS = 0, 1, 2, 3, 4, 5
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


def se2_exp(xi):
    """Pose2::Expmap, batched: (..., 3) [vx, vy, w] -> (..., 3) [x, y, theta]"""
    vx, vy, w = xi[..., 0], xi[..., 1], xi[..., 2]
    small = np.abs(w) < 1e-10
    ws = np.where(small, 1.0, w)
    a = np.where(small, 1.0 - w * w / 6.0, np.sin(w) / ws)
    b = np.where(small, 0.5 * w, (1.0 - np.cos(w)) / ws)
    return np.stack([a * vx - b * vy, b * vx + a * vy, w], -1)


def se2_compose(a, b):
    c, s = np.cos(a[..., 2]), np.sin(a[..., 2])
    return np.stack([a[..., 0] + c * b[..., 0] - s * b[..., 1], a[..., 1] + s * b[..., 0] + c * b[..., 1],
                     a[..., 2] + b[..., 2]], -1)


def pose2_range_chain(N, L=8, seed=0, dt=0.1, rate=0.44):
    """C4': SE(2) trajectory on a large figure of eight, odometry, interpolated ranges to L landmarks."""
    rng = np.random.default_rng(SEED_BASE + 4 + seed)
    i = np.arange(N - 1)
    twist = np.stack([1.0 + 0.2 * np.sin(0.003 * i), 0.05 * np.cos(0.007 * i), 0.15 * np.sin(0.0011 * i)], -1)
    rel = se2_exp(dt * twist)                              # true relative motions
    truth = np.zeros((N, 3))
    for k in range(N - 1):                                 # sequential composition (cheap: N 3-vectors)
        truth[k + 1] = se2_compose(truth[k], rel[k])
    sig_odo = np.array([1e-3, 1e-3, np.pi * 1e-3])
    odo = rel + sig_odo * rng.standard_normal((N - 1, 3))  # measured = true + noise (first-order chart)
    dead = np.zeros((N, 3))
    for k in range(N - 1):
        dead[k + 1] = se2_compose(dead[k], odo[k])
    centre, span = truth[:, :2].mean(0), np.ptp(truth[:, :2], axis=0).max() + 10.0
    lmk = centre + span * (rng.random((L, 2)) - 0.5)
    has = rng.random(N - 1) < rate
    left = np.nonzero(has)[0].astype(np.int32)
    tau = dt * rng.random(len(left))
    lm = rng.integers(0, L, len(left)).astype(np.int32)
    # pose at the measurement time on the true (constant-twist per interval) trajectory
    at = se2_compose(truth[left], se2_exp(tau[:, None] * twist[left]))
    z = np.linalg.norm(lmk[lm] - at[:, :2], axis=1) + 0.5 * rng.standard_normal(len(left))
    return dict(kind=POSE2, name="C4' pose2 GP prior + odometry + interpolated ranges", N=N, qc=0.01 * np.eye(3),
                pose=dead, vel=np.zeros((N, 3)), truth=truth, landmarks=lmk + 0.5 * rng.standard_normal((L, 2)),
                landmark_truth=lmk, linear=False,
                gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.full(N - 1, dt),
                between_left=np.arange(N - 1, dtype=np.int32), between_meas=odo, between_sig=np.tile(sig_odo, (N - 1, 1)),
                prior_idx=np.array([0], dtype=np.int32), prior_pose=truth[:1].copy(), prior_sig=np.array([[1.0, 1.0, np.pi]]),
                lprior_idx=np.arange(L, dtype=np.int32), lprior=lmk.copy(), lprior_sig=np.full((L, 2), 1.0),
                range_left=left, range_lm=lm, range_z=z, range_sigma=np.full(len(left), 0.5),
                range_dt=np.full(len(left), dt), range_tau=tau)


def se2_prefix(rel):
    """Inclusive prefix composition of a batch of SE(2) elements (N x 3 [x, y, theta]) by doubling."""
    x = rel.copy()
    n, shift = len(x), 1
    while shift < n:
        x[shift:] = se2_compose(x[:-shift], x[shift:])
        shift *= 2
    return x


def pose2_local_landmarks_chain(N, L=None, seed=0, dt=0.1, rate=0.44, window=200, anchor=256):
    """Config C4 (BASELINE: 1e6 poses + 5e4 range landmarks, Plaza-scaled): the Plaza recipe (matlab/PlazaPose2.m:40-46,
    :55-66, :147-178) on a long SE(2) drive past L = N / 20 landmarks, each visible only while the robot is within
    `window` / 2 states of its closest approach, so that ~window / 20 landmarks are in view at any time and every
    landmark collects about rate * 20 = 8.8 interpolated range factors.  Initial values: odometry dead reckoning
    (PlazaPose2.m:196-202) re-anchored at the true pose every `anchor` states: a 1e6-pose drive is never dead-reckoned
    open loop from its first pose (the heading noise of the Plaza recipe, pi 1e-3 per step, alone accumulates to 3 rad),
    and beyond a few hundred states the drift mirrors landmarks across the path, which leaves range-only Gauss-Newton
    creeping along a flat valley (anchor = 0 or >= N: open loop)."""
    rng = np.random.default_rng(SEED_BASE + 44 + seed)
    if L is None:
        L = max(N // 20, 1)
    i = np.arange(N - 1)
    twist = np.stack([1.0 + 0.2 * np.sin(0.003 * i), 0.05 * np.cos(0.007 * i), 0.05 * np.sin(0.0011 * i)], -1)
    rel = se2_exp(dt * twist)
    truth = np.zeros((N, 3))
    truth[1:] = se2_prefix(rel)
    sig_odo = np.array([1e-3, 1e-3, np.pi * 1e-3])
    odo = rel + sig_odo * rng.standard_normal((N - 1, 3))
    dead = np.zeros((N, 3))
    if anchor and anchor < N:
        for a0 in range(0, N, anchor):                     # dead reckoning inside each window, from its true first pose
            a1 = min(a0 + anchor, N)
            dead[a0] = truth[a0]
            if a1 - a0 > 1:
                loc = se2_prefix(odo[a0:a1 - 1])
                dead[a0 + 1:a1] = se2_compose(np.broadcast_to(truth[a0], loc.shape), loc)
    else:
        dead[1:] = se2_prefix(odo)
    centre = np.minimum(((np.arange(L) + 0.5) * (N / L)).astype(np.int64), N - 1)     # closest-approach state
    # 5-15 m to either side of the path: every range stays many sigma away from zero (a range measured near zero makes
    # the factor's minimum a ring and Gauss-Newton oscillate around it)
    side = np.where(rng.random(L) < 0.5, -1.0, 1.0) * (5.0 + 10.0 * rng.random(L))
    th = truth[centre, 2]
    lmk = truth[centre, :2] + side[:, None] * np.stack([-np.sin(th), np.cos(th)], -1)
    has = rng.random(N - 1) < rate
    left = np.nonzero(has)[0].astype(np.int32)
    tau = dt * rng.random(len(left))
    # a visible landmark: closest approach within window / 2 states of the interval
    pos = (left + 0.5) * (L / N) - 0.5                       # fractional landmark index at this interval
    half = 0.5 * window * (L / N)
    lo = np.clip(np.ceil(pos - half), 0, L - 1).astype(np.int64)
    hi = np.clip(np.floor(pos + half), 0, L - 1).astype(np.int64)
    hi = np.maximum(hi, lo)
    lm = (lo + (rng.random(len(left)) * (hi - lo + 1)).astype(np.int64)).astype(np.int32)
    at = se2_compose(truth[left], se2_exp(tau[:, None] * twist[left]))
    z = np.linalg.norm(lmk[lm] - at[:, :2], axis=1) + 0.5 * rng.standard_normal(len(left))
    return dict(kind=POSE2, name="C4 pose2 GP prior + odometry + interpolated ranges to locally visible landmarks", N=N,
                qc=0.01 * np.eye(3), pose=dead, vel=np.zeros((N, 3)), truth=truth,
                landmarks=lmk + 0.5 * rng.standard_normal((L, 2)), landmark_truth=lmk, linear=False,
                gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.full(N - 1, dt),
                between_left=np.arange(N - 1, dtype=np.int32), between_meas=odo, between_sig=np.tile(sig_odo, (N - 1, 1)),
                prior_idx=np.array([0], dtype=np.int32), prior_pose=truth[:1].copy(), prior_sig=np.array([[1.0, 1.0, np.pi]]),
                lprior_idx=np.arange(L, dtype=np.int32), lprior=lmk.copy(), lprior_sig=np.full((L, 2), 1.0),
                range_left=left, range_lm=lm, range_z=z, range_sigma=np.full(len(left), 0.5),
                range_dt=np.full(len(left), dt), range_tau=tau)


def rot3_attitude_chain(N, per_interval=4, seed=0, dt=0.1, qc_sigma=1.0, acc_sigma=0.1, refs=1):
    """C5 (reference-faithful variant): SO(3) GP chain with interpolated attitude (accelerometer) factors.
    refs = 1: every factor observes the body z axis, as matlab/GPAHRSexample.m:169-173 does with the accelerometer alone
    (heading is then held only by the prior on x0 and the GP smoothness: a nearly flat direction, Gauss-Newton does not
    converge from a perturbed start).  refs = 2: factors alternate between the body z and the body x axis
    (accelerometer + magnetometer, both are GPInterpolatedAttitudeFactorRot3 with their own bRef): fully observable."""
    rng = np.random.default_rng(SEED_BASE + 5 + seed)
    i = np.arange(N - 1)
    omega = np.stack([0.3 * np.sin(0.004 * i), 0.2 * np.cos(0.006 * i + 0.5), 0.4 + 0.1 * np.sin(0.002 * i)], -1)
    dR = so3_exp(dt * omega)
    R = np.zeros((N, 3, 3))
    R[0] = np.eye(3)
    # inclusive prefix product by doubling, as for SE(3)
    P, _ = se3_prefix(dR, np.zeros((N - 1, 3)))
    R[1:] = P
    M = (N - 1) * per_interval
    left = np.repeat(np.arange(N - 1), per_interval).astype(np.int32)
    tau = dt * (np.tile(np.arange(per_interval), N - 1) + rng.random(M)) / per_interval
    Rm = R[left] @ so3_exp(tau[:, None] * omega[left])     # true attitude at the measurement times
    bref = np.array([0.0, 0.0, 1.0])                       # body reference axis (Unit3(0, 0, 1), AttitudeFactorRot3.h:48)
    brefs = np.tile(bref, (M, 1))
    if refs == 2:
        brefs[1::2] = [1.0, 0.0, 0.0]
    nz = np.einsum("nij,nj->ni", Rm, brefs) + acc_sigma * rng.standard_normal((M, 3))
    nz /= np.linalg.norm(nz, axis=1, keepdims=True)        # measured nav-frame direction of the body axis
    init = R @ so3_exp(0.05 * rng.standard_normal((N, 3)))
    return dict(kind=ROT3, name="C5 rot3 GP prior + interpolated attitude", N=N, qc=qc_sigma ** 2 * np.eye(3),
                pose=init.reshape(N, 9), vel=np.zeros((N, 3)), truth=R.reshape(N, 9),
                gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.full(N - 1, dt),
                prior_idx=np.array([0], dtype=np.int32), prior_pose=R[:1].reshape(1, 9), prior_sig=np.full((1, 3), 1e-2),
                vprior_idx=np.array([0], dtype=np.int32), vprior=omega[:1].copy(), vprior_sig=np.full((1, 3), 0.1),
                att_left=left, att_nz=nz, att_bref=brefs, att_sigma=np.full((M, 2), acc_sigma),
                att_dt=np.full(M, dt), att_tau=tau)


def pose3_gps_chain(N, per_interval=4, seed=0, dt=0.1, keep_odometry=False):
    """C5 (SE(3) variant): the C3 chain with GPInterpolatedGPSFactorPose3 at 4x the state rate, without odometry
    (position fixes alone: the attitude is weakly observable, Gauss-Newton creeps) or with it (keep_odometry)."""
    p = pose3_chain(N, seed=seed, dt=dt)
    rng = np.random.default_rng(SEED_BASE + 55 + seed)
    for k in ("between_left", "between_meas", "between_sig"):
        if not keep_odometry:
            p.pop(k)
    M = (N - 1) * per_interval
    left = np.repeat(np.arange(N - 1), per_interval).astype(np.int32)
    tau = dt * (np.tile(np.arange(per_interval), N - 1) + rng.random(M)) / per_interval
    # positions along the dead-reckoned path (close to the truth: odometry noise 1e-3) + GPS noise
    t0, t1 = p["pose"][left, 9:12], p["pose"][left + 1, 9:12]
    w = (tau / dt)[:, None]
    p.update(name="C5b pose3 GP prior + interpolated GPS", gps_left=left, gps_meas=(1 - w) * t0 + w * t1 + 0.05 * rng.standard_normal((M, 3)),
             gps_sigma=np.full((M, 3), 0.05), gps_dt=np.full(M, dt), gps_tau=tau)
    return p


def _so3_right_jacobian(w):
    """Rot3::ExpmapDerivative, batched (GTSAM's closed form; series near zero)"""
    th2 = np.sum(w * w, -1)
    th = np.sqrt(np.maximum(th2, 1e-300))
    W = _skew(w)
    a = np.where(th2 > 1e-16, (1.0 - np.cos(th)) / np.maximum(th2, 1e-300), 0.5 - th2 / 24.0)
    b = np.where(th2 > 1e-16, (th - np.sin(th)) / np.maximum(th2 * th, 1e-300), 1.0 / 6.0 - th2 / 120.0)
    return np.eye(3) - a[..., None, None] * W + b[..., None, None] * (W @ W)


def rot3_bias_ahrs_chain(N, seed=0, dt=0.1, qc_sigma=1.0, gyro_sigma=0.03, acc_sigma=0.1, acc_every=4, bias_walk=1e-4):
    """C5 as matlab/GPAHRSexample.m:69-209 builds it, at any size: one (rotation, gyroscope bias | angular velocity) state per
    time stamp (GPSLAM_ROT3_BIAS), per interval one gtsam::AHRSFactor of a single pre-integrated gyroscope sample
    (PreintegratedAhrsMeasurements::integrateMeasurement, biasHat = 0) + BetweenFactorVector on the bias +
    GaussianProcessPriorRot3; a Rot3AttitudeFactor (accelerometer, = the interpolated factor at tau = dt) on every
    acc_every-th state, alternating between the body z and x axes so that the heading is observable; priors on the first state."""
    rng = np.random.default_rng(SEED_BASE + 7 + seed)
    i = np.arange(N - 1)
    omega = np.stack([0.3 * np.sin(0.004 * i), 0.2 * np.cos(0.006 * i + 0.5), 0.4 + 0.1 * np.sin(0.002 * i)], -1)
    R = np.zeros((N, 3, 3))
    R[0] = np.eye(3)
    R[1:], _ = se3_prefix(so3_exp(dt * omega), np.zeros((N - 1, 3)))
    bias_true = np.array([0.02, -0.01, 0.015]) + bias_walk * np.cumsum(rng.standard_normal((N, 3)), axis=0)
    # one gyroscope sample per interval: omega_m = omega + bias + noise;  incrR = Exp(omega_m dt), D = Jr(omega_m dt)
    wm = omega + bias_true[:-1] + gyro_sigma * rng.standard_normal((N - 1, 3))
    dR = so3_exp(wm * dt)
    dRdb = -dt * _so3_right_jacobian(wm * dt)
    cov = np.tile(gyro_sigma ** 2 * dt * np.eye(3), (N - 1, 1, 1))
    pose = np.zeros((N, 12))
    pose[:, :9] = (R @ so3_exp(0.03 * rng.standard_normal((N, 3)))).reshape(N, 9)          # noisy initial attitudes, zero bias
    vel = np.zeros((N, 6))
    vel[:-1, :3] = wm
    vel[-1, :3] = wm[-1]
    truth = np.zeros((N, 12))
    truth[:, :9], truth[:, 9:] = R.reshape(N, 9), bias_true
    right = np.arange(acc_every, N, acc_every)                                              # states with an accelerometer sample
    M = len(right)
    bref = np.tile([0.0, 0.0, 1.0], (M, 1))
    bref[1::2] = [1.0, 0.0, 0.0]
    nz = np.einsum("nij,nj->ni", R[right], bref) + acc_sigma * rng.standard_normal((M, 3))
    nz /= np.linalg.norm(nz, axis=1, keepdims=True)
    ident_bias0 = np.concatenate([np.eye(3).reshape(9), np.zeros(3)])
    return dict(kind=ROT3_BIAS, name="C5 AHRS: rot3 + gyroscope bias chain", N=N, qc=qc_sigma ** 2 * np.eye(3), pose=pose, vel=vel, truth=truth,
                gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.full(N - 1, dt),
                prior_idx=np.array([0], dtype=np.int32), prior_pose=truth[:1].copy(), prior_sig=np.array([[0.1, 0.1, 0.1, 1e-2, 1e-2, 1e-2]]),
                between_left=np.arange(N - 1, dtype=np.int32), between_meas=np.tile(ident_bias0, (N - 1, 1)),
                between_sig=np.tile([np.inf] * 3 + [10 * bias_walk] * 3, (N - 1, 1)),
                ahrs_left=np.arange(N - 1, dtype=np.int32), ahrs_dR=dR.reshape(-1, 9), ahrs_dRdb=dRdb.reshape(-1, 9), ahrs_bias_hat=np.zeros((N - 1, 3)),
                ahrs_dt=np.full(N - 1, dt), ahrs_cov=cov.reshape(-1, 9),
                att_left=(right - 1).astype(np.int32), att_nz=nz, att_bref=bref, att_sigma=np.full((M, 2), acc_sigma),
                att_dt=np.full(M, dt), att_tau=np.full(M, dt))


def add_loop_closures(problem, pairs, seed=0, sigma=None):
    """Loop closures on a problem: gtsam::BetweenFactor<Pose>(x_first, x_second, measured) between NON-adjacent states (any order of
    the two), measured = truth_first^-1 truth_second + noise.  pairs: K x 2 state indices.  Adds closure_first / _second / _meas / _sig."""
    p = dict(problem)
    kind = p["kind"]
    rng = np.random.default_rng(SEED_BASE + 77 + seed)
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    a, b = pairs[:, 0], pairs[:, 1]
    truth = np.asarray(p["truth"], dtype=np.float64) if "truth" in p else np.asarray(p["pose"], dtype=np.float64)
    if kind in (LINEAR2, LINEAR3):
        d = truth.shape[1]
        sig = np.full(d, 0.05) if sigma is None else np.asarray(sigma, dtype=np.float64)
        meas = truth[b] - truth[a] + sig * rng.standard_normal((len(a), d))
    elif kind == POSE2:
        sig = np.array([0.02, 0.02, 0.01]) if sigma is None else np.asarray(sigma, dtype=np.float64)
        c, s = np.cos(truth[a, 2]), np.sin(truth[a, 2])
        dx, dy = truth[b, 0] - truth[a, 0], truth[b, 1] - truth[a, 1]
        rel = np.stack([c * dx + s * dy, -s * dx + c * dy, truth[b, 2] - truth[a, 2]], -1)
        meas = se2_compose(rel, se2_exp(sig * rng.standard_normal((len(a), 3))))
    elif kind == POSE3:
        sig = np.array([0.01, 0.01, 0.01, 0.02, 0.02, 0.02]) if sigma is None else np.asarray(sigma, dtype=np.float64)
        Ra, Rb = truth[a, :9].reshape(-1, 3, 3), truth[b, :9].reshape(-1, 3, 3)
        R = np.einsum("nji,njk->nik", Ra, Rb)
        t = np.einsum("nji,nj->ni", Ra, truth[b, 9:] - truth[a, 9:])
        nR, nt = se3_exp(sig * rng.standard_normal((len(a), 6)))
        meas = flat_pose3(np.einsum("nij,njk->nik", R, nR), t + np.einsum("nij,nj->ni", R, nt))
    else:
        raise ValueError("loop closures: LINEAR2 / LINEAR3 / POSE2 / POSE3 problems")
    p.update(closure_first=a.copy(), closure_second=b.copy(), closure_meas=np.ascontiguousarray(meas), closure_sig=np.tile(sig, (len(a), 1)))
    return p


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
    if "landmarks" in p:
        solver.set_landmarks(p["landmarks"])
        if "lprior_idx" in p:
            solver.add_landmark_priors(p["lprior_idx"], p["lprior"], p["lprior_sig"])
        if "range_left" in p:
            solver.add_interp_range(p["range_left"], p["range_lm"], p["range_z"], p["range_sigma"], p["range_dt"], p["range_tau"])
    if "gps_left" in p:
        solver.add_interp_gps(p["gps_left"], p["gps_meas"], p["gps_sigma"], p["gps_dt"], p["gps_tau"])
    if "ahrs_left" in p:
        solver.add_ahrs(p["ahrs_left"], p["ahrs_dR"], p["ahrs_dRdb"], p["ahrs_bias_hat"], p["ahrs_dt"], p["ahrs_cov"], None)
    if "att_left" in p:
        solver.add_interp_attitude(p["att_left"], p["att_nz"], p["att_bref"], p["att_sigma"], p["att_dt"], p["att_tau"])
    if "closure_first" in p:
        solver.add_between_pairs(p["closure_first"], p["closure_second"], p["closure_meas"], p["closure_sig"])
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
