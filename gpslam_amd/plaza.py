"""The Plaza range-only SLAM recipe (BASELINE config 1): the caller of the hot path.

Host-side restatement of what matlab/PlazaPose2.m does around the optimizer -- range bias fit and outlier rejection
(matlab/range_measure_fit.m), graph construction (PlazaPose2.m:49-204), the LM stopping rule (:208-228) and the
three error metrics (:236-274) -- written against the ChainSolver call surface, so the same description drives the
HIP library and, in the tests, the CPU oracle.  Pure numpy; no GPU and no oracle imports here.

Dataset arrays (tests/golden/plaza2.npz, made by tests/golden/make_plaza_fixture.py):
  GT (N, 4) time x y heading | DR (N-1, 3) time distance dheading | TD (M, 4) time sender landmark-id range |
  TL (L, 3) landmark-id x y | init_heading_offset
"""
import numpy as np

from . import synthetic as S


def load(path):
    d = np.load(path)
    return {k: np.asarray(d[k], dtype=np.float64) for k in ("GT", "DR", "TD", "TL")} | {
        "init_heading_offset": float(d["init_heading_offset"])}


def range_measure_fit(GT, TL, TD, outlier_limit=2.0):
    """Linear bias fit of the range measurements against ground truth, twice (all, then inliers only).

    matlab/range_measure_fit.m:1-93: every measurement is attached to the nearer of the two poses that bracket its
    time stamp (the earlier one on a tie, :36-40), true range = distance from that ground-truth position to the
    landmark (:50-56), least squares `true = a * measured + b` (:59-62), outliers are |fit - true| > 2 m (:65-74),
    and the fit is repeated on the inliers (:91-93).  Returns ((a, b), outlier mask).
    """
    T = GT[:, 0]
    t = TD[:, 0]
    hi = np.searchsorted(T, t, side="left")          # first pose with T >= t (the scan of :30-47)
    if np.any(hi == 0) or np.any(hi >= len(T)):
        raise ValueError("range measurement outside the ground-truth time span")
    pose = np.where(np.abs(T[hi - 1] - t) <= np.abs(T[hi] - t), hi - 1, hi)
    ids = TL[:, 0].astype(np.int64)
    lut = -np.ones(ids.max() + 1, dtype=np.int64)
    lut[ids] = np.arange(len(ids))
    lm = lut[TD[:, 2].astype(np.int64)]
    true_range = np.linalg.norm(GT[pose, 1:3] - TL[lm, 1:3], axis=1)
    measured = TD[:, 3]

    def fit(mask):
        A = np.stack([measured[mask], np.ones(int(mask.sum()))], axis=1)
        return np.linalg.lstsq(A, true_range[mask], rcond=None)[0]

    trans = fit(np.ones(len(measured), dtype=bool))
    outlier = np.abs(trans[0] * measured + trans[1] - true_range) > outlier_limit
    trans = fit(~outlier)
    return trans, outlier


def _se2_compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + c * b[0] - s * b[1], a[1] + s * b[0] + c * b[1], a[2] + b[2]])


def build_problem(data, linear=False, add_odometry=True, init_ground_truth=False, first_pose_prior=True,
                  first_vel_prior=False, landmark_prior=True, qc_sigma=0.1, odom_sigmas=(1e-3, 1e-3, np.pi * 1e-3),
                  first_prior_sigmas=(1.0, 1.0, np.pi), range_sigma=0.5, landmark_prior_sigma=1.0):
    """Problem description of the Plaza graph (PlazaPose2.m:26-204, the script's default switches as defaults).

    linear=False: SE(2) states, BetweenFactorPose2 odometry, GaussianProcessPriorPose2, GPInterpolatedRangeFactorPose2.
    linear=True:  [x y theta] vector states, OdometryFactor2DLinear, GaussianProcessPriorLinear<3>,
                  GPInterpolatedRangeFactor2DLinear.
    """
    GT, DR, TD, TL = data["GT"], data["DR"], data["TD"], data["TL"]
    off = data["init_heading_offset"]
    N, L = len(GT), len(TL)
    T = GT[:, 0]
    trans, outlier = range_measure_fit(GT, TL, TD)
    ids = TL[:, 0].astype(np.int64)
    lut = -np.ones(ids.max() + 1, dtype=np.int64)
    lut[ids] = np.arange(L)

    # dead reckoning (:84-123): odometry increment k is Pose2(distance, 0, dheading) between poses k and k + 1
    odo = np.stack([DR[:, 1], np.zeros(N - 1), DR[:, 2]], axis=1)
    first = np.array([GT[0, 1], GT[0, 2], GT[0, 3] + off])
    dead = np.zeros((N, 3))
    dead[0] = first
    if linear:
        # the linear variant accumulates world-frame x, y steps of the SE(2) dead reckoning plus the raw heading step
        se2 = first.copy()
        for k in range(N - 1):
            nxt = _se2_compose(se2, odo[k])
            dead[k + 1] = dead[k] + np.array([nxt[0] - se2[0], nxt[1] - se2[1], odo[k, 2]])
            se2 = nxt
    else:
        for k in range(N - 1):
            dead[k + 1] = _se2_compose(dead[k], odo[k])
    truth = np.stack([GT[:, 1], GT[:, 2], GT[:, 3] + off], axis=1)

    # range measurements (:147-178): a measurement at time t joins the first interval (k-1, k), k >= 1, whose right
    # end has T[k] >= t; tau is measured from the interval's left end; outliers are dropped
    t = TD[:, 0]
    right = np.maximum(np.searchsorted(T, t, side="left"), 1)
    keep = (~outlier) & (right < N)
    right = right[keep]
    left = right - 1
    p = dict(kind=S.LINEAR3 if linear else S.POSE2, name="plaza", N=N, linear=linear,
             qc=(qc_sigma ** 2) * np.eye(3), pose=truth.copy() if init_ground_truth else dead.copy(),
             vel=np.zeros((N, 3)), truth=truth, landmark_truth=TL[:, 1:3].copy(), landmarks=TL[:, 1:3].copy(),
             gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=np.diff(T),
             range_left=left.astype(np.int32), range_lm=lut[TD[keep, 2].astype(np.int64)].astype(np.int32),
             range_z=trans[0] * TD[keep, 3] + trans[1], range_sigma=np.full(int(keep.sum()), range_sigma),
             range_dt=T[right] - T[left], range_tau=t[keep] - T[left], range_trans=trans, range_outliers=outlier)
    if add_odometry:
        p.update(between_left=np.arange(N - 1, dtype=np.int32), between_meas=odo,
                 between_sig=np.tile(np.asarray(odom_sigmas, dtype=np.float64), (N - 1, 1)))
    if first_pose_prior:
        p.update(prior_idx=np.zeros(1, dtype=np.int32), prior_pose=first[None, :],
                 prior_sig=np.asarray(first_prior_sigmas, dtype=np.float64)[None, :])
    if first_vel_prior:
        p.update(vprior_idx=np.zeros(1, dtype=np.int32), vprior=np.zeros((1, 3)),
                 vprior_sig=np.asarray(first_prior_sigmas, dtype=np.float64)[None, :])
    if landmark_prior:
        p.update(lprior_idx=np.arange(L, dtype=np.int32), lprior=TL[:, 1:3].copy(),
                 lprior_sig=np.full((L, 2), landmark_prior_sigma))
    return p


def apply(p, solver):
    """Feed a Plaza problem to a ChainSolver-like object created for p["kind"] with landmark_dim = 2."""
    solver.set_qc(p["qc"])
    solver.set_states(p["pose"], p["vel"])
    solver.set_landmarks(p["landmarks"])
    if "prior_idx" in p:
        solver.add_pose_priors(p["prior_idx"], p["prior_pose"], p["prior_sig"])
    if "vprior_idx" in p:
        solver.add_vel_priors(p["vprior_idx"], p["vprior"], p["vprior_sig"])
    if "lprior_idx" in p:
        solver.add_landmark_priors(p["lprior_idx"], p["lprior"], p["lprior_sig"])
    if "between_left" in p:
        if p["linear"]:
            solver.add_odometry2d(p["between_left"], p["between_meas"], p["between_sig"])
        else:
            solver.add_between(p["between_left"], p["between_meas"], p["between_sig"])
    solver.add_gp_priors(p["gp_left"], p["gp_dt"])
    solver.add_interp_range(p["range_left"], p["range_lm"], p["range_z"], p["range_sigma"], p["range_dt"], p["range_tau"])
    solver.compile()
    return solver


def optimize(solver, stop_rel_err=1e-6, lambda_initial=1e-5, max_iterations=100):
    """The script's outer loop (PlazaPose2.m:217-228): LevenbergMarquardtOptimizer::iterate() until the relative
    error decrease drops to stop_rel_err.  Returns the error after every iteration (element 0 = initial error)."""
    lam = lambda_initial
    errors = [float(solver.error())]
    last = 1e20
    while (last - errors[-1]) / last > stop_rel_err and len(errors) <= max_iterations:
        last = errors[-1]
        out = solver.iterate_lm(lam)
        rc, st, lam = out[0], out[1], out[2]
        if rc != 0:
            raise RuntimeError("iterate_lm failed: %d" % rc)
        errors.append(float(st.error_after))
    return errors


def metrics(p, pose, landmarks):
    """Average position error (m), average absolute heading error (deg), average landmark error (m) -- :236-274."""
    pose, landmarks = np.asarray(pose), np.asarray(landmarks)
    pos = np.linalg.norm(pose[:, :2] - p["truth"][:, :2], axis=1).mean()
    rot = pose[:, 2] - p["truth"][:, 2]
    rot = (rot + np.pi) % (2.0 * np.pi) - np.pi
    land = np.linalg.norm(landmarks - p["landmark_truth"], axis=1).mean()
    return dict(position_m=float(pos), rotation_deg=float(np.abs(rot).mean() * 180.0 / np.pi), landmark_m=float(land))
