"""The AHRS recipe (second half): asynchronous gyroscope + accelerometer fusion with
bias states, the caller of the hot path for the GPSLAM_ROT3_BIAS chain.

Host-side restatement of what matlab/GPAHRSexample.m does around the optimizer -- the measurement loop that decides the
state time stamps and pre-integrates the gyroscope (:88-209), the two graphs (gyro-only for initialisation, full), the
initial values (:222-250) and the stopping rule (:253-266) -- written against the ChainSolver call surface, so the same
description drives the HIP library and, in the tests, the CPU oracle.  Pure numpy; no GPU and no oracle imports here.

gtsam::PreintegratedAhrsMeasurements / AHRSFactor are GTSAM 4.0 classes (gtsam/navigation/AHRSFactor.{h,cpp},
PreintegratedRotation.{h,cpp}; not under /root/reference): `Preintegrated` restates integrateMeasurement; the factor
itself is evaluated on the device (gpslam_hip_add_ahrs).

State layout of a GPSLAM_ROT3_BIAS chain (include/gpslam_hip.h): pose = [R row-major (9) | bias (3)],
velocity = [omega (3) | 0 0 0].

Dataset arrays (tests/golden/ahrs_imu.npz, made by tests/golden/make_ahrs_fixture.py):
  IMU (n, 8) seq, time, gyro x y z, acc x y z | MOCAP (m, 9) seq, time, position x y z, orientation x y z w
"""
import numpy as np

INF = np.inf


def load(path):
    d = np.load(path)
    return {k: np.asarray(d[k], dtype=np.float64) for k in ("IMU", "MOCAP")}


# ---------------------------------------------------------------- small SO(3) helpers (host side, recipe only)

def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(w):
    """Rot3::Expmap (Rodrigues) and its right Jacobian ExpmapDerivative, GTSAM's theta^2 <= eps branches."""
    th2 = float(np.dot(w, w))
    W = skew(w)
    if th2 <= np.finfo(float).eps:
        return np.eye(3) + W, np.eye(3)
    th = np.sqrt(th2)
    K = W / th
    R = np.eye(3) + np.sin(th) * K + (2.0 * np.sin(th / 2.0) ** 2) * (K @ K)
    J = np.eye(3) - ((1.0 - np.cos(th)) / th) * K + (1.0 - np.sin(th) / th) * (K @ K)
    return R, J


def rot_from_quaternion(w, x, y, z):
    """gtsam::Rot3::Quaternion(w, x, y, z) (GPAHRSexample.m:61)."""
    n = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot_ypr(R):
    """gtsam::Rot3::ypr() = (yaw, pitch, roll) of R = Rz(y) Ry(p) Rx(r) (GPAHRSexample.m:62)."""
    R = np.asarray(R).reshape(3, 3)
    return np.array([np.arctan2(R[1, 0], R[0, 0]), np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0])),
                     np.arctan2(R[2, 1], R[2, 2])])


def rot_from_ypr(y, p, r):
    """gtsam::Rot3::Ypr(y, p, r) (GPAHRSexample.m:118)."""
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    Rz = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]])
    Rx = np.array([[1.0, 0.0, 0.0], [0.0, cr, -sr], [0.0, sr, cr]])
    return Rz @ Ry @ Rx


class Preintegrated:
    """gtsam::PreintegratedAhrsMeasurements(biasHat, measuredOmegaCovariance) (GPAHRSexample.m:188).

    integrate() = PreintegratedAhrsMeasurements::integrateMeasurement (GTSAM 4.0 AHRSFactor.cpp) on top of
    PreintegratedRotation::integrateMeasurement:
        incrR = Expmap((omega - biasHat) dt), D = ExpmapDerivative(.)
        deltaTij += dt;  deltaRij = deltaRij incrR;  delRdelBiasOmega = incrR^T delRdelBiasOmega - D dt
        preintMeasCov = incrR^T preintMeasCov incrR + gyroCov dt
    """

    def __init__(self, bias_hat, gyro_cov):
        self.bias_hat = np.asarray(bias_hat, dtype=np.float64).reshape(3)
        self.gyro_cov = np.asarray(gyro_cov, dtype=np.float64).reshape(3, 3)
        self.delta_R = np.eye(3)
        self.dR_dbias = np.zeros((3, 3))
        self.delta_tij = 0.0
        self.cov = np.zeros((3, 3))

    def integrate(self, omega, dt):
        incr, D = so3_exp((np.asarray(omega, dtype=np.float64) - self.bias_hat) * dt)
        self.delta_tij += dt
        self.delta_R = self.delta_R @ incr
        self.dR_dbias = incr.T @ self.dR_dbias - D * dt
        self.cov = incr.T @ self.cov @ incr + self.gyro_cov * dt


# ---------------------------------------------------------------- the recipe

def build_problem(data, use_gyro=True, use_acc=True, dataset_max_time=50.0, gyro_dt=0.005, acc_dt=0.02,
                  qc_sigma=100.0, bias_prior_sigma=1e-2, bias_between_sigma=1e-4, gyro_cov=1e-3, acc_sigma=0.1,
                  first_rot_prior_sigma=0.1):
    """Graph description of matlab/GPAHRSexample.m:69-209 (settings :11-33 as keyword defaults).

    Returns a dict of arrays; `apply(p, solver, gyro_only)` adds them to a ChainSolver / oracle Chain.
    """
    IMU, MOCAP = data["IMU"], data["MOCAP"]
    q0 = MOCAP[0, 5:9]                                        # ATT = MOCAP(:, [1:2, 6:9]); Rot3.Quaternion(w, x, y, z) (:52, :61)
    R_first = rot_from_ypr(*rot_ypr(rot_from_quaternion(q0[3], q0[0], q0[1], q0[2])))   # Rot3.Ypr(ATT_YPR(1, :)) (:118)
    bias_hat = np.zeros(3)
    gcov = np.eye(3) * gyro_cov
    nr_imu = IMU.shape[0]

    state_time, state_meas_idx = [], []
    ahrs = {k: [] for k in ("left", "delta_R", "dR_dbias", "delta_tij", "cov")}
    gp_left, gp_dt = [], []
    att_left, att_dt, att_tau, att_b = [], [], [], []
    cached_acc, nr_acc = [], 0
    meas_time, last_gyro, last_acc = 0.0, 0.0, 0.0
    pim = None
    meas_idx = 0
    while meas_time < dataset_max_time and meas_idx < nr_imu:
        meas_time = IMU[meas_idx, 1]
        delta_t = IMU[meas_idx, 1] - IMU[meas_idx - 1, 1] if meas_idx > 0 else 0.0
        if use_acc and meas_time - last_acc >= acc_dt:                       # :103-108
            cached_acc.append(meas_idx)
            last_acc = meas_time
            nr_acc += 1
        first = len(state_time) == 0
        if first or meas_idx == nr_imu - 1 or meas_time - last_gyro >= gyro_dt:   # :114
            if not first:
                dt = meas_time - last_gyro
                pim.integrate(IMU[meas_idx, 2:5], delta_t)                   # :128
                left = len(state_time) - 1
                ahrs["left"].append(left)                                    # AHRSFactor(x_{k-1}, x_k, b_{k-1}, pim) (:131-137)
                ahrs["delta_R"].append(pim.delta_R.copy()); ahrs["dR_dbias"].append(pim.dR_dbias.copy())
                ahrs["delta_tij"].append(pim.delta_tij); ahrs["cov"].append(pim.cov.copy())
                gp_left.append(left); gp_dt.append(dt)                       # GaussianProcessPriorRot3 (:149-152)
                for acc_idx in cached_acc:                                   # :155-177
                    # the script passes the acceleration of the CURRENT sample (meas_idx) also to the cached,
                    # interpolated factors ("TODO" at :166): restated as it is
                    att_left.append(left); att_dt.append(dt); att_b.append(IMU[meas_idx, 5:8].copy())
                    # acc_idx == meas_idx: Rot3AttitudeFactor on x_k = the interpolated factor at tau = dt
                    att_tau.append(dt if acc_idx == meas_idx else IMU[acc_idx, 1] - last_gyro)
                cached_acc = []
            pim = Preintegrated(bias_hat, gcov)                              # :188
            last_gyro = meas_time
            state_time.append(meas_time)
            state_meas_idx.append(meas_idx)
        else:
            pim.integrate(IMU[meas_idx, 2:5], delta_t)                       # :198
        meas_idx += 1
    N = len(state_time)
    M = len(att_left)
    p = {
        "N": N, "state_time": np.array(state_time), "state_meas_idx": np.array(state_meas_idx, dtype=np.int64),
        "nr_acc": nr_acc, "use_gyro": use_gyro, "R_first": R_first,
        "Qc": np.eye(3) * qc_sigma ** 2,                                     # noiseModel.Diagonal.Sigmas(100) -> covariance
        # PriorFactorRot3(x_1) + PriorFactorVector(b_1) (:118-121): one 6-row (rotation, bias) prior
        "prior_idx": np.array([0], dtype=np.int32),
        "prior_pose": np.concatenate([R_first.reshape(9), np.zeros(3)])[None, :],
        "prior_sig": np.array([[first_rot_prior_sigma] * 3 + [bias_prior_sigma] * 3]),
        # BetweenFactorVector(b_{k-1}, b_k, 0) (:143): rotation half switched off
        "between_left": np.arange(N - 1, dtype=np.int32),
        "between_meas": np.tile(np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), (N - 1, 1)),
        "between_sig": np.tile(np.array([INF] * 3 + [bias_between_sigma] * 3), (N - 1, 1)),
        "ahrs_left": np.array(ahrs["left"], dtype=np.int32),
        "ahrs_delta_R": np.array(ahrs["delta_R"]).reshape(-1, 9), "ahrs_dR_dbias": np.array(ahrs["dR_dbias"]).reshape(-1, 9),
        "ahrs_bias_hat": np.zeros((N - 1, 3)), "ahrs_delta_tij": np.array(ahrs["delta_tij"]),
        "ahrs_cov": np.array(ahrs["cov"]).reshape(-1, 9),
        "gp_left": np.array(gp_left, dtype=np.int32), "gp_dt": np.array(gp_dt),
        "att_left": np.array(att_left, dtype=np.int32), "att_dt": np.array(att_dt), "att_tau": np.array(att_tau),
        "att_nZ": np.tile(np.array([0.0, 0.0, 1.0]), (M, 1)),                # Unit3(Point3(0, 0, 1)) (:161)
        "att_bRef": np.array(att_b).reshape(M, 3),
        "att_sig": np.full((M, 2), acc_sigma),
    }
    return p


def initial_values(p):
    """init_values of :222-228: identity rotations, zero biases (and zero velocities, :247-250)."""
    N = p["N"]
    pose = np.tile(np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), (N, 1))
    return pose, np.zeros((N, 6))


def apply(p, solver, gyro_only=False):
    """Add the recipe's factors to `solver` (states must be set).  gyro_only: the initialisation graph of :116-147
    (priors, AHRS factors, bias random walk; no GP prior, no accelerometer) -- its velocity slots are tied down by unit
    priors, which GTSAM does not need because that graph has no velocity variables at all."""
    N = p["N"]
    solver.set_qc(p["Qc"])
    solver.add_pose_priors(p["prior_idx"], p["prior_pose"], p["prior_sig"])
    solver.add_between(p["between_left"], p["between_meas"], p["between_sig"])
    if gyro_only or p["use_gyro"]:
        solver.add_ahrs(p["ahrs_left"], p["ahrs_delta_R"], p["ahrs_dR_dbias"], p["ahrs_bias_hat"], p["ahrs_delta_tij"],
                        p["ahrs_cov"])
    if gyro_only:
        solver.add_vel_priors(np.arange(N, dtype=np.int32), np.zeros((N, 6)), np.ones((N, 6)))
    else:
        solver.add_gp_priors(p["gp_left"], p["gp_dt"])
        if len(p["att_left"]):
            solver.add_interp_attitude(p["att_left"], p["att_nZ"], p["att_bRef"], p["att_sig"], p["att_dt"], p["att_tau"])
    solver.compile()


def optimize_default(solver, params):
    """optimizer.optimize() with default GaussNewtonParams / LevenbergMarquardtParams (:231-238); `params` = the
    backend's default parameter block with use_lm set as the script's useGaussNewton switch says."""
    return solver.optimize(params)


def iterate_until(solver, stop_rel_err=1e-6, max_iterations=100, lm=True, lambda_initial=1e-5):
    """The loop of :253-266: iterate while the relative error decrease exceeds optimizeStopRelErr."""
    last, lam, it = 1e20, lambda_initial, 0
    err = solver.error()
    trace = [err]
    while (last - err) / last > stop_rel_err and it < max_iterations:
        last = err
        if lm:
            _rc, st, lam = solver.iterate_lm(lam)[:3]
        else:
            _rc, st = solver.iterate_gn()
        err = st.error_after
        trace.append(err)
        it += 1
    return it, trace


def ground_truth_ypr(data, times):
    """MOCAP attitude as yaw / pitch / roll at the given times (nearest sample), as plotted at :296-331."""
    M = data["MOCAP"]
    idx = np.clip(np.searchsorted(M[:, 1], times), 1, len(M) - 1)
    idx = np.where(np.abs(M[idx - 1, 1] - times) <= np.abs(M[idx, 1] - times), idx - 1, idx)
    return np.array([rot_ypr(rot_from_quaternion(q[3], q[0], q[1], q[2])) for q in M[idx, 5:9]])
