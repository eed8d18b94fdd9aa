"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, from __graft_entry__.smoke() and from
bench.py's cpu_baseline leg -- never from gpslam_amd/.  See oracle/gpslam_oracle.h for what
the oracle restates and which parts of it are pinned by the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

LINEAR2, LINEAR3, POSE2, POSE3, ROT3, ROT3_BIAS = 0, 1, 2, 3, 4, 5
CHART_EXPMAP, CHART_FIRST_ORDER = 0, 1
POSE_DIM = {LINEAR2: 2, LINEAR3: 3, POSE2: 3, POSE3: 12, ROT3: 9, ROT3_BIAS: 12}
TANGENT_DIM = {LINEAR2: 2, LINEAR3: 3, POSE2: 3, POSE3: 6, ROT3: 3, ROT3_BIAS: 6}


def build(force=False):
    """Compile oracle/liboracle.so with gcc (seconds)."""
    srcs = ["orc_lie.c", "orc_gp.c", "orc_factors.c", "orc_chain.c", "gpslam_oracle.h", "orc_math.h"]
    if not force and os.path.exists(_LIB_PATH):
        so_m = os.path.getmtime(_LIB_PATH)
        if all(os.path.getmtime(os.path.join(_HERE, s)) <= so_m for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_pose3_range.restype = C.c_double
        _lib.orc_pose2_range.restype = C.c_double
        for n in ("orc_interp_range_pose2", "orc_interp_range_pose3", "orc_interp_range_2dlinear",
                  "orc_range_2dlinear", "orc_range_pose2", "orc_interp_range_pose3vw"):
            getattr(_lib, n).restype = C.c_double
        _lib.orc_chain_create.restype = C.c_void_p
    return _lib


class Stats(C.Structure):
    _fields_ = [("error_before", C.c_double), ("error_after", C.c_double), ("delta_inf_norm", C.c_double),
                ("lambda_", C.c_double), ("iterations", C.c_int32), ("status", C.c_int32),
                ("accepted", C.c_int32), ("trials", C.c_int32), ("last_trial_error", C.c_double)]


class Params(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("relative_error_tol", C.c_double),
                ("absolute_error_tol", C.c_double), ("error_tol", C.c_double), ("delta_tol", C.c_double),
                ("lambda_initial", C.c_double), ("lambda_factor", C.c_double), ("lambda_upper_bound", C.c_double),
                ("lambda_lower_bound", C.c_double), ("min_model_fidelity", C.c_double), ("use_lm", C.c_int32),
                ("pad", C.c_int32)]


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def set_threads(n=0):
    """OpenMP threads for factor evaluation (0 = all cores); results do not depend on the count."""
    lib().orc_set_threads(int(n))
    return lib().orc_get_threads()


def _d(a):
    """contiguous float64 array + pointer (None -> NULL)"""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, C.c_void_p(a.ctypes.data)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, C.c_void_p(a.ctypes.data)


def _out(*shape):
    a = np.zeros(shape, dtype=np.float64)
    return a, C.c_void_p(a.ctypes.data)


def call(name, *args):
    """Generic call: numpy arrays are passed as double*, python floats as double, ints as int."""
    keep, cargs = [], []
    for a in args:
        if a is None:
            cargs.append(None)
        elif isinstance(a, np.ndarray):
            if a.dtype == np.int32:
                cargs.append(C.c_void_p(a.ctypes.data))
            else:
                assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "pass contiguous float64"
                cargs.append(C.c_void_p(a.ctypes.data))
            keep.append(a)
        elif isinstance(a, (float, np.floating)):
            cargs.append(C.c_double(float(a)))
        elif isinstance(a, (int, np.integer)):
            cargs.append(C.c_int(int(a)))
        else:
            cargs.append(a)
    return getattr(lib(), name)(*cargs)


# ---------------------------------------------------------------- small conveniences used by the tests

def A(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def rot3_ypr(y, p, r):
    R = np.zeros(9)
    call("orc_rot3_ypr", float(y), float(p), float(r), R)
    return R


def pose3(ypr, t):
    return np.concatenate([rot3_ypr(*ypr), A(t)])


def ahrs_factor(Ri, Rj, bias, prm, jac=True):
    """gtsam::AHRSFactor::evaluateError; prm = [deltaRij 9 | delRdelBiasOmega 9 | biasHat 3 | deltaTij | omegaCoriolis 3]."""
    e, H1, H2, H3 = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3))
    call("orc_ahrs_factor", A(Ri), A(Rj), A(bias), A(prm), e, H1 if jac else None, H2 if jac else None, H3 if jac else None)
    return e, H1, H2, H3


def ahrs_preintegrate(omegas, dts, bias_hat, gyro_cov):
    """PreintegratedAhrsMeasurements after integrating the given samples: (deltaRij, delRdelBiasOmega, deltaTij, cov)."""
    st = np.zeros(28)
    call("orc_ahrs_preint_reset", st)
    bh, gc = A(bias_hat), A(gyro_cov)
    for w, dt in zip(np.asarray(omegas, dtype=np.float64), dts):
        call("orc_ahrs_preint_integrate", st, bh, gc, A(w), float(dt))
    return st[0:9].reshape(3, 3).copy(), st[9:18].reshape(3, 3).copy(), float(st[18]), st[19:28].reshape(3, 3).copy()


def lambda_psi(D, Qc, dt, tau):
    n = 2 * D
    Lam, Psi = np.zeros((n, n)), np.zeros((n, n))
    Qc = A(Qc)
    assert call("orc_calcLambda", D, Qc, float(dt), float(tau), Lam) == 0
    assert call("orc_calcPsi", D, Qc, float(dt), float(tau), Psi) == 0
    return Lam, Psi


def retract(kind, x, delta, chart=CHART_EXPMAP):
    out = np.zeros(POSE_DIM[kind])
    call("orc_retract", kind, chart, A(x), A(delta), out)
    return out


def local(kind, x, y, chart=CHART_EXPMAP):
    out = np.zeros(TANGENT_DIM[kind])
    call("orc_local", kind, chart, A(x), A(y), out)
    return out


def gp_prior(kind, p1, v1, p2, v2, dt, jac=True):
    """(e, [H1..H4]) of the GP prior for one factor; unwhitened, row-major."""
    d = TANGENT_DIM[kind]
    b = 2 * d
    e = np.zeros(b)
    H = [np.zeros((b, d)) for _ in range(4)] if jac else [None] * 4
    p1, v1, p2, v2 = A(p1), A(v1), A(p2), A(v2)
    if kind in (LINEAR2, LINEAR3):
        call("orc_gp_prior_linear", d, p1, v1, p2, v2, float(dt), e, *H)
    else:
        name = {POSE2: "orc_gp_prior_pose2", POSE3: "orc_gp_prior_pose3", ROT3: "orc_gp_prior_rot3"}[kind]
        call(name, p1, v1, p2, v2, float(dt), e, *H)
    return e, H


def gp_prior_vw(p1, v1, w1, p2, v2, w2, dt, jac=True):
    """GaussianProcessPriorPose3VW::evaluateError: (e, [H1 (12x6), H2, H3 (12x3), H4 (12x6), H5, H6 (12x3)])."""
    e = np.zeros(12)
    H = [np.zeros((12, n)) for n in (6, 3, 3, 6, 3, 3)] if jac else [None] * 6
    call("orc_gp_prior_pose3vw", A(p1), A(v1), A(w1), A(p2), A(v2), A(w2), float(dt), e, *H)
    return e, H


def interpolate_vw(Lam, Psi, p1, v1, w1, p2, v2, w2, jac=True):
    """GaussianProcessInterpolatorPose3VW::interpolatePose: (pose, [H1 (6x6), H2, H3 (6x3), H4 (6x6), H5, H6])."""
    out = np.zeros(12)
    H = [np.zeros((6, n)) for n in (6, 3, 3, 6, 3, 3)] if jac else [None] * 6
    call("orc_interp_pose3vw", A(Lam), A(Psi), A(p1), A(v1), A(w1), A(p2), A(v2), A(w2), out, *H)
    return out, H


def _k9(K):
    K = np.asarray(K, dtype=np.float64).ravel()
    return np.concatenate([K, np.zeros(9 - len(K))])       # Cal3_S2 (5) or Cal3DS2 (9: + k1, k2, p1, p2)


def pinhole_project(cam, K, point):
    """PinholeCamera<Cal3_S2 | Cal3DS2>(cam, K).project(point); raises on a cheirality violation."""
    uv = np.zeros(2)
    if call("orc_pinhole_project_ds2", A(cam), A(_k9(K)), A(point), uv, None, None) != 0:
        raise ValueError("point behind the camera")
    return uv


def interp_projection(Lam, Psi, meas, K, sensor, p1, v1, p2, v2, land, jac=True):
    """GPInterpolatedProjectionFactorPose3<Cal3_S2>::evaluateError: (e, [H1..H4 (2x6), H5 (2x3)], behind_camera)."""
    e = np.zeros(2)
    H = [np.zeros((2, 6)) for _ in range(4)] + [np.zeros((2, 3))] if jac else [None] * 5
    behind = call("orc_interp_projection_pose3_ds2", A(Lam), A(Psi), A(meas), A(_k9(K)), None if sensor is None else A(sensor),
                  A(p1), A(v1), A(p2), A(v2), A(land), e, *H)
    return e, H, bool(behind)


def interpolate(kind, Lam, Psi, p1, v1, p2, v2, jac=True):
    d = TANGENT_DIM[kind]
    out = np.zeros(POSE_DIM[kind])
    H = [np.zeros((d, d)) for _ in range(4)] if jac else [None] * 4
    Lam, Psi, p1, v1, p2, v2 = A(Lam), A(Psi), A(p1), A(v1), A(p2), A(v2)
    if kind in (LINEAR2, LINEAR3):
        call("orc_interp_linear", d, Lam, Psi, p1, v1, p2, v2, out, *H)
    else:
        name = {POSE2: "orc_interp_pose2", POSE3: "orc_interp_pose3", ROT3: "orc_interp_rot3"}[kind]
        call(name, Lam, Psi, p1, v1, p2, v2, out, *H)
    return out, H


class Chain:
    """Oracle-side chain problem; same call surface as gpslam_amd.ChainSolver."""

    def __init__(self, kind, chart=CHART_EXPMAP, landmark_dim=0, velocity_world=False):
        self.kind, self.chart, self.ld = kind, chart, landmark_dim
        self.d, self.pd = TANGENT_DIM[kind], POSE_DIM[kind]
        self.b = 2 * self.d
        self.N = self.L = 0
        self.n_gp = 0
        self._h = C.c_void_p(lib().orc_chain_create(kind, chart, landmark_dim))
        assert self._h
        if velocity_world:
            assert lib().orc_chain_set_velocity_world(self._h, 1) == 0

    def __del__(self):
        try:
            if self._h:
                lib().orc_chain_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_qc(self, Qc):
        return call("orc_chain_set_qc", self._h, A(Qc))

    def set_states(self, pose, vel):
        pose, vel = A(pose), A(vel)
        self.N = pose.shape[0]
        return call("orc_chain_set_states", self._h, self.N, pose, vel)

    def get_states(self):
        pose, vel = np.zeros((self.N, self.pd)), np.zeros((self.N, self.d))
        call("orc_chain_get_states", self._h, pose, vel)
        return pose, vel

    def set_landmarks(self, pts):
        pts = A(pts).reshape(-1, self.ld)
        self.L = pts.shape[0]
        return call("orc_chain_set_landmarks", self._h, self.L, pts)

    def get_landmarks(self):
        pts = np.zeros((self.L, self.ld))
        call("orc_chain_get_landmarks", self._h, pts)
        return pts

    def add_gp_priors(self, left, dt):
        left, _ = _i(left)
        self.n_gp += len(left)
        return call("orc_chain_add_gp_priors", self._h, len(left), left, A(dt))

    def add_gp_priors_qc(self, left, dt, Qc):
        """one Qc per factor (count x d x d)"""
        left, _ = _i(left)
        self.n_gp += len(left)
        return call("orc_chain_add_gp_priors_qc", self._h, len(left), left, A(dt), A(Qc))

    def set_meas_covariance(self, kind, cov):
        """full Gaussian covariances (count x rows x rows) for the most recently added factors of one GPSLAM_MEAS_* kind"""
        cov = A(cov)
        return call("orc_chain_set_meas_covariance", self._h, 5 + int(kind), cov.shape[0], self.MEAS_ROWS[kind], cov)

    def add_pose_priors(self, idx, prior, sigmas):
        idx, _ = _i(idx)
        return call("orc_chain_add_pose_priors", self._h, len(idx), idx, A(prior), A(sigmas))

    def add_vel_priors(self, idx, prior, sigmas):
        idx, _ = _i(idx)
        return call("orc_chain_add_vel_priors", self._h, len(idx), idx, A(prior), A(sigmas))

    def add_between(self, left, measured, sigmas):
        left, _ = _i(left)
        return call("orc_chain_add_between", self._h, len(left), left, A(measured), A(sigmas))

    def add_between_pairs(self, first, second, measured, sigmas):
        """gtsam::BetweenFactor<Pose>(x_first, x_second, measured) between any two states (loop closures)"""
        first, _ = _i(first)
        second, _ = _i(second)
        assert len(first) == len(second)
        return call("orc_chain_add_between_pairs", self._h, len(first), first, second, A(measured), A(sigmas))

    def add_landmark_priors(self, idx, prior, sigmas):
        idx, _ = _i(idx)
        return call("orc_chain_add_landmark_priors", self._h, len(idx), idx, A(prior), A(sigmas))

    def add_interp_range(self, left, landmark, z, sigma, dt, tau, sensor=None):
        left, _ = _i(left)
        landmark, _ = _i(landmark)
        return call("orc_chain_add_interp_range", self._h, len(left), left, landmark, A(z), A(sigma), A(dt), A(tau),
                    None if sensor is None else A(sensor))

    def add_range(self, idx, landmark, z, sigma):
        idx, _ = _i(idx)
        landmark, _ = _i(landmark)
        return call("orc_chain_add_range", self._h, len(idx), idx, landmark, A(z), A(sigma))

    def add_interp_attitude(self, left, nZ, bRef, sigma, dt, tau):
        left, _ = _i(left)
        return call("orc_chain_add_interp_attitude", self._h, len(left), left, A(nZ), A(bRef), A(sigma), A(dt), A(tau))

    def add_ahrs(self, left, delta_R, dR_dbias, bias_hat, delta_tij, cov, omega_coriolis=None):
        left, _ = _i(left)
        return call("orc_chain_add_ahrs", self._h, len(left), left, A(delta_R), A(dR_dbias), A(bias_hat), A(delta_tij),
                    A(cov), None if omega_coriolis is None else A(omega_coriolis))

    def add_interp_gps(self, left, measured, sigmas, dt, tau, sensor=None):
        left, _ = _i(left)
        return call("orc_chain_add_interp_gps", self._h, len(left), left, A(measured), A(sigmas), A(dt), A(tau),
                    None if sensor is None else A(sensor))

    def add_interp_projection(self, left, landmark, measured, sigmas, dt, tau, K, sensor=None):
        left, _ = _i(left)
        landmark, _ = _i(landmark)
        K = _k9(K)
        return call("orc_chain_add_interp_projection_ds2", self._h, len(left), left, landmark, A(measured), A(sigmas), A(dt),
                    A(tau), A(K), None if sensor is None else A(sensor))

    def add_odometry2d(self, left, measured, sigmas):
        left, _ = _i(left)
        return call("orc_chain_add_odometry2d", self._h, len(left), left, A(measured), A(sigmas))

    def add_bearing_range(self, idx, landmark, bearing, rng, sigmas):
        idx, _ = _i(idx)
        landmark, _ = _i(landmark)
        return call("orc_chain_add_bearing_range", self._h, len(idx), idx, landmark, A(bearing), A(rng), A(sigmas))

    def compile(self):
        return 0

    def linearize_gp(self):
        e = np.zeros((self.n_gp, self.b))
        H = np.zeros((self.n_gp, 4, self.b, self.d))
        call("orc_chain_linearize_gp", self._h, e, H)
        return e, H

    MEAS_ROWS = {0: 1, 1: 1, 2: 2, 3: 3, 4: 3, 5: 2, 6: 2, 7: 3}

    def linearize_meas(self, kind, count):
        """Unwhitened (e, J) per factor of one measurement kind (the HIP ABI's GPSLAM_MEAS_* numbering)."""
        rows = self.MEAS_ROWS[kind]
        e = np.zeros((count, rows))
        J = np.zeros((count, rows, 2 * self.b + 3))
        n = call("orc_chain_linearize_meas", self._h, 5 + int(kind), e, J)
        assert n == count, (n, count)
        return e, J

    def error(self):
        out = C.c_double(0.0)
        rc = lib().orc_chain_error(self._h, C.byref(out))
        assert rc == 0, rc
        return out.value

    def normal_equations(self):
        N, b, nl = self.N, self.b, self.L * self.ld
        D, O, g = np.zeros((N, b, b)), np.zeros((N, b, b)), np.zeros((N, b))
        B = np.zeros((N, b, nl)) if nl else None
        HLL = np.zeros((nl, nl)) if nl else None
        gL = np.zeros(nl) if nl else None
        rc = call("orc_chain_normal_equations", self._h, D, O, g, B, HLL, gL)
        assert rc == 0, rc
        return D, O, g, B, HLL, gL

    def iterate_gn(self):
        st = Stats()
        rc = lib().orc_chain_iterate_gn(self._h, C.byref(st))
        return rc, st

    def iterate_lm(self, lam, params=None):
        st = Stats()
        p = params or default_params(use_lm=1)
        lam_c = C.c_double(lam)
        rc = lib().orc_chain_iterate_lm(self._h, C.byref(lam_c), C.byref(p), C.byref(st))
        return rc, st, lam_c.value

    def optimize(self, params=None):
        st = Stats()
        p = params or default_params()
        rc = lib().orc_chain_optimize(self._h, C.byref(p), C.byref(st))
        return rc, st


def force_envelope_solver(on):
    """tests: every chain through the envelope Cholesky (the solver of graphs with loop closures)"""
    lib().orc_force_envelope_solver(1 if on else 0)


def block_tridiag_solve(D, O, g):
    D, O, g = A(D), A(O), A(g)
    N, b = g.shape
    x = np.zeros((N, b))
    rc = call("orc_block_tridiag_solve", N, b, D, O, g, x)
    assert rc == 0, rc
    return x
