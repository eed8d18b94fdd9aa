"""ctypes binding of libgpslam_hip.so: one ChainSolver = one gpslam_hip_handle = one GPU stream.

Mirrors the reference's usage pattern (NonlinearFactorGraph::add / Values::insert / optimizer.iterate(),
e.g. gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:172-188) on arrays instead of per-factor objects.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

LINEAR2, LINEAR3, POSE2, POSE3, ROT3, ROT3_BIAS = 0, 1, 2, 3, 4, 5
CHART_EXPMAP, CHART_FIRST_ORDER = 0, 1
FP64, FP32 = 0, 1
# gpslam_hip_config_v2.plan: kernel families compile() is told to use instead of its default choice (include/gpslam_hip.h)
PLAN_UNFUSED_LEVEL0, PLAN_COLUMN_LEVEL0, PLAN_LEVELS_OF_FOUR, PLAN_FS_TWO_LAUNCHES, PLAN_GP_ROWS, PLAN_GENERIC_QC, PLAN_MEAS_ROWS, PLAN_SEPARATE_RETRACT = 1, 2, 4, 8, 16, 32, 64, 128
# Test harness hook of this PYTHON mirror (the library itself reads no environment): plan bits OR-ed into every ChainSolver a
# process creates, so that tests/test_gpu_switches.py can re-run whole parity suites on the fallback kernel families.
_DEFAULT_PLAN = int(os.environ.get("GPSLAM_PY_DEFAULT_PLAN", "0"))
POSE_DIM = {LINEAR2: 2, LINEAR3: 3, POSE2: 3, POSE3: 12, ROT3: 9, ROT3_BIAS: 12}
TANGENT_DIM = {LINEAR2: 2, LINEAR3: 3, POSE2: 3, POSE3: 6, ROT3: 3, ROT3_BIAS: 6}

# every symbol include/gpslam_hip.h declares
ABI_SYMBOLS = [
    "gpslam_hip_create", "gpslam_hip_destroy", "gpslam_hip_default_params", "gpslam_hip_last_error",
    "gpslam_hip_stream", "gpslam_hip_set_stream", "gpslam_hip_set_states", "gpslam_hip_get_states", "gpslam_hip_set_landmarks",
    "gpslam_hip_get_landmarks", "gpslam_hip_set_qc", "gpslam_hip_add_gp_priors", "gpslam_hip_add_pose_priors",
    "gpslam_hip_add_vel_priors", "gpslam_hip_add_between", "gpslam_hip_add_landmark_priors",
    "gpslam_hip_add_interp_range", "gpslam_hip_add_range", "gpslam_hip_add_interp_attitude",
    "gpslam_hip_add_interp_gps", "gpslam_hip_add_odometry2d", "gpslam_hip_add_bearing_range", "gpslam_hip_compile",
    "gpslam_hip_linearize_gp", "gpslam_hip_error", "gpslam_hip_iterate_gn", "gpslam_hip_iterate_lm",
    "gpslam_hip_optimize", "gpslam_hip_normal_equations", "gpslam_hip_get_rows", "gpslam_hip_block_tridiag_solve",
    "gpslam_hip_last_timing", "gpslam_hip_run_gn", "gpslam_hip_time_kernel", "gpslam_hip_interface_send", "gpslam_hip_interface_recv",
    "gpslam_hip_iterate_phase1", "gpslam_hip_iterate_phase2", "gpslam_hip_set_halo_state",
    "gpslam_hip_interpolate_poses", "gpslam_hip_add_interp_projection", "gpslam_hip_add_interp_projection_ds2", "gpslam_hip_iterate_phase2a",
    "gpslam_hip_iterate_phase2b", "gpslam_hip_landmark_reduce_buffer", "gpslam_hip_lm_begin",
    "gpslam_hip_lm_trial_phase1", "gpslam_hip_lm_trial_phase2", "gpslam_hip_lm_reject", "gpslam_hip_clear_factors", "gpslam_hip_segment_plan", "gpslam_hip_linearize_meas", "gpslam_hip_interpolate_poses_jac",
    "gpslam_hip_add_ahrs", "gpslam_hip_plan_info", "gpslam_hip_fs_set_split", "gpslam_hip_fs_split_info", "gpslam_hip_fs_set_top",
    "gpslam_hip_fs_interface", "gpslam_hip_fs_phase1", "gpslam_hip_fs_phase2", "gpslam_hip_fs_lm_trial_phase1",
    "gpslam_hip_fs_lm_trial_phase2", "gpslam_hip_add_gp_priors_qc", "gpslam_hip_set_meas_covariance",
    "gpslam_hip_interpolate_velocities", "gpslam_hip_body_centric_velocity", "gpslam_hip_last_level0_ms",
    "gpslam_hip_lm_decide", "gpslam_hip_set_collectives", "gpslam_hip_create_v2", "gpslam_hip_abi_version", "gpslam_hip_struct_size",
    "gpslam_hip_add_between_pairs", "gpslam_hip_set_level0_stamps",
]
# the version of include/gpslam_hip.h this binding's structs mirror (GPSLAM_HIP_ABI_MAJOR / _MINOR); load_library() checks the library's
ABI_MAJOR, ABI_MINOR = 2, 2
STRUCT_CONFIG, STRUCT_CONFIG_V2, STRUCT_STATS, STRUCT_PARAMS = 0, 1, 2, 3


class GpslamHipError(RuntimeError):
    pass


class Config(C.Structure):
    """gpslam_hip_config (v1: the eight knobs as anonymous words); kept for callers of gpslam_hip_create"""
    _fields_ = [("manifold", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32), ("chart", C.c_int32),
                ("landmark_dim", C.c_int32), ("chunk", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
                ("reserved", C.c_int32 * 8)]


class ConfigV2(C.Structure):
    """gpslam_hip_config_v2: struct_size first, every knob by name (include/gpslam_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("manifold", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32), ("chart", C.c_int32),
                ("landmark_dim", C.c_int32), ("chunk", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32),
                ("force_sharded", C.c_int32), ("upper_chunk", C.c_int32), ("top_blocks", C.c_int32), ("velocity", C.c_int32),
                ("segment_length", C.c_int32), ("force_segmented", C.c_int32), ("plan", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("error_before", C.c_double), ("error_after", C.c_double), ("delta_inf_norm", C.c_double),
                ("lambda_", C.c_double), ("iterations", C.c_int32), ("status", C.c_int32),
                ("accepted", C.c_int32), ("trials", C.c_int32), ("last_trial_error", C.c_double)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Params(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("relative_error_tol", C.c_double),
                ("absolute_error_tol", C.c_double), ("error_tol", C.c_double), ("delta_tol", C.c_double),
                ("lambda_initial", C.c_double), ("lambda_factor", C.c_double), ("lambda_upper_bound", C.c_double),
                ("lambda_lower_bound", C.c_double), ("min_model_fidelity", C.c_double), ("use_lm", C.c_int32),
                ("pad", C.c_int32)]


_lib = None


def load_library():
    """Load libgpslam_hip.so, building it with hipcc if needed.  Raises if that is impossible."""
    global _lib
    if _lib is None:
        path = os.environ.get("GPSLAM_LIB") or _build.build()     # GPSLAM_LIB: an alternative build (A/B timing of kernel variants)
        if not os.path.exists(path):
            raise GpslamHipError("libgpslam_hip.so is missing and could not be built; there is no CPU fallback")
        lib = C.CDLL(path)
        lib.gpslam_hip_last_error.restype = C.c_char_p
        lib.gpslam_hip_stream.restype = C.c_void_p
        check_abi(lib, path)
        _lib = lib
    return _lib


def check_abi(lib, path="libgpslam_hip.so"):
    """The structs below are written into by the library: a library built from another major version of the header, or one whose
    struct sizes differ from this file's, is refused at load time (gpslam_hip_stats grew by 8 bytes in round 5 with
    nothing to tell an old caller)."""
    if not hasattr(lib, "gpslam_hip_abi_version"):
        raise GpslamHipError("%s exports no gpslam_hip_abi_version: built from a header older than ABI 2.0" % path)
    lib.gpslam_hip_abi_version.restype = C.c_uint32
    lib.gpslam_hip_struct_size.restype = C.c_size_t
    v = lib.gpslam_hip_abi_version()
    if (v >> 16) != ABI_MAJOR or (v & 0xffff) < ABI_MINOR:
        raise GpslamHipError("%s speaks ABI %d.%d, this binding %d.%d" % (path, v >> 16, v & 0xffff, ABI_MAJOR, ABI_MINOR))
    for which, ty in ((STRUCT_CONFIG, Config), (STRUCT_CONFIG_V2, ConfigV2), (STRUCT_STATS, Stats), (STRUCT_PARAMS, Params)):
        n = lib.gpslam_hip_struct_size(which)
        if n != C.sizeof(ty):
            raise GpslamHipError("%s: sizeof(%s) is %d in the library, %d in this binding" % (path, ty.__name__, n, C.sizeof(ty)))


def lm_decide(s6, lam, lambda_factor=10.0, lambda_upper_bound=1e5, lambda_lower_bound=0.0, min_model_fidelity=1e-3,
              relative_error_tol=1e-5):
    """gpslam_hip_lm_decide (host arithmetic, no GPU): one tryLambda decision from the reduced scalars of a trial.
    Returns (accepted, done, new lambda)."""
    p = Params()
    load_library().gpslam_hip_default_params(C.byref(p))
    p.use_lm = 1
    p.lambda_factor, p.lambda_upper_bound, p.lambda_lower_bound = lambda_factor, lambda_upper_bound, lambda_lower_bound
    p.min_model_fidelity, p.relative_error_tol = min_model_fidelity, relative_error_tol
    s = (C.c_double * 6)(*[float(v) for v in s6])
    lam_c, acc, done = C.c_double(lam), C.c_int32(0), C.c_int32(0)
    rc = load_library().gpslam_hip_lm_decide(s, C.byref(p), C.byref(lam_c), C.byref(acc), C.byref(done))
    if rc:
        raise GpslamHipError("lm_decide: %d" % rc)
    return bool(acc.value), bool(done.value), lam_c.value


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class ChainSolver:
    def __init__(self, kind, chart=CHART_EXPMAP, landmark_dim=0, device=0, chunk=0, rank=0, nranks=1,
                 force_sharded=False, upper_chunk=0, top_blocks=0, velocity_world=False, segment_length=0,
                 force_segmented=False, precision=0, plan=0):
        self.lib = load_library()
        self.kind, self.chart, self.ld = kind, chart, landmark_dim
        self.d, self.pd = TANGENT_DIM[kind], POSE_DIM[kind]
        self.b = 2 * self.d
        self.N = self.L = self.n_gp = 0
        cfg = ConfigV2(struct_size=C.sizeof(ConfigV2), manifold=kind, precision=precision, device=device, chart=chart,
                       landmark_dim=landmark_dim, chunk=chunk, rank=rank, nranks=nranks,
                       force_sharded=1 if force_sharded else 0, upper_chunk=upper_chunk, top_blocks=top_blocks,
                       velocity=1 if velocity_world else 0,        # GPSLAM_VELOCITY_WORLD_VW: the *Pose3VW factor family
                       segment_length=segment_length,              # segmented landmark elimination: states per segment (0 = automatic)
                       force_segmented=1 if force_segmented else 0,   # ... for any landmark count (default: only beyond the dense border)
                       plan=plan | _DEFAULT_PLAN)                  # GPSLAM_PLAN_* bits
        self._h = C.c_void_p()
        rc = self.lib.gpslam_hip_create_v2(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise GpslamHipError("gpslam_hip_create_v2 failed (%d): no usable HIP device, and there is no CPU fallback" % rc)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gpslam_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            msg = self.lib.gpslam_hip_last_error(self._h)
            raise GpslamHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
        return rc

    # ---- variables
    def set_qc(self, Qc):
        Qc = _f64(Qc)
        return self._chk(self.lib.gpslam_hip_set_qc(self._h, _p(Qc)), "set_qc")

    def set_states(self, pose, vel):
        pose, vel = _f64(pose).reshape(-1, self.pd), _f64(vel).reshape(-1, self.d)
        self.N = pose.shape[0]
        return self._chk(self.lib.gpslam_hip_set_states(self._h, self.N, _p(pose), _p(vel)), "set_states")

    def get_states(self):
        pose, vel = np.zeros((self.N, self.pd)), np.zeros((self.N, self.d))
        self._chk(self.lib.gpslam_hip_get_states(self._h, _p(pose), _p(vel)), "get_states")
        return pose, vel

    def set_landmarks(self, pts):
        pts = _f64(pts).reshape(-1, self.ld)
        self.L = pts.shape[0]
        return self._chk(self.lib.gpslam_hip_set_landmarks(self._h, self.L, _p(pts)), "set_landmarks")

    def get_landmarks(self):
        pts = np.zeros((self.L, self.ld))
        self._chk(self.lib.gpslam_hip_get_landmarks(self._h, _p(pts)), "get_landmarks")
        return pts

    # ---- factors
    def add_gp_priors(self, left, dt):
        left, dt = _i32(left), _f64(dt)
        self.n_gp += len(left)
        return self._chk(self.lib.gpslam_hip_add_gp_priors(self._h, len(left), _p(left), _p(dt)), "add_gp_priors")

    def add_gp_priors_qc(self, left, dt, Qc):
        """GP priors with one Qc_model per factor (count x d x d), as the reference's constructors take it."""
        left, dt, Qc = _i32(left), _f64(dt), _f64(Qc)
        self.n_gp += len(left)
        return self._chk(self.lib.gpslam_hip_add_gp_priors_qc(self._h, len(left), _p(left), _p(dt), _p(Qc)), "add_gp_priors_qc")

    def set_meas_covariance(self, kind, cov):
        """noiseModel::Gaussian::Covariance (count x rows x rows) on the most recently added factors of one MEAS_* kind."""
        cov = _f64(cov)
        return self._chk(self.lib.gpslam_hip_set_meas_covariance(self._h, int(kind), cov.shape[0], _p(cov)), "set_meas_covariance")

    def add_pose_priors(self, idx, prior, sigmas):
        idx, prior, sigmas = _i32(idx), _f64(prior), _f64(sigmas)
        return self._chk(self.lib.gpslam_hip_add_pose_priors(self._h, len(idx), _p(idx), _p(prior), _p(sigmas)),
                         "add_pose_priors")

    def add_vel_priors(self, idx, prior, sigmas):
        idx, prior, sigmas = _i32(idx), _f64(prior), _f64(sigmas)
        return self._chk(self.lib.gpslam_hip_add_vel_priors(self._h, len(idx), _p(idx), _p(prior), _p(sigmas)),
                         "add_vel_priors")

    def add_between(self, left, measured, sigmas):
        left, measured, sigmas = _i32(left), _f64(measured), _f64(sigmas)
        return self._chk(self.lib.gpslam_hip_add_between(self._h, len(left), _p(left), _p(measured), _p(sigmas)),
                         "add_between")

    def add_between_pairs(self, first, second, measured, sigmas):
        """gtsam::BetweenFactor<Pose>(x_first, x_second, measured) between any two states: loop closures (consecutive pairs are
        ordinary chain factors)"""
        first, second, measured, sigmas = _i32(first), _i32(second), _f64(measured), _f64(sigmas)
        if len(first) != len(second):
            raise ValueError("add_between_pairs: first and second differ in length")
        return self._chk(self.lib.gpslam_hip_add_between_pairs(self._h, len(first), _p(first), _p(second), _p(measured), _p(sigmas)),
                         "add_between_pairs")

    def add_landmark_priors(self, idx, prior, sigmas):
        idx, prior, sigmas = _i32(idx), _f64(prior), _f64(sigmas)
        return self._chk(self.lib.gpslam_hip_add_landmark_priors(self._h, len(idx), _p(idx), _p(prior), _p(sigmas)),
                         "add_landmark_priors")

    def add_interp_range(self, left, landmark, z, sigma, dt, tau, sensor=None):
        left, landmark = _i32(left), _i32(landmark)
        z, sigma, dt, tau = _f64(z), _f64(sigma), _f64(dt), _f64(tau)
        sensor = None if sensor is None else _f64(sensor)
        return self._chk(self.lib.gpslam_hip_add_interp_range(self._h, len(left), _p(left), _p(landmark), _p(z),
                                                              _p(sigma), _p(dt), _p(tau), _p(sensor)),
                         "add_interp_range")

    def add_range(self, idx, landmark, z, sigma):
        idx, landmark, z, sigma = _i32(idx), _i32(landmark), _f64(z), _f64(sigma)
        return self._chk(self.lib.gpslam_hip_add_range(self._h, len(idx), _p(idx), _p(landmark), _p(z), _p(sigma)),
                         "add_range")

    def add_interp_attitude(self, left, nZ, bRef, sigma, dt, tau):
        left = _i32(left)
        nZ, bRef, sigma, dt, tau = _f64(nZ), _f64(bRef), _f64(sigma), _f64(dt), _f64(tau)
        return self._chk(self.lib.gpslam_hip_add_interp_attitude(self._h, len(left), _p(left), _p(nZ), _p(bRef),
                                                                 _p(sigma), _p(dt), _p(tau)), "add_interp_attitude")

    def add_ahrs(self, left, delta_R, dR_dbias, bias_hat, delta_tij, cov, omega_coriolis=None):
        """gtsam::AHRSFactor(x_left, x_left+1, b_left, pim, omegaCoriolis) -- matlab/GPAHRSexample.m:131-137; the
        arrays are the PreintegratedAhrsMeasurements state per factor (gpslam_amd.ahrs.Preintegrated)."""
        left = _i32(left)
        delta_R, dR_dbias, bias_hat = _f64(delta_R), _f64(dR_dbias), _f64(bias_hat)
        delta_tij, cov = _f64(delta_tij), _f64(cov)
        cor = None if omega_coriolis is None else _f64(omega_coriolis)
        return self._chk(self.lib.gpslam_hip_add_ahrs(self._h, len(left), _p(left), _p(delta_R), _p(dR_dbias), _p(bias_hat),
                                                      _p(delta_tij), _p(cov), None if cor is None else _p(cor)), "add_ahrs")

    def add_interp_gps(self, left, measured, sigmas, dt, tau, sensor=None):
        left = _i32(left)
        measured, sigmas, dt, tau = _f64(measured), _f64(sigmas), _f64(dt), _f64(tau)
        sensor = None if sensor is None else _f64(sensor)
        return self._chk(self.lib.gpslam_hip_add_interp_gps(self._h, len(left), _p(left), _p(measured), _p(sigmas),
                                                            _p(dt), _p(tau), _p(sensor)), "add_interp_gps")

    def add_interp_projection(self, left, landmark, measured, sigmas, dt, tau, K, sensor=None):
        left, landmark = _i32(left), _i32(landmark)
        measured, sigmas, dt, tau, K = _f64(measured), _f64(sigmas), _f64(dt), _f64(tau), _f64(K)
        sensor = None if sensor is None else _f64(sensor)
        if K.size == 9:      # gtsam::Cal3DS2: fx, fy, s, u0, v0, k1, k2, p1, p2
            return self._chk(self.lib.gpslam_hip_add_interp_projection_ds2(
                self._h, len(left), _p(left), _p(landmark), _p(measured), _p(sigmas), _p(dt), _p(tau), _p(K),
                None if sensor is None else _p(sensor)), "add_interp_projection_ds2")
        return self._chk(self.lib.gpslam_hip_add_interp_projection(
            self._h, len(left), _p(left), _p(landmark), _p(measured), _p(sigmas), _p(dt), _p(tau), _p(K),
            None if sensor is None else _p(sensor)), "add_interp_projection")

    def add_odometry2d(self, left, measured, sigmas):
        left, measured, sigmas = _i32(left), _f64(measured), _f64(sigmas)
        return self._chk(self.lib.gpslam_hip_add_odometry2d(self._h, len(left), _p(left), _p(measured), _p(sigmas)),
                         "add_odometry2d")

    def add_bearing_range(self, idx, landmark, bearing, rng, sigmas):
        idx, landmark = _i32(idx), _i32(landmark)
        bearing, rng, sigmas = _f64(bearing), _f64(rng), _f64(sigmas)
        return self._chk(self.lib.gpslam_hip_add_bearing_range(self._h, len(idx), _p(idx), _p(landmark), _p(bearing),
                                                               _p(rng), _p(sigmas)), "add_bearing_range")

    def clear_factors(self):
        self.n_gp = 0
        return self._chk(self.lib.gpslam_hip_clear_factors(self._h), "clear_factors")

    def compile(self):
        return self._chk(self.lib.gpslam_hip_compile(self._h), "compile")

    # ---- hot path
    def linearize_gp(self, jac=True):
        e = np.zeros((self.n_gp, self.b))
        H = np.zeros((self.n_gp, 4, self.b, self.d)) if jac else None
        self._chk(self.lib.gpslam_hip_linearize_gp(self._h, _p(e), _p(H)), "linearize_gp")
        return e, H

    MEAS_ROWS = {0: 1, 1: 1, 2: 2, 3: 3, 4: 3, 5: 2, 6: 2, 7: 3}

    def linearize_meas(self, kind, count):
        """Unwhitened (e, J) of the `count` measurement factors of one kind (MEAS_* order of include/gpslam_hip.h):
        e (count, rows), J (count, rows, 4d + 3) = [H1 | H2 | H3 | H4 | H5 padded to 3]."""
        rows = self.MEAS_ROWS[kind]
        e = np.zeros((count, rows))
        J = np.zeros((count, rows, 2 * self.b + 3))
        n = self._chk(self.lib.gpslam_hip_linearize_meas(self._h, int(kind), _p(e), _p(J)), "linearize_meas")
        assert n == count, (n, count)
        return e, J

    def error(self):
        out = C.c_double(0.0)
        self._chk(self.lib.gpslam_hip_error(self._h, C.byref(out)), "error")
        return out.value

    def iterate_gn(self):
        st = Stats()
        rc = self.lib.gpslam_hip_iterate_gn(self._h, C.byref(st))
        self._chk(rc, "iterate_gn")
        return rc, st

    def iterate_lm(self, lam, params=None):
        st = Stats()
        p = params or self.default_params(use_lm=1)
        lam_c = C.c_double(lam)
        rc = self.lib.gpslam_hip_iterate_lm(self._h, C.byref(lam_c), C.byref(p), C.byref(st))
        self._chk(rc, "iterate_lm")
        return rc, st, lam_c.value

    def optimize(self, params=None):
        st = Stats()
        p = params or self.default_params()
        rc = self.lib.gpslam_hip_optimize(self._h, C.byref(p), C.byref(st))
        self._chk(rc, "optimize")
        return rc, st

    def run_gn(self, iters, timed=False):
        st = Stats()
        t = np.zeros(5) if timed else None
        self._chk(self.lib.gpslam_hip_run_gn(self._h, int(iters), C.byref(st), _p(t)), "run_gn")
        return st, t

    def default_params(self, **kw):
        p = Params()
        self.lib.gpslam_hip_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    # ---- inspection
    def normal_equations(self):
        N, b, nl = self.N, self.b, self.L * self.ld
        D, O, g = np.zeros((N, b, b)), np.zeros((N, b, b)), np.zeros((N, b))
        B = np.zeros((N, b, nl)) if nl else None
        self._chk(self.lib.gpslam_hip_normal_equations(self._h, _p(D), _p(O), _p(g), _p(B)), "normal_equations")
        return D, O, g, B

    def get_rows(self):
        """(rowLR, rowE, rowM, rowLm) of the current linearisation (whitened)."""
        n = C.c_int32(0)
        self._chk(self.lib.gpslam_hip_get_rows(self._h, C.byref(n), None, None, None, None), "get_rows")
        M = n.value
        LR, E = np.zeros((M, 2 * self.b)), np.zeros(M)
        Mm = np.zeros((M, self.ld)) if self.ld else None
        Lm = np.full(M, -1, dtype=np.int32) if self.ld else None
        self._chk(self.lib.gpslam_hip_get_rows(self._h, C.byref(n), _p(LR), _p(E), _p(Mm), _p(Lm)), "get_rows")
        return LR, E, Mm, Lm

    def block_tridiag_solve(self, D, O, g):
        D, O, g = _f64(D), _f64(O), _f64(g)
        x = np.zeros_like(g)
        self._chk(self.lib.gpslam_hip_block_tridiag_solve(self._h, g.shape[0], _p(D), _p(O), _p(g), _p(x)),
                  "block_tridiag_solve")
        return x

    def interpolate_poses(self, left, dt, tau):
        """Batched interpolatePose of the current estimate: (count, pose_dim)."""
        left, dt, tau = _i32(left), _f64(dt), _f64(tau)
        out = np.zeros((len(left), self.pd))
        self._chk(self.lib.gpslam_hip_interpolate_poses(self._h, len(left), _p(left), _p(dt), _p(tau), _p(out)),
                  "interpolate_poses")
        return out

    def interpolate_velocities(self, left, dt, tau, jac=False):
        """Batched GaussianProcessInterpolatorLinear::interpolateVelocity of the current estimate: (count, d) [, H (count, 4, d, d)]."""
        left, dt, tau = _i32(left), _f64(dt), _f64(tau)
        out = np.zeros((len(left), self.d))
        H = np.zeros((len(left), 4, self.d, self.d)) if jac else None
        self._chk(self.lib.gpslam_hip_interpolate_velocities(self._h, len(left), _p(left), _p(dt), _p(tau), _p(out), _p(H)),
                  "interpolate_velocities")
        return (out, H) if jac else out

    def body_centric_velocity(self, pose1, pose2, dt, spatial=False):
        """getBodyCentricVb (spatial=False) / getBodyCentricVs of pose pairs (count x 12 each): (count, 6)."""
        pose1, pose2 = _f64(pose1).reshape(-1, 12), _f64(pose2).reshape(-1, 12)
        dt = _f64(np.broadcast_to(np.asarray(dt, dtype=np.float64), (len(pose1),)))
        out = np.zeros((len(pose1), 6))
        self._chk(self.lib.gpslam_hip_body_centric_velocity(self._h, 1 if spatial else 0, len(pose1), _p(pose1), _p(pose2), _p(dt), _p(out)),
                  "body_centric_velocity")
        return out

    def plan_info(self):
        """What compile() chose: dict(levels, chunk0, chunk_upper, fused, structured_gp, rows_full, rows_compact, R)."""
        out = (C.c_int32 * 8)()
        self._chk(self.lib.gpslam_hip_plan_info(self._h, out), "plan_info")
        return dict(zip(("levels", "chunk0", "chunk_upper", "fused", "structured_gp", "rows_full", "rows_compact", "R"), list(out)))

    def segment_plan(self):
        """dict of the segmented landmark elimination's plan (active, C, K, NB, NC, NCP, levels, links)."""
        out = np.zeros(8, dtype=np.int32)
        self._chk(self.lib.gpslam_hip_segment_plan(self._h, _p(out)), "segment_plan")
        return dict(zip(("active", "C", "K", "NB", "NC", "NCP", "levels", "links"), (int(v) for v in out)))

    def interpolate_poses_jac(self, left, dt, tau):
        """interpolatePose with H1..H4: (poses (count, pose_dim), H (count, 4, d, d))."""
        left, dt, tau = _i32(left), _f64(dt), _f64(tau)
        out = np.zeros((len(left), self.pd))
        H = np.zeros((len(left), 4, self.d, self.d))
        self._chk(self.lib.gpslam_hip_interpolate_poses_jac(self._h, len(left), _p(left), _p(dt), _p(tau), _p(out), _p(H)),
                  "interpolate_poses_jac")
        return out, H

    def set_level0_stamps(self, on=True):
        """timed iterations stamp the fused level-0 launch with its own dispatch events (exact duration; perturbs the other phases)"""
        return self._chk(self.lib.gpslam_hip_set_level0_stamps(self._h, 1 if on else 0), "set_level0_stamps")

    def last_level0_ms(self):
        """device ms of the level-0 forward launch, summed over the iterations of the last timed run_gn / iterate_gn"""
        out = C.c_double(0.0)
        self._chk(self.lib.gpslam_hip_last_level0_ms(self._h, C.byref(out)), "last_level0_ms")
        return out.value

    def last_timing(self):
        t = np.zeros(5)
        self.lib.gpslam_hip_last_timing(self._h, _p(t))
        return t

    def time_kernel(self, which, reps=10):
        out = C.c_double(0.0)
        self._chk(self.lib.gpslam_hip_time_kernel(self._h, int(which), int(reps), C.byref(out)), "time_kernel")
        return out.value

    def stream(self):
        return self.lib.gpslam_hip_stream(self._h)

    def set_stream(self, hip_stream):
        return self._chk(self.lib.gpslam_hip_set_stream(self._h, C.c_void_p(hip_stream)), "set_stream")

    def set_collectives(self, all_gather, all_reduce_sum=None):
        """gpslam_hip_set_collectives: hand the library the host's collectives, after which iterate_gn / run_gn / iterate_lm /
        optimize / error work on this rank's handle (or split piece) and return the whole chain's statistics.
            all_gather(send_ptr, recv_ptr, bytes_per_rank, hip_stream)     all_reduce_sum(buf_ptr, n_doubles, hip_stream)
        are Python callables working on raw device pointers, on the handle's stream; an exception becomes GPSLAM_E_COMM."""
        def guard(fn):
            def run(_user, *args):
                try:
                    fn(*args)
                    return 0
                except Exception:       # noqa: BLE001 -- no exception may cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return 1
            return run
        self._cb_gather = ALL_GATHER_FN(guard(all_gather)) if all_gather else ALL_GATHER_FN()
        self._cb_reduce = ALL_REDUCE_FN(guard(all_reduce_sum)) if all_reduce_sum else ALL_REDUCE_FN()
        return self._chk(self.lib.gpslam_hip_set_collectives(self._h, self._cb_gather, self._cb_reduce, None), "set_collectives")

    # ---- segment sharding (nranks > 1)
    def set_halo_state(self, pose, vel):
        pose, vel = _f64(pose), _f64(vel)
        return self._chk(self.lib.gpslam_hip_set_halo_state(self._h, _p(pose), _p(vel)), "set_halo_state")

    def interface_buffers(self):
        """(send_ptr, send_bytes, recv_ptr, recv_bytes): device pointers of this rank's interface record and of
        the gathered records of all ranks."""
        sp, rp = C.c_void_p(), C.c_void_p()
        sb, rb = C.c_size_t(), C.c_size_t()
        self._chk(self.lib.gpslam_hip_interface_send(self._h, C.byref(sp), C.byref(sb)), "interface_send")
        self._chk(self.lib.gpslam_hip_interface_recv(self._h, C.byref(rp), C.byref(rb)), "interface_recv")
        return sp.value, sb.value, rp.value, rb.value

    def iterate_phase1(self, lam=0.0):
        return self._chk(self.lib.gpslam_hip_iterate_phase1(self._h, C.c_double(lam)), "iterate_phase1")

    def iterate_phase2a(self):
        return self._chk(self.lib.gpslam_hip_iterate_phase2a(self._h), "iterate_phase2a")

    def iterate_phase2b(self, want_stats=True):
        st = Stats()
        self._chk(self.lib.gpslam_hip_iterate_phase2b(self._h, C.byref(st) if want_stats else None), "iterate_phase2b")
        return st

    def lm_begin(self):
        return self._chk(self.lib.gpslam_hip_lm_begin(self._h), "lm_begin")

    def lm_trial_phase1(self, lam):
        return self._chk(self.lib.gpslam_hip_lm_trial_phase1(self._h, C.c_double(lam)), "lm_trial_phase1")

    def lm_trial_phase2(self):
        """[error, trial error, |delta|_inf, delta.g, |delta|^2, indefinite flag] of this rank."""
        out = np.zeros(6)
        self._chk(self.lib.gpslam_hip_lm_trial_phase2(self._h, _p(out)), "lm_trial_phase2")
        return out

    def lm_reject(self):
        return self._chk(self.lib.gpslam_hip_lm_reject(self._h), "lm_reject")

    def landmark_reduce_buffer(self):
        """(device pointer, bytes) of this rank's landmark Schur complement [S | gL]; (None, 0) without landmarks."""
        ptr, nb = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.gpslam_hip_landmark_reduce_buffer(self._h, C.byref(ptr), C.byref(nb)), "landmark_reduce_buffer")
        return ptr.value, nb.value

    # ---- config 4 across GPUs: pieces of a chain joined at shared cut states (include/gpslam_hip.h, fs_set_split)
    def fs_set_split(self, rank, nranks, first_lm=(), last_lm=()):
        f = np.ascontiguousarray(first_lm, dtype=np.int32)
        l = np.ascontiguousarray(last_lm, dtype=np.int32)
        return self._chk(self.lib.gpslam_hip_fs_set_split(self._h, C.c_int32(rank), C.c_int32(nranks), _p(f) if len(f) else None, C.c_int32(len(f)),
                                                          _p(l) if len(l) else None, C.c_int32(len(l))), "fs_set_split")

    def fs_split_info(self):
        out = (C.c_int32 * 4)()
        self._chk(self.lib.gpslam_hip_fs_split_info(self._h, out), "fs_split_info")
        return dict(fat_block=out[0], fat_blocks=out[1], segment_length=out[2], nb_top=out[3])

    def fs_set_top(self, nb_top):
        return self._chk(self.lib.gpslam_hip_fs_set_top(self._h, C.c_int32(nb_top)), "fs_set_top")

    def fs_interface(self):
        """(send pointer, bytes, recv pointer, bytes) of the interface record and of the gathered records."""
        sp, rp = C.c_void_p(), C.c_void_p()
        sb, rb = C.c_size_t(), C.c_size_t()
        self._chk(self.lib.gpslam_hip_fs_interface(self._h, C.byref(sp), C.byref(sb), C.byref(rp), C.byref(rb)), "fs_interface")
        return sp.value, sb.value, rp.value, rb.value

    def fs_phase1(self, lam=0.0):
        return self._chk(self.lib.gpslam_hip_fs_phase1(self._h, C.c_double(lam)), "fs_phase1")

    def fs_phase2(self, want_stats=True):
        st = Stats()
        self._chk(self.lib.gpslam_hip_fs_phase2(self._h, C.byref(st) if want_stats else None), "fs_phase2")
        return st

    def fs_lm_trial_phase1(self, lam):
        return self._chk(self.lib.gpslam_hip_fs_lm_trial_phase1(self._h, C.c_double(lam)), "fs_lm_trial_phase1")

    def fs_lm_trial_phase2(self):
        """[error, trial error, |delta|_inf, delta.g, |delta|^2, indefinite flag] of this piece."""
        out = np.zeros(6)
        self._chk(self.lib.gpslam_hip_fs_lm_trial_phase2(self._h, _p(out)), "fs_lm_trial_phase2")
        return out

    def iterate_phase2(self, want_stats=True):
        st = Stats()
        self._chk(self.lib.gpslam_hip_iterate_phase2(self._h, C.byref(st) if want_stats else None), "iterate_phase2")
        return st
