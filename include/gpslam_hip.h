/*
 * gpslam_hip.h -- C ABI of the MI355X-native GP-SLAM Gauss-Newton / Levenberg-Marquardt inner loop.
 *
 * This is the drop-in boundary.  In the reference (gtrll/gpslam) the hot path
 * sits behind GTSAM's NonlinearFactor virtual interface: every factor overrides
 *     gtsam::Vector evaluateError(x1, ..., boost::optional<gtsam::Matrix&> H1, ...)
 * (e.g. gpslam/gp/GaussianProcessPriorPose3.h:60-64) and GTSAM's optimizers call it factor by
 * factor, whiten, eliminate and retract on the CPU
 * (call sites: gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:185-188, matlab/PlazaPose2.m:208-228).
 * Here the whole chain-structured graph lives in HBM behind an opaque handle and each of those steps
 * is a batched HIP kernel.  The C++ host classes in gpslam_amd/host/ (same names and ctor signatures
 * as gpslam.h / GTSAM) and the Python mirror in gpslam_amd/ bind to exactly these entry points.
 *
 * Conventions
 *  - All functions return 0 on success, <0 on failure (GPSLAM_E_*); no exceptions cross the ABI.
 *  - The caller owns every host buffer; the library owns all device memory behind the handle.
 *  - One handle = one HIP stream; a handle is thread-compatible, not thread-safe.
 *  - Host arrays are row-major doubles.  Pose layouts (per state):
 *        LINEAR2/LINEAR3: D doubles;   POSE2: (x, y, theta);   ROT3: R row-major (9);
 *        POSE3: R row-major (9) then t (3).   Velocities: d doubles, d = tangent dimension
 *        (POSE3: (omega, v) rotation first; POSE2: (vx, vy, omega)).
 *  - State i is the pair (pose_i, vel_i); the variable ordering is the explicit chain order
 *    [x0, v0, x1, v1, ..., l0, l1, ...].  Factors that couple two states couple i and i+1 only
 *    (the GP Markov property, gpslam/gp/GaussianProcessPriorPose3.h:43-47).
 *  - Noise models: the GP prior uses Q(dt) built from Qc (gpslam/gp/GPutils.h:24-41) -- the handle's shared one or one per
 *    factor (add_gp_priors_qc); every other factor takes diagonal sigmas, or a full Gaussian covariance through
 *    gpslam_hip_set_meas_covariance.
 */
#ifndef GPSLAM_HIP_H
#define GPSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpslam_hip_handle gpslam_hip_handle;

enum { GPSLAM_LINEAR2 = 0, GPSLAM_LINEAR3 = 1, GPSLAM_POSE2 = 2, GPSLAM_POSE3 = 3, GPSLAM_ROT3 = 4, GPSLAM_ROT3_BIAS = 5 };
/* GPSLAM_ROT3_BIAS: the AHRS state of matlab/GPAHRSexample.m:69-202 (gtsam keys x_i Rot3, v_i Vector3, b_i Vector3):
 *   pose slot (12 doubles) = [R row-major (9) | gyroscope bias b_i (3)],  tangent (dtheta, db)
 *   velocity slot (6)      = [angular velocity v_i (3) | 0 0 0],           the three pads stay zero
 * set_qc takes the 3 x 3 Qc of GaussianProcessPriorRot3; add_gp_priors = GaussianProcessPriorRot3 on (x, v);
 * add_pose_priors = [PriorFactorRot3 ; PriorFactorVector(b)] and add_between = [BetweenFactorRot3 ; BetweenFactorVector(b)]
 * with 6 sigmas each, INFINITY switching a half off; add_vel_priors: 6 sigmas, the last three ignored;
 * add_interp_attitude = GPInterpolatedAttitudeFactorRot3; gpslam_hip_add_ahrs = gtsam::AHRSFactor. */
/* retract chart of the state update (and, for POSE2, of PriorFactor / BetweenFactor's Local):
 * EXPMAP       x Exp(delta) on every manifold (GTSAM >= 4.1 defaults; GTSAM_ROT3_EXPMAP / GTSAM_POSE3_EXPMAP /
 *              SLOW_BUT_CORRECT_EXPMAP builds of 4.0)
 * FIRST_ORDER  GTSAM 4.0's default charts: POSE2 x * Pose2(dx, dy, dtheta); ROT3 R * Cayley(w);
 *              POSE3 (R * Cayley(w), t + R v).  All charts agree to first order: the fixed point is the same. */
enum { GPSLAM_CHART_EXPMAP = 0, GPSLAM_CHART_FIRST_ORDER = 1 };
enum { GPSLAM_FP64 = 0, GPSLAM_FP32 = 1 };
/* velocity parameterisation of a POSE3 chain (config_v2.velocity; v1: config.reserved[3]):
 * BODY      6-vector body-frame velocity (w, v): GaussianProcessPriorPose3 / InterpolatorPose3 (gpslam.h:32-35, :72-77)
 * WORLD_VW  world-frame translational v and rotational w, stored as [v; w]: GaussianProcessPriorPose3VW,
 *           GaussianProcessInterpolatorPose3VW, GPInterpolatedGPSFactorPose3VW (gpslam.h:38-41, :65-70;
 *           gpslam/gp/GaussianProcessPriorPose3VW.h:62-117).  gtsam keys (x_i, v_i, w_i) map to one state. */
enum { GPSLAM_VELOCITY_BODY = 0, GPSLAM_VELOCITY_WORLD_VW = 1 };

enum {
  GPSLAM_OK = 0,
  GPSLAM_E_INVALID = -1,      /* bad argument / wrong manifold for this factor */
  GPSLAM_E_HIP = -2,          /* a HIP runtime call failed */
  GPSLAM_E_NOT_SPD = -3,      /* a pivot block was not positive definite (indeterminate system) */
  GPSLAM_E_NOT_COMPILED = -4, /* gpslam_hip_compile() has not been called since the last change */
  GPSLAM_E_UNSUPPORTED = -5,
  GPSLAM_E_NAN = -6,
  GPSLAM_E_COMM = -7          /* a collective supplied through gpslam_hip_set_collectives reported a failure */
};

/* ---- ABI version (round 6).  Everything a caller's struct layouts depend on is behind one number: the library reports the
 * version it was built from (gpslam_hip_abi_version) and the sizes of its structs (gpslam_hip_struct_size); a binding checks
 * both once at load time (gpslam_amd/chain.py: load_library; gpslam_amd/host/gpslam_host.hpp: check_abi) instead of finding out
 * through a corrupted stack.  History: 1.0 rounds 1-4 (gpslam_hip_stats 48 bytes); 1.1 round 5 (stats 56 bytes: trials,
 * last_trial_error; GPSLAM_E_COMM; plan bits 64, 128) -- shipped without a version symbol; 2.0 this header: gpslam_hip_config_v2 +
 * gpslam_hip_create_v2 (named fields, struct_size first), gpslam_hip_abi_version, gpslam_hip_struct_size.  The v1 config and
 * gpslam_hip_create stay, bit for bit; 2.1 gpslam_hip_add_between_pairs (loop closures); 2.2 gpslam_hip_set_level0_stamps.  A MAJOR bump changes a struct or a
 * signature, a MINOR bump only adds. */
#define GPSLAM_HIP_ABI_MAJOR 2
#define GPSLAM_HIP_ABI_MINOR 2
#define GPSLAM_HIP_ABI_VERSION ((GPSLAM_HIP_ABI_MAJOR << 16) | GPSLAM_HIP_ABI_MINOR)
uint32_t gpslam_hip_abi_version(void);
enum { GPSLAM_STRUCT_CONFIG = 0, GPSLAM_STRUCT_CONFIG_V2 = 1, GPSLAM_STRUCT_STATS = 2, GPSLAM_STRUCT_PARAMS = 3 };
/* sizeof() of the named struct in the library's build (0: unknown struct) */
size_t gpslam_hip_struct_size(int32_t which);

/* What a binding written against gpslam.h's class list fills in (gpslam_hip_create_v2).  struct_size = sizeof(gpslam_hip_config_v2)
 * of the CALLER's header: a library that knows a longer struct reads the fields the caller has and takes defaults (0) for the rest;
 * a library that knows a shorter one accepts the call only if the bytes it does not know are zero (GPSLAM_E_UNSUPPORTED otherwise). */
typedef struct {
  uint32_t struct_size;     /* sizeof(gpslam_hip_config_v2) */
  int32_t manifold;         /* GPSLAM_LINEAR2 .. GPSLAM_ROT3_BIAS */
  int32_t precision;        /* GPSLAM_FP64 | GPSLAM_FP32 (a tolerance mode, see gpslam_hip_config.precision) */
  int32_t device;           /* HIP device ordinal */
  int32_t chart;            /* GPSLAM_CHART_* */
  int32_t landmark_dim;     /* 0 (no landmarks), 2 or 3 */
  int32_t chunk;            /* level-0 chunk length of the partitioned solver; 0 = default */
  int32_t rank, nranks;     /* contiguous-segment sharding: this handle owns segment `rank` of `nranks` */
  int32_t force_sharded;    /* 1: the sharded code path on a single segment (self-test of the exchange plumbing)      v1 reserved[0] */
  int32_t upper_chunk;      /* chunk length of the solver levels above level 0; 0 / 1 = default (LDS-resident groups)  v1 reserved[1] */
  int32_t top_blocks;       /* size of the sequential top level; 0 = default                                          v1 reserved[2] */
  int32_t velocity;         /* GPSLAM_VELOCITY_* (POSE3 only): body (w, v) or the *Pose3VW family's world [v; w]      v1 reserved[3] */
  int32_t segment_length;   /* segmented landmark elimination: states per segment; 0 = the smallest that fits         v1 reserved[4] */
  int32_t force_segmented;  /* 1: the segmented landmark elimination for any landmark count                           v1 reserved[5] */
  int32_t plan;             /* mask of GPSLAM_PLAN_* bits; 0 = the default plan                                       v1 reserved[6] */
} gpslam_hip_config_v2;

/* v1 (rounds 1-5): the same eight knobs as anonymous words.  Kept so that callers built against the old header keep working. */
typedef struct {
  int32_t manifold;      /* GPSLAM_LINEAR2 .. GPSLAM_ROT3 */
  int32_t precision;     /* GPSLAM_FP64, or GPSLAM_FP32: fp32 Jacobian rows (evaluated re-centred, with the exact derivative in place of the
                          * h = 1e-6 difference), fp64 residual, normal equations and solver.  A TOLERANCE mode
                          * (north_star's 1e-5-relative sweep, half the row-table footprint), not a throughput mode: the structured
                          * GP-prior records of the fp64 path are fp64-only, fp32 rows are slower per iteration on every measured mix */
  int32_t device;        /* HIP device ordinal */
  int32_t chart;         /* GPSLAM_CHART_* */
  int32_t landmark_dim;  /* 0 (no landmarks), 2 or 3 */
  int32_t chunk;         /* level-0 chunk length of the partitioned solver; 0 = default */
  int32_t rank, nranks;  /* contiguous-segment sharding: this handle owns segment `rank` of `nranks` */
  int32_t reserved[8];   /* [0] force the sharded code path, [1] upper-level chunk length, [2] sequential top size,
                          * [3] GPSLAM_VELOCITY_* (POSE3 only), [4] segment length of the segmented landmark elimination
                          * (0 = smallest that fits), [5] 1 = use the segmented landmark elimination for any landmark
                          * count (default: only when the landmarks do not fit the dense border), [6] mask of GPSLAM_PLAN_* bits
                          * (below; 0 = the default plan); [7] must be 0 */
} gpslam_hip_config;
/* config_v2.plan (v1: config.reserved[6]): kernel families compile() may be told to use instead of its default choice.  Every one of them is the
 * path some graphs take anyway (chains with landmark columns, pinned hierarchy shapes, wide landmark borders, graphs whose
 * full-width rows are not all GP priors); the bits exist so that tests run them on plain chains and so that a maintainer
 * can A/B them per handle (gpslam_hip_plan_info shows what a handle ended up with).  No reference counterpart. */
enum {
  GPSLAM_PLAN_UNFUSED_LEVEL0 = 1,     /* assembly and level-0 elimination as two launches (k_assemble_ghost + k_chunk_forward_rows) */
  GPSLAM_PLAN_COLUMN_LEVEL0 = 2,      /* column-layout level-0 elimination (k_chunk_forward; implies the two launches) */
  GPSLAM_PLAN_LEVELS_OF_FOUR = 4,     /* upper hierarchy as one launch per level of chunks of four instead of LDS-resident cyclic reduction */
  GPSLAM_PLAN_FS_TWO_LAUNCHES = 8,    /* segmented landmark elimination: border sweep and Schur complement as two launches through Y */
  GPSLAM_PLAN_GP_ROWS = 16,           /* GP priors as plain Jacobian rows instead of structured records */
  GPSLAM_PLAN_GENERIC_QC = 32,        /* SE(3) records: the general (upper-triangular) chol(Qc^-1) form even when Qc is diagonal */
  GPSLAM_PLAN_MEAS_ROWS = 64,         /* SE(3) records: interpolated GPS factors as plain 24-column rows (k_fused_level0<3>) instead of 16-double lines (<4>) */
  GPSLAM_PLAN_SEPARATE_RETRACT = 128  /* gpslam_hip_run_gn: every iteration ends with its own k_retract launch instead of leaving the retraction to the next K1 */
};

/* per-call statistics; mirrors what GTSAM's optimizers expose (error(), iterations(), lambda()) */
typedef struct {
  double error_before;    /* 0.5 * sum |R e|^2 at the linearisation point */
  double error_after;     /* after the accepted update */
  double delta_inf_norm;  /* max |delta| over all variables */
  double lambda;          /* LM damping after the step */
  int32_t iterations;
  int32_t status;
  int32_t accepted;
  int32_t trials;         /* iterate_lm: lambdas tried by this call (0 elsewhere) */
  double last_trial_error;/* iterate_lm: error at the last lambda tried, kept or not; lets a caller see how far the cost moved on a
                           * trial that was not kept (a change at rounding level makes the fidelity test a ratio of two rounding
                           * errors: tests compare lambda schedules only above it) */
} gpslam_hip_stats;

/* GaussNewtonParams / LevenbergMarquardtParams (GTSAM names; defaults = GTSAM 4.0 defaults) */
typedef struct {
  int32_t max_iterations;     /* 100 */
  double relative_error_tol;  /* 1e-5 */
  double absolute_error_tol;  /* 1e-5 */
  double error_tol;           /* 0 */
  double delta_tol;           /* if > 0 additionally stop when |delta|_inf < delta_tol */
  double lambda_initial;      /* 1e-5 */
  double lambda_factor;       /* 10 */
  double lambda_upper_bound;  /* 1e5 */
  double lambda_lower_bound;  /* 0 */
  double min_model_fidelity;  /* 1e-3 */
  int32_t use_lm;             /* 0 Gauss-Newton, 1 Levenberg-Marquardt */
  int32_t pad;
} gpslam_hip_params;

/* ---- life cycle ---- */
int gpslam_hip_create(const gpslam_hip_config *cfg, gpslam_hip_handle **out);
int gpslam_hip_create_v2(const gpslam_hip_config_v2 *cfg, gpslam_hip_handle **out);
int gpslam_hip_destroy(gpslam_hip_handle *h);
void gpslam_hip_default_params(gpslam_hip_params *p);
const char *gpslam_hip_last_error(const gpslam_hip_handle *h);
/* the HIP stream (hipStream_t) all kernels of this handle are launched on */
void *gpslam_hip_stream(gpslam_hip_handle *h);
/* run on a caller-owned stream instead (e.g. the stream RCCL collectives are ordered against) */
int gpslam_hip_set_stream(gpslam_hip_handle *h, void *hip_stream);

/* ---- variables (replaces gtsam::Values::insert / at, e.g. testGaussianProcessPriorPose3.cpp:179-183) ---- */
int gpslam_hip_set_states(gpslam_hip_handle *h, int32_t N, const double *pose, const double *vel);
int gpslam_hip_get_states(gpslam_hip_handle *h, double *pose, double *vel);
int gpslam_hip_set_landmarks(gpslam_hip_handle *h, int32_t L, const double *pts);
int gpslam_hip_get_landmarks(gpslam_hip_handle *h, double *pts);
/* power-spectral density Qc (d x d, SPD) shared by all GP priors/interpolators
 * (replaces the Qc_model ctor argument + getQc, gpslam/gp/GPutils.cpp:16-20) */
int gpslam_hip_set_qc(gpslam_hip_handle *h, const double *Qc);

/* ---- factors (replace NonlinearFactorGraph::add of the named class) ---- */
/* GaussianProcessPrior{Linear,Pose2,Pose3,Rot3}(key_i, vel_i, key_i+1, vel_i+1, dt, Qc_model)
 * gpslam/gp/GaussianProcessPriorPose3.h:43-49 and siblings */
int gpslam_hip_add_gp_priors(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt);
/* ... with ONE Qc_model PER FACTOR, as the reference's constructors take it (gpslam/gp/GaussianProcessPriorPose3.h:43-49,
 * GaussianProcessPriorLinear.h:45-52): Qc = count x d x d (GPSLAM_ROT3_BIAS: 3 x 3), each SPD.  Factors added through
 * gpslam_hip_add_gp_priors use the handle's shared Qc (set_qc).  A graph with several distinct Qc is linearised with one
 * launch per distinct Qc.  (The interpolated measurement factors do not need their Qc_model at all: in
 * Psi = Q(tau) Phi(dt - tau)^T Q^-1(dt) = (A(tau) Phi2^T A^-1(dt)) (x) (Qc Qc^-1) and Lambda = Phi(tau) - Psi Phi(dt)
 * -- gpslam/gp/GPutils.h:54-71 -- Qc cancels for every SPD Qc, so their constructors' Qc_model argument has no effect
 * on evaluateError; GPInterpolatedRangeFactorPose2.h:46-54.) */
int gpslam_hip_add_gp_priors_qc(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt, const double *Qc);
/* gtsam::PriorFactor<Pose> on x_idx, diagonal sigmas (count x d) */
int gpslam_hip_add_pose_priors(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const double *prior,
                               const double *sigmas);
/* gtsam::PriorFactor<Vector> on v_idx */
int gpslam_hip_add_vel_priors(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const double *prior,
                              const double *sigmas);
/* gtsam::BetweenFactor<Pose>(x_left, x_left+1, measured) (matlab/PlazaPose2.m:125) */
int gpslam_hip_add_between(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *measured,
                           const double *sigmas);
/* gtsam::BetweenFactor<Pose>(x_first, x_second, measured) between ANY two states of the chain -- a loop closure (round 6; ABI 2.1).
 * The reference's factors take arbitrary keys (gpslam/gp/GaussianProcessPriorPose3.h:43-47) and GTSAM eliminates whatever graph
 * they form; here everything but closures couples state i with i + 1, and a closure is applied to the chain solve as a low-rank
 * correction: its d whitened rows ride through the block-tridiagonal solver as d extra right-hand sides behind the landmark
 * columns (Sherman-Morrison-Woodbury; kernels.hpp "loop closures").  Pairs with second == first + 1 are ordinary chain factors
 * (add_between).  Capacity: 1 + landmarks * landmark_dim + closures * d <= 28 right-hand sides (Pose2 / Rot3 / Linear3: 9 closures
 * without landmarks, Pose3: 4); compile() answers GPSLAM_E_UNSUPPORTED beyond that, on fp32 handles, on sharded handles and on
 * the segmented landmark path.  measured: count x pose_dim, sigmas: count x d.  Gauss-Newton, Levenberg-Marquardt, optimize and
 * error include the closures; gpslam_hip_normal_equations reports the chain part H0 (the closures' blocks are not in D / O / g). */
int gpslam_hip_add_between_pairs(gpslam_hip_handle *h, int32_t count, const int32_t *first, const int32_t *second,
                                 const double *measured, const double *sigmas);
/* gtsam::PriorFactor<Point> on landmark idx (matlab/PlazaPose2.m:63) */
int gpslam_hip_add_landmark_priors(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const double *prior,
                                   const double *sigmas);
/* GPInterpolatedRangeFactor{Pose2,Pose3,2DLinear}(z, model, Qc, x_i, v_i, x_i+1, v_i+1, l, dt, tau[, body_P_sensor])
 * gpslam/slam/GPInterpolatedRangeFactorPose2.h:46-54; sensor: body_P_sensor of THIS call's `count` factors or NULL.
 * Every factor keeps its own body_P_sensor as in the reference (GPInterpolatedRangeFactorPose3.h:46-54): calls with
 * different sensor poses may be mixed freely on one handle. */
int gpslam_hip_add_interp_range(gpslam_hip_handle *h, int32_t count, const int32_t *left, const int32_t *landmark,
                                const double *z, const double *sigma, const double *dt, const double *tau,
                                const double *sensor);
/* RangeFactorPose2 / RangeFactor2DLinear (gpslam/slam/RangeFactor2DLinear.h:30-37) */
int gpslam_hip_add_range(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const int32_t *landmark,
                         const double *z, const double *sigma);
/* GPInterpolatedAttitudeFactorRot3(keys, dt, tau, Qc, model, nZ, bRef) -- GPInterpolatedAttitudeFactorRot3.h:44-51 */
int gpslam_hip_add_interp_attitude(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *nZ,
                                   const double *bRef, const double *sigma, const double *dt, const double *tau);
/* gtsam::AHRSFactor(x_left, x_left+1, b_left, PreintegratedAhrsMeasurements, omegaCoriolis) -- GTSAM 4.0
 * gtsam/navigation/AHRSFactor.h (third party; call site matlab/GPAHRSexample.m:128-137).  GPSLAM_ROT3_BIAS handles only.
 * Per factor, the state of the pre-integration after its last integrateMeasurement():
 *   delta_R (9, deltaRij row-major), dR_dbias (9, delRdelBiasOmega), bias_hat (3), delta_tij (1),
 *   cov (9, preintMeasCov: the factor's Gaussian noise model, SPD)
 * omega_coriolis: 3 doubles shared by the call, or NULL (the recipe passes zeros).
 * gpslam_amd/host/gpslam_host.hpp and gpslam_amd/ahrs.py restate PreintegratedAhrsMeasurements::integrateMeasurement. */
int gpslam_hip_add_ahrs(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *delta_R,
                        const double *dR_dbias, const double *bias_hat, const double *delta_tij, const double *cov,
                        const double *omega_coriolis);
/* GPInterpolatedGPSFactorPose3 -- gpslam/slam/GPInterpolatedGPSFactorPose3.h:46-54 */
int gpslam_hip_add_interp_gps(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *measured,
                              const double *sigmas, const double *dt, const double *tau, const double *sensor);
/* GPInterpolatedProjectionFactorPose3<Cal3_S2>(measured, cam_model, Qc_model, x_i, v_i, x_i+1, v_i+1, l, delta_t, tau, K,
 * body_P_sensor) -- gpslam/slam/GPInterpolatedProjectionFactorPose3.h:64-139: reprojection error of 3-D landmark l
 * through a pinhole camera riding on the GP-interpolated pose.  measured: count x 2 pixels, sigmas: count x 2,
 * K = {fx, fy, s, u0, v0} (gtsam::Cal3_S2, kept per factor: calls may differ), sensor = body_P_sensor (12 doubles) or NULL.
 * throwCheirality = false semantics: a landmark behind the camera contributes error 2 fx and zero Jacobians. */
int gpslam_hip_add_interp_projection(gpslam_hip_handle *h, int32_t count, const int32_t *left, const int32_t *landmark,
                                     const double *measured, const double *sigmas, const double *dt, const double *tau,
                                     const double *K, const double *sensor);
/* The same factor with CALIBRATION = gtsam::Cal3DS2 (the reference class is a template over it,
 * GPInterpolatedProjectionFactorPose3.h:29): K9 = {fx, fy, s, u0, v0, k1, k2, p1, p2}, radial (k1, k2) and tangential
 * (p1, p2) distortion of the intrinsic point before the linear calibration (Cal3DS2_Base::uncalibrate).  Zero coefficients
 * give the Cal3_S2 factor bit for bit; both kinds may share a graph. */
int gpslam_hip_add_interp_projection_ds2(gpslam_hip_handle *h, int32_t count, const int32_t *left, const int32_t *landmark,
                                         const double *measured, const double *sigmas, const double *dt, const double *tau,
                                         const double *K9, const double *sensor);
/* OdometryFactor2DLinear(x_left, x_left+1, measured) -- gpslam/slam/OdometryFactor2DLinear.h:38-40 */
int gpslam_hip_add_odometry2d(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *measured,
                              const double *sigmas);
/* RangeBearingFactor2DLinear(x, l, range, bearing) -- gpslam/slam/RangeBearingFactor2DLinear.h:33-37 */
int gpslam_hip_add_bearing_range(gpslam_hip_handle *h, int32_t count, const int32_t *idx, const int32_t *landmark,
                                 const double *bearing, const double *range, const double *sigmas);

/* noiseModel::Gaussian::Covariance on measurement factors.  The reference's constructors take any gtsam::SharedNoiseModel
 * (gpslam/slam/GPInterpolatedGPSFactorPose3.h:46-54, GPInterpolatedRangeFactorPose3.h:46-54, OdometryFactor2DLinear.h:38-40);
 * the add_* calls above take diagonal sigmas.  This call replaces the noise model of the `count` most recently added factors
 * of `kind` (GPSLAM_MEAS_*, not AHRS: that one takes its covariance itself) by full covariances, cov = count x rows x rows
 * (rows = residual dimension of the kind), each SPD: the rows are whitened by R = chol_upper(cov^-1), R^T R = cov^-1, as
 * gtsam::noiseModel::Gaussian does. */
int gpslam_hip_set_meas_covariance(gpslam_hip_handle *h, int32_t kind, int32_t count, const double *cov);

/* drop every factor added so far (states, landmarks and Qc stay); the handle needs a new compile() */
int gpslam_hip_clear_factors(gpslam_hip_handle *h);

/* Graph compile: classify, sort by left state, pack SoA parameter arrays, size the solver hierarchy.
 * Must be called after the last add_* / set_states and before any of the calls below.  Every stored factor index is
 * re-validated against the CURRENT number of states / landmarks (GPSLAM_E_INVALID if set_states / set_landmarks
 * shrank the problem under factors that still refer to the removed variables). */
int gpslam_hip_compile(gpslam_hip_handle *h);

/* ---- the hot path ---- */
/* Batched evaluateError + H1..H4 of every GP prior, in the order they were added (unwhitened, exactly what
 * NoiseModelFactor::unwhitenedError returns): errors count x 2d; jacobians count x 4 x 2d x d row-major
 * (may be NULL).  Returns the factor count. */
int gpslam_hip_linearize_gp(gpslam_hip_handle *h, double *errors, double *jacobians);
/* The same for the measurement factors of one kind (kind = GPSLAM_MEAS_*), in the order they were added: errors
 * count x rows (unwhitened evaluateError), jacobians count x rows x (4d + 3) = per row [H1 (d) | H2 (d) | H3 (d) | H4 (d) |
 * H5 (landmark, zero padded to 3)] (may be NULL); single-state factors leave H3 / H4 zero, factors without a landmark H5
 * (e.g. gpslam/slam/GPInterpolatedRangeFactorPose3.h:64-98, GPInterpolatedAttitudeFactorRot3.h:61-83).  Returns the count. */
enum { GPSLAM_MEAS_INTERP_RANGE = 0, GPSLAM_MEAS_RANGE = 1, GPSLAM_MEAS_INTERP_ATTITUDE = 2, GPSLAM_MEAS_INTERP_GPS = 3,
       GPSLAM_MEAS_ODOMETRY2D = 4, GPSLAM_MEAS_BEARING_RANGE = 5, GPSLAM_MEAS_INTERP_PROJECTION = 6, GPSLAM_MEAS_AHRS = 7 };
int gpslam_hip_linearize_meas(gpslam_hip_handle *h, int32_t kind, double *errors, double *jacobians);
/* total graph error 0.5 * sum |R e|^2 (NonlinearFactorGraph::error) */
int gpslam_hip_error(gpslam_hip_handle *h, double *err);
/* one GaussNewtonOptimizer::iterate(): linearize, assemble, solve, retract, error */
int gpslam_hip_iterate_gn(gpslam_hip_handle *h, gpslam_hip_stats *st);
/* one LevenbergMarquardtOptimizer::iterate() */
int gpslam_hip_iterate_lm(gpslam_hip_handle *h, double *lambda, const gpslam_hip_params *p, gpslam_hip_stats *st);
/* The decision of ONE lambda trial of LevenbergMarquardtOptimizer::iterate() (GTSAM 4.0.x tryLambda, third party, as recalled --
 * parity unpinned; the reference's call sites: matlab/PlazaPose2.m:210-226, matlab/GPAHRSexample.m:259-264), host arithmetic
 * only, no handle.  s6 = {error at the linearisation point, error after the trial step, |delta|_inf, delta . g, |delta|^2,
 * indefinite flag} (lm_trial_phase2's out6 after the reduction over ranks; iterate_lm's own scalars).
 *   rho = (s6[0] - s6[1]) / (0.5 delta.g + 0.5 lambda |delta|^2) > minModelFidelity: *accepted = 1, lambda /= lambdaFactor
 *       (not below lambdaLowerBound), *done = 1;
 *   otherwise, |s6[0] - s6[1]| < relativeErrorTol * s6[0]:  *accepted = 0, *done = 1, lambda untouched ("relative cost
 *       reduction is small": the search for a lambda ends, the values stay);
 *   otherwise lambda *= lambdaFactor, *accepted = 0, *done = (lambda >= lambdaUpperBound).
 * A caller that owns the loop (sharded / split handles) calls lm_reject whenever *accepted == 0 and tries again while
 * *done == 0.  gpslam_hip_iterate_lm takes its branches here as well. */
int gpslam_hip_lm_decide(const double *s6, const gpslam_hip_params *p, double *lambda, int32_t *accepted, int32_t *done);
/* NonlinearOptimizer::optimize() with GTSAM's stop rules (+ optional |delta|_inf rule) */
int gpslam_hip_optimize(gpslam_hip_handle *h, const gpslam_hip_params *p, gpslam_hip_stats *st);

/* ---- inspection (parity tests, profiling) ---- */
/* normal equations of the current linearisation: D (N x b x b), O (N x b x b), g (N x b), b = 2d;
 * with landmarks also B (N x b x nl), nl = L * landmark_dim.  Any pointer may be NULL. */
int gpslam_hip_normal_equations(gpslam_hip_handle *h, double *D, double *O, double *g, double *B);
/* whitened Jacobian rows of the current linearisation (what NoiseModelFactor::linearize produces after
 * WhitenSystem): rowLR M x 4d = [d/dx_left | d/dx_right], rowE M, rowM M x landmark_dim, rowLm M (landmark id or -1).
 * Row order: first the rows with velocity columns, grouped by left state (inside a state: GP priors, velocity priors,
 * then interpolated range, range, attitude, GPS, odometry2d, bearing-range, projection, each in the order they were
 * added); then the velocity-free rows (pose priors, between factors; the device keeps them in a half-width table),
 * again grouped by left state, expanded to full width.  Any output pointer may be NULL; n_rows receives M. */
int gpslam_hip_get_rows(gpslam_hip_handle *h, int32_t *n_rows, double *rowLR, double *rowE, double *rowM,
                        int32_t *rowLm);
/* solve the block-tridiagonal SPD system D/O/g (host arrays as above, N x b) with the device solver; x: N x b */
int gpslam_hip_block_tridiag_solve(gpslam_hip_handle *h, int32_t N, const double *D, const double *O,
                                   const double *g, double *x);
/* Batched GaussianProcessInterpolator{Linear<D>,Pose2,Pose3,Rot3}::interpolatePose (gpslam.h:57-86; e.g.
 * gpslam/gp/GaussianProcessInterpolatorPose3.h:57-105 without the Jacobians): the pose of the current estimate at
 * time tau[q] after state left[q], inside the interval (left[q], left[q] + 1) of duration dt[q], with the handle's
 * Qc.  out_pose: count x pose_dim, rows in the layout of gpslam_hip_get_states.  Dense trajectory output after
 * an optimisation is the use the MATLAB wrapper exports the interpolators for. */
int gpslam_hip_interpolate_poses(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                 const double *tau, double *out_pose);
/* ... with the Jacobians of the interpolated pose with respect to (pose_left, vel_left, pose_right, vel_right):
 * out_H count x 4 x d x d = H1..H4 of interpolatePose (gpslam/gp/GaussianProcessInterpolatorPose3.h:82-98, gpslam.h:57-86) */
int gpslam_hip_interpolate_poses_jac(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                     const double *tau, double *out_pose, double *out_H);
/* GaussianProcessInterpolatorLinear<D>::interpolateVelocity (gpslam.h:193, gpslam/gp/GaussianProcessInterpolatorLinear.h:106-126)
 * of the current estimate, batched like interpolate_poses: out_vel count x d; out_H (may be NULL) count x 4 x d x d = H1..H4
 * (:117-120, the lower blocks of Lambda and Psi).  GPSLAM_LINEAR2 / GPSLAM_LINEAR3 handles; the reference declares the method
 * for its Lie-group interpolators without implementing it (GaussianProcessInterpolatorPose3.h:118-123): GPSLAM_E_UNSUPPORTED. */
int gpslam_hip_interpolate_velocities(gpslam_hip_handle *h, int32_t count, const int32_t *left, const double *dt,
                                      const double *tau, double *out_vel, double *out_H);
/* getBodyCentricVb / getBodyCentricVs (gpslam.h:161-164, gpslam/gp/Pose3utils.cpp:17-24), batched: the MATLAB scripts
 * initialise the velocity of every state with them.  which = 0: Vb = Logmap(pose1^-1 pose2) / dt, 1: Vs = Logmap(pose2
 * pose1^-1) / dt;  pose1, pose2: count x 12 (R row-major, t), dt: count, out: count x 6 (omega, v).  Any handle: its device
 * and stream are used, its graph is not touched. */
int gpslam_hip_body_centric_velocity(gpslam_hip_handle *h, int32_t which, int32_t count, const double *pose1,
                                     const double *pose2, const double *dt, double *out);
/* What compile() chose for the chain solver (introspection for tests and tuning; no reference counterpart):
 * out8 = {levels of the hierarchy, level-0 chunk length, upper chunk length, assembly fused into the level-0 elimination
 * (k_fused_level0) 0/1, GP priors handed to the assembly as structured records instead of Jacobian rows 0/1 (SE(3): inside the fused
 * kernel; SE(2) / SO(3) / 3-D linear: where k_assemble_ghost runs; 2: SE(3) records AND the interpolated GPS factors' rows as
 * 16-double lines whose right halves the assembly wave forms from the interval's record -- round 5), rows in the
 * full-width table, rows in the compact table, right-hand-side columns R}. */
int gpslam_hip_plan_info(gpslam_hip_handle *h, int32_t out8[8]);
/* the plan of the segmented landmark elimination chosen by compile(): out = {active (0 / 1), segment length C, fat
 * blocks K, fat block size NB, border columns NC per segment, NC rounded up to MFMA tiles, cyclic-reduction levels,
 * link blocks} */
int gpslam_hip_segment_plan(gpslam_hip_handle *h, int32_t out8[8]);
/* time (ms) of the last iterate call's phases measured with hipEvents on the handle's stream:
 * out[0] linearize, out[1] assemble, out[2] solve, out[3] retract+error, out[4] total */
int gpslam_hip_last_timing(gpslam_hip_handle *h, double *out5);
/* device time (ms) of the LEVEL-0 FORWARD launch -- on Pose3 chains k_fused_level0, the dominant kernel -- summed over the
 * iterations of the last timed run_gn / iterate_gn: the kernel as it runs INSIDE an iteration (behind the linearisation,
 * caches as an iteration leaves them), which is what bench.py's roofline fraction is computed from (time_kernel's isolated
 * launches are faster).  The fused launch carries its own start / stop events (the dispatch's time stamps, what a profiler's
 * kernel trace reports); the two-launch level 0 is bracketed by events on the stream.  0 on the segmented landmark path. */
int gpslam_hip_last_level0_ms(gpslam_hip_handle *h, double *ms);
/* on = 1: the fused level-0 launch of a TIMED iteration carries its own start / stop events (hipExtLaunchKernelGGL: the dispatch's
 * time stamps -- gpslam_hip_last_level0_ms then reports what a profiler's kernel trace reports, where events recorded around the
 * launch add their marker packets: 149 us for 142).  A stamped dispatch costs the iteration ~10 us elsewhere, so the other phase
 * times of such a run are not to be quoted; off (the default) nothing changes.  ABI 2.2. */
int gpslam_hip_set_level0_stamps(gpslam_hip_handle *h, int32_t on);
/* run `iters` Gauss-Newton iterations back to back with no host synchronisation in between (the benchmark
 * loop); per-phase device time is accumulated in out5 (ms, summed over iters) when out5 != NULL.
 * The error of the state an iteration produces is the error the next iteration's linearisation evaluates, so
 * inside the run it is computed once (by that linearisation) and only the last iteration is followed by an
 * error-only pass; st->error_before / error_after refer to the last iteration. */
int gpslam_hip_run_gn(gpslam_hip_handle *h, int32_t iters, gpslam_hip_stats *st, double *out5);

/* average device time (ms, hipEvents on the handle's stream) of ONE launch of a hot kernel over `reps` launches:
 * which = 0 GP-prior linearisation (K1), 1 normal-equation assembly (K3), 2 level-0 forward elimination (K4),
 * 3 level-0 back-substitution, 4 retract (K6).  Inputs are re-created (untimed) before every timed launch.
 * Pose3 chains without landmarks run the assembly INSIDE the level-0 elimination (k_fused_level0): there
 * which = 1 reports 0.0 (no such launch in an iteration) and which = 2 times the fused kernel. */
int gpslam_hip_time_kernel(gpslam_hip_handle *h, int32_t which, int32_t reps, double *avg_ms);

/* ---- segment sharding (nranks > 1): one exchange per iteration, performed by the host via RCCL ---- */
/* device pointer + size (bytes) of this rank's interface record (written by phase 1) */
int gpslam_hip_interface_send(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes);
/* device pointer + size of the gathered records of all ranks (nranks * bytes; filled by the host's all-gather) */
int gpslam_hip_interface_recv(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes);
/* phase 1: linearize + assemble + local elimination down to the rank separator -> interface record */
int gpslam_hip_iterate_phase1(gpslam_hip_handle *h, double lambda);
/* phase 2 (after the all-gather): reduced solve, back-substitution, retract, local error (st may be NULL: no error
 * pass, no host synchronisation).  Chains with landmarks on more than one rank run it in two halves with one
 * all-reduce (sum) of the landmark buffer in between -- SURVEY.md section 8(e) collective (3):
 *   phase2a: reduced solve + back-substitution + this rank's share of the landmark Schur complement [S | gL]
 *   all-reduce of gpslam_hip_landmark_reduce_buffer (doubles)
 *   phase2b: landmark solve (redundant on every rank), chain correction, retract of states / halo / landmarks */
int gpslam_hip_iterate_phase2(gpslam_hip_handle *h, gpslam_hip_stats *st);
int gpslam_hip_iterate_phase2a(gpslam_hip_handle *h);
int gpslam_hip_iterate_phase2b(gpslam_hip_handle *h, gpslam_hip_stats *st);
int gpslam_hip_landmark_reduce_buffer(gpslam_hip_handle *h, void **dev_ptr, size_t *bytes);
/* LevenbergMarquardtOptimizer::iterate on a sharded chain: the accept / reject decisions need global sums, so the
 * loop is the caller's (gpslam_amd/sharded.py: ShardedSolver.iterate_lm mirrors gpslam_hip_iterate_lm line by line):
 *   lm_begin;  repeat { lm_trial_phase1(lambda); all-gather records; iterate_phase2a; [all-reduce landmark buffer];
 *   lm_trial_phase2(out6); all-reduce out6 (sum of [0], [1], [3], [4]; max of [2], [5]); accept, or lm_reject and a
 *   larger lambda }.  out6 = {error, trial error, |delta|_inf, delta . g, |delta|^2, indefinite flag} of this rank. */
int gpslam_hip_lm_begin(gpslam_hip_handle *h);
int gpslam_hip_lm_trial_phase1(gpslam_hip_handle *h, double lambda);
int gpslam_hip_lm_trial_phase2(gpslam_hip_handle *h, double *out6);
int gpslam_hip_lm_reject(gpslam_hip_handle *h);
/* ---- chains with many locally visible landmarks (the segmented landmark elimination, BASELINE config 4) across GPUs ----
 * The chain is split into P pieces that OVERLAP in one state: piece r holds the states [s_r, s_(r+1)] (both ends included),
 * every factor whose left / only state lies in [s_r, s_(r+1)) (the last piece also those of the final state), and the
 * landmarks its factors touch, in its own numbering.  A landmark seen from both sides of s_(r+1) is listed -- in the same
 * order -- as a `last` landmark of piece r and a `first` landmark of piece r + 1 (its prior, if any, is given to ONE of the
 * two); the shared state and these landmarks form a fat separator both pieces hold, and both apply the same update to it.
 * Handles are created with nranks = 1 (no halo); before compile():  fs_set_split;  after compile(): fs_split_info ->
 * out4 = {this piece's fat block size NB, fat blocks, segment length, nb_top}, then fs_set_top(max of NB over the pieces).
 * One Gauss-Newton iteration:  fs_phase1(lambda)  -> interface record [Dff | H | Dll | g_first | g_last] in `send`
 * (3 nb_top^2 + 2 nb_top doubles);  ONE all-gather of the records into `recv` (piece order);  fs_phase2: every piece solves
 * the (P + 1)-block system of the shared separators redundantly, back-substitutes, retracts its states and landmarks
 * (st: THIS piece's error terms and |delta|_inf; NULL skips the error pass).  GPInterpolatedRangeFactorPose2.h:64-98 over
 * matlab/PlazaPose2.m:147-178's graph, at BASELINE config 4's size on more than one GPU. */
int gpslam_hip_fs_set_split(gpslam_hip_handle *h, int32_t rank, int32_t nranks, const int32_t *first_lm, int32_t n_first,
                            const int32_t *last_lm, int32_t n_last);
int gpslam_hip_fs_split_info(gpslam_hip_handle *h, int32_t out4[4]);
int gpslam_hip_fs_set_top(gpslam_hip_handle *h, int32_t nb_top);
int gpslam_hip_fs_interface(gpslam_hip_handle *h, void **send, size_t *send_bytes, void **recv, size_t *recv_bytes);
int gpslam_hip_fs_phase1(gpslam_hip_handle *h, double lambda);
int gpslam_hip_fs_phase2(gpslam_hip_handle *h, gpslam_hip_stats *st);
/* LevenbergMarquardtOptimizer::iterate on a split chain (matlab/PlazaPose2.m:217-229 optimises this graph with it): the
 * caller's loop, as for nranks > 1 below --  gpslam_hip_lm_begin;  repeat { fs_lm_trial_phase1(lambda); all-gather of the
 * records; fs_lm_trial_phase2(out6); all-reduce of out6 (sum of [0], [1], [3], [4]; max of [2], [5]); accept, or
 * gpslam_hip_lm_reject and a larger lambda }.  out6 as for gpslam_hip_lm_trial_phase2; a shared state / landmark enters
 * |delta|^2 on the piece to its right only, delta . g adds up over the pieces as it is. */
int gpslam_hip_fs_lm_trial_phase1(gpslam_hip_handle *h, double lambda);
int gpslam_hip_fs_lm_trial_phase2(gpslam_hip_handle *h, double *out6);
/* ---- the whole optimiser loop on a sharded handle or a split piece (round 5) ----
 * The library does not link a communication library: the ONE data-path collective of an iteration is the
 * host's to perform.  With the two callbacks below registered, gpslam_hip_iterate_gn / _run_gn / _iterate_lm / _optimize /
 * _error accept handles with nranks > 1 and pieces of a split chain: they run the phases above in order and call
 *   all_gather(user, send_dev, recv_dev, bytes_per_rank, hip_stream)  -- every rank's `bytes_per_rank` bytes at send_dev into
 *       recv_dev, rank order, enqueued on hip_stream (the handle's stream): ncclAllGather(send, recv, bytes, ncclChar, comm, stream)
 *   all_reduce_sum(user, buf_dev, n_doubles, hip_stream)              -- in-place sum of doubles (the landmark Schur complement
 *       of a chain with a dense landmark border; may be NULL for chains without landmarks and for split pieces)
 * and return the statistics of the WHOLE chain (errors summed, |delta|_inf maximised over the ranks), identical on every rank.
 * A callback returns 0, anything else ends the call with GPSLAM_E_COMM.  Collectives per call:
 *   iterate_gn      1 all-gather of the interface records (+ 1 all-reduce with a landmark border) + 1 all-gather of 64 B of scalars
 *                   (with or without statistics: it is also how a rank learns that another one failed);
 *   run_gn(K)       K x (the record all-gather (+ the all-reduce)) + ONE all-gather of 64 B at the end: the statistics of the last
 *                   iteration, the non-positive-pivot flag of ANY iteration (sticky through the run) and the first failure of any rank;
 *   iterate_lm      per lambda trial: the record all-gather (+ the all-reduce) and ONE all-gather of 64 B carrying the trial's six
 *                   scalars {error, trial error, |delta|_inf, delta . g, |delta|^2, indefinite flag} -- the scalars of a trial
 *                   exist only after its back-substitution, which needs the gathered records, so they cannot ride the record;
 *                   every rank then takes the branch of gpslam_hip_lm_decide on identical reduced numbers;
 *   optimize        GTSAM's loop (NonlinearOptimizer::defaultOptimize) over the two above.
 * Every rank must make the same calls in the same order (they are collective).
 * FAILURES ON ONE RANK (round 6).  A rank whose local work fails inside one of these calls -- a HIP error, a phase's status, its own
 * all_reduce_sum callback -- does NOT leave the collective sequence: it records the code, skips its remaining local work, goes on making
 * every collective call of the iteration, and the code travels in a spare slot of the 64-byte scalar gather; EVERY rank then returns
 * that code (a non-positive pivot anywhere: GPSLAM_E_NOT_SPD everywhere, as before).  What can still leave ranks out of step, and is
 * the caller's to prevent or to time out: a rank that never enters the call (GPSLAM_E_NOT_COMPILED, no callbacks registered, a
 * split piece without gpslam_hip_fs_set_top: checked before the first collective), an all_gather callback that fails or hangs on
 * one rank only, and a HIP failure inside the scalar gather's own two 64-byte copies.
 * A single-rank handle that was forced onto the sharded code path (config_v2.force_sharded) needs no callbacks.  LevenbergMarquardtOptimizer::iterate on the reference's
 * landmark graph: matlab/PlazaPose2.m:210-226. */
typedef int (*gpslam_hip_all_gather_fn)(void *user, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *hip_stream);
typedef int (*gpslam_hip_all_reduce_sum_fn)(void *user, void *buf_dev, size_t n_doubles, void *hip_stream);
int gpslam_hip_set_collectives(gpslam_hip_handle *h, gpslam_hip_all_gather_fn all_gather, gpslam_hip_all_reduce_sum_fn all_reduce_sum,
                               void *user);
/* halo: the first state of the right neighbour (pose_dim + d doubles), kept in sync by the library after init */
int gpslam_hip_set_halo_state(gpslam_hip_handle *h, const double *pose, const double *vel);

#ifdef __cplusplus
}
#endif
#endif /* GPSLAM_HIP_H */
