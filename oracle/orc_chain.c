/*
 * orc_chain.c -- CPU oracle for the Gauss-Newton / Levenberg-Marquardt loop over a GP
 * trajectory chain: linearise every factor, whiten, assemble the block-tridiagonal (+ landmark
 * border) normal equations, solve by sequential block Cholesky in the explicit chain ordering
 * [x0,v0,x1,v1,...,l0,l1,...], retract, re-evaluate the error.
 *
 * TEST INFRASTRUCTURE ONLY (see gpslam_oracle.h).
 *
 * In the reference this loop is GTSAM (external, not under /root/reference):
 *   NoiseModelFactor::linearize -> noiseModel::Gaussian::WhitenSystem -> GaussianFactorGraph::optimize
 *   -> Values::retract, driven by GaussNewtonOptimizer / LevenbergMarquardtOptimizer; call sites
 *   gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:185-188, matlab/PlazaPose2.m:208-228.
 * Semantics restated from SURVEY.md Appendix A.  PARITY UNPINNED for whitening/error scaling,
 * iteration counts, LM schedule and stop rules (nothing in the reference's tests fixes them);
 * the fixed points of the reference's 2-state optimisation tests are pinned (tests/golden).
 */
#include "gpslam_oracle.h"
#include "orc_math.h"

#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum {
  F_GP = 0, F_POSE_PRIOR, F_VEL_PRIOR, F_BETWEEN, F_LM_PRIOR, F_INTERP_RANGE, F_RANGE, F_INTERP_ATT,
  F_INTERP_GPS, F_ODOM2D, F_BEARING_RANGE, F_INTERP_PROJ, F_AHRS,
  F_CLOSURE   /* gtsam::BetweenFactor<Pose> between ANY two states (idx, idx2): a loop closure */
};

typedef struct {
  int type;
  int idx;          /* state index (left state for binary-in-time factors) */
  int idx2;         /* F_CLOSURE: the second state (any state but idx) */
  int lm;           /* landmark index or -1 */
  double meas[12];  /* measurement / prior value */
  double sig[6];    /* diagonal sigmas */
  double dt, tau;
  double aux[9];    /* attitude: nZ(3), bRef(3); projection: fx, fy, s, u0, v0, k1, k2, p1, p2 */
  int has_sensor;
  double sensor[12];
  double ahrs[34];  /* F_AHRS: the 25 parameters of orc_ahrs_factor + square-root information R (3 x 3 upper triangular) */
  int has_qc;       /* F_GP: this factor's own Qc_model (GaussianProcessPriorPose3.h:43-49 takes one per factor) */
  double qc[36];
  int has_cov;      /* measurement factors: noiseModel::Gaussian::Covariance instead of diagonal sigmas */
  double sqi[9];    /* ... its square-root information R, R^T R = cov^-1 (rows x rows, upper triangular) */
} orc_factor;

struct orc_chain {
  int kind, chart, d, pd, b, ld;
  int vw;      /* Pose3 only: state velocity = world-frame [v; w] (GaussianProcessPriorPose3VW family) */
  int N, L;
  double *pose, *vel, *lmk;
  double Qc[36];
  int nf, capf;
  orc_factor *f;
};

#define MAXR 12 /* max residual rows */
#define MAXB 12 /* max state block */

void orc_default_params(orc_params *p) {
  p->max_iterations = 100;
  p->relative_error_tol = 1e-5;
  p->absolute_error_tol = 1e-5;
  p->error_tol = 0.0;
  p->delta_tol = 0.0;
  p->lambda_initial = 1e-5;
  p->lambda_factor = 10.0;
  p->lambda_upper_bound = 1e5;
  p->lambda_lower_bound = 0.0;
  p->min_model_fidelity = 1e-3;
  p->use_lm = 0;
  p->pad = 0;
}

orc_chain *orc_chain_create(int kind, int chart, int landmark_dim) {
  if (orc_tangent_dim(kind) < 0) return NULL;
  orc_chain *c = (orc_chain *)calloc(1, sizeof(orc_chain));
  c->kind = kind;
  c->chart = chart;
  c->d = orc_tangent_dim(kind);
  c->pd = orc_pose_dim(kind);
  c->b = 2 * c->d;
  c->ld = landmark_dim;
  orc_eye(c->d, c->Qc);
  return c;
}

void orc_chain_destroy(orc_chain *c) {
  if (!c) return;
  free(c->pose); free(c->vel); free(c->lmk); free(c->f); free(c);
}

int orc_chain_set_qc(orc_chain *c, const double *Qc) {
  if (c->kind == ORC_ROT3_BIAS) {   /* 3 x 3 Qc of GaussianProcessPriorRot3; bias / pad components: identity (see gp_eval) */
    orc_eye(6, c->Qc);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) c->Qc[i * 6 + j] = Qc[i * 3 + j];
    return 0;
  }
  orc_copy(c->d * c->d, Qc, c->Qc);
  return 0;
}
int orc_chain_set_velocity_world(orc_chain *c, int on) {
  if (on && c->kind != ORC_POSE3) return -2;
  c->vw = on ? 1 : 0;
  return 0;
}

int orc_chain_set_states(orc_chain *c, int N, const double *pose, const double *vel) {
  free(c->pose); free(c->vel);
  c->N = N;
  c->pose = (double *)malloc(sizeof(double) * (size_t)N * c->pd);
  c->vel = (double *)malloc(sizeof(double) * (size_t)N * c->d);
  orc_copy(N * c->pd, pose, c->pose);
  orc_copy(N * c->d, vel, c->vel);
  return 0;
}
int orc_chain_get_states(const orc_chain *c, double *pose, double *vel) {
  if (pose) orc_copy(c->N * c->pd, c->pose, pose);
  if (vel) orc_copy(c->N * c->d, c->vel, vel);
  return 0;
}
int orc_chain_set_landmarks(orc_chain *c, int L, const double *pts) {
  free(c->lmk);
  c->L = L;
  c->lmk = (double *)malloc(sizeof(double) * (size_t)(L > 0 ? L : 1) * (c->ld > 0 ? c->ld : 1));
  if (L > 0) orc_copy(L * c->ld, pts, c->lmk);
  return 0;
}
int orc_chain_get_landmarks(const orc_chain *c, double *pts) {
  if (c->L > 0) orc_copy(c->L * c->ld, c->lmk, pts);
  return 0;
}

static orc_factor *new_factor(orc_chain *c, int type) {
  if (c->nf == c->capf) {
    c->capf = c->capf ? 2 * c->capf : 1024;
    c->f = (orc_factor *)realloc(c->f, sizeof(orc_factor) * (size_t)c->capf);
  }
  orc_factor *f = &c->f[c->nf++];
  memset(f, 0, sizeof(*f));
  f->type = type;
  f->lm = -1;
  return f;
}

int orc_chain_add_gp_priors(orc_chain *c, int count, const int32_t *left, const double *dt) {
  for (int k = 0; k < count; k++) { orc_factor *f = new_factor(c, F_GP); f->idx = left[k]; f->dt = dt[k]; }
  return 0;
}
/* GaussianProcessPrior*(key1 .. key4, delta_t, Qc_model): one Qc per factor, as the reference's constructors take it
 * (gpslam/gp/GaussianProcessPriorPose3.h:43-49).  Qc: count x d x d (ROT3_BIAS: 3 x 3, padded as in orc_chain_set_qc). */
int orc_chain_add_gp_priors_qc(orc_chain *c, int count, const int32_t *left, const double *dt, const double *Qc) {
  const int dq = (c->kind == ORC_ROT3_BIAS) ? 3 : c->d;
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_GP);
    f->idx = left[k];
    f->dt = dt[k];
    f->has_qc = 1;
    const double *q = Qc + (size_t)k * dq * dq;
    if (c->kind == ORC_ROT3_BIAS) {
      orc_eye(6, f->qc);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) f->qc[i * 6 + j] = q[i * 3 + j];
    } else {
      orc_copy(c->d * c->d, q, f->qc);
    }
  }
  return 0;
}
/* noiseModel::Gaussian::Covariance on the `count` most recently added factors of one measurement type (the reference's
 * constructors take any gtsam::SharedNoiseModel, e.g. GPInterpolatedGPSFactorPose3.h:46-54): whitening by
 * R = chol_upper(cov^-1), as gtsam::noiseModel::Gaussian does.  cov: count x rows x rows. */
int orc_chain_set_meas_covariance(orc_chain *c, int type, int count, int rows, const double *cov) {
  if (rows < 1 || rows > 3 || type < F_INTERP_RANGE || type == F_AHRS) return -2;
  int k = count;
  for (int i = c->nf - 1; i >= 0 && k > 0; i--) {
    orc_factor *f = &c->f[i];
    if (f->type != type) continue;
    k--;
    double inv[9];
    if (orc_inv(rows, cov + (size_t)k * rows * rows, inv)) return -1;
    orc_copy(rows * rows, inv, f->sqi);
    if (orc_chol_upper(rows, f->sqi)) return -1;
    f->has_cov = 1;
  }
  return k == 0 ? 0 : -2;
}
int orc_chain_add_pose_priors(orc_chain *c, int count, const int32_t *idx, const double *prior, const double *sigmas) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_POSE_PRIOR);
    f->idx = idx[k];
    orc_copy(c->pd, prior + (size_t)k * c->pd, f->meas);
    orc_copy(c->d, sigmas + (size_t)k * c->d, f->sig);
  }
  return 0;
}
int orc_chain_add_vel_priors(orc_chain *c, int count, const int32_t *idx, const double *prior, const double *sigmas) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_VEL_PRIOR);
    f->idx = idx[k];
    orc_copy(c->d, prior + (size_t)k * c->d, f->meas);
    orc_copy(c->d, sigmas + (size_t)k * c->d, f->sig);
  }
  return 0;
}
int orc_chain_add_between(orc_chain *c, int count, const int32_t *left, const double *measured, const double *sigmas) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_BETWEEN);
    f->idx = left[k];
    orc_copy(c->pd, measured + (size_t)k * c->pd, f->meas);
    orc_copy(c->d, sigmas + (size_t)k * c->d, f->sig);
  }
  return 0;
}
/* gtsam::BetweenFactor<Pose>(x_first, x_second, measured, diagonal model) between any two states -- what a loop closure is in a
 * GTSAM graph (third party; the reference's factors take arbitrary keys the same way, gpslam/gp/GaussianProcessPriorPose3.h:43-47).
 * Such a factor breaks the block-tridiagonal pattern: chains that hold one are solved by the envelope Cholesky below. */
int orc_chain_add_between_pairs(orc_chain *c, int count, const int32_t *first, const int32_t *second, const double *measured,
                                const double *sigmas) {
  for (int k = 0; k < count; k++) {
    if (first[k] == second[k] || first[k] < 0 || second[k] < 0) return -2;
    orc_factor *f = new_factor(c, F_CLOSURE);
    f->idx = first[k];
    f->idx2 = second[k];
    orc_copy(c->pd, measured + (size_t)k * c->pd, f->meas);
    orc_copy(c->d, sigmas + (size_t)k * c->d, f->sig);
  }
  return 0;
}
int orc_chain_add_landmark_priors(orc_chain *c, int count, const int32_t *idx, const double *prior,
                                  const double *sigmas) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_LM_PRIOR);
    f->lm = idx[k];
    orc_copy(c->ld, prior + (size_t)k * c->ld, f->meas);
    orc_copy(c->ld, sigmas + (size_t)k * c->ld, f->sig);
  }
  return 0;
}
int orc_chain_add_interp_range(orc_chain *c, int count, const int32_t *left, const int32_t *landmark, const double *z,
                               const double *sigma, const double *dt, const double *tau, const double *sensor) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_INTERP_RANGE);
    f->idx = left[k]; f->lm = landmark[k]; f->meas[0] = z[k]; f->sig[0] = sigma[k]; f->dt = dt[k]; f->tau = tau[k];
    if (sensor) { f->has_sensor = 1; orc_copy(c->pd, sensor, f->sensor); }
  }
  return 0;
}
int orc_chain_add_range(orc_chain *c, int count, const int32_t *idx, const int32_t *landmark, const double *z,
                        const double *sigma) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_RANGE);
    f->idx = idx[k]; f->lm = landmark[k]; f->meas[0] = z[k]; f->sig[0] = sigma[k];
  }
  return 0;
}
int orc_chain_add_interp_attitude(orc_chain *c, int count, const int32_t *left, const double *nZ, const double *bRef,
                                  const double *sigma, const double *dt, const double *tau) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_INTERP_ATT);
    f->idx = left[k];
    /* gtsam::Unit3 normalises its argument on construction (GPInterpolatedAttitudeFactorRot3.h:47-48) */
    for (int part = 0; part < 2; part++) {
      const double *v = (part == 0 ? nZ : bRef) + 3 * (size_t)k;
      double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (int q = 0; q < 3; q++) f->aux[3 * part + q] = v[q] / n;
    }
    f->sig[0] = sigma[2 * (size_t)k]; f->sig[1] = sigma[2 * (size_t)k + 1];
    f->dt = dt[k]; f->tau = tau[k];
  }
  return 0;
}
int orc_chain_add_interp_gps(orc_chain *c, int count, const int32_t *left, const double *measured,
                             const double *sigmas, const double *dt, const double *tau, const double *sensor) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_INTERP_GPS);
    f->idx = left[k];
    orc_copy(3, measured + 3 * (size_t)k, f->meas);
    orc_copy(3, sigmas + 3 * (size_t)k, f->sig);
    f->dt = dt[k]; f->tau = tau[k];
    if (sensor) { f->has_sensor = 1; orc_copy(c->pd, sensor, f->sensor); }
  }
  return 0;
}
int orc_chain_add_odometry2d(orc_chain *c, int count, const int32_t *left, const double *measured,
                             const double *sigmas) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_ODOM2D);
    f->idx = left[k];
    orc_copy(3, measured + 3 * (size_t)k, f->meas);
    orc_copy(3, sigmas + 3 * (size_t)k, f->sig);
  }
  return 0;
}
int orc_chain_add_bearing_range(orc_chain *c, int count, const int32_t *idx, const int32_t *landmark,
                                const double *bearing, const double *range, const double *sigmas) {
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_BEARING_RANGE);
    f->idx = idx[k]; f->lm = landmark[k]; f->meas[0] = bearing[k]; f->meas[1] = range[k];
    f->sig[0] = sigmas[2 * (size_t)k]; f->sig[1] = sigmas[2 * (size_t)k + 1];
  }
  return 0;
}
int orc_chain_add_interp_projection_ds2(orc_chain *c, int count, const int32_t *left, const int32_t *landmark,
                                        const double *measured, const double *sigmas, const double *dt, const double *tau,
                                        const double *K9, const double *sensor);
int orc_chain_add_interp_projection(orc_chain *c, int count, const int32_t *left, const int32_t *landmark,
                                    const double *measured, const double *sigmas, const double *dt, const double *tau,
                                    const double *K, const double *sensor) {
  const double K9[9] = {K[0], K[1], K[2], K[3], K[4], 0.0, 0.0, 0.0, 0.0};
  return orc_chain_add_interp_projection_ds2(c, count, left, landmark, measured, sigmas, dt, tau, K9, sensor);
}
int orc_chain_add_interp_projection_ds2(orc_chain *c, int count, const int32_t *left, const int32_t *landmark,
                                        const double *measured, const double *sigmas, const double *dt, const double *tau,
                                        const double *K, const double *sensor) {
  if (c->kind != ORC_POSE3 || c->ld != 3 || c->vw) return -2;
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_INTERP_PROJ);
    f->idx = left[k]; f->lm = landmark[k];
    f->meas[0] = measured[2 * (size_t)k]; f->meas[1] = measured[2 * (size_t)k + 1];
    f->sig[0] = sigmas[2 * (size_t)k]; f->sig[1] = sigmas[2 * (size_t)k + 1];
    f->dt = dt[k]; f->tau = tau[k];
    orc_copy(9, K, f->aux);                       /* fx, fy, s, u0, v0, k1, k2, p1, p2 */
    if (sensor) { f->has_sensor = 1; orc_copy(12, sensor, f->sensor); }
  }
  return 0;
}

int orc_chain_add_ahrs(orc_chain *c, int count, const int32_t *left, const double *delta_R, const double *dR_dbias,
                       const double *bias_hat, const double *delta_tij, const double *cov, const double *omega_coriolis) {
  if (c->kind != ORC_ROT3_BIAS) return -2;
  for (int k = 0; k < count; k++) {
    orc_factor *f = new_factor(c, F_AHRS);
    f->idx = left[k];
    orc_copy(9, delta_R + 9 * (size_t)k, f->ahrs);
    orc_copy(9, dR_dbias + 9 * (size_t)k, f->ahrs + 9);
    orc_copy(3, bias_hat + 3 * (size_t)k, f->ahrs + 18);
    f->ahrs[21] = delta_tij[k];
    if (omega_coriolis) orc_copy(3, omega_coriolis, f->ahrs + 22);
    /* noiseModel::Gaussian::Covariance(preintMeasCov): R with R^T R = cov^-1 */
    double inv[9];
    if (orc_inv(3, cov + 9 * (size_t)k, inv)) return -1;
    orc_copy(9, inv, f->ahrs + 25);
    if (orc_chol_upper(3, f->ahrs + 25)) return -1;
    f->sig[0] = f->sig[1] = f->sig[2] = 1.0;
  }
  return 0;
}

/* ------------------------------------------------------------------ factor evaluation */

static void gp_eval(const orc_chain *c, const orc_factor *f, double *e, double *H1, double *H2, double *H3, double *H4) {
  const double *p1 = c->pose + (size_t)f->idx * c->pd, *p2 = p1 + c->pd;
  const double *v1 = c->vel + (size_t)f->idx * c->d, *v2 = v1 + c->d;
  switch (c->kind) {
    case ORC_LINEAR2: orc_gp_prior_linear(2, p1, v1, p2, v2, f->dt, e, H1, H2, H3, H4); break;
    case ORC_LINEAR3: orc_gp_prior_linear(3, p1, v1, p2, v2, f->dt, e, H1, H2, H3, H4); break;
    case ORC_POSE2: orc_gp_prior_pose2(p1, v1, p2, v2, f->dt, e, H1, H2, H3, H4); break;
    case ORC_POSE3:
      if (c->vw) orc_gp_prior_pose3vw_packed(p1, v1, p2, v2, f->dt, e, H1, H2, H3, H4);
      else orc_gp_prior_pose3(p1, v1, p2, v2, f->dt, e, H1, H2, H3, H4);
      break;
    case ORC_ROT3: orc_gp_prior_rot3(p1, v1, p2, v2, f->dt, e, H1, H2, H3, H4); break;
    case ORC_ROT3_BIAS: {
      /* GaussianProcessPriorRot3 on (x, v) (GPAHRSexample.m:149-152); the bias has no GP prior; the pads are pinned by
       * unit rows: e = [r - v1 dt; pad1; v2 - v1; pad2], rows / columns in the 6-wide (theta, bias | omega, pad) layout */
      double e3[6], A1[18], A2[18], A3[18], A4[18];
      orc_gp_prior_rot3(p1, v1, p2, v2, f->dt, e3, H1 ? A1 : NULL, H2 ? A2 : NULL, H3 ? A3 : NULL, H4 ? A4 : NULL);
      for (int i = 0; i < 3; i++) { e[i] = e3[i]; e[3 + i] = v1[3 + i]; e[6 + i] = e3[3 + i]; e[9 + i] = v2[3 + i]; }
      double *H[4] = {H1, H2, H3, H4};
      const double *A[4] = {A1, A2, A3, A4};
      for (int m = 0; m < 4; m++) {
        if (!H[m]) continue;
        orc_zero(72, H[m]);
        for (int half = 0; half < 2; half++)
          for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) H[m][(6 * half + r) * 6 + q] = A[m][(3 * half + r) * 3 + q];
      }
      if (H2) for (int q = 0; q < 3; q++) H2[(3 + q) * 6 + 3 + q] = 1.0;     /* d pad1 / d pad1 */
      if (H4) for (int q = 0; q < 3; q++) H4[(9 + q) * 6 + 3 + q] = 1.0;     /* d pad2 / d pad2 */
    } break;
  }
}

/* Evaluate one factor: rows, WHITENED error we[rows], whitened Jacobians w.r.t. the left state block
 * JL (rows x b, [pose | vel]), the right state block JR (rows x b) and the landmark Jm (rows x ld).
 * Whitening per GTSAM NoiseModelFactor::linearize / Gaussian::WhitenSystem: A <- R A, e <- R e.
 * uses_right / uses_lm tell which blocks are live.  want_jac = 0 skips Jacobians. */
static int factor_eval(const orc_chain *c, const orc_factor *f, int want_jac, double *we, double *JL, double *JR,
                       double *Jm, int *uses_right, int *uses_lm) {
  const int d = c->d, b = c->b, pd = c->pd, ld = c->ld;
  const double *p1 = c->pose + (size_t)f->idx * pd, *v1 = c->vel + (size_t)f->idx * d;
  const double *p2 = p1 + pd, *v2 = v1 + d;
  const double *pt = (f->lm >= 0) ? c->lmk + (size_t)f->lm * ld : NULL;
  double e[MAXR], H1_[MAXR * 6], H2_[MAXR * 6], H3_[MAXR * 6], H4_[MAXR * 6], H5_[MAXR * 3];
  double *H1 = want_jac ? H1_ : NULL, *H2 = want_jac ? H2_ : NULL, *H3 = want_jac ? H3_ : NULL;
  double *H4 = want_jac ? H4_ : NULL, *H5 = want_jac ? H5_ : NULL;
  int rows = 0;
  *uses_right = 0;
  *uses_lm = 0;
  if (want_jac) {
    orc_zero(MAXR * b, JL);
    orc_zero(MAXR * b, JR);
    if (ld > 0) orc_zero(MAXR * ld, Jm);
  }
#define PUT(J, H, col0, w)                                        \
  for (int r_ = 0; r_ < rows; r_++)                                \
    for (int q_ = 0; q_ < (w); q_++) (J)[r_ * ((J) == Jm ? ld : b) + (col0) + q_] = (H)[r_ * (w) + q_]

  double Lam[144], Psi[144];
  switch (f->type) {
    case F_GP: {
      rows = b;
      gp_eval(c, f, e, H1, H2, H3, H4);
      *uses_right = 1;
      double R[144], t[MAXR];
      if (orc_gp_whitening(d, f->has_qc ? f->qc : c->Qc, f->dt, R)) return -1;
      orc_mm(b, b, 1, R, e, t);
      orc_copy(b, t, we);
      if (want_jac) {
        double A[144], RA[144];
        for (int r = 0; r < b; r++)
          for (int q = 0; q < d; q++) { A[r * b + q] = H1[r * d + q]; A[r * b + d + q] = H2[r * d + q]; }
        orc_mm(b, b, b, R, A, RA);
        orc_copy(b * b, RA, JL);
        for (int r = 0; r < b; r++)
          for (int q = 0; q < d; q++) { A[r * b + q] = H3[r * d + q]; A[r * b + d + q] = H4[r * d + q]; }
        orc_mm(b, b, b, R, A, RA);
        orc_copy(b * b, RA, JR);
      }
      return rows;
    }
    case F_POSE_PRIOR:
      rows = d;
      orc_prior_factor(c->kind, c->chart, f->meas, p1, e, H1);
      if (want_jac) PUT(JL, H1, 0, d);
      break;
    case F_VEL_PRIOR:
      rows = d;
      for (int i = 0; i < d; i++) e[i] = v1[i] - f->meas[i];
      if (want_jac) { orc_eye(d, H1); PUT(JL, H1, d, d); }
      break;
    case F_BETWEEN:
      rows = d;
      orc_between_factor(c->kind, c->chart, f->meas, p1, p2, e, H1, H2);
      *uses_right = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JR, H2, 0, d); }
      break;
    case F_CLOSURE:   /* JL: columns of state idx, JR: columns of state idx2 (uses_right = 2: "right" is not idx + 1) */
      rows = d;
      orc_between_factor(c->kind, c->chart, f->meas, p1, c->pose + (size_t)f->idx2 * pd, e, H1, H2);
      *uses_right = 2;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JR, H2, 0, d); }
      break;
    case F_LM_PRIOR:
      rows = ld;
      for (int i = 0; i < ld; i++) e[i] = pt[i] - f->meas[i];
      *uses_lm = 1;
      if (want_jac) { orc_eye(ld, H5); PUT(Jm, H5, 0, ld); }
      break;
    case F_INTERP_RANGE:
      rows = 1;
      if (orc_calcLambda(d, c->Qc, f->dt, f->tau, Lam) || orc_calcPsi(d, c->Qc, f->dt, f->tau, Psi)) return -1;
      if (c->kind == ORC_POSE2)
        e[0] = orc_interp_range_pose2(Lam, Psi, f->meas[0], f->has_sensor ? f->sensor : NULL, p1, v1, p2, v2, pt,
                                      H1, H2, H3, H4, H5);
      else if (c->kind == ORC_POSE3)
        e[0] = (c->vw ? orc_interp_range_pose3vw : orc_interp_range_pose3)(
            Lam, Psi, f->meas[0], f->has_sensor ? f->sensor : NULL, p1, v1, p2, v2, pt, H1, H2, H3, H4, H5);
      else if (c->kind == ORC_LINEAR3)
        e[0] = orc_interp_range_2dlinear(Lam, Psi, f->meas[0], p1, v1, p2, v2, pt, H1, H2, H3, H4,
                                         H5);
      else return -2;
      *uses_right = 1;
      *uses_lm = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JL, H2, d, d); PUT(JR, H3, 0, d); PUT(JR, H4, d, d); PUT(Jm, H5, 0, ld); }
      break;
    case F_RANGE:
      rows = 1;
      if (c->kind == ORC_POSE2) e[0] = orc_range_pose2(f->meas[0], p1, pt, H1, H5);
      else if (c->kind == ORC_LINEAR3) e[0] = orc_range_2dlinear(f->meas[0], p1, pt, H1, H5);
      else if (c->kind == ORC_POSE3) e[0] = orc_pose3_range(p1, pt, H1, H5) - f->meas[0];
      else return -2;
      *uses_lm = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(Jm, H5, 0, ld); }
      break;
    case F_INTERP_ATT:
      rows = 2;
      if (c->kind == ORC_ROT3_BIAS) {   /* the rotation part of the AHRS state; Qc = the 3 x 3 corner */
        double Q3[9];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Q3[i * 3 + j] = c->Qc[i * 6 + j];
        if (orc_calcLambda(3, Q3, f->dt, f->tau, Lam) || orc_calcPsi(3, Q3, f->dt, f->tau, Psi)) return -1;
        orc_interp_attitude_rot3(Lam, Psi, f->aux, f->aux + 3, p1, v1, p2, v2, e, H1, H2, H3, H4);
        *uses_right = 1;
        if (want_jac) { PUT(JL, H1, 0, 3); PUT(JL, H2, d, 3); PUT(JR, H3, 0, 3); PUT(JR, H4, d, 3); }
        break;
      }
      if (c->kind != ORC_ROT3) return -2;
      if (orc_calcLambda(d, c->Qc, f->dt, f->tau, Lam) || orc_calcPsi(d, c->Qc, f->dt, f->tau, Psi)) return -1;
      orc_interp_attitude_rot3(Lam, Psi, f->aux, f->aux + 3, p1, v1, p2, v2, e, H1, H2, H3, H4);
      *uses_right = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JL, H2, d, d); PUT(JR, H3, 0, d); PUT(JR, H4, d, d); }
      break;
    case F_AHRS: {
      if (c->kind != ORC_ROT3_BIAS) return -2;
      rows = 3;
      double A[9], B[9], C3[9], t[9];
      orc_ahrs_factor(p1, p2, p1 + 9, f->ahrs, e, want_jac ? A : NULL, want_jac ? B : NULL, want_jac ? C3 : NULL);
      *uses_right = 1;
      /* Gaussian noise model: whiten by the full R (the diagonal step below then multiplies by 1) */
      const double *R = f->ahrs + 25;
      double we3[3];
      orc_mm(3, 3, 1, R, e, we3);
      orc_copy(3, we3, e);
      if (want_jac) {
        orc_mm(3, 3, 3, R, A, t); PUT(JL, t, 0, 3);
        orc_mm(3, 3, 3, R, C3, t); PUT(JL, t, 3, 3);
        orc_mm(3, 3, 3, R, B, t); PUT(JR, t, 0, 3);
      }
    } break;
    case F_INTERP_GPS:
      if (c->kind != ORC_POSE3) return -2;
      rows = 3;
      if (orc_calcLambda(d, c->Qc, f->dt, f->tau, Lam) || orc_calcPsi(d, c->Qc, f->dt, f->tau, Psi)) return -1;
      (c->vw ? orc_interp_gps_pose3vw : orc_interp_gps_pose3)(Lam, Psi, f->meas, f->has_sensor ? f->sensor : NULL, p1,
                                                                 v1, p2, v2, e, H1, H2, H3, H4);
      *uses_right = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JL, H2, d, d); PUT(JR, H3, 0, d); PUT(JR, H4, d, d); }
      break;
    case F_ODOM2D:
      if (c->kind != ORC_LINEAR3) return -2;
      rows = 3;
      orc_odometry_2dlinear(f->meas, p1, p2, e, H1, H2);
      *uses_right = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JR, H2, 0, d); }
      break;
    case F_BEARING_RANGE:
      if (c->kind != ORC_LINEAR3) return -2;
      rows = 2;
      orc_range_bearing_2dlinear(f->meas[0], f->meas[1], p1, pt, e, H1, H5);
      *uses_lm = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(Jm, H5, 0, ld); }
      break;
    case F_INTERP_PROJ:
      rows = 2;
      if (orc_calcLambda(d, c->Qc, f->dt, f->tau, Lam) || orc_calcPsi(d, c->Qc, f->dt, f->tau, Psi)) return -1;
      orc_interp_projection_pose3_ds2(Lam, Psi, f->meas, f->aux, f->has_sensor ? f->sensor : NULL, p1, v1, p2, v2, pt, e,
                                  H1, H2, H3, H4, H5);
      *uses_right = 1;
      *uses_lm = 1;
      if (want_jac) { PUT(JL, H1, 0, d); PUT(JL, H2, d, d); PUT(JR, H3, 0, d); PUT(JR, H4, d, d); PUT(Jm, H5, 0, ld); }
      break;
    default: return -2;
  }
#undef PUT
  if (f->has_cov) {   /* Gaussian noise model: rows <- R rows (R upper triangular: row r needs rows >= r, so ascending in place) */
    const double *Rw = f->sqi;
    for (int r = 0; r < rows; r++) {
      double acc = 0.0;
      for (int q = r; q < rows; q++) acc += Rw[r * rows + q] * e[q];
      we[r] = acc;
      if (want_jac) {
        for (int col = 0; col < b; col++) {
          double aL = 0.0, aR = 0.0;
          for (int q = r; q < rows; q++) { aL += Rw[r * rows + q] * JL[q * b + col]; aR += Rw[r * rows + q] * JR[q * b + col]; }
          JL[r * b + col] = aL; JR[r * b + col] = aR;
        }
        for (int col = 0; col < ld; col++) {
          double am = 0.0;
          for (int q = r; q < rows; q++) am += Rw[r * rows + q] * Jm[q * ld + col];
          Jm[r * ld + col] = am;
        }
      }
    }
    return rows;
  }
  /* diagonal noise model: R = diag(1/sigma) */
  for (int r = 0; r < rows; r++) {
    double w = 1.0 / f->sig[r];
    we[r] = e[r] * w;
    if (want_jac) {
      for (int q = 0; q < b; q++) { JL[r * b + q] *= w; JR[r * b + q] *= w; }
      for (int q = 0; q < ld; q++) Jm[r * ld + q] *= w;
    }
  }
  return rows;
}

int orc_chain_linearize_gp(const orc_chain *c, double *errors, double *jac) {
  const int d = c->d, b = c->b;
  size_t k = 0;
  for (int i = 0; i < c->nf; i++) {
    const orc_factor *f = &c->f[i];
    if (f->type != F_GP) continue;
    double *e = errors + k * b;
    double *H = jac ? jac + k * 4 * b * d : NULL;
    gp_eval(c, f, e, H, H ? H + b * d : NULL, H ? H + 2 * b * d : NULL, H ? H + 3 * b * d : NULL);
    k++;
  }
  return (int)k;
}

/* Threads (OpenMP): the factors are EVALUATED in parallel -- what GTSAM built with TBB does in
 * NonlinearFactorGraph::linearize / error -- into per-factor storage and then accumulated serially in factor order, so
 * every result is bit-identical for any thread count.  The elimination stays sequential: the elimination tree of a
 * chain in chain order is a path.  orc_set_threads(0) = all cores (default 1: the tests and the 1-core baseline). */
static int g_threads = 1;
void orc_set_threads(int n) {
#ifdef _OPENMP
  g_threads = n > 0 ? n : omp_get_num_procs();
#else
  (void)n;
  g_threads = 1;
#endif
}
int orc_get_threads(void) { return g_threads; }

/* Unwhitened evaluateError + Jacobians of every measurement factor of one type (F_INTERP_RANGE ...), in the order
 * they were added: errors count x rows, jac count x rows x (2b + 3) = per row [H1 H2 | H3 H4 | H5 zero-padded to 3].
 * The per-factor counterpart of gpslam_hip_linearize_meas: factor_eval's diagonally whitened rows times sigma. */
int orc_chain_linearize_meas(const orc_chain *c, int type, double *errors, double *jac) {
  const int b = c->b, ld = c->ld, W = 2 * b + 3;
  size_t k = 0;
  if (type < F_INTERP_RANGE) return -1;
  for (int i = 0; i < c->nf; i++) {
    const orc_factor *f = &c->f[i];
    if (f->type != type) continue;
    double we[MAXR], JL[MAXR * MAXB], JR[MAXR * MAXB], Jm[MAXR * 3];
    int ur, ul;
    int rows = factor_eval(c, f, 1, we, JL, JR, Jm, &ur, &ul);
    if (rows < 0) return rows;
    if (type == F_AHRS) {   /* full (non-diagonal) whitening: evaluate the factor itself for the unwhitened values */
      double A[9], B[9], C3[9];
      const double *p1 = c->pose + (size_t)f->idx * c->pd;
      orc_ahrs_factor(p1, p1 + c->pd, p1 + 9, f->ahrs, we, A, B, C3);
      orc_zero(MAXR * b, JL);
      orc_zero(MAXR * b, JR);
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) { JL[r * b + q] = A[r * 3 + q]; JL[r * b + 3 + q] = C3[r * 3 + q]; JR[r * b + q] = B[r * 3 + q]; }
    }
    if (f->has_cov) {   /* undo the full whitening: solve R x = (whitened) from the last row up */
      const double *Rw = f->sqi;
      for (int r = rows - 1; r >= 0; r--) {
        double acc = we[r];
        for (int q = r + 1; q < rows; q++) acc -= Rw[r * rows + q] * we[q];
        we[r] = acc / Rw[r * rows + r];
        for (int col = 0; col < b; col++) {
          double aL = JL[r * b + col], aR = JR[r * b + col];
          for (int q = r + 1; q < rows; q++) { aL -= Rw[r * rows + q] * JL[q * b + col]; aR -= Rw[r * rows + q] * JR[q * b + col]; }
          JL[r * b + col] = aL / Rw[r * rows + r]; JR[r * b + col] = aR / Rw[r * rows + r];
        }
        for (int col = 0; col < ld; col++) {
          double am = Jm[r * ld + col];
          for (int q = r + 1; q < rows; q++) am -= Rw[r * rows + q] * Jm[q * ld + col];
          Jm[r * ld + col] = am / Rw[r * rows + r];
        }
      }
    }
    for (int r = 0; r < rows; r++) {
      const double sg = f->has_cov ? 1.0 : f->sig[r];
      errors[k * rows + r] = we[r] * sg;
      double *o = jac + (k * rows + r) * W;
      for (int q = 0; q < W; q++) o[q] = 0.0;
      for (int q = 0; q < b; q++) { o[q] = JL[r * b + q] * sg; o[b + q] = JR[r * b + q] * sg; }
      for (int q = 0; q < ld; q++) o[2 * b + q] = Jm[r * ld + q] * sg;
    }
    k++;
  }
  return (int)k;
}

int orc_chain_error(const orc_chain *c, double *err) {
  double *fe = (double *)malloc(sizeof(double) * (size_t)(c->nf > 0 ? c->nf : 1));
  int bad = 0;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int i = 0; i < c->nf; i++) {
    double we[MAXR];
    int ur, ul;
    int rows = factor_eval(c, &c->f[i], 0, we, NULL, NULL, NULL, &ur, &ul);
    if (rows < 0) { bad = rows; fe[i] = 0.0; }
    else fe[i] = 0.5 * orc_dot(rows, we, we);   /* NoiseModelFactor::error = 0.5 |R e|^2 */
  }
  double total = 0.0;
  for (int i = 0; i < c->nf; i++) total += fe[i];
  free(fe);
  if (bad) return bad;
  *err = total;
  return 0;
}

/* ------------------------------------------------------------------ normal equations */

typedef struct {
  int N, b, nl;
  double *D, *O, *g, *B, *HLL, *gL;
  double err;
  /* loop closures: block k = H[clo_hi[k], clo_lo[k]] (b x b, rows of the later state), clo_lo < clo_hi */
  int nclo;
  int *clo_lo, *clo_hi;
  double *clo_H;
} orc_neq;

static void neq_free(orc_neq *q) {
  free(q->D); free(q->O); free(q->g); free(q->B); free(q->HLL); free(q->gL);
  free(q->clo_lo); free(q->clo_hi); free(q->clo_H);
}

static int build_neq(const orc_chain *c, orc_neq *q) {
  const int N = c->N, b = c->b, ld = c->ld, nl = c->L * ld;
  memset(q, 0, sizeof(*q));
  q->N = N; q->b = b; q->nl = nl;
  q->D = (double *)calloc((size_t)N * b * b, sizeof(double));
  q->O = (double *)calloc((size_t)N * b * b, sizeof(double));
  q->g = (double *)calloc((size_t)N * b, sizeof(double));
  if (nl > 0) {
    q->B = (double *)calloc((size_t)N * b * nl, sizeof(double));
    q->HLL = (double *)calloc((size_t)nl * nl, sizeof(double));
    q->gL = (double *)calloc((size_t)nl, sizeof(double));
  }
  {
    int nclo = 0;
    for (int k = 0; k < c->nf; k++) nclo += (c->f[k].type == F_CLOSURE);
    if (nclo > 0) {
      q->clo_lo = (int *)calloc((size_t)nclo, sizeof(int));
      q->clo_hi = (int *)calloc((size_t)nclo, sizeof(int));
      q->clo_H = (double *)calloc((size_t)nclo * b * b, sizeof(double));
    }
  }
  double total = 0.0;
  /* pass 1 (parallel): evaluate every factor into its own slot; pass 2 (serial, factor order): accumulate */
  typedef struct { double we[MAXR], JL[MAXR * MAXB], JR[MAXR * MAXB], Jm[MAXR * 3]; int rows, ur, ul; } lin_t;
  lin_t *lin = (lin_t *)malloc(sizeof(lin_t) * (size_t)(c->nf > 0 ? c->nf : 1));
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int k = 0; k < c->nf; k++)
    lin[k].rows = factor_eval(c, &c->f[k], 1, lin[k].we, lin[k].JL, lin[k].JR, lin[k].Jm, &lin[k].ur, &lin[k].ul);
  for (int k = 0; k < c->nf; k++) {
    const orc_factor *f = &c->f[k];
    const double *we = lin[k].we, *JL = lin[k].JL, *JR = lin[k].JR, *Jm = lin[k].Jm;
    const int ur = lin[k].ur, ul = lin[k].ul, rows = lin[k].rows;
    if (rows < 0) { free(lin); neq_free(q); return rows; }
    total += 0.5 * orc_dot(rows, we, we);
    const int i = f->idx;
    const int is_lm_only = (f->type == F_LM_PRIOR);
    double T[MAXB * MAXB], tv[MAXB];
    if (!is_lm_only) {
      orc_mtm(rows, b, b, JL, JL, T);
      orc_axpy(b * b, 1.0, T, q->D + (size_t)i * b * b);
      orc_mtm(rows, b, 1, JL, we, tv);
      orc_axpy(b, -1.0, tv, q->g + (size_t)i * b);
      if (ur == 2) {   /* loop closure: diagonal blocks and gradient where they belong, the coupling block in the closure list */
        const int j = f->idx2;
        if (j < 0 || j >= N || i < 0 || i >= N) { free(lin); neq_free(q); return -2; }
        orc_mtm(rows, b, b, JR, JR, T);
        orc_axpy(b * b, 1.0, T, q->D + (size_t)j * b * b);
        orc_mtm(rows, b, 1, JR, we, tv);
        orc_axpy(b, -1.0, tv, q->g + (size_t)j * b);
        double *Hk = q->clo_H + (size_t)q->nclo * b * b;
        if (j > i) orc_mtm(rows, b, b, JR, JL, Hk);   /* H[j, i] = JR^T JL */
        else orc_mtm(rows, b, b, JL, JR, Hk);          /* H[i, j] = JL^T JR */
        q->clo_lo[q->nclo] = i < j ? i : j;
        q->clo_hi[q->nclo] = i < j ? j : i;
        q->nclo++;
      } else if (ur) {
        orc_mtm(rows, b, b, JR, JR, T);
        orc_axpy(b * b, 1.0, T, q->D + (size_t)(i + 1) * b * b);
        orc_mtm(rows, b, b, JR, JL, T);   /* O[i] = H[i+1, i] */
        orc_axpy(b * b, 1.0, T, q->O + (size_t)i * b * b);
        orc_mtm(rows, b, 1, JR, we, tv);
        orc_axpy(b, -1.0, tv, q->g + (size_t)(i + 1) * b);
      }
    }
    if (ul) {
      const int l0 = f->lm * ld;
      double M[3 * 3], ml[3];
      orc_mtm(rows, ld, ld, Jm, Jm, M);
      for (int r = 0; r < ld; r++)
        for (int s = 0; s < ld; s++) q->HLL[(size_t)(l0 + r) * nl + l0 + s] += M[r * ld + s];
      orc_mtm(rows, ld, 1, Jm, we, ml);
      for (int r = 0; r < ld; r++) q->gL[l0 + r] -= ml[r];
      if (!is_lm_only) {
        double X[MAXB * 3];
        orc_mtm(rows, b, ld, JL, Jm, X);
        for (int r = 0; r < b; r++)
          for (int s = 0; s < ld; s++) q->B[((size_t)i * b + r) * nl + l0 + s] += X[r * ld + s];
        if (ur) {
          orc_mtm(rows, b, ld, JR, Jm, X);
          for (int r = 0; r < b; r++)
            for (int s = 0; s < ld; s++) q->B[((size_t)(i + 1) * b + r) * nl + l0 + s] += X[r * ld + s];
        }
      }
    }
  }
  free(lin);
  q->err = total;
  return 0;
}

int orc_chain_normal_equations(const orc_chain *c, double *D, double *O, double *g, double *B, double *HLL,
                               double *gL) {
  orc_neq q;
  int rc = build_neq(c, &q);
  if (rc) return rc;
  if (D) orc_copy(q.N * q.b * q.b, q.D, D);
  if (O) orc_copy(q.N * q.b * q.b, q.O, O);
  if (g) orc_copy(q.N * q.b, q.g, g);
  if (q.nl > 0) {
    if (B) memcpy(B, q.B, sizeof(double) * (size_t)q.N * q.b * q.nl);
    if (HLL) memcpy(HLL, q.HLL, sizeof(double) * (size_t)q.nl * q.nl);
    if (gL) orc_copy(q.nl, q.gL, gL);
  }
  neq_free(&q);
  return 0;
}

/* ------------------------------------------------------------------ bordered block-tridiagonal solve */

/* solve U^T z = r (U upper b x b) for nrhs columns stored row-major r(b x nrhs), in place */
static void utsolve(int b, const double *U, int nrhs, double *r) {
  for (int i = 0; i < b; i++) {
    for (int k = 0; k < i; k++) {
      double u = U[k * b + i];
      if (u != 0.0) for (int j = 0; j < nrhs; j++) r[i * nrhs + j] -= u * r[k * nrhs + j];
    }
    double dinv = 1.0 / U[i * b + i];
    for (int j = 0; j < nrhs; j++) r[i * nrhs + j] *= dinv;
  }
}
/* solve U z = r in place */
static void usolve(int b, const double *U, int nrhs, double *r) {
  for (int i = b - 1; i >= 0; i--) {
    for (int k = i + 1; k < b; k++) {
      double u = U[i * b + k];
      if (u != 0.0) for (int j = 0; j < nrhs; j++) r[i * nrhs + j] -= u * r[k * nrhs + j];
    }
    double dinv = 1.0 / U[i * b + i];
    for (int j = 0; j < nrhs; j++) r[i * nrhs + j] *= dinv;
  }
}

/* dense SPD solve (n x n) in place on rhs */
static int dense_spd_solve(int n, double *A, double *rhs) {
  for (int j = 0; j < n; j++) {
    double dd = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) dd -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
    if (!(dd > 0.0)) return -1;
    dd = sqrt(dd);
    A[(size_t)j * n + j] = dd;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)j * n + i];
      for (int k = 0; k < j; k++) s -= A[(size_t)k * n + j] * A[(size_t)k * n + i];
      A[(size_t)j * n + i] = s / dd;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = rhs[i];
    for (int k = 0; k < i; k++) s -= A[(size_t)k * n + i] * rhs[k];
    rhs[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = rhs[i];
    for (int k = i + 1; k < n; k++) s -= A[(size_t)i * n + k] * rhs[k];
    rhs[i] = s / A[(size_t)i * n + i];
  }
  return 0;
}

/* Solve [[H, B],[B^T, HLL]] [x; xl] = [g; gL], H block tridiagonal (D, O), adding lambda*I (LM damping,
 * GTSAM diagonalDamping=false) to every variable.  Elimination order = chain order then landmarks. */
static int bordered_solve(const orc_neq *q, double lambda, double *x, double *xl) {
  const int N = q->N, b = q->b, nl = q->nl, w = 1 + nl;
  double *U = (double *)malloc(sizeof(double) * (size_t)N * b * b);      /* chol factors, D~ = U^T U */
  double *E = (double *)malloc(sizeof(double) * (size_t)N * b * b);      /* E_i = O_i U_i^-1 */
  double *Y = (double *)malloc(sizeof(double) * (size_t)N * b * w);      /* [y | W] per block, b x (1+nl) */
  int rc = 0;
  for (int i = 0; i < N; i++) {
    double *Ui = U + (size_t)i * b * b, *Yi = Y + (size_t)i * b * w;
    orc_copy(b * b, q->D + (size_t)i * b * b, Ui);
    for (int r = 0; r < b; r++) Ui[r * b + r] += lambda;
    for (int r = 0; r < b; r++) {
      Yi[r * w] = q->g[(size_t)i * b + r];
      for (int s = 0; s < nl; s++) Yi[r * w + 1 + s] = q->B[((size_t)i * b + r) * nl + s];
    }
    if (i > 0) {
      const double *Ep = E + (size_t)(i - 1) * b * b, *Yp = Y + (size_t)(i - 1) * b * w;
      for (int r = 0; r < b; r++)
        for (int s = 0; s < b; s++) {
          double acc = 0.0;
          for (int k = 0; k < b; k++) acc += Ep[r * b + k] * Ep[s * b + k];
          Ui[r * b + s] -= acc;
        }
      for (int r = 0; r < b; r++)
        for (int s = 0; s < w; s++) {
          double acc = 0.0;
          for (int k = 0; k < b; k++) acc += Ep[r * b + k] * Yp[k * w + s];
          Yi[r * w + s] -= acc;
        }
    }
    if (orc_chol_upper(b, Ui)) { rc = -3; goto done; }
    utsolve(b, Ui, w, Yi);
    if (i + 1 < N) {
      /* E_i = O_i U_i^-1  <=>  U_i^T E_i^T = O_i^T */
      double Et[MAXB * MAXB];
      orc_tr(b, b, q->O + (size_t)i * b * b, Et);
      utsolve(b, Ui, b, Et);
      orc_tr(b, b, Et, E + (size_t)i * b * b);
    }
  }
  if (nl > 0) {
    double *S = (double *)malloc(sizeof(double) * (size_t)nl * nl);
    memcpy(S, q->HLL, sizeof(double) * (size_t)nl * nl);
    for (int s = 0; s < nl; s++) { S[(size_t)s * nl + s] += lambda; xl[s] = q->gL[s]; }
    for (int i = 0; i < N; i++) {
      const double *Yi = Y + (size_t)i * b * w;
      for (int r = 0; r < b; r++) {
        const double *row = Yi + r * w;
        for (int s = 0; s < nl; s++) {
          double ws = row[1 + s];
          if (ws == 0.0) continue;
          xl[s] -= ws * row[0];
          for (int t = 0; t < nl; t++) S[(size_t)s * nl + t] -= ws * row[1 + t];
        }
      }
    }
    if (dense_spd_solve(nl, S, xl)) { free(S); rc = -4; goto done; }
    free(S);
  }
  for (int i = N - 1; i >= 0; i--) {
    double z[MAXB];
    const double *Yi = Y + (size_t)i * b * w;
    for (int r = 0; r < b; r++) {
      double s = Yi[r * w];
      for (int t = 0; t < nl; t++) s -= Yi[r * w + 1 + t] * xl[t];
      z[r] = s;
    }
    if (i + 1 < N) {
      const double *Ei = E + (size_t)i * b * b, *xn = x + (size_t)(i + 1) * b;
      for (int r = 0; r < b; r++) {
        double s = 0.0;
        for (int k = 0; k < b; k++) s += Ei[k * b + r] * xn[k];
        z[r] -= s;
      }
    }
    usolve(b, U + (size_t)i * b * b, 1, z);
    orc_copy(b, z, x + (size_t)i * b);
  }
done:
  free(U); free(E); free(Y);
  return rc;
}

/* Envelope (skyline) Cholesky of the whole system in the order [x0, v0, x1, v1, ..., l0, l1, ...] -- the solver of chains that hold
 * loop closures, whose coupling blocks H[hi, lo] lie outside the block-tridiagonal pattern.  Row r of the lower triangle is stored
 * from its first non-zero column first[r] to the diagonal; fill stays inside the envelope, so the factorisation is exact and costs
 * b (span b)^2 / 2 for the b rows of a closure's later state on top of the chain's N b^3.  Deliberately a different elimination
 * than the product's (which keeps the chain solver and applies the closures as a low-rank correction). */
static int skyline_solve(const orc_neq *q, double lambda, double *x, double *xl) {
  const int N = q->N, b = q->b, nl = q->nl, nx = N * b, n = nx + nl;
  int *first = (int *)malloc(sizeof(int) * (size_t)n);
  size_t *ptr = (size_t *)malloc(sizeof(size_t) * (size_t)(n + 1));
  for (int s = 0; s < N; s++)
    for (int r = 0; r < b; r++) first[s * b + r] = (s > 0 ? s - 1 : 0) * b;
  for (int k = 0; k < q->nclo; k++)
    for (int r = 0; r < b; r++) {
      int *f = &first[q->clo_hi[k] * b + r];
      if (q->clo_lo[k] * b < *f) *f = q->clo_lo[k] * b;
    }
  for (int l = 0; l < nl; l++) {
    int f = nx;   /* the landmark block itself is dense */
    for (int rs = 0; rs < nx && f == nx; rs++)
      if (q->B[(size_t)rs * nl + l] != 0.0) f = rs - rs % b;
    first[nx + l] = f;
  }
  ptr[0] = 0;
  for (int r = 0; r < n; r++) ptr[r + 1] = ptr[r] + (size_t)(r - first[r] + 1);
  double *A = (double *)calloc(ptr[n], sizeof(double));
  double *y = (double *)malloc(sizeof(double) * (size_t)n);
#define SKY(r, c) A[ptr[r] + (size_t)((c) - first[r])]
  for (int s = 0; s < N; s++)
    for (int r = 0; r < b; r++) {
      const int R = s * b + r;
      for (int cc = 0; cc <= r; cc++) SKY(R, s * b + cc) = q->D[((size_t)s * b + r) * b + cc];
      SKY(R, R) += lambda;
      if (s > 0)
        for (int cc = 0; cc < b; cc++) SKY(R, (s - 1) * b + cc) = q->O[((size_t)(s - 1) * b + r) * b + cc];   /* O[s-1] = H[s, s-1] */
      y[R] = q->g[(size_t)s * b + r];
    }
  for (int k = 0; k < q->nclo; k++)
    for (int r = 0; r < b; r++)
      for (int cc = 0; cc < b; cc++) SKY(q->clo_hi[k] * b + r, q->clo_lo[k] * b + cc) += q->clo_H[((size_t)k * b + r) * b + cc];
  for (int l = 0; l < nl; l++) {
    const int R = nx + l;
    for (int rs = first[R]; rs < nx; rs++) SKY(R, rs) = q->B[(size_t)rs * nl + l];
    for (int m = 0; m <= l; m++) SKY(R, nx + m) = q->HLL[(size_t)l * nl + m];
    SKY(R, R) += lambda;
    y[R] = q->gL[l];
  }
  int rc = 0;
  for (int r = 0; r < n && !rc; r++) {
    for (int cc = first[r]; cc <= r; cc++) {
      const int k0 = first[r] > first[cc] ? first[r] : first[cc];
      double s = SKY(r, cc);
      const double *Lr = &SKY(r, k0), *Lc = &SKY(cc, k0);
      for (int k = 0; k < cc - k0; k++) s -= Lr[k] * Lc[k];
      if (cc < r) SKY(r, cc) = s / SKY(cc, cc);
      else if (!(s > 0.0)) rc = (r < nx) ? -3 : -4;
      else SKY(r, r) = sqrt(s);
    }
  }
  if (!rc) {
    for (int r = 0; r < n; r++) {
      double s = y[r];
      for (int cc = first[r]; cc < r; cc++) s -= SKY(r, cc) * y[cc];
      y[r] = s / SKY(r, r);
    }
    for (int r = n - 1; r >= 0; r--) {
      const double v = y[r] / SKY(r, r);
      y[r] = v;
      for (int cc = first[r]; cc < r; cc++) y[cc] -= SKY(r, cc) * v;
    }
    orc_copy(nx, y, x);
    for (int l = 0; l < nl; l++) xl[l] = y[nx + l];
  }
#undef SKY
  free(A); free(y); free(first); free(ptr);
  return rc;
}

/* chains: the sequential block Cholesky with the landmark border; chains with loop closures: the envelope Cholesky */
static int g_force_skyline = 0;
void orc_force_envelope_solver(int on) { g_force_skyline = on; }   /* tests: the second solver on graphs the first one serves */
static int solve_neq(const orc_neq *q, double lambda, double *x, double *xl) {
  if (q->nclo > 0 || g_force_skyline) return skyline_solve(q, lambda, x, xl);
  return bordered_solve(q, lambda, x, xl);
}

int orc_block_tridiag_solve(int N, int b, const double *D, const double *O, const double *g, double *x) {
  orc_neq q;
  memset(&q, 0, sizeof(q));
  q.N = N; q.b = b; q.nl = 0;
  q.D = (double *)D; q.O = (double *)O; q.g = (double *)g;
  return bordered_solve(&q, 0.0, x, NULL);
}

/* ------------------------------------------------------------------ GN / LM */

static void apply_update(orc_chain *c, const double *x, const double *xl, double *dinf) {
  const int d = c->d, b = c->b, pd = c->pd;
  double m = 0.0;
  for (int i = 0; i < c->N; i++) {
    double np[12];
    orc_retract(c->kind, c->chart, c->pose + (size_t)i * pd, x + (size_t)i * b, np);
    orc_copy(pd, np, c->pose + (size_t)i * pd);
    for (int k = 0; k < d; k++) c->vel[(size_t)i * d + k] += x[(size_t)i * b + d + k];
    for (int k = 0; k < b; k++) if (fabs(x[(size_t)i * b + k]) > m) m = fabs(x[(size_t)i * b + k]);
  }
  for (int k = 0; k < c->L * c->ld; k++) {
    c->lmk[k] += xl[k];
    if (fabs(xl[k]) > m) m = fabs(xl[k]);
  }
  *dinf = m;
}

/* GaussNewtonOptimizer::iterate: linearize -> solve -> retract -> error */
int orc_chain_iterate_gn(orc_chain *c, orc_stats *st) {
  orc_neq q;
  memset(st, 0, sizeof(*st));
  int rc = build_neq(c, &q);
  if (rc) { st->status = rc; return rc; }
  double *x = (double *)malloc(sizeof(double) * (size_t)c->N * c->b);
  double *xl = (double *)calloc((size_t)(q.nl > 0 ? q.nl : 1), sizeof(double));
  rc = solve_neq(&q, 0.0, x, xl);
  st->error_before = q.err;
  if (rc == 0) {
    apply_update(c, x, xl, &st->delta_inf_norm);
    orc_chain_error(c, &st->error_after);
    st->iterations = 1;
    st->accepted = 1;
  }
  st->status = rc;
  free(x); free(xl);
  neq_free(&q);
  return rc;
}

/* LevenbergMarquardtOptimizer::iterate with the 4.0 defaults (diagonalDamping = false,
 * useFixedLambdaFactor = true): linearise once; then tryLambda until it says stop:
 *   damp with lambda*I; solve; if solved and linErr(0) - linErr(delta) >= 0: retract, newErr,
 *     costChange = err - newErr;
 *     if the linear decrease > 1e-20: rho = costChange / linear decrease, success = rho > minModelFidelity;
 *     if |costChange| < relativeErrorTol * err: stopSearchingLambda  (the small-cost-change stop, round 5);
 *   success            -> keep the step, lambda /= factor (not below lambdaLowerBound), done;
 *   else, not stopping -> lambda *= factor; done (giving up) if lambda >= lambdaUpperBound, else try again;
 *   else (stopping)    -> done, lambda and the values untouched ("relative cost reduction is small").
 * As recalled from GTSAM 4.0.x LevenbergMarquardtOptimizer::tryLambda (the reference's call sites:
 * matlab/PlazaPose2.m:210-226, matlab/GPAHRSexample.m:259-264); GTSAM is not in this image: PARITY UNPINNED.
 * st->trials counts the lambdas tried by this call. */
int orc_chain_iterate_lm(orc_chain *c, double *lambda, const orc_params *p, orc_stats *st) {
  orc_neq q;
  memset(st, 0, sizeof(*st));
  int rc = build_neq(c, &q);
  if (rc) { st->status = rc; return rc; }
  const int nx = c->N * c->b, nl = q.nl;
  double *x = (double *)malloc(sizeof(double) * (size_t)nx);
  double *xl = (double *)calloc((size_t)(nl > 0 ? nl : 1), sizeof(double));
  double *pose0 = (double *)malloc(sizeof(double) * (size_t)c->N * c->pd);
  double *vel0 = (double *)malloc(sizeof(double) * (size_t)c->N * c->d);
  double *lm0 = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1));
  orc_copy(c->N * c->pd, c->pose, pose0);
  orc_copy(c->N * c->d, c->vel, vel0);
  if (nl > 0) orc_copy(nl, c->lmk, lm0);
  st->error_before = q.err;
  st->error_after = q.err;
  st->last_trial_error = q.err;
  for (;;) {
    rc = solve_neq(&q, *lambda, x, xl);
    int ok = 0, stop_searching = 0;
    st->trials++;
    if (rc == 0) {
      /* model decrease: linErr(0) - linErr(delta) = 0.5 delta.g + 0.5 lambda |delta|^2 since (H + lambda I) delta = g */
      double dg = orc_dot(nx, x, q.g) + (nl > 0 ? orc_dot(nl, xl, q.gL) : 0.0);
      double dd = orc_dot(nx, x, x) + (nl > 0 ? orc_dot(nl, xl, xl) : 0.0);
      double lin_change = 0.5 * dg + 0.5 * (*lambda) * dd;
      if (lin_change >= 0.0) {
        double dinf, new_err;
        apply_update(c, x, xl, &dinf);
        orc_chain_error(c, &new_err);
        double cost_change = q.err - new_err;
        st->last_trial_error = new_err;
        if (lin_change > 1e-20 && cost_change / lin_change > p->min_model_fidelity) {
          ok = 1;
          st->error_after = new_err;
          st->delta_inf_norm = dinf;
        } else {
          orc_copy(c->N * c->pd, pose0, c->pose);
          orc_copy(c->N * c->d, vel0, c->vel);
          if (nl > 0) orc_copy(nl, lm0, c->lmk);
        }
        if (fabs(cost_change) < p->relative_error_tol * q.err) stop_searching = 1;
      }
    }
    if (ok) {
      *lambda /= p->lambda_factor;
      if (*lambda < p->lambda_lower_bound) *lambda = p->lambda_lower_bound;
      st->accepted = 1;
      break;
    }
    if (stop_searching) break;
    *lambda *= p->lambda_factor;
    if (*lambda >= p->lambda_upper_bound) break;
  }
  st->iterations = 1;
  st->lambda = *lambda;
  st->status = 0;
  free(x); free(xl); free(pose0); free(vel0); free(lm0);
  neq_free(&q);
  return 0;
}

/* NonlinearOptimizer::defaultOptimize + checkConvergence:
 * do { cur = error(); iterate(); } while (iters < max && !converged(cur, new)) */
int orc_chain_optimize(orc_chain *c, const orc_params *p, orc_stats *st) {
  orc_stats it;
  double lambda = p->lambda_initial;
  double err0;
  int iters = 0, rc = orc_chain_error(c, &err0);
  memset(st, 0, sizeof(*st));
  if (rc) { st->status = rc; return rc; }
  st->error_before = err0;
  double new_err = err0, dinf = 0.0;
  if (err0 <= p->error_tol) { st->error_after = err0; return 0; }
  for (;;) {
    double cur = new_err;
    rc = p->use_lm ? orc_chain_iterate_lm(c, &lambda, p, &it) : orc_chain_iterate_gn(c, &it);
    if (rc) { st->status = rc; break; }
    iters++;
    new_err = it.error_after;
    dinf = it.delta_inf_norm;
    if (iters >= p->max_iterations) break;
    if (new_err <= p->error_tol) break;
    double abs_dec = cur - new_err, rel_dec = abs_dec / cur;
    if (rel_dec <= p->relative_error_tol || abs_dec <= p->absolute_error_tol) break;
    if (p->delta_tol > 0.0 && dinf < p->delta_tol) break;
    if (p->use_lm && !it.accepted) break;
  }
  st->iterations = iters;
  st->error_after = new_err;
  st->delta_inf_norm = dinf;
  st->lambda = lambda;
  return rc;
}
