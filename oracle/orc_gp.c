/*
 * orc_gp.c -- CPU restatement of gpslam/gp: GP kernel matrices, SE(3) Jacobian utilities,
 * the four GP priors and the four GP interpolators.
 *
 * TEST INFRASTRUCTURE ONLY (see gpslam_oracle.h).  Each function cites the reference
 * lines it follows (paths relative to /root/reference).
 */
#include "gpslam_oracle.h"
#include "orc_math.h"

#include <float.h>

/* ------------------------------------------------------------------ GPutils */

/* Kronecker helper: out(2D x 2D) = [[a*M, b*M],[c*M, d*M]] */
static void kron2(int D, double a, double b, double c, double d, const double *M, double *out) {
  int n = 2 * D;
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) {
      double m = M[i * D + j];
      out[i * n + j] = a * m;
      out[i * n + D + j] = b * m;
      out[(D + i) * n + j] = c * m;
      out[(D + i) * n + D + j] = d * m;
    }
}

/* calcQ -- gpslam/gp/GPutils.h:24-30 */
void orc_calcQ(int D, const double *Qc, double tau, double *Q) {
  kron2(D, 1.0 / 3 * pow(tau, 3.0), 1.0 / 2 * pow(tau, 2.0), 1.0 / 2 * pow(tau, 2.0), tau, Qc, Q);
}

/* calcQ_inv -- gpslam/gp/GPutils.h:33-41 */
int orc_calcQ_inv(int D, const double *Qc, double tau, double *Qinv) {
  double Qc_inv[36];
  if (orc_inv(D, Qc, Qc_inv)) return -1;
  kron2(D, 12.0 * pow(tau, -3.0), (-6.0) * pow(tau, -2.0), (-6.0) * pow(tau, -2.0), 4.0 * pow(tau, -1.0),
        Qc_inv, Qinv);
  return 0;
}

/* calcPhi -- gpslam/gp/GPutils.h:44-51 */
void orc_calcPhi(int D, double tau, double *Phi) {
  int n = 2 * D;
  orc_eye(n, Phi);
  for (int i = 0; i < D; i++) Phi[i * n + D + i] = tau;
}

/* Q(tau) * Phi(dt - tau)^T * Qinv(dt): shared by calcLambda and calcPsi */
static int psi_core(int D, const double *Qc, double dt, double tau, double *Psi) {
  int n = 2 * D;
  double Q[144], Phi[144], PhiT[144], Qinv[144], T1[144];
  orc_calcQ(D, Qc, tau, Q);
  orc_calcPhi(D, dt - tau, Phi);
  orc_tr(n, n, Phi, PhiT);
  if (orc_calcQ_inv(D, Qc, dt, Qinv)) return -1;
  orc_mm(n, n, n, Q, PhiT, T1);
  orc_mm(n, n, n, T1, Qinv, Psi);
  return 0;
}

/* calcPsi -- gpslam/gp/GPutils.h:64-71 */
int orc_calcPsi(int D, const double *Qc, double dt, double tau, double *Psi) {
  return psi_core(D, Qc, dt, tau, Psi);
}

/* calcLambda -- gpslam/gp/GPutils.h:54-61 */
int orc_calcLambda(int D, const double *Qc, double dt, double tau, double *Lambda) {
  int n = 2 * D;
  double Psi[144], Phi[144], T[144];
  if (psi_core(D, Qc, dt, tau, Psi)) return -1;
  orc_calcPhi(D, dt, Phi);
  orc_mm(n, n, n, Psi, Phi, T);
  orc_calcPhi(D, tau, Lambda);
  for (int i = 0; i < n * n; i++) Lambda[i] -= T[i];
  return 0;
}

/* getQc -- gpslam/gp/GPutils.cpp:16-20: Qc = (R^T R)^-1 from the noise model's sqrt information */
int orc_getQc(int D, const double *R, double *Qc) {
  double RtR[36];
  orc_mtm(D, D, D, R, R, RtR);
  return orc_inv(D, RtR, Qc);
}

/* Whitening matrix of the GP prior's noise model:
 * noiseModel::Gaussian::Covariance(calcQ(Qc, dt)) (GaussianProcessPriorPose3.h:46) stores
 * R = chol_upper(Q^-1), R^T R = Q^-1 (GTSAM Gaussian::Information). */
int orc_gp_whitening(int D, const double *Qc, double dt, double *R) {
  int n = 2 * D;
  double Q[144];
  orc_calcQ(D, Qc, dt, Q);
  if (orc_inv(n, Q, R)) return -1;
  /* symmetrise rounding noise before the factorisation */
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      double m = 0.5 * (R[i * n + j] + R[j * n + i]);
      R[i * n + j] = m;
      R[j * n + i] = m;
    }
  return orc_chol_upper(n, R);
}

/* ------------------------------------------------------------------ Pose3utils */

/* rightJacobianRot3 / rightJacobianRot3inv -- Pose3utils.cpp:203-212, :215-224 (same as SO3 derivs) */
void orc_rightJacobianRot3(const double w[3], double J[9]) { orc_rot3_expmap_derivative(w, J); }
void orc_rightJacobianRot3inv(const double w[3], double J[9]) { orc_rot3_logmap_derivative(w, J); }

/* rightJacobianPose3Q -- Pose3utils.cpp:92-113 */
void orc_rightJacobianPose3Q(const double xi[6], double Q[9]) { orc_pose3_Q(xi, Q); }

/* rightJacobianPose3 -- Pose3utils.cpp:182-189 */
void orc_rightJacobianPose3(const double xi[6], double J[36]) { orc_pose3_expmap_derivative(xi, J); }

/* rightJacobianPose3inv -- Pose3utils.cpp:192-200 */
void orc_rightJacobianPose3inv(const double xi[6], double J[36]) { orc_pose3_logmap_derivative_xi(xi, J); }

/* jacobianMethodNumercialDiff(rightJacobianPose3inv, xi, x, dxi) -- Pose3utils.cpp:167-179
 * column i = (F(xi + h e_i) - F(xi - h e_i)) / (2h) * x, default h = 1e-6 (Pose3utils.h:57) */
void orc_jacobianNumDiff_Pose3inv(const double xi[6], const double x[6], double dxi, double Diff[36]) {
  for (int i = 0; i < 6; i++) {
    double xp[6], xn[6], Jp[36], Jn[36], col[6];
    orc_copy(6, xi, xp);
    orc_copy(6, xi, xn);
    xp[i] += dxi;
    orc_rightJacobianPose3inv(xp, Jp);
    xn[i] -= dxi;
    orc_rightJacobianPose3inv(xn, Jn);
    for (int k = 0; k < 36; k++) Jp[k] = (Jp[k] - Jn[k]) / (2.0 * dxi);
    orc_mm(6, 6, 1, Jp, x, col);
    for (int r = 0; r < 6; r++) Diff[r * 6 + i] = col[r];
  }
}

/* getBodyCentricVb / Vs -- Pose3utils.cpp:17-24 */
void orc_getBodyCentricVb(const double p1[12], const double p2[12], double dt, double v[6]) {
  double inv[12], c[12];
  orc_pose3_inverse(p1, inv, NULL);
  orc_pose3_compose(inv, p2, c, NULL, NULL);
  orc_pose3_logmap(c, v, NULL);
  orc_scale(6, 1.0 / dt, v);
}
void orc_getBodyCentricVs(const double p1[12], const double p2[12], double dt, double v[6]) {
  double inv[12], c[12];
  orc_pose3_inverse(p1, inv, NULL);
  orc_pose3_compose(p2, inv, c, NULL, NULL);
  orc_pose3_logmap(c, v, NULL);
  orc_scale(6, 1.0 / dt, v);
}

/* ------------------------------------------------------------------ GP priors */

/* GaussianProcessPriorLinear<D>::evaluateError -- GaussianProcessPriorLinear.h:63-83 */
void orc_gp_prior_linear(int D, const double *p1, const double *v1, const double *p2, const double *v2, double dt,
                         double *e, double *H1, double *H2, double *H3, double *H4) {
  int n = 2 * D;
  for (int i = 0; i < D; i++) {
    e[i] = p1[i] + dt * v1[i] - p2[i];
    e[D + i] = v1[i] - v2[i];
  }
  if (H1) { orc_zero(n * D, H1); for (int i = 0; i < D; i++) H1[i * D + i] = 1.0; }
  if (H2) { orc_zero(n * D, H2); for (int i = 0; i < D; i++) { H2[i * D + i] = dt; H2[(D + i) * D + i] = 1.0; } }
  if (H3) { orc_zero(n * D, H3); for (int i = 0; i < D; i++) H3[i * D + i] = -1.0; }
  if (H4) { orc_zero(n * D, H4); for (int i = 0; i < D; i++) H4[(D + i) * D + i] = -1.0; }
}

/* shared tail of the d=3 priors (Pose2, Rot3): e = [r - v1 dt; v2 - v1],
 * H1 = [J1; 0], H2 = [-dt I; -I], H3 = [J3; 0], H4 = [0; I]
 * -- GaussianProcessPriorPose2.h:76-81, GaussianProcessPriorRot3.h:73-78 */
static void prior3_tail(const double r[3], const double *v1, const double *v2, double dt, const double *J1,
                        const double *J3, double *e, double *H1, double *H2, double *H3, double *H4) {
  for (int i = 0; i < 3; i++) {
    e[i] = r[i] - v1[i] * dt;
    e[3 + i] = v2[i] - v1[i];
  }
  if (H1) { orc_zero(18, H1); orc_blk(3, 3, J1, 3, 0, 0, H1, 3, 0, 0); }
  if (H2) { orc_zero(18, H2); for (int i = 0; i < 3; i++) { H2[i * 3 + i] = -dt; H2[(3 + i) * 3 + i] = -1.0; } }
  if (H3) { orc_zero(18, H3); orc_blk(3, 3, J3, 3, 0, 0, H3, 3, 0, 0); }
  if (H4) { orc_zero(18, H4); for (int i = 0; i < 3; i++) H4[(3 + i) * 3 + i] = 1.0; }
}

/* GaussianProcessPriorPose2::evaluateError -- GaussianProcessPriorPose2.h:58-82 */
void orc_gp_prior_pose2(const double *p1, const double *v1, const double *p2, const double *v2, double dt,
                        double *e, double *H1, double *H2, double *H3, double *H4) {
  double Hinv[9], Hc1[9], Hc2[9], Hlog[9], inv[3], btw[3], r[3], T[9], J1[9], J3[9];
  orc_pose2_inverse(p1, inv, Hinv);
  orc_pose2_compose(inv, p2, btw, Hc1, Hc2);
  orc_pose2_logmap(btw, r, Hlog);
  orc_mm(3, 3, 3, Hlog, Hc1, T);
  orc_mm(3, 3, 3, T, Hinv, J1);
  orc_mm(3, 3, 3, Hlog, Hc2, J3);
  prior3_tail(r, v1, v2, dt, J1, J3, e, H1, H2, H3, H4);
}

/* GaussianProcessPriorRot3::evaluateError -- GaussianProcessPriorRot3.h:58-79 */
void orc_gp_prior_rot3(const double *R1, const double *v1, const double *R2, const double *v2, double dt,
                       double *e, double *H1, double *H2, double *H3, double *H4) {
  double Hinv[9], Hc1[9], Hc2[9], Hlog[9], inv[9], btw[9], r[3], T[9], J1[9], J3[9];
  orc_rot3_inverse(R1, inv, Hinv);
  orc_rot3_compose(inv, R2, btw, Hc1, Hc2);
  orc_rot3_logmap(btw, r, Hlog);
  orc_mm(3, 3, 3, Hlog, Hc1, T);
  orc_mm(3, 3, 3, T, Hinv, J1);
  orc_mm(3, 3, 3, Hlog, Hc2, J3);
  prior3_tail(r, v1, v2, dt, J1, J3, e, H1, H2, H3, H4);
}

/* GaussianProcessPriorPose3::evaluateError -- GaussianProcessPriorPose3.h:60-98 */
void orc_gp_prior_pose3(const double *p1, const double *v1, const double *p2, const double *v2, double dt,
                        double *e, double *H1, double *H2, double *H3, double *H4) {
  double Hinv[36], Hc1[36], Hc2[36], Hlog[36], inv[12], btw[12], r[6], Jinv[36], Jv2[6];
  orc_pose3_inverse(p1, inv, Hinv);
  orc_pose3_compose(inv, p2, btw, Hc1, Hc2);
  orc_pose3_logmap(btw, r, Hlog);                      /* :72 */
  orc_rightJacobianPose3inv(r, Jinv);                  /* :76 */
  if (H1) {                                            /* :79-84 */
    double T[36], J_Ti[36], FD[36], Jdiff[36];
    orc_mm(6, 6, 6, Hlog, Hc1, T);
    orc_mm(6, 6, 6, T, Hinv, J_Ti);
    orc_jacobianNumDiff_Pose3inv(r, v2, 1e-6, FD);
    orc_mm(6, 6, 6, FD, J_Ti, Jdiff);
    orc_blk(6, 6, J_Ti, 6, 0, 0, H1, 6, 0, 0);
    orc_blk(6, 6, Jdiff, 6, 0, 0, H1, 6, 6, 0);
  }
  if (H2) {                                            /* :86 */
    orc_zero(72, H2);
    for (int i = 0; i < 6; i++) { H2[i * 6 + i] = -dt; H2[(6 + i) * 6 + i] = -1.0; }
  }
  if (H3) {                                            /* :88-93 */
    double J_Ti1[36], FD[36], Jdiff[36];
    orc_mm(6, 6, 6, Hlog, Hc2, J_Ti1);
    orc_jacobianNumDiff_Pose3inv(r, v2, 1e-6, FD);
    orc_mm(6, 6, 6, FD, J_Ti1, Jdiff);
    orc_blk(6, 6, J_Ti1, 6, 0, 0, H3, 6, 0, 0);
    orc_blk(6, 6, Jdiff, 6, 0, 0, H3, 6, 6, 0);
  }
  if (H4) {                                            /* :95 */
    orc_zero(72, H4);
    orc_blk(6, 6, Jinv, 6, 0, 0, H4, 6, 6, 0);
  }
  orc_mm(6, 6, 1, Jinv, v2, Jv2);                      /* :97 */
  for (int i = 0; i < 6; i++) {
    e[i] = r[i] - v1[i] * dt;
    e[6 + i] = Jv2[i] - v1[i];
  }
}

/* convertVWtoVb -- gpslam/gp/Pose3utils.cpp:47-64: world-frame translational / rotational velocity (v, w) to the
 * body-frame 6-velocity [R^T w; R^T v].  Rot3::unrotate(p, H1, H2): q = R^T p, H1 = skew(q), H2 = R^T (GTSAM).
 * Hv, Hw: 6x3; Hpose: 6x6; any may be NULL. */
void orc_convertVWtoVb(const double v[3], const double w[3], const double pose[12], double v6[6], double *Hv,
                       double *Hw, double *Hpose) {
  const double *R = pose;
  double qw[3], qv[3];
  for (int j = 0; j < 3; j++) {
    qw[j] = R[0 * 3 + j] * w[0] + R[1 * 3 + j] * w[1] + R[2 * 3 + j] * w[2];
    qv[j] = R[0 * 3 + j] * v[0] + R[1 * 3 + j] * v[1] + R[2 * 3 + j] * v[2];
  }
  for (int j = 0; j < 3; j++) { v6[j] = qw[j]; v6[3 + j] = qv[j]; }
  if (Hv) {                                             /* :58  [0; Hrv], Hrv = R^T */
    orc_zero(18, Hv);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Hv[(3 + i) * 3 + j] = R[j * 3 + i];
  }
  if (Hw) {                                             /* :59  [Hrw; 0] */
    orc_zero(18, Hw);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Hw[i * 3 + j] = R[j * 3 + i];
  }
  if (Hpose) {                                          /* :60  [Hpw 0; Hpv 0], Hp* = skew(q*) */
    orc_zero(36, Hpose);
    const double sw[9] = {0, -qw[2], qw[1], qw[2], 0, -qw[0], -qw[1], qw[0], 0};
    const double sv[9] = {0, -qv[2], qv[1], qv[2], 0, -qv[0], -qv[1], qv[0], 0};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Hpose[i * 6 + j] = sw[i * 3 + j]; Hpose[(3 + i) * 6 + j] = sv[i * 3 + j]; }
  }
}

/* GaussianProcessPriorPose3VW::evaluateError -- GaussianProcessPriorPose3VW.h:62-117.
 * H1, H4: 12x6; H2, H3, H5, H6: 12x3. */
void orc_gp_prior_pose3vw(const double *p1, const double *vel1, const double *omega1, const double *p2,
                          const double *vel2, const double *omega2, double dt, double *e, double *H1, double *H2,
                          double *H3, double *H4, double *H5, double *H6) {
  double Hinv[36], Hc1[36], Hc2[36], Hlog[36], inv[12], btw[12], r[6], Jinv[36], Jv2[6];
  double H1v[18], H1w[18], H2v[18], H2w[18], H1p[36], H2p[36], v1[6], v2[6], Hv1[72], Hv2[72];
  orc_pose3_inverse(p1, inv, Hinv);
  orc_pose3_compose(inv, p2, btw, Hc1, Hc2);
  orc_pose3_logmap(btw, r, Hlog);                      /* :75-78 */
  orc_rightJacobianPose3inv(r, Jinv);                  /* :80 */
  orc_convertVWtoVb(vel1, omega1, p1, v1, H1v, H1w, H1p);   /* :87-88 */
  orc_convertVWtoVb(vel2, omega2, p2, v2, H2v, H2w, H2p);
  orc_zero(72, Hv1);                                   /* :89-90 */
  orc_zero(72, Hv2);
  for (int i = 0; i < 6; i++) { Hv1[i * 6 + i] = -dt; Hv1[(6 + i) * 6 + i] = -1.0; }
  orc_blk(6, 6, Jinv, 6, 0, 0, Hv2, 6, 6, 0);
  if (H1) {                                            /* :97-102 */
    double T[36], J_Ti[36], FD[36], Jdiff[36];
    orc_mm(6, 6, 6, Hlog, Hc1, T);
    orc_mm(6, 6, 6, T, Hinv, J_Ti);
    orc_jacobianNumDiff_Pose3inv(r, v2, 1e-6, FD);
    orc_mm(6, 6, 6, FD, J_Ti, Jdiff);
    for (int i = 0; i < 36; i++) { H1[i] = J_Ti[i] - dt * H1p[i]; H1[36 + i] = Jdiff[i] - H1p[i]; }
  }
  if (H2) orc_mm(12, 6, 3, Hv1, H1v, H2);              /* :104 */
  if (H3) orc_mm(12, 6, 3, Hv1, H1w, H3);              /* :105 */
  if (H4) {                                            /* :107-112 */
    double J_Ti1[36], FD[36], Jdiff[36], JH[36];
    orc_mm(6, 6, 6, Hlog, Hc2, J_Ti1);
    orc_jacobianNumDiff_Pose3inv(r, v2, 1e-6, FD);
    orc_mm(6, 6, 6, FD, J_Ti1, Jdiff);
    orc_mm(6, 6, 6, Jinv, H2p, JH);
    for (int i = 0; i < 36; i++) { H4[i] = J_Ti1[i]; H4[36 + i] = Jdiff[i] + JH[i]; }
  }
  if (H5) orc_mm(12, 6, 3, Hv2, H2v, H5);              /* :114 */
  if (H6) orc_mm(12, 6, 3, Hv2, H2w, H6);              /* :115 */
  orc_mm(6, 6, 1, Jinv, v2, Jv2);                      /* :117 */
  for (int i = 0; i < 6; i++) {
    e[i] = r[i] - v1[i] * dt;
    e[6 + i] = Jv2[i] - v1[i];
  }
}

/* ------------------------------------------------------------------ GP interpolators */

/* GaussianProcessInterpolatorLinear<D>::interpolatePose -- GaussianProcessInterpolatorLinear.h:70-90
 * Lambda/Psi: 2D x 2D from the ctor (:52-63). */
void orc_interp_linear(int D, const double *Lambda, const double *Psi, const double *p1, const double *v1,
                       const double *p2, const double *v2, double *pose, double *H1, double *H2, double *H3,
                       double *H4) {
  int n = 2 * D;
  for (int i = 0; i < D; i++) {
    double s = 0.0;
    for (int j = 0; j < D; j++)
      s += Lambda[i * n + j] * p1[j] + Lambda[i * n + D + j] * v1[j] + Psi[i * n + j] * p2[j] +
           Psi[i * n + D + j] * v2[j];
    pose[i] = s;
  }
  if (H1) orc_blk(D, D, Lambda, n, 0, 0, H1, D, 0, 0);
  if (H2) orc_blk(D, D, Lambda, n, 0, D, H2, D, 0, 0);
  if (H3) orc_blk(D, D, Psi, n, 0, 0, H3, D, 0, 0);
  if (H4) orc_blk(D, D, Psi, n, 0, D, H4, D, 0, 0);
}

/* GaussianProcessInterpolatorLinear<D>::interpolateVelocity -- :106-126 (bottom D rows) */
void orc_interp_linear_velocity(int D, const double *Lambda, const double *Psi, const double *p1, const double *v1,
                                const double *p2, const double *v2, double *vel) {
  int n = 2 * D;
  for (int i = 0; i < D; i++) {
    double s = 0.0;
    for (int j = 0; j < D; j++)
      s += Lambda[(D + i) * n + j] * p1[j] + Lambda[(D + i) * n + D + j] * v1[j] + Psi[(D + i) * n + j] * p2[j] +
           Psi[(D + i) * n + D + j] * v2[j];
    vel[i] = s;
  }
}

/* xi(3) = Lambda[0:3, :] * [0; v1] + Psi[0:3, :] * [r; v2'] for the d=3 manifolds */
static void interp3_xi(const double *Lambda, const double *Psi, const double *v1, const double *r, const double *v2,
                       double xi[3]) {
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
    for (int j = 0; j < 3; j++) s += Lambda[i * 6 + 3 + j] * v1[j] + Psi[i * 6 + j] * r[j] + Psi[i * 6 + 3 + j] * v2[j];
    xi[i] = s;
  }
}

/* shared Jacobian tail of the d=3 interpolators:
 *   Hexpr1 = Hcomp22 * Hexp;  H1 = Hcomp21 + Hexpr1*Psi11*J1;  H2 = Hexpr1*Lambda12;
 *   H3 = Hexpr1*Psi11*J3;     H4 = Hexpr1*Psi12
 * -- GaussianProcessInterpolatorPose2.h:79-83, GaussianProcessInterpolatorRot3.h:76-80 */
static void interp3_jac(const double *Lambda, const double *Psi, const double *Hc21, const double *Hc22,
                        const double *Hexp, const double *J1, const double *J3, double *H1, double *H2, double *H3,
                        double *H4) {
  double He[9], L12[9], P11[9], P12[9], T[9], T2[9];
  orc_mm(3, 3, 3, Hc22, Hexp, He);
  orc_blk(3, 3, Lambda, 6, 0, 3, L12, 3, 0, 0);
  orc_blk(3, 3, Psi, 6, 0, 0, P11, 3, 0, 0);
  orc_blk(3, 3, Psi, 6, 0, 3, P12, 3, 0, 0);
  orc_mm(3, 3, 3, He, P11, T);
  if (H1) { orc_mm(3, 3, 3, T, J1, T2); for (int i = 0; i < 9; i++) H1[i] = Hc21[i] + T2[i]; }
  if (H2) orc_mm(3, 3, 3, He, L12, H2);
  if (H3) orc_mm(3, 3, 3, T, J3, H3);
  if (H4) orc_mm(3, 3, 3, He, P12, H4);
}

/* GaussianProcessInterpolatorPose2::interpolatePose -- GaussianProcessInterpolatorPose2.h:56-89 */
void orc_interp_pose2(const double *Lambda, const double *Psi, const double *p1, const double *v1, const double *p2,
                      const double *v2, double *pose, double *H1, double *H2, double *H3, double *H4) {
  double Hinv[9], Hc11[9], Hc12[9], Hlog[9], inv[3], btw[3], r[3], xi[3], ex[3], Hexp[9], Hc21[9], Hc22[9];
  orc_pose2_inverse(p1, inv, Hinv);
  orc_pose2_compose(inv, p2, btw, Hc11, Hc12);
  orc_pose2_logmap(btw, r, Hlog);
  interp3_xi(Lambda, Psi, v1, r, v2, xi);
  orc_pose2_expmap(xi, ex, Hexp);
  orc_pose2_compose(p1, ex, pose, Hc21, Hc22);
  if (H1 || H2 || H3 || H4) {
    double T[9], J1[9], J3[9];
    orc_mm(3, 3, 3, Hlog, Hc11, T);
    orc_mm(3, 3, 3, T, Hinv, J1);
    orc_mm(3, 3, 3, Hlog, Hc12, J3);
    interp3_jac(Lambda, Psi, Hc21, Hc22, Hexp, J1, J3, H1, H2, H3, H4);
  }
}

/* GaussianProcessInterpolatorRot3::interpolatePose -- GaussianProcessInterpolatorRot3.h:56-86 */
void orc_interp_rot3(const double *Lambda, const double *Psi, const double *R1, const double *v1, const double *R2,
                     const double *v2, double *rot, double *H1, double *H2, double *H3, double *H4) {
  double Hinv[9], Hc11[9], Hc12[9], Hlog[9], inv[9], btw[9], r[3], xi[3], ex[9], Hexp[9], Hc21[9], Hc22[9];
  orc_rot3_inverse(R1, inv, Hinv);
  orc_rot3_compose(inv, R2, btw, Hc11, Hc12);
  orc_rot3_logmap(btw, r, Hlog);
  interp3_xi(Lambda, Psi, v1, r, v2, xi);
  orc_rot3_expmap(xi, ex, Hexp);
  orc_rot3_compose(R1, ex, rot, Hc21, Hc22);
  if (H1 || H2 || H3 || H4) {
    double T[9], J1[9], J3[9];
    orc_mm(3, 3, 3, Hlog, Hc11, T);
    orc_mm(3, 3, 3, T, Hinv, J1);
    orc_mm(3, 3, 3, Hlog, Hc12, J3);
    interp3_jac(Lambda, Psi, Hc21, Hc22, Hexp, J1, J3, H1, H2, H3, H4);
  }
}

/* GaussianProcessInterpolatorPose3::interpolatePose -- GaussianProcessInterpolatorPose3.h:57-105
 * Lambda/Psi are 12x12 (ctor :43-50). */
void orc_interp_pose3(const double *Lambda, const double *Psi, const double *p1, const double *v1, const double *p2,
                      const double *v2, double *pose, double *H1, double *H2, double *H3, double *H4) {
  double Hinv[36], Hc11[36], Hc12[36], Hlog[36], inv[12], btw[12], r[6], Jinv[36], Jv2[6];
  double r1[12], r2[12], xi[6], ex[12], Hexp[36], Hc21[36], Hc22[36];
  orc_pose3_inverse(p1, inv, Hinv);
  orc_pose3_compose(inv, p2, btw, Hc11, Hc12);
  orc_pose3_logmap(btw, r, Hlog);                      /* :68 */
  orc_rightJacobianPose3inv(r, Jinv);                  /* :72 */
  orc_mm(6, 6, 1, Jinv, v2, Jv2);
  for (int i = 0; i < 6; i++) { r1[i] = 0.0; r1[6 + i] = v1[i]; r2[i] = r[i]; r2[6 + i] = Jv2[i]; }   /* :58,:73 */
  for (int i = 0; i < 6; i++) {
    double s = 0.0;
    for (int j = 0; j < 12; j++) s += Lambda[i * 12 + j] * r1[j] + Psi[i * 12 + j] * r2[j];
    xi[i] = s;
  }
  orc_pose3_expmap(xi, ex, Hexp);
  orc_pose3_compose(p1, ex, pose, Hc21, Hc22);         /* :79 */
  if (H1 || H2 || H3 || H4) {
    double He[36], Psi1[72], FD[36];
    orc_mm(6, 6, 6, Hc22, Hexp, He);                   /* :80 */
    orc_blk(6, 12, Psi, 12, 0, 0, Psi1, 12, 0, 0);
    orc_jacobianNumDiff_Pose3inv(r, v2, 1e-6, FD);
    if (H1) {                                          /* :82-87 */
      double T[36], tmp[36], FDtmp[36], dr2[72], PD[36], HPD[36];
      orc_mm(6, 6, 6, Hlog, Hc11, T);
      orc_mm(6, 6, 6, T, Hinv, tmp);
      orc_mm(6, 6, 6, FD, tmp, FDtmp);
      orc_blk(6, 6, tmp, 6, 0, 0, dr2, 6, 0, 0);
      orc_blk(6, 6, FDtmp, 6, 0, 0, dr2, 6, 6, 0);
      orc_mm(6, 12, 6, Psi1, dr2, PD);
      orc_mm(6, 6, 6, He, PD, HPD);
      for (int i = 0; i < 36; i++) H1[i] = Hc21[i] + HPD[i];
    }
    if (H2) {                                          /* :89 */
      double L12[36];
      orc_blk(6, 6, Lambda, 12, 0, 6, L12, 6, 0, 0);
      orc_mm(6, 6, 6, He, L12, H2);
    }
    if (H3) {                                          /* :91-96 */
      double tmp[36], FDtmp[36], dr2[72], PD[36];
      orc_mm(6, 6, 6, Hlog, Hc12, tmp);
      orc_mm(6, 6, 6, FD, tmp, FDtmp);
      orc_blk(6, 6, tmp, 6, 0, 0, dr2, 6, 0, 0);
      orc_blk(6, 6, FDtmp, 6, 0, 0, dr2, 6, 6, 0);
      orc_mm(6, 12, 6, Psi1, dr2, PD);
      orc_mm(6, 6, 6, He, PD, H3);
    }
    if (H4) {                                          /* :98 */
      double P12[36], T[36];
      orc_blk(6, 6, Psi, 12, 0, 6, P12, 6, 0, 0);
      orc_mm(6, 6, 6, He, P12, T);
      orc_mm(6, 6, 6, T, Jinv, H4);
    }
  }
}

/* GaussianProcessInterpolatorPose3VW::interpolatePose -- GaussianProcessInterpolatorPose3VW.h:58-124.
 * H1, H4: 6x6; H2, H3, H5, H6: 6x3. */
void orc_interp_pose3vw(const double *Lambda, const double *Psi, const double *p1, const double *v1,
                        const double *omega1, const double *p2, const double *v2, const double *omega2, double *pose,
                        double *H1, double *H2, double *H3, double *H4, double *H5, double *H6) {
  double Hinv[36], Hc11[36], Hc12[36], Hlog[36], inv[12], btw[12], r[6], Jinv[36], Jv2[6];
  double H1v[18], H1w[18], H2v[18], H2w[18], H1p[36], H2p[36], vel1[6], vel2[6];
  double r1[12], r2[12], xi[6], ex[12], Hexp[36], Hc21[36], Hc22[36];
  const int want = H1 || H2 || H3 || H4 || H5 || H6;
  orc_pose3_inverse(p1, inv, Hinv);
  orc_pose3_compose(inv, p2, btw, Hc11, Hc12);
  orc_pose3_logmap(btw, r, Hlog);                      /* :68-71 */
  orc_rightJacobianPose3inv(r, Jinv);                  /* :73 */
  orc_convertVWtoVb(v1, omega1, p1, vel1, H1v, H1w, H1p);   /* :79-80 */
  orc_convertVWtoVb(v2, omega2, p2, vel2, H2v, H2w, H2p);
  orc_mm(6, 6, 1, Jinv, vel2, Jv2);
  for (int i = 0; i < 6; i++) { r1[i] = 0.0; r1[6 + i] = vel1[i]; r2[i] = r[i]; r2[6 + i] = Jv2[i]; }   /* :85-86 */
  for (int i = 0; i < 6; i++) {
    double s = 0.0;
    for (int j = 0; j < 12; j++) s += Lambda[i * 12 + j] * r1[j] + Psi[i * 12 + j] * r2[j];
    xi[i] = s;
  }
  orc_pose3_expmap(xi, ex, Hexp);
  orc_pose3_compose(p1, ex, pose, Hc21, Hc22);         /* :92 */
  if (want) {
    double He[36], L12[36], P12[36], Hvel1[36], Hvel2[36], T[36], Psi1[72], FD[36];
    orc_mm(6, 6, 6, Hc22, Hexp, He);                   /* :93 */
    orc_blk(6, 6, Lambda, 12, 0, 6, L12, 6, 0, 0);
    orc_blk(6, 6, Psi, 12, 0, 6, P12, 6, 0, 0);
    orc_mm(6, 6, 6, He, L12, Hvel1);                   /* :94 */
    orc_mm(6, 6, 6, He, P12, T);
    orc_mm(6, 6, 6, T, Jinv, Hvel2);                   /* :95 */
    orc_blk(6, 12, Psi, 12, 0, 0, Psi1, 12, 0, 0);
    orc_jacobianNumDiff_Pose3inv(r, vel2, 1e-6, FD);
    if (H1) {                                          /* :97-102 */
      double A[36], tmp[36], FDtmp[36], dr2[72], PD[36], HPD[36], HH[36];
      orc_mm(6, 6, 6, Hlog, Hc11, A);
      orc_mm(6, 6, 6, A, Hinv, tmp);
      orc_mm(6, 6, 6, FD, tmp, FDtmp);
      orc_blk(6, 6, tmp, 6, 0, 0, dr2, 6, 0, 0);
      orc_blk(6, 6, FDtmp, 6, 0, 0, dr2, 6, 6, 0);
      orc_mm(6, 12, 6, Psi1, dr2, PD);
      orc_mm(6, 6, 6, He, PD, HPD);
      orc_mm(6, 6, 6, Hvel1, H1p, HH);
      for (int i = 0; i < 36; i++) H1[i] = Hc21[i] + HPD[i] + HH[i];
    }
    if (H2) orc_mm(6, 6, 3, Hvel1, H1v, H2);           /* :104 */
    if (H3) orc_mm(6, 6, 3, Hvel1, H1w, H3);           /* :105 */
    if (H4) {                                          /* :107-112 */
      double tmp[36], FDtmp[36], dr2[72], PD[36], HPD[36], HH[36];
      orc_mm(6, 6, 6, Hlog, Hc12, tmp);
      orc_mm(6, 6, 6, FD, tmp, FDtmp);
      orc_blk(6, 6, tmp, 6, 0, 0, dr2, 6, 0, 0);
      orc_blk(6, 6, FDtmp, 6, 0, 0, dr2, 6, 6, 0);
      orc_mm(6, 12, 6, Psi1, dr2, PD);
      orc_mm(6, 6, 6, He, PD, HPD);
      orc_mm(6, 6, 6, Hvel2, H2p, HH);
      for (int i = 0; i < 36; i++) H4[i] = HPD[i] + HH[i];
    }
    if (H5) orc_mm(6, 6, 3, Hvel2, H2v, H5);           /* :114 */
    if (H6) orc_mm(6, 6, 3, Hvel2, H2w, H6);           /* :115 */
  }
}

/* The VW factors with the two 3-vectors of a state stored as one 6-vector s = [v; w] and their Jacobians packed
 * side by side ([H_v | H_w], rows x 6): the call shape of the body-velocity versions, used by the chain problem. */
void orc_gp_prior_pose3vw_packed(const double *p1, const double *s1, const double *p2, const double *s2, double dt,
                                 double *e, double *H1, double *H2, double *H3, double *H4) {
  double Hv1[36], Hw1[36], Hv2[36], Hw2[36];
  const int jac = H1 || H2 || H3 || H4;
  orc_gp_prior_pose3vw(p1, s1, s1 + 3, p2, s2, s2 + 3, dt, e, jac ? H1 : NULL, jac ? Hv1 : NULL, jac ? Hw1 : NULL,
                       jac ? H3 : NULL, jac ? Hv2 : NULL, jac ? Hw2 : NULL);
  if (jac) {
    for (int i = 0; i < 12; i++)
      for (int j = 0; j < 3; j++) {
        H2[i * 6 + j] = Hv1[i * 3 + j]; H2[i * 6 + 3 + j] = Hw1[i * 3 + j];
        H4[i * 6 + j] = Hv2[i * 3 + j]; H4[i * 6 + 3 + j] = Hw2[i * 3 + j];
      }
  }
}
void orc_interp_pose3vw_packed(const double *Lambda, const double *Psi, const double *p1, const double *s1,
                               const double *p2, const double *s2, double *pose, double *H1, double *H2, double *H3,
                               double *H4) {
  double Hv1[18], Hw1[18], Hv2[18], Hw2[18];
  const int jac = H1 || H2 || H3 || H4;
  orc_interp_pose3vw(Lambda, Psi, p1, s1, s1 + 3, p2, s2, s2 + 3, pose, jac ? H1 : NULL, jac ? Hv1 : NULL,
                     jac ? Hw1 : NULL, jac ? H3 : NULL, jac ? Hv2 : NULL, jac ? Hw2 : NULL);
  if (jac) {
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 3; j++) {
        H2[i * 6 + j] = Hv1[i * 3 + j]; H2[i * 6 + 3 + j] = Hw1[i * 3 + j];
        H4[i * 6 + j] = Hv2[i * 3 + j]; H4[i * 6 + 3 + j] = Hw2[i * 3 + j];
      }
  }
}
