/*
 * orc_factors.c -- CPU restatement of gpslam/slam measurement factors plus the GTSAM
 * PriorFactor / BetweenFactor the reference's graphs use beside them.
 *
 * TEST INFRASTRUCTURE ONLY (see gpslam_oracle.h).  Paths relative to /root/reference.
 */
#include "gpslam_oracle.h"
#include "orc_math.h"

/* H_k (rows x d) = Hpose (rows x d) * Hint_k (d x d):
 * GaussianProcessInterpolator*::updatePoseJacobians, e.g. GaussianProcessInterpolatorPose3.h:108-116 */
static void update_pose_jacobians(int rows, int d, const double *Hpose, const double *Hint1, const double *Hint2,
                                  const double *Hint3, const double *Hint4, double *H1, double *H2, double *H3,
                                  double *H4) {
  if (H1) orc_mm(rows, d, d, Hpose, Hint1, H1);
  if (H2) orc_mm(rows, d, d, Hpose, Hint2, H2);
  if (H3) orc_mm(rows, d, d, Hpose, Hint3, H3);
  if (H4) orc_mm(rows, d, d, Hpose, Hint4, H4);
}

/* GPInterpolatedRangeFactorPose2::evaluateError -- GPInterpolatedRangeFactorPose2.h:64-98
 * sensor: optional body_P_sensor (Pose2, 3 doubles) or NULL. */
double orc_interp_range_pose2(const double *Lambda, const double *Psi, double measured, const double *sensor,
                              const double *p1, const double *v1, const double *p2, const double *v2,
                              const double *point, double *H1, double *H2, double *H3, double *H4, double *H5) {
  double Hi1[9], Hi2[9], Hi3[9], Hi4[9], pose[3], Hpose[3], hx;
  int want = H1 || H2 || H3 || H4;
  orc_interp_pose2(Lambda, Psi, p1, v1, p2, v2, pose, want ? Hi1 : NULL, want ? Hi2 : NULL, want ? Hi3 : NULL,
                   want ? Hi4 : NULL);
  if (sensor) {
    double H0[9], sp[3], Hr[3];
    orc_pose2_compose(pose, sensor, sp, H0, NULL);
    hx = orc_pose2_range(sp, point, Hr, H5);
    orc_mm(1, 3, 3, Hr, H0, Hpose);
  } else {
    hx = orc_pose2_range(pose, point, Hpose, H5);
  }
  if (want) update_pose_jacobians(1, 3, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
  return hx - measured;
}

/* GPInterpolatedRangeFactorPose3::evaluateError -- GPInterpolatedRangeFactorPose3.h:64-98 */
double orc_interp_range_pose3(const double *Lambda, const double *Psi, double measured, const double *sensor,
                              const double *p1, const double *v1, const double *p2, const double *v2,
                              const double *point, double *H1, double *H2, double *H3, double *H4, double *H5) {
  double Hi1[36], Hi2[36], Hi3[36], Hi4[36], pose[12], Hpose[6], hx;
  int want = H1 || H2 || H3 || H4;
  orc_interp_pose3(Lambda, Psi, p1, v1, p2, v2, pose, want ? Hi1 : NULL, want ? Hi2 : NULL, want ? Hi3 : NULL,
                   want ? Hi4 : NULL);
  if (sensor) {
    double H0[36], sp[12], Hr[6];
    orc_pose3_compose(pose, sensor, sp, H0, NULL);
    hx = orc_pose3_range(sp, point, Hr, H5);
    orc_mm(1, 6, 6, Hr, H0, Hpose);
  } else {
    hx = orc_pose3_range(pose, point, Hpose, H5);
  }
  if (want) update_pose_jacobians(1, 6, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
  return hx - measured;
}

/* GPInterpolatedRangeFactor2DLinear::evaluateError -- GPInterpolatedRangeFactor2DLinear.h:60-88
 * pose vectors are [x, y, theta]; Lambda/Psi are 6x6. */
double orc_interp_range_2dlinear(const double *Lambda, const double *Psi, double measured, const double *p1,
                                 const double *v1, const double *p2, const double *v2, const double *point,
                                 double *H1, double *H2, double *H3, double *H4, double *H5) {
  double Hi1[9], Hi2[9], Hi3[9], Hi4[9], pose[3];
  orc_interp_linear(3, Lambda, Psi, p1, v1, p2, v2, pose, Hi1, Hi2, Hi3, Hi4);
  double d[2] = {point[0] - pose[0], point[1] - pose[1]};
  double r = sqrt(d[0] * d[0] + d[1] * d[1]);
  double H[2] = {d[0] / r, d[1] / r};
  if (H1 || H2 || H3 || H4) {
    double Hpose[3] = {-H[0], -H[1], 0.0};                       /* :82 */
    update_pose_jacobians(1, 3, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
  }
  if (H5) { H5[0] = H[0]; H5[1] = H[1]; }                         /* :86 */
  return r - measured;
}

/* GPInterpolatedAttitudeFactorRot3::evaluateError -- GPInterpolatedAttitudeFactorRot3.h:61-83 */
void orc_interp_attitude_rot3(const double *Lambda, const double *Psi, const double *nZ, const double *bRef,
                              const double *R1, const double *v1, const double *R2, const double *v2, double *e,
                              double *H1, double *H2, double *H3, double *H4) {
  double Hi1[9], Hi2[9], Hi3[9], Hi4[9], rot[9], Hrot[6];
  int want = H1 || H2 || H3 || H4;
  orc_interp_rot3(Lambda, Psi, R1, v1, R2, v2, rot, want ? Hi1 : NULL, want ? Hi2 : NULL, want ? Hi3 : NULL,
                  want ? Hi4 : NULL);
  orc_attitude_error(rot, nZ, bRef, e, want ? Hrot : NULL);
  if (want) update_pose_jacobians(2, 3, Hrot, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
}

/* GPInterpolatedGPSFactorPose3::evaluateError -- GPInterpolatedGPSFactorPose3.h:66-95 */
void orc_interp_gps_pose3(const double *Lambda, const double *Psi, const double *measured, const double *sensor,
                          const double *p1, const double *v1, const double *p2, const double *v2, double *e,
                          double *H1, double *H2, double *H3, double *H4) {
  double Hi1[36], Hi2[36], Hi3[36], Hi4[36], pose[12], Hpose[18], t[3];
  int want = H1 || H2 || H3 || H4;
  orc_interp_pose3(Lambda, Psi, p1, v1, p2, v2, pose, want ? Hi1 : NULL, want ? Hi2 : NULL, want ? Hi3 : NULL,
                   want ? Hi4 : NULL);
  if (sensor) {
    double H0[36], sp[12], Ht[18];
    orc_pose3_compose(pose, sensor, sp, H0, NULL);
    orc_pose3_translation(sp, t, Ht);
    orc_mm(3, 6, 6, Ht, H0, Hpose);
  } else {
    orc_pose3_translation(pose, t, Hpose);
  }
  for (int i = 0; i < 3; i++) e[i] = t[i] - measured[i];
  if (want) update_pose_jacobians(3, 6, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
}

/* GPInterpolatedGPSFactorPose3VW::evaluateError -- GPInterpolatedGPSFactorPose3VW.h:73-103 (state velocities packed
 * as s = [v; w], Jacobians [H_v | H_w]) */
void orc_interp_gps_pose3vw(const double *Lambda, const double *Psi, const double *measured, const double *sensor,
                            const double *p1, const double *s1, const double *p2, const double *s2, double *e,
                            double *H1, double *H2, double *H3, double *H4) {
  double Hi1[36], Hi2[36], Hi3[36], Hi4[36], pose[12], Hpose[18], t[3];
  int want = H1 || H2 || H3 || H4;
  orc_interp_pose3vw_packed(Lambda, Psi, p1, s1, p2, s2, pose, want ? Hi1 : NULL, want ? Hi2 : NULL,
                            want ? Hi3 : NULL, want ? Hi4 : NULL);
  if (sensor) {
    double H0[36], sp[12], Ht[18];
    orc_pose3_compose(pose, sensor, sp, H0, NULL);
    orc_pose3_translation(sp, t, Ht);
    orc_mm(3, 6, 6, Ht, H0, Hpose);
  } else {
    orc_pose3_translation(pose, t, Hpose);
  }
  for (int i = 0; i < 3; i++) e[i] = t[i] - measured[i];
  if (want) update_pose_jacobians(3, 6, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
}

/* Range to a landmark from the VW-interpolated pose: GPInterpolatedRangeFactorPose3.h:64-98 with the interpolator
 * of GaussianProcessInterpolatorPose3VW.h (the reference ships no such class; it is the same composition). */
double orc_interp_range_pose3vw(const double *Lambda, const double *Psi, double measured, const double *sensor,
                                const double *p1, const double *s1, const double *p2, const double *s2,
                                const double *point, double *H1, double *H2, double *H3, double *H4, double *H5) {
  double Hi1[36], Hi2[36], Hi3[36], Hi4[36], pose[12], Hpose[6], hx;
  int want = H1 || H2 || H3 || H4;
  orc_interp_pose3vw_packed(Lambda, Psi, p1, s1, p2, s2, pose, want ? Hi1 : NULL, want ? Hi2 : NULL,
                            want ? Hi3 : NULL, want ? Hi4 : NULL);
  if (sensor) {
    double H0[36], sp[12], Hr[6];
    orc_pose3_compose(pose, sensor, sp, H0, NULL);
    hx = orc_pose3_range(sp, point, Hr, H5);
    orc_mm(1, 6, 6, Hr, H0, Hpose);
  } else {
    hx = orc_pose3_range(pose, point, Hpose, H5);
  }
  if (want) update_pose_jacobians(1, 6, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
  return hx - measured;
}

/* PinholeCamera<CALIBRATION>::project(point, Dpose, Dpoint) (GTSAM CalibratedCamera.cpp, Cal3_S2.cpp, Cal3DS2_Base.cpp; call
 * site GPInterpolatedProjectionFactorPose3.h:106-118, a template over CALIBRATION, :29):  q = R^T (p - t) must have q.z > 0
 * (else CheiralityException), pn = (q.x, q.y) / q.z, then CALIBRATION::uncalibrate:
 *   Cal3_S2   uv = (fx pn.x + s pn.y + u0, fy pn.y + v0),                                  K = [fx, fy, s, u0, v0]
 *   Cal3DS2   (x, y) = pn, rr = x^2 + y^2, g = 1 + k1 rr + k2 rr^2,                        K9 = [fx, fy, s, u0, v0, k1, k2, p1, p2]
 *             pd = (g x + 2 p1 x y + p2 (rr + 2 x^2), g y + 2 p2 x y + p1 (rr + 2 y^2)), uv = Cal3_S2(pd)
 *             (radial-tangential distortion; Cal3DS2_Base::uncalibrate, recalled from GTSAM 4.0 -- no reference test uses it:
 *              parity unpinned, held here to central differences and to Cal3_S2 at zero distortion)
 * Dpn_pose = [uv', -1-u^2, v, -d, 0, d u; 1+v^2, -uv', -u, 0, -d, d v] with (u, v) = pn, d = 1/q.z (PinholeBase::Dpose),
 * Dpn_point = d [Rt_0 - u Rt_2; Rt_1 - v Rt_2], Rt = R^T (PinholeBase::Dpoint), Duv_pn = [fx, s; 0, fy] D(pd)/D(pn).
 * Returns 0, or 1 on a cheirality violation (outputs untouched). */
int orc_pinhole_project_ds2(const double cam[12], const double K[9], const double point[3], double uv[2], double *Dpose,
                            double *Dpoint) {
  double q[3];
  orc_pose3_transform_to(cam, point, q, NULL, NULL);
  if (q[2] <= 0.0) return 1;
  const double d = 1.0 / q[2], u = q[0] * d, v = q[1] * d;
  const double k1 = K[5], k2 = K[6], p1 = K[7], p2 = K[8];
  const double xx = u * u, yy = v * v, xy = u * v, rr = xx + yy;
  const double g = 1.0 + k1 * rr + k2 * rr * rr;
  const double pdx = g * u + 2.0 * p1 * xy + p2 * (rr + 2.0 * xx);
  const double pdy = g * v + 2.0 * p2 * xy + p1 * (rr + 2.0 * yy);
  uv[0] = K[0] * pdx + K[2] * pdy + K[3];
  uv[1] = K[1] * pdy + K[4];
  const double dgx = 2.0 * k1 * u + 4.0 * k2 * rr * u, dgy = 2.0 * k1 * v + 4.0 * k2 * rr * v;
  const double D2[4] = {g + u * dgx + 2.0 * p1 * v + 6.0 * p2 * u, u * dgy + 2.0 * p1 * u + 2.0 * p2 * v,
                        v * dgx + 2.0 * p2 * v + 2.0 * p1 * u, g + v * dgy + 2.0 * p2 * u + 6.0 * p1 * v};
  const double Kk[4] = {K[0], K[2], 0.0, K[1]};
  double Dk[4];
  orc_mm(2, 2, 2, Kk, D2, Dk);
  if (Dpose) {
    const double Dp[12] = {u * v, -1.0 - u * u, v, -d, 0.0, d * u,
                           1.0 + v * v, -u * v, -u, 0.0, -d, d * v};
    orc_mm(2, 2, 6, Dk, Dp, Dpose);
  }
  if (Dpoint) {
    double Dp[6];
    for (int j = 0; j < 3; j++) {   /* Rt(i, j) = R(j, i) = cam[3 * j + i] */
      Dp[j] = d * (cam[3 * j + 0] - u * cam[3 * j + 2]);
      Dp[3 + j] = d * (cam[3 * j + 1] - v * cam[3 * j + 2]);
    }
    orc_mm(2, 2, 3, Dk, Dp, Dpoint);
  }
  return 0;
}
int orc_pinhole_project(const double cam[12], const double K[5], const double point[3], double uv[2], double *Dpose,
                        double *Dpoint) {
  const double K9[9] = {K[0], K[1], K[2], K[3], K[4], 0.0, 0.0, 0.0, 0.0};
  return orc_pinhole_project_ds2(cam, K9, point, uv, Dpose, Dpoint);
}

/* GPInterpolatedProjectionFactorPose3<Cal3_S2>::evaluateError -- GPInterpolatedProjectionFactorPose3.h:82-139.
 * A landmark behind the camera does not throw (throwCheirality = false, the default): the error is 2 fx in both
 * components and every Jacobian is zero (:122-138).  H1..H4: 2x6, H5: 2x3.  Returns 1 in that case, else 0. */
int orc_interp_projection_pose3_ds2(const double *Lambda, const double *Psi, const double measured[2], const double K[9],
                                    const double *sensor, const double *p1, const double *v1, const double *p2,
                                    const double *v2, const double *point, double *e, double *H1, double *H2, double *H3,
                                    double *H4, double *H5);
int orc_interp_projection_pose3(const double *Lambda, const double *Psi, const double measured[2], const double K[5],
                                const double *sensor, const double *p1, const double *v1, const double *p2,
                                const double *v2, const double *point, double *e, double *H1, double *H2, double *H3,
                                double *H4, double *H5) {
  const double K9[9] = {K[0], K[1], K[2], K[3], K[4], 0.0, 0.0, 0.0, 0.0};
  return orc_interp_projection_pose3_ds2(Lambda, Psi, measured, K9, sensor, p1, v1, p2, v2, point, e, H1, H2, H3, H4, H5);
}
/* the same for GPInterpolatedProjectionFactorPose3<Cal3DS2> (K9 = Cal3_S2's five, then k1, k2, p1, p2) */
int orc_interp_projection_pose3_ds2(const double *Lambda, const double *Psi, const double measured[2], const double K[9],
                                    const double *sensor, const double *p1, const double *v1, const double *p2,
                                    const double *v2, const double *point, double *e, double *H1, double *H2, double *H3,
                                    double *H4, double *H5) {
  double Hi1[36], Hi2[36], Hi3[36], Hi4[36], pose[12], cam[12], H0[36], Hcam[12], Hpose[12], uv[2];
  int want = H1 || H2 || H3 || H4;
  orc_interp_pose3(Lambda, Psi, p1, v1, p2, v2, pose, want ? Hi1 : NULL, want ? Hi2 : NULL, want ? Hi3 : NULL,
                   want ? Hi4 : NULL);
  if (sensor) orc_pose3_compose(pose, sensor, cam, H0, NULL);
  else orc_copy(12, pose, cam);
  if (orc_pinhole_project_ds2(cam, K, point, uv, Hcam, H5)) {
    e[0] = e[1] = 2.0 * K[0];
    if (H1) orc_zero(12, H1);
    if (H2) orc_zero(12, H2);
    if (H3) orc_zero(12, H3);
    if (H4) orc_zero(12, H4);
    if (H5) orc_zero(6, H5);
    return 1;
  }
  e[0] = uv[0] - measured[0];
  e[1] = uv[1] - measured[1];
  if (want) {
    if (sensor) orc_mm(2, 6, 6, Hcam, H0, Hpose);
    else orc_copy(12, Hcam, Hpose);
    update_pose_jacobians(2, 6, Hpose, Hi1, Hi2, Hi3, Hi4, H1, H2, H3, H4);
  }
  return 0;
}

/* RangeFactor2DLinear::evaluateError -- RangeFactor2DLinear.h:43-56 */
double orc_range_2dlinear(double measured, const double *pose, const double *point, double *H1, double *H2) {
  double d[2] = {point[0] - pose[0], point[1] - pose[1]};
  double r = sqrt(d[0] * d[0] + d[1] * d[1]);
  double H[2] = {d[0] / r, d[1] / r};
  if (H1) { H1[0] = -H[0]; H1[1] = -H[1]; H1[2] = 0.0; }
  if (H2) { H2[0] = H[0]; H2[1] = H[1]; }
  return r - measured;
}

/* RangeFactorPose2 = gtsam::RangeFactor<Pose2,Point2> -- RangeFactorPose2.h:15 */
double orc_range_pose2(double measured, const double *pose, const double *point, double *H1, double *H2) {
  return orc_pose2_range(pose, point, H1, H2) - measured;
}

/* RangeBearingFactor2DLinear::evaluateError -- RangeBearingFactor2DLinear.h:47-84
 * e = [Rot2::Logmap(bearing.between(atan2(rel))), |point - t| - range] */
void orc_range_bearing_2dlinear(double bearing, double range, const double *pose, const double *point, double *e,
                                double *H1, double *H2) {
  double rel[2];
  orc_pose2_transform_to(pose, point, rel);
  double expect_theta = atan2(rel[1], rel[0]);
  double d[2] = {point[0] - pose[0], point[1] - pose[1]};
  double expect_d = sqrt(d[0] * d[0] + d[1] * d[1]);
  double Hnorm[2] = {d[0] / expect_d, d[1] / expect_d};
  if (H1 || H2) {
    double tmp[2];
    if (expect_d > 1e-5) {                                        /* :62 */
      double d2 = expect_d * expect_d;
      tmp[0] = -rel[1] / d2;
      tmp[1] = rel[0] / d2;
    } else {
      tmp[0] = 0.0;
      tmp[1] = 0.0;
    }
    double c = cos(pose[2]), s = sin(pose[2]);
    double Rt[4] = {c, s, -s, c};                                 /* pose2.r().transpose() */
    if (H1) {
      double M[6] = {-Rt[0], -Rt[1], rel[1], -Rt[2], -Rt[3], -rel[0]};   /* [-R^T, t], t = (rel.y, -rel.x) */
      orc_mm(1, 2, 3, tmp, M, H1);                                /* H11 */
      H1[3] = -Hnorm[0]; H1[4] = -Hnorm[1]; H1[5] = 0.0;          /* H21 */
    }
    if (H2) {
      orc_mm(1, 2, 2, tmp, Rt, H2);                               /* H12 */
      H2[2] = Hnorm[0]; H2[3] = Hnorm[1];                         /* H22 */
    }
  }
  double diff = expect_theta - bearing;
  e[0] = atan2(sin(diff), cos(diff));
  e[1] = expect_d - range;
}

/* OdometryFactor2DLinear::evaluateError -- OdometryFactor2DLinear.h:50-75 */
void orc_odometry_2dlinear(const double *measured, const double *pose1, const double *pose2, double *e, double *H1,
                           double *H2) {
  double dv[3] = {pose2[0] - pose1[0], pose2[1] - pose1[1], pose2[2] - pose1[2]};
  double c = cos(pose1[2]), s = sin(pose1[2]);
  /* Rot2(theta).unrotate(p): q = R^T p; Hrot = (q.y, -q.x)^T; Hp = R^T */
  double q[2] = {c * dv[0] + s * dv[1], -s * dv[0] + c * dv[1]};
  if (H1) {
    double A[9] = {-c, -s, q[1], s, -c, -q[0], 0.0, 0.0, -1.0};
    orc_copy(9, A, H1);
  }
  if (H2) {
    double A[9] = {c, s, 0.0, -s, c, 0.0, 0.0, 0.0, 1.0};
    orc_copy(9, A, H2);
  }
  e[0] = q[0] - measured[0];
  e[1] = q[1] - measured[1];
  e[2] = dv[2] - measured[2];
}

/* ---------------------------------------------------------------- GTSAM PriorFactor / BetweenFactor
 * used at gpslam/gp/tests/testGaussianProcessPriorPose3.cpp:173-174, matlab/PlazaPose2.m:63,:80,:125.
 * kind: one of ORC_LINEAR* / ORC_POSE2 / ORC_POSE3 / ORC_ROT3; chart: ORC_CHART_EXPMAP or
 * ORC_CHART_FIRST_ORDER (Pose2 only: GTSAM's default Pose2 chart).
 *   PriorFactor:   e = Local(prior, x),  H = dLocal/dx
 *   BetweenFactor: hx = between(x1, x2), e = Local(measured, hx), H1 = Hlocal * (-Ad(hx^-1)), H2 = Hlocal */

int orc_pose_dim(int kind) {
  switch (kind) {
    case ORC_LINEAR2: return 2;
    case ORC_LINEAR3: return 3;
    case ORC_POSE2: return 3;
    case ORC_POSE3: return 12;
    case ORC_ROT3: return 9;
    case ORC_ROT3_BIAS: return 12;
    default: return -1;
  }
}
int orc_tangent_dim(int kind) {
  switch (kind) {
    case ORC_LINEAR2: return 2;
    case ORC_LINEAR3: return 3;
    case ORC_POSE2: return 3;
    case ORC_POSE3: return 6;
    case ORC_ROT3: return 3;
    case ORC_ROT3_BIAS: return 6;
    default: return -1;
  }
}

/* v = Local_origin(h), H = dv/dh (d x d) */
static void chart_local(int kind, int chart, const double *h, double *v, double *H) {
  switch (kind) {
    case ORC_POSE2:
      if (chart == ORC_CHART_FIRST_ORDER) {
        v[0] = h[0]; v[1] = h[1]; v[2] = atan2(sin(h[2]), cos(h[2]));
        if (H) {
          double c = cos(h[2]), s = sin(h[2]);
          double A[9] = {c, -s, 0.0, s, c, 0.0, 0.0, 0.0, 1.0};   /* topLeft = h.rotation().matrix() (GTSAM Pose2::ChartAtOrigin::Local) */
          orc_copy(9, A, H);
        }
      } else {
        orc_pose2_logmap(h, v, H);
      }
      break;
    case ORC_POSE3: orc_pose3_logmap(h, v, H); break;
    case ORC_ROT3: orc_rot3_logmap(h, v, H); break;
    default: break;
  }
}

/* Rot3::CayleyChart::Retract (GTSAM 4.0 geometry/Rot3M.cpp, recalled: SURVEY.md Appendix A "Default retract in 4.0:
 * Cayley"): R(w) = (I + W/2)(I - W/2)^-1 written out; "parity unpinned" like every chart default */
static void orc_rot3_cayley(const double *w, double *R) {
  const double x = w[0], y = w[1], z = w[2];
  const double x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
  const double f = 1.0 / (4.0 + x2 + y2 + z2), f2 = 2.0 * f;
  R[0] = (4 + x2 - y2 - z2) * f; R[1] = (xy - 2 * z) * f2; R[2] = (xz + 2 * y) * f2;
  R[3] = (xy + 2 * z) * f2; R[4] = (4 - x2 + y2 - z2) * f; R[5] = (yz - 2 * x) * f2;
  R[6] = (xz - 2 * y) * f2; R[7] = (yz + 2 * x) * f2; R[8] = (4 - x2 - y2 + z2) * f;
}

void orc_retract(int kind, int chart, const double *x, const double *delta, double *out) {
  switch (kind) {
    case ORC_LINEAR2:
    case ORC_LINEAR3: {
      int d = orc_tangent_dim(kind);
      for (int i = 0; i < d; i++) out[i] = x[i] + delta[i];
    } break;
    case ORC_POSE2: {
      double ex[3];
      if (chart == ORC_CHART_FIRST_ORDER) { ex[0] = delta[0]; ex[1] = delta[1]; ex[2] = delta[2]; }
      else orc_pose2_expmap(delta, ex, NULL);
      orc_pose2_compose(x, ex, out, NULL, NULL);
    } break;
    case ORC_POSE3: {
      double ex[12];
      if (chart == ORC_CHART_FIRST_ORDER) {   /* GTSAM 4.0 default: Pose3(Rot3::Retract(w) [Cayley], v) */
        orc_rot3_cayley(delta, ex);
        ex[9] = delta[3]; ex[10] = delta[4]; ex[11] = delta[5];
      } else {
        orc_pose3_expmap(delta, ex, NULL);
      }
      orc_pose3_compose(x, ex, out, NULL, NULL);
    } break;
    case ORC_ROT3: {
      double ex[9];
      if (chart == ORC_CHART_FIRST_ORDER) orc_rot3_cayley(delta, ex);
      else orc_rot3_expmap(delta, ex, NULL);
      orc_rot3_compose(x, ex, out, NULL, NULL);
    } break;
    case ORC_ROT3_BIAS:   /* SO(3) x R^3: rotation by the Rot3 chart, bias additive */
      orc_retract(ORC_ROT3, chart, x, delta, out);
      for (int i = 0; i < 3; i++) out[9 + i] = x[9 + i] + delta[3 + i];
      break;
    default: break;
  }
}

/* local coordinates of y around x: Local_origin(x^-1 y) */
void orc_local(int kind, int chart, const double *x, const double *y, double *v) {
  switch (kind) {
    case ORC_LINEAR2:
    case ORC_LINEAR3: {
      int d = orc_tangent_dim(kind);
      for (int i = 0; i < d; i++) v[i] = y[i] - x[i];
    } break;
    case ORC_POSE2: { double inv[3], h[3]; orc_pose2_inverse(x, inv, NULL); orc_pose2_compose(inv, y, h, NULL, NULL); chart_local(kind, chart, h, v, NULL); } break;
    case ORC_POSE3: { double inv[12], h[12]; orc_pose3_inverse(x, inv, NULL); orc_pose3_compose(inv, y, h, NULL, NULL); chart_local(kind, chart, h, v, NULL); } break;
    case ORC_ROT3: { double inv[9], h[9]; orc_rot3_inverse(x, inv, NULL); orc_rot3_compose(inv, y, h, NULL, NULL); chart_local(kind, chart, h, v, NULL); } break;
    case ORC_ROT3_BIAS:
      orc_local(ORC_ROT3, chart, x, y, v);
      for (int i = 0; i < 3; i++) v[3 + i] = y[9 + i] - x[9 + i];
      break;
    default: break;
  }
}

/* embed a 3 x 3 rotation block and +-identity on the bias into the 6 x 6 Jacobian of the (rotation, bias) product */
static void embed_rb(const double *H3, double sgn, double *H) {
  orc_zero(36, H);
  for (int r = 0; r < 3; r++) {
    for (int q = 0; q < 3; q++) H[r * 6 + q] = H3[r * 3 + q];
    H[(3 + r) * 6 + 3 + r] = sgn;
  }
}

void orc_prior_factor(int kind, int chart, const double *prior, const double *x, double *e, double *H) {
  int d = orc_tangent_dim(kind);
  switch (kind) {
    case ORC_LINEAR2:
    case ORC_LINEAR3:
      for (int i = 0; i < d; i++) e[i] = x[i] - prior[i];
      if (H) orc_eye(d, H);
      break;
    case ORC_POSE2: { double inv[3], h[3]; orc_pose2_inverse(prior, inv, NULL); orc_pose2_compose(inv, x, h, NULL, NULL); chart_local(kind, chart, h, e, H); } break;
    case ORC_POSE3: { double inv[12], h[12]; orc_pose3_inverse(prior, inv, NULL); orc_pose3_compose(inv, x, h, NULL, NULL); chart_local(kind, chart, h, e, H); } break;
    case ORC_ROT3: { double inv[9], h[9]; orc_rot3_inverse(prior, inv, NULL); orc_rot3_compose(inv, x, h, NULL, NULL); chart_local(kind, chart, h, e, H); } break;
    case ORC_ROT3_BIAS: {   /* [PriorFactorRot3; PriorFactorVector(bias)] (GPAHRSexample.m:118-121) */
      double H3[9];
      orc_prior_factor(ORC_ROT3, chart, prior, x, e, H ? H3 : NULL);
      for (int i = 0; i < 3; i++) e[3 + i] = x[9 + i] - prior[9 + i];
      if (H) embed_rb(H3, 1.0, H);
    } break;
    default: break;
  }
}

void orc_between_factor(int kind, int chart, const double *measured, const double *x1, const double *x2, double *e,
                        double *H1, double *H2) {
  int d = orc_tangent_dim(kind);
  switch (kind) {
    case ORC_LINEAR2:
    case ORC_LINEAR3:
      for (int i = 0; i < d; i++) e[i] = (x2[i] - x1[i]) - measured[i];
      if (H1) { orc_eye(d, H1); orc_scale(d * d, -1.0, H1); }
      if (H2) orc_eye(d, H2);
      break;
    case ORC_POSE2: {
      double inv[3], hx[3], minv[3], h[3], Hl[9];
      orc_pose2_inverse(x1, inv, NULL);
      orc_pose2_compose(inv, x2, hx, NULL, NULL);
      orc_pose2_inverse(measured, minv, NULL);
      orc_pose2_compose(minv, hx, h, NULL, NULL);
      chart_local(kind, chart, h, e, Hl);
      if (H1) { double hxinv[3], Ad[9]; orc_pose2_inverse(hx, hxinv, NULL); orc_pose2_adjoint(hxinv, Ad); orc_mm(3, 3, 3, Hl, Ad, H1); orc_scale(9, -1.0, H1); }
      if (H2) orc_copy(9, Hl, H2);
    } break;
    case ORC_POSE3: {
      double inv[12], hx[12], minv[12], h[12], Hl[36];
      orc_pose3_inverse(x1, inv, NULL);
      orc_pose3_compose(inv, x2, hx, NULL, NULL);
      orc_pose3_inverse(measured, minv, NULL);
      orc_pose3_compose(minv, hx, h, NULL, NULL);
      chart_local(kind, chart, h, e, Hl);
      if (H1) { double hxinv[12], Ad[36]; orc_pose3_inverse(hx, hxinv, NULL); orc_pose3_adjoint(hxinv, Ad); orc_mm(6, 6, 6, Hl, Ad, H1); orc_scale(36, -1.0, H1); }
      if (H2) orc_copy(36, Hl, H2);
    } break;
    case ORC_ROT3: {
      double inv[9], hx[9], minv[9], h[9], Hl[9];
      orc_rot3_inverse(x1, inv, NULL);
      orc_rot3_compose(inv, x2, hx, NULL, NULL);
      orc_rot3_inverse(measured, minv, NULL);
      orc_rot3_compose(minv, hx, h, NULL, NULL);
      chart_local(kind, chart, h, e, Hl);
      if (H1) { double hxT[9]; orc_tr(3, 3, hx, hxT); orc_mm(3, 3, 3, Hl, hxT, H1); orc_scale(9, -1.0, H1); }   /* Ad(R^-1) = R^T */
      if (H2) orc_copy(9, Hl, H2);
    } break;
    case ORC_ROT3_BIAS: {   /* [BetweenFactorRot3; BetweenFactorVector(bias)] (GPAHRSexample.m:143) */
      double A[9], B[9];
      orc_between_factor(ORC_ROT3, chart, measured, x1, x2, e, H1 ? A : NULL, H2 ? B : NULL);
      for (int i = 0; i < 3; i++) e[3 + i] = (x2[9 + i] - x1[9 + i]) - measured[9 + i];
      if (H1) embed_rb(A, -1.0, H1);
      if (H2) embed_rb(B, 1.0, H2);
    } break;
    default: break;
  }
}

/* ---------------------------------------------------------------- gtsam::AHRSFactor (GTSAM 4.0, third party)
 * gtsam/navigation/AHRSFactor.cpp: PreintegratedAhrsMeasurements::integrateMeasurement / predict and
 * AHRSFactor::evaluateError; gtsam/navigation/PreintegratedRotation.cpp: integrateMeasurement, biascorrectedDeltaRij,
 * integrateCoriolis.  Call sites: matlab/GPAHRSexample.m:128-137 (integrate, factor), :188 (new pim).
 * NOT under /root/reference -- restated from the published algorithm:
 *   incrR = Exp((omega_meas - biasHat) dt), D = ExpmapDerivative(that);  deltaTij += dt;  deltaRij = deltaRij incrR;
 *   delRdelBiasOmega = incrR^T delRdelBiasOmega - D dt;   preintMeasCov = incrR^T preintMeasCov incrR + gyroCov dt
 *   predict(bias) = Log(deltaRij Exp(delRdelBiasOmega (bias - biasHat)))
 *   fR = Log(Exp(predict - Ri^T omegaCoriolis deltaTij)^T Ri^T Rj) */
void orc_ahrs_preint_reset(double *st) {
  orc_zero(28, st);
  st[0] = st[4] = st[8] = 1.0;
}
void orc_ahrs_preint_integrate(double *st, const double *bias_hat, const double *gyro_cov, const double *omega, double dt) {
  double th[3], incr[9], D[9], incrT[9], t[9], u[9];
  for (int i = 0; i < 3; i++) th[i] = (omega[i] - bias_hat[i]) * dt;
  orc_rot3_expmap(th, incr, D);
  st[18] += dt;
  orc_mm(3, 3, 3, st, incr, t);
  orc_copy(9, t, st);
  orc_tr(3, 3, incr, incrT);
  orc_mm(3, 3, 3, incrT, st + 9, t);
  for (int i = 0; i < 9; i++) st[9 + i] = t[i] - D[i] * dt;
  orc_mm(3, 3, 3, incrT, st + 19, t);     /* Fr = d(deltaRij incrR)/d deltaRij = incrR^T */
  orc_mm(3, 3, 3, t, incr, u);
  for (int i = 0; i < 9; i++) st[19 + i] = u[i] + gyro_cov[i] * dt;
}

void orc_ahrs_factor(const double *Ri, const double *Rj, const double *bias, const double *prm, double *e, double *H1,
                     double *H2, double *H3) {
  const double *dR = prm, *D = prm + 9, *bh = prm + 18, dtij = prm[21], *wc = prm + 22;
  double binc[3], bio[3], ex[9], Jbio[9], bc[9], om[3], Jom[9];
  for (int i = 0; i < 3; i++) binc[i] = bias[i] - bh[i];
  orc_mm(3, 3, 1, D, binc, bio);
  orc_rot3_expmap(bio, ex, Jbio);                 /* deltaRij.expmap(v, none, H): H = ExpmapDerivative(v) */
  orc_mm(3, 3, 3, dR, ex, bc);
  orc_rot3_logmap(bc, om, Jom);
  double RiT[9], cor[3], com[3];
  orc_tr(3, 3, Ri, RiT);
  orc_mm(3, 3, 1, RiT, wc, cor);
  for (int i = 0; i < 3; i++) { cor[i] *= dtij; com[i] = om[i] - cor[i]; }
  double cdR[9], Dexp[9], cdRT[9], aR[9], fRrot[9], Dlog[9];
  orc_rot3_expmap(com, cdR, Dexp);
  orc_mm(3, 3, 3, RiT, Rj, aR);
  orc_tr(3, 3, cdR, cdRT);
  orc_mm(3, 3, 3, cdRT, aR, fRrot);
  orc_rot3_logmap(fRrot, e, Dlog);
  if (H1 || H3) {
    double fRt[9], t[9], u[9];
    orc_tr(3, 3, fRrot, fRt);
    if (H1) {
      double S[9], aRT[9];
      orc_skew(cor, S);
      orc_mm(3, 3, 3, Dexp, S, t);               /* -D_coriolis */
      orc_mm(3, 3, 3, fRt, t, u);
      orc_tr(3, 3, aR, aRT);
      for (int i = 0; i < 9; i++) u[i] -= aRT[i];
      orc_mm(3, 3, 3, Dlog, u, H1);
    }
    if (H3) {
      double v[9];
      orc_mm(3, 3, 3, Jbio, D, t);               /* biascorrectedDeltaRij's H */
      orc_mm(3, 3, 3, Jom, t, u);                /* predict's H */
      orc_mm(3, 3, 3, Dexp, u, t);               /* JbiasOmega */
      orc_mm(3, 3, 3, fRt, t, v);
      orc_mm(3, 3, 3, Dlog, v, H3);
      orc_scale(9, -1.0, H3);
    }
  }
  if (H2) orc_copy(9, Dlog, H2);
}
