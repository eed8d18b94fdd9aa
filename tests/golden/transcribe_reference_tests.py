#!/usr/bin/env python3
"""Writes tests/golden/reference_tests.json.

The fixture is DATA: the inputs, expected outputs and tolerances of the reference's own unit
tests (gtrll/gpslam, gpslam/gp/tests/*.cpp and gpslam/slam/tests/*.cpp), transcribed by hand
with the file:line each case comes from.  Nothing here is reference source code and nothing is
computed by the reference (it cannot be built in this image: GTSAM/Eigen/Boost are absent).
Where the reference's expected Jacobian is `numericalDerivative11(evaluateError)`, the case
records the finite-difference step and tolerance the reference used; the test recomputes that
numerical derivative from the error function under test, exactly as the reference does.

Pose encodings:  pose3 = {"ypr": [yaw, pitch, roll], "t": [x, y, z]}  (Rot3::Ypr, Point3)
                 rot3  = {"ypr": [...]},  pose2 = [x, y, theta],  vectors = lists.
Run:  python tests/golden/transcribe_reference_tests.py
"""
import json
import math
import os

PI = math.pi
GP = "gpslam/gp/tests/"
SL = "gpslam/slam/tests/"


def P3(y, p, r, x, yy, z):
    return {"ypr": [y, p, r], "t": [x, yy, z]}


def R3(y, p, r):
    return {"ypr": [y, p, r]}


Z3, Z6 = [0, 0, 0], [0, 0, 0, 0, 0, 0]

# ---- GaussianProcessPriorPose3VW (dt = 0.1, Qc = 0.01 I6: testGaussianProcessPriorPose3VW.cpp:35-36).
# The last case transcribes the reference's statements literally: `w1` is assigned twice and `w2` keeps the value of the
# previous case (:124-125).
gp_prior_vw = [
    dict(src=GP + "testGaussianProcessPriorPose3VW.cpp:49-74", dt=0.1, p1=P3(0, 0, 0, 0, 0, 0), v1=[0, 0, 0], w1=[0, 0, 0],
         p2=P3(0, 0, 0, 0, 0, 0), v2=[0, 0, 0], w2=[0, 0, 0], expect=[0] * 12, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 6),
    dict(src=GP + "testGaussianProcessPriorPose3VW.cpp:77-103", dt=0.1, p1=P3(0, 0, 0, 0, 0, 0), v1=[1, 0, 0], w1=[0, 0, 0],
         p2=P3(0, 0, 0, 0.1, 0, 0), v2=[1, 0, 0], w2=[0, 0, 0], expect=[0] * 12, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 6),
    dict(src=GP + "testGaussianProcessPriorPose3VW.cpp:106-132", dt=0.1, p1=P3(0, 0, 0, 0, 0, 0), v1=[0, 0, 0], w1=[0, 0, 1],
         p2=P3(0.1, 0, 0, 0, 0, 0), v2=[0, 0, 0], w2=[0, 0, 1], expect=[0] * 12, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 6),
    dict(src=GP + "testGaussianProcessPriorPose3VW.cpp:135-139", dt=0.1, p1=P3(-0.1, 1.2, 0.3, -4.0, 2.0, 14.0),
         v1=[2, 3, 1], w1=[0, 6, 4], p2=P3(2.4, -2.5, 3.7, 9.0, -8.0, -7.0), v2=[1, 3, 8], w2=[0, 0, 1], expect=None,
         fd=1e-6, tol_H=[1e-5, 1e-6, 1e-6, 1e-5, 1e-6, 1e-6]),
]

# ---- GaussianProcessInterpolatorPose3VW (dt = 0.1, tau = 0.03, Qc = 0.01 I6: ...InterpolatorPose3VW.cpp:33-35);
# last case literal again: v1 / w1 assigned twice, v2 / w2 left from the previous case (:127-128)
interpolator_vw = [
    dict(src=GP + "testGaussianProcessInterpolatorPose3VW.cpp:39-64", dt=0.1, tau=0.03, qc=0.01, p1=P3(0, 0, 0, 0, 0, 0),
         v1=[0, 0, 0], w1=[0, 0, 0], p2=P3(0, 0, 0, 0, 0, 0), v2=[0, 0, 0], w2=[0, 0, 0], expect=P3(0, 0, 0, 0, 0, 0),
         tol_e=1e-6, fd=1e-6, tol_H=[1e-8] * 6),
    dict(src=GP + "testGaussianProcessInterpolatorPose3VW.cpp:67-93", dt=0.1, tau=0.03, qc=0.01, p1=P3(0, 0, 0, 0, 0, 0),
         v1=[1, 2, 0], w1=[0, 0, 0], p2=P3(0, 0, 0, 0.1, 0.2, 0), v2=[1, 2, 0], w2=[0, 0, 0],
         expect=P3(0, 0, 0, 0.03, 0.06, 0), tol_e=1e-6, fd=1e-6, tol_H=[1e-8] * 6),
    dict(src=GP + "testGaussianProcessInterpolatorPose3VW.cpp:96-121", dt=0.1, tau=0.03, qc=0.01, p1=P3(0, 0, 0, 0, 0, 0),
         v1=[0, 0, 0], w1=[0, 0, 1], p2=P3(0.1, 0, 0, 0, 0, 0), v2=[0, 0, 0], w2=[0, 0, 1],
         expect=P3(0.03, 0, 0, 0, 0, 0), tol_e=1e-6, fd=1e-6, tol_H=[1e-8] * 6),
    dict(src=GP + "testGaussianProcessInterpolatorPose3VW.cpp:125-148", dt=0.1, tau=0.03, qc=0.01,
         p1=P3(0.4, -0.8, 0.2, 3, -8, 2), v1=[0.6, 0.3, -0.9], w1=[0.4, -0.2, 0.8], p2=P3(0.1, 0.3, -0.5, -9, 3, 4),
         v2=[0, 0, 0], w2=[0, 0, 1], expect=None, fd=1e-6, tol_H=[1e-8] * 6),
]

# ---- GPInterpolatedProjectionFactorPose3<Cal3_S2> (dt = 0.1, tau = 0.04, Qc = 0.001 I6, sigma = 0.1;
# K1 = Cal3_S2() = (fx, fy, s, u0, v0) = (1, 1, 0, 0, 0), K2 = Cal3_S2(50, 50, 0, 40, 30);
# testGPInterpolatedProjectionFactorPose3.cpp:39-47).  meas given as a dict = "project `land` through the camera at
# true_pose * sensor", as the reference builds it (:131-134).
PROJ = SL + "testGPInterpolatedProjectionFactorPose3.cpp"
K1, K2 = [1, 1, 0, 0, 0], [50, 50, 0, 40, 30]
PSENS = P3(1.0, 0.4, 0.5, 0.3, 0.6, -0.7)
interp_projection = [
    dict(src=PROJ + ":58-83", dt=0.1, tau=0.04, qc=0.001, K=K1, sensor=None, p1=P3(0, 0, 0, 0, 0, 0), v1=Z6,
         p2=P3(0, 0, 0, 0, 0, 0), v2=Z6, land=[0, 0, 10], meas=[0, 0], expect=[0, 0], tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=PROJ + ":87-112", dt=0.1, tau=0.04, qc=0.001, K=K1, sensor=None, p1=P3(0, 0, 0, -0.04, 0, 0),
         v1=[0, 0, 0, 1, 0, 0], p2=P3(0, 0, 0, 0.06, 0, 0), v2=[0, 0, 0, 1, 0, 0], land=[0, 0, 10], meas=[0, 0],
         expect=[0, 0], tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=PROJ + ":116-141", dt=0.1, tau=0.04, qc=0.001, K=K1, sensor=None, p1=P3(-0.04, 0, 0, 0, 0, 0),
         v1=[0, 0, 1, 0, 0, 0], p2=P3(0.06, 0, 0, 0, 0, 0), v2=[0, 0, 1, 0, 0, 0], land=[0, 0, 10], meas=[0, 0],
         expect=[0, 0], tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=PROJ + ":146-176", dt=0.1, tau=0.04, qc=0.001, K=K2, sensor=PSENS, p1=P3(0, 0, 0, 0, 0, 0),
         v1=[0, 0, 0, 15, 0, 0], p2=P3(0, 0, 0, 1.5, 0, 0), v2=[0, 0, 0, 15, 0, 0], land=[3.4, 1.2, 10],
         meas=dict(true_pose=P3(0, 0, 0, 0.6, 0, 0)), expect=[0, 0], tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
]
# optimisation (:180-262): two poses with priors, GP prior (dt 0.1, Qc 0.01 I6), three projections of one landmark at
# tau = 0.02 / 0.06 / 0.09 generated from cameras at x = 0.2 / 0.6 / 0.9; Gauss-Newton must recover everything to 1e-6
projection_optimization = dict(
    src=PROJ + ":180-262", dt=0.1, qc=0.01, K=K2, prior_sigma=0.01, cam_sigma=0.1, taus=[0.02, 0.06, 0.09],
    cam_poses=[P3(0, 0, 0, 0.2, 0, 0), P3(0, 0, 0, 0.6, 0, 0), P3(0, 0, 0, 0.9, 0, 0)],
    p1=P3(0, 0, 0, 0, 0, 0), p2=P3(0, 0, 0, 1, 0, 0), v1=[0, 0, 0, 10, 0, 0], v2=[0, 0, 0, 10, 0, 0], land=[3.4, 1.2, 20],
    p1_init=P3(0.1, 0.2, 0.4, 0.2, 0.3, -0.2), p2_init=P3(-0.1, -0.2, -0.4, 1.2, -0.3, 0.2),
    v1_init=[-0.3, 0, 0, 0.7, 0, 0.2], v2_init=[0, 0, 0.4, 1.2, 0, -0.1], land_init=[3.3, 1.3, 18], tol=1e-6)

# ---- GPInterpolatedGPSFactorPose3 (dt = 0.1, tau = 0.04, Qc = 0.001 I6, sigma 0.1, body_T_sensor = Ypr(1.0, 0.4, 0.5),
# (0.3, 0.6, -0.7): testGPInterpolatedGPSFactorPose3.cpp:40-46).  meas as a dict = translation of true_pose * sensor.
GPSF = SL + "testGPInterpolatedGPSFactorPose3.cpp"
GSENS = P3(1.0, 0.4, 0.5, 0.3, 0.6, -0.7)
interp_gps = [
    dict(src=GPSF + ":55-78", dt=0.1, tau=0.04, qc=0.001, sensor=None, p1=P3(0, 0, 0, 0, 0, 0), v1=Z6, p2=P3(0, 0, 0, 0, 0, 0),
         v2=Z6, meas=[0, 0, 0], expect=[0, 0, 0], tol_e=1e-6, fd=1e-4, tol_H=[1e-6] * 4),
    dict(src=GPSF + ":82-105", dt=0.1, tau=0.04, qc=0.001, sensor=None, p1=P3(0, 0, 0, -0.04, 0, 0), v1=[0, 0, 0, 1, 0, 0],
         p2=P3(0, 0, 0, 0.06, 0, 0), v2=[0, 0, 0, 1, 0, 0], meas=[0, 0, 0], expect=[0, 0, 0], tol_e=1e-6, fd=1e-4,
         tol_H=[1e-6] * 4),
    dict(src=GPSF + ":109-132", dt=0.1, tau=0.04, qc=0.001, sensor=None, p1=P3(-0.04, 0, 0, 0, 0, 0), v1=[0, 0, 1, 0, 0, 0],
         p2=P3(0.06, 0, 0, 0, 0, 0), v2=[0, 0, 1, 0, 0, 0], meas=[0, 0, 0], expect=[0, 0, 0], tol_e=1e-6, fd=1e-4,
         tol_H=[1e-6] * 4),
    dict(src=GPSF + ":136-162", dt=0.1, tau=0.04, qc=0.001, sensor=GSENS, p1=P3(0, 0, 0, 0, 0, 0), v1=[0, 0, 0, 15, 0, 0],
         p2=P3(0, 0, 0, 1.5, 0, 0), v2=[0, 0, 0, 15, 0, 0], meas=dict(true_pose=P3(0, 0, 0, 0.6, 0, 0)), expect=[0, 0, 0],
         tol_e=1e-6, fd=1e-4, tol_H=[1e-6] * 4),
    dict(src=GPSF + ":166-188", dt=0.1, tau=0.04, qc=0.001, sensor=GSENS, p1=P3(1.3, 2.4, 1.2, 0.2, 0.3, 0.4),
         v1=[1.0, 2.0, 0.4, 15, 0.3, 0.2], p2=P3(0.5, 6.5, 1.1, 1.5, 0.7, 0.5), v2=[2.0, 0.2, 0.1, 17, 0.4, 0.7],
         meas=[0, 0, 0], expect=None, fd=1e-4, tol_H=[1e-6] * 4),
]
# optimisation (:193-262): loose prior on x1 (sigma 100), tight velocity priors, GP prior (dt 0.1, Qc 0.01 I6), three GPS
# fixes at tau = -0.1 / 0.05 / 0.2 (two of them extrapolate) on the line the true trajectory follows
gps_optimization = dict(
    src=GPSF + ":193-262", dt=0.1, qc=0.01, prior_sigma=0.01, loose_sigma=100.0, gps_sigma=0.1, taus=[-0.1, 0.05, 0.2],
    meas=[[-1, 0, 0], [0.5, 0, 0], [2, 0, 0]], p1=P3(0, 0, 0, 0, 0, 0), p2=P3(0, 0, 0, 1, 0, 0), v1=[0, 0, 0, 10, 0, 0],
    v2=[0, 0, 0, 10, 0, 0], p1_init=P3(0.1, 0.1, -0.1, 0.04, 0.1, -0.06), p2_init=P3(-0.1, 0.1, -0.1, 1.05, -0.1, 0.1),
    v1_init=[-0.1, 0, 0, 9.8, 0, 0.2], v2_init=[0, 0, 0.2, 9.7, 0, -0.1], tol=1e-6)

# ---- GPInterpolatedGPSFactorPose3VW (same dt / tau / Qc; body_T_sensor = Ypr(1.4, 4.4, -0.5), (0.3, 0.6, -0.7) and a
# rotation-only variant: testGPInterpolatedGPSFactorPose3VW.cpp:40-46).  Last case literal (v1 / w1 assigned twice, v2 / w2
# left from the previous case, :249-250).
GPSV = SL + "testGPInterpolatedGPSFactorPose3VW.cpp"
VSENS, VSENS_ROT = P3(1.4, 4.4, -0.5, 0.3, 0.6, -0.7), P3(1.4, 4.4, -0.5, 0, 0, 0)
interp_gps_vw = [
    dict(src=GPSV + ":56-83", sensor=None, p1=P3(0, 0, 0, 0, 0, 0), v1=Z3, w1=Z3, p2=P3(0, 0, 0, 0, 0, 0), v2=Z3, w2=Z3,
         meas=[0, 0, 0], expect=[0, 0, 0]),
    dict(src=GPSV + ":87-114", sensor=None, p1=P3(0, 0, 0, -0.04, 0.04, 0), v1=[1, -1, 0], w1=Z3, p2=P3(0, 0, 0, 0.06, -0.06, 0),
         v2=[1, -1, 0], w2=Z3, meas=[0, 0, 0], expect=[0, 0, 0]),
    dict(src=GPSV + ":118-145", sensor=None, p1=P3(-0.04, 0, 0, 0, 0, 0), v1=Z3, w1=[0, 0, 1], p2=P3(0.06, 0, 0, 0, 0, 0), v2=Z3,
         w2=[0, 0, 1], meas=[0, 0, 0], expect=[0, 0, 0]),
    dict(src=GPSV + ":149-178", sensor=VSENS, p1=P3(0, 0, 0, 1, 0, 0), v1=[15, 5, 0], w1=Z3, p2=P3(0, 0, 0, 2.5, 0.5, 0),
         v2=[15, 5, 0], w2=Z3, meas=dict(true_pose=P3(0, 0, 0, 1.6, 0.2, 0)), expect=[0, 0, 0]),
    dict(src=GPSV + ":180-210", sensor=VSENS_ROT, p1=P3(0, 0, 0, 1, 0, 0), v1=[15, 5, 0], w1=Z3, p2=P3(0, 0, 0, 2.5, 0.5, 0),
         v2=[15, 5, 0], w2=Z3, meas=[0, 0, 0], expect=[1.6, 0.2, 0]),
    dict(src=GPSV + ":213-243", sensor=VSENS, p1=P3(0, 0, 0, 0, 0, 0), v1=Z3, w1=[0, 0, 10], p2=P3(1.0, 0, 0, 0, 0, 0), v2=Z3,
         w2=[0, 0, 10], meas=dict(true_pose=P3(0.4, 0, 0, 0, 0, 0)), expect=[0, 0, 0]),
    dict(src=GPSV + ":246-270", sensor=VSENS, p1=P3(0.4, -0.8, 0.2, 3, -8, 2), v1=[0.6, 0.3, -0.9], w1=[0.4, -0.2, 0.8],
         p2=P3(0.1, 0.3, -0.5, -9, 3, 4), v2=Z3, w2=[0, 0, 10], meas=[0, 0, 0], expect=None),
]
for _c in interp_gps_vw:
    _c.update(dt=0.1, tau=0.04, qc=0.001, tol_e=1e-6, fd=1e-4, tol_H=[1e-6] * 6)

# ---- convertVWtoVb known answers (testPose3Utils.cpp:345-420): body 6-velocity [w_b; v_b] of world (v, w) at `pose`;
# the Jacobians are checked against central differences of the same function (:359-367)
vw_conversion = [
    dict(src=GP + "testPose3Utils.cpp:355-368", pose=P3(0, 0, 0, 0, 0, 0), v=[1, 0, 0], w=[1, 0, 0], v6=[1, 0, 0, 1, 0, 0]),
    dict(src=GP + "testPose3Utils.cpp:371-384", pose=P3(0, 0, 0, 3, 4, 5), v=[1, 2, 3], w=[1, -2, -3], v6=[1, -2, -3, 1, 2, 3]),
    dict(src=GP + "testPose3Utils.cpp:388-401", pose=P3(PI / 2, 0, 0, 0, 0, 0), v=[1, 0, 0], w=[0, 0, 1], v6=[0, 0, 1, 0, -1, 0]),
    dict(src=GP + "testPose3Utils.cpp:404-417", pose=P3(PI / 2, 0, 0, 0, 0, 0), v=[-3, 4, 0], w=[0, 0, 1], v6=[0, 0, 1, 4, 3, 0]),
]

gp_prior = [
    # ---- GaussianProcessPriorPose3 (dt = 0.1, Qc = 0.01 I6: testGaussianProcessPriorPose3.cpp:29-30)
    dict(src=GP + "testGaussianProcessPriorPose3.cpp:43-65", kind="pose3", dt=0.1, p1=P3(0, 0, 0, 0, 0, 0), v1=Z6,
         p2=P3(0, 0, 0, 0, 0, 0), v2=Z6, expect=[0] * 12, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorPose3.cpp:69-91", kind="pose3", dt=0.1, p1=P3(0, 0, 0, 0, 0, 0),
         v1=[0, 0, 0, 1, 0, 0], p2=P3(0, 0, 0, 0.1, 0, 0), v2=[0, 0, 0, 1, 0, 0], expect=[0] * 12, tol_e=1e-6,
         fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorPose3.cpp:95-117", kind="pose3", dt=0.1, p1=P3(0, 0, 0, 0, 0, 0),
         v1=[0, 0, 1, 0, 0, 0], p2=P3(0.1, 0, 0, 0, 0, 0), v2=[0, 0, 1, 0, 0, 0], expect=[0] * 12, tol_e=1e-6,
         fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorPose3.cpp:121-142", kind="pose3", dt=0.1,
         p1=P3(-0.1, 1.2, 0.3, -4.0, 2.0, 14.0), v1=[2, 3, 1, 5, 4, 9], p2=P3(2.4, -2.5, 3.7, 9.0, -8.0, -7.0),
         v2=[1, 3, 8, 0, 6, 4], expect=None, fd=1e-6, tol_H=[1e-5, 1e-6, 1e-5, 1e-6]),
    # ---- GaussianProcessPriorPose2 (dt = 0.1, Qc = 0.01 I3: testGaussianProcessPriorPose2.cpp:29-30)
    dict(src=GP + "testGaussianProcessPriorPose2.cpp:43-65", kind="pose2", dt=0.1, p1=[0, 0, 0], v1=Z3, p2=[0, 0, 0],
         v2=Z3, expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorPose2.cpp:69-91", kind="pose2", dt=0.1, p1=[0, 0, 0], v1=[1, 0, 0],
         p2=[0.1, 0, 0], v2=[1, 0, 0], expect=[0] * 6, tol_e=1e-6, fd=1e-4, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorPose2.cpp:95-117", kind="pose2", dt=0.1, p1=[0, 0, 0], v1=[0, 0, 1],
         p2=[0, 0, 0.1], v2=[0, 0, 1], expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorPose2.cpp:121-141", kind="pose2", dt=0.1, p1=[-0.1, 1.2, 0.3],
         v1=[5, 4, 9], p2=[2.4, -2.5, 3.7], v2=[0, 6, 4], expect=None, fd=1e-6, tol_H=[1e-6] * 4),
    # ---- GaussianProcessPriorRot3 (testGaussianProcessPriorRot3.cpp:29-30)
    dict(src=GP + "testGaussianProcessPriorRot3.cpp:43-65", kind="rot3", dt=0.1, p1=R3(0, 0, 0), v1=Z3,
         p2=R3(0, 0, 0), v2=Z3, expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorRot3.cpp:69-91", kind="rot3", dt=0.1, p1=R3(0, 0, 0), v1=[0, 0, 1],
         p2=R3(0.1, 0, 0), v2=[0, 0, 1], expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorRot3.cpp:94-116", kind="rot3", dt=0.1, p1=R3(0, 0, 0), v1=[1, 0, 0],
         p2=R3(0, 0, 0.1), v2=[1, 0, 0], expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorRot3.cpp:120-141", kind="rot3", dt=0.1, p1=R3(-0.1, 1.2, 0.3),
         v1=[2, 3, 1], p2=R3(2.4, -2.5, 3.7), v2=[1, 3, 8], expect=None, fd=1e-6, tol_H=[1e-5, 1e-6, 1e-5, 1e-6]),
    # ---- GaussianProcessPriorLinear<3> (testGaussianProcessPriorLinear.cpp:32-33)
    dict(src=GP + "testGaussianProcessPriorLinear.cpp:45-67", kind="linear3", dt=0.1, p1=Z3, v1=Z3, p2=Z3, v2=Z3,
         expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorLinear.cpp:70-92", kind="linear3", dt=0.1, p1=Z3, v1=[1, 0, 0],
         p2=[0.1, 0, 0], v2=[1, 0, 0], expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorLinear.cpp:95-117", kind="linear3", dt=0.1, p1=Z3, v1=[0, 0, 1],
         p2=[0, 0, 0.1], v2=[0, 0, 1], expect=[0] * 6, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessPriorLinear.cpp:120-140", kind="linear3", dt=0.1, p1=[2, -5, 7],
         v1=[-1, 2, -9], p2=[-8, 4, -8], v2=[3, -4, 7], expect=None, fd=1e-6, tol_H=[1e-6] * 4),
]

# interpolators: dt = 0.1, tau = 0.03, Qc = 0.01 I
interpolator = [
    dict(src=GP + "testGaussianProcessInterpolatorPose3.cpp:33-56", kind="pose3", dt=0.1, tau=0.03, qc=0.01,
         p1=P3(0, 0, 0, 0, 0, 0), v1=Z6, p2=P3(0, 0, 0, 0, 0, 0), v2=Z6, expect=P3(0, 0, 0, 0, 0, 0), tol_e=1e-6,
         fd=1e-6, tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose3.cpp:60-83", kind="pose3", dt=0.1, tau=0.03, qc=0.01,
         p1=P3(0, 0, 0, 0, 0, 0), v1=[0, 0, 0, 1, 0, 0], p2=P3(0, 0, 0, 0.1, 0, 0), v2=[0, 0, 0, 1, 0, 0],
         expect=P3(0, 0, 0, 0.03, 0, 0), tol_e=1e-6, fd=1e-4, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose3.cpp:87-110", kind="pose3", dt=0.1, tau=0.03, qc=0.01,
         p1=P3(0, 0, 0, 0, 0, 0), v1=[0, 0, 1, 0, 0, 0], p2=P3(0.1, 0, 0, 0, 0, 0), v2=[0, 0, 1, 0, 0, 0],
         expect=P3(0.03, 0, 0, 0, 0, 0), tol_e=1e-6, fd=1e-6, tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose3.cpp:114-135", kind="pose3", dt=0.1, tau=0.03, qc=0.01,
         p1=P3(0.4, -0.8, 0.2, 3, -8, 2), v1=[0.1, -0.2, -1.4, 0.5, 0.9, 0.7], p2=P3(0.1, 0.3, -0.5, -9, 3, 4),
         v2=[0.6, 0.3, -0.9, 0.4, -0.2, 0.8], expect=None, fd=1e-6, tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose2.cpp:33-56", kind="pose2", dt=0.1, tau=0.03, qc=0.01,
         p1=[0, 0, 0], v1=Z3, p2=[0, 0, 0], v2=Z3, expect=[0, 0, 0], tol_e=1e-6, fd=1e-6, tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose2.cpp:60-83", kind="pose2", dt=0.1, tau=0.03, qc=0.01,
         p1=[0, 0, 0], v1=[1, 0, 0], p2=[0.1, 0, 0], v2=[1, 0, 0], expect=[0.03, 0, 0], tol_e=1e-6, fd=1e-4,
         tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose2.cpp:87-110", kind="pose2", dt=0.1, tau=0.03, qc=0.01,
         p1=[0, 0, 0], v1=[0, 0, 1], p2=[0, 0, 0.1], v2=[0, 0, 1], expect=[0, 0, 0.03], tol_e=1e-6, fd=1e-6,
         tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorPose2.cpp:114-135", kind="pose2", dt=0.1, tau=0.03, qc=0.01,
         p1=[3, -8, 2], v1=[0.5, 0.9, 0.7], p2=[-9, 3, 4], v2=[0.6, -0.2, 0.8], expect=None, fd=1e-6,
         tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorRot3.cpp:33-56", kind="rot3", dt=0.1, tau=0.03, qc=0.01,
         p1=R3(0, 0, 0), v1=Z3, p2=R3(0, 0, 0), v2=Z3, expect=R3(0, 0, 0), tol_e=1e-6, fd=1e-6, tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorRot3.cpp:60-83", kind="rot3", dt=0.1, tau=0.03, qc=0.01,
         p1=R3(0, 0, 0), v1=[1, 0, 0], p2=R3(0, 0, 0.1), v2=[1, 0, 0], expect=R3(0, 0, 0.03), tol_e=1e-6, fd=1e-6,
         tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorRot3.cpp:87-110", kind="rot3", dt=0.1, tau=0.03, qc=0.01,
         p1=R3(0, 0, 0), v1=[0, 0, 1], p2=R3(0.1, 0, 0), v2=[0, 0, 1], expect=R3(0.03, 0, 0), tol_e=1e-6, fd=1e-6,
         tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorRot3.cpp:114-135", kind="rot3", dt=0.1, tau=0.03, qc=0.01,
         p1=R3(0.4, -0.8, 0.2), v1=[0.1, -0.2, -1.4], p2=R3(0.1, 0.3, -0.5), v2=[0.6, 0.3, -0.9], expect=None,
         fd=1e-6, tol_H=[1e-8] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorLinear.cpp:58-81", kind="linear3", dt=0.1, tau=0.03, qc=0.01,
         p1=Z3, v1=Z3, p2=Z3, v2=Z3, expect=[0, 0, 0], tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorLinear.cpp:84-107", kind="linear3", dt=0.1, tau=0.03, qc=0.01,
         p1=Z3, v1=[10, 0, 0], p2=[1, 0, 0], v2=[10, 0, 0], expect=[0.3, 0, 0], tol_e=1e-6, fd=1e-6,
         tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorLinear.cpp:110-133", kind="linear3", dt=0.1, tau=0.03, qc=0.01,
         p1=Z3, v1=[0, 0, 3], p2=[0, 0, 0.3], v2=[0, 0, 3], expect=[0, 0, 0.09], tol_e=1e-6, fd=1e-6,
         tol_H=[1e-6] * 4),
    dict(src=GP + "testGaussianProcessInterpolatorLinear.cpp:136-157", kind="linear3", dt=0.1, tau=0.03, qc=0.01,
         p1=[2, -5, 7], v1=[-1, 2, -9], p2=[-8, 4, -8], v2=[3, -4, 7], expect=None, fd=1e-6, tol_H=[1e-6] * 4),
]

# interpolated range factors: dt = 0.1, tau = 0.04, Qc = 0.001 I, sigma 0.1
_true_r2 = math.hypot(3.4 - 0.6, 1.2)           # Pose2(0.6,0,0).range(Point2(3.4,1.2)), RangeFactorPose2.cpp:146-149
interp_range = [
    dict(src=SL + "testGPInterpolatedRangeFactorPose2.cpp:55-80", kind="pose2", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, 0], v1=Z3, p2=[0, 0, 0], v2=Z3, land=[0, 10], meas=10, sensor=None, expect=0.0, tol_e=1e-6,
         fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose2.cpp:84-109", kind="pose2", dt=0.1, tau=0.04, qc=0.001,
         p1=[-0.04, 0, 0], v1=[1, 0, 0], p2=[0.06, 0, 0], v2=[1, 0, 0], land=[0, 10], meas=10, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose2.cpp:113-138", kind="pose2", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, -0.04], v1=[0, 0, 1], p2=[0, 0, 0.06], v2=[0, 0, 1], land=[0, 10], meas=10, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose2.cpp:142-169", kind="pose2", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, 0], v1=[15, 0, 0], p2=[1.5, 0, 0], v2=[15, 0, 0], land=[3.4, 1.2], meas=_true_r2, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose2.cpp:172-195", kind="pose2", dt=0.1, tau=0.04, qc=0.001,
         p1=[5.34, 7.1, -4.32], v1=[15, 21.3, 32], p2=[1.5, -2.2, 3.0], v2=[-15, 4.2, -30], land=[3.4, 1.2],
         meas=_true_r2, sensor=None, expect=None, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose3.cpp:56-81", kind="pose3", dt=0.1, tau=0.04, qc=0.001,
         p1=P3(0, 0, 0, 0, 0, 0), v1=Z6, p2=P3(0, 0, 0, 0, 0, 0), v2=Z6, land=[0, 0, 10], meas=10, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose3.cpp:85-110", kind="pose3", dt=0.1, tau=0.04, qc=0.001,
         p1=P3(0, 0, 0, -0.04, 0, 0), v1=[0, 0, 0, 1, 0, 0], p2=P3(0, 0, 0, 0.06, 0, 0), v2=[0, 0, 0, 1, 0, 0],
         land=[0, 0, 10], meas=10, sensor=None, expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactorPose3.cpp:114-139", kind="pose3", dt=0.1, tau=0.04, qc=0.001,
         p1=P3(-0.04, 0, 0, 0, 0, 0), v1=[0, 0, 1, 0, 0, 0], p2=P3(0.06, 0, 0, 0, 0, 0), v2=[0, 0, 1, 0, 0, 0],
         land=[0, 0, 10], meas=10, sensor=None, expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    # with body_T_sensor = Pose3(Ypr(1.0,0.4,0.5), (0.3,0.6,-0.7)) (:45); meas = (true_pose * sensor).range(land)
    dict(src=SL + "testGPInterpolatedRangeFactorPose3.cpp:144-172", kind="pose3", dt=0.1, tau=0.04, qc=0.001,
         p1=P3(0, 0, 0, 0, 0, 0), v1=[0, 0, 0, 15, 0, 0], p2=P3(0, 0, 0, 1.5, 0, 0), v2=[0, 0, 0, 15, 0, 0],
         land=[3.4, 1.2, 10], meas={"true_pose": P3(0, 0, 0, 0.6, 0, 0)}, sensor=P3(1.0, 0.4, 0.5, 0.3, 0.6, -0.7),
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6, 1e-6, 1e-6, 1e-5, 1e-6]),
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:55-80", kind="linear3", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, 0], v1=Z3, p2=[0, 0, 0], v2=Z3, land=[0, 10], meas=10, sensor=None, expect=0.0, tol_e=1e-6,
         fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:84-109", kind="linear3", dt=0.1, tau=0.04, qc=0.001,
         p1=[-0.04, 0, 0], v1=[1, 0, 0], p2=[0.06, 0, 0], v2=[1, 0, 0], land=[0, 10], meas=10, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:113-138", kind="linear3", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, -0.04], v1=[0, 0, 1], p2=[0, 0, 0.06], v2=[0, 0, 1], land=[0, 10], meas=10, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:142-167", kind="linear3", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, 10 * PI - 0.04], v1=[0, 0, 1], p2=[0, 0, 10 * PI + 0.06], v2=[0, 0, 1], land=[0, 10], meas=10,
         sensor=None, expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:171-198", kind="linear3", dt=0.1, tau=0.04, qc=0.001,
         p1=[0, 0, 0], v1=[15, 0, 0], p2=[1.5, 0, 0], v2=[15, 0, 0], land=[3.4, 1.2], meas=_true_r2, sensor=None,
         expect=0.0, tol_e=1e-6, fd=1e-6, tol_H=[1e-6] * 5),
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:201-224", kind="linear3", dt=0.1, tau=0.04, qc=0.001,
         p1=[5.34, 7.1, -4.32], v1=[15, 21.3, 32], p2=[1.5, -2.2, 3.0], v2=[-15, 4.2, -30], land=[3.4, 1.2],
         meas=_true_r2, sensor=None, expect=None, fd=1e-6, tol_H=[1e-6] * 5),
]

range2d = [
    dict(src=SL + "testRangeFactor2DLinear.cpp:39-44", pose=[0, 0, 0], land=[0, 0], meas=0.0, expect=0.0,
         check_H=False),
    dict(src=SL + "testRangeFactor2DLinear.cpp:47-60", pose=[3, 4, 5], land=[7, 7], meas=5.0, expect=0.0,
         check_H=True),
    dict(src=SL + "testRangeFactor2DLinear.cpp:63-76", pose=[13.1, -4.8, 1.5], land=[-5.4, 6.6], meas=13.1,
         expect=8.630393461693233, check_H=True),
]
bearing_range2d = [
    dict(src=SL + "testRangeBearingFactor2DLinear.cpp:39-52", pose=[3, 4, 0], land=[7, 7], range=5.0,
         bearing=0.643501108793284, expect=[0, 0]),
    dict(src=SL + "testRangeBearingFactor2DLinear.cpp:55-68", pose=[13.1, -4.8, 1.5], land=[-5.4, 6.6], range=13.1,
         bearing=0.0, expect=[1.089334716657378, 8.630393461693233]),
]
odometry2d = [
    dict(src=SL + "testOdometryFactor2DLinear.cpp:38-44", pose1=[0, 0, 0], pose2=[0, 0, 0], meas=[0, 0, 0],
         expect=[0, 0, 0], check_H=False),
    dict(src=SL + "testOdometryFactor2DLinear.cpp:47-61", pose1=[0, 0, 0], pose2=[1, 0, 0], meas=[1, 0, 0],
         expect=[0, 0, 0], check_H=True),
    dict(src=SL + "testOdometryFactor2DLinear.cpp:65-79", pose1=[42, 24, 1.570796326794897],
         pose2=[42, 25, 2.570796326794897], meas=[1, 0, 1.0], expect=[0, 0, 0], check_H=True),
]

# testPose3Utils.cpp:89-164 (dt = 0.1)
H = PI / 2.0
body_centric_velocity = [
    dict(src=GP + "testPose3Utils.cpp:96-103", p1=P3(0, 0, 0, 0, 0, 0), p2=P3(0, 0, 0, 0, 0, 0), vb=Z6, vs=Z6),
    dict(src=GP + "testPose3Utils.cpp:106-113", p1=P3(0, 0, 0, 0, 0, 0), p2=P3(0, 0, 0, 0.1, 0, 0),
         vb=[0, 0, 0, 1, 0, 0], vs=[0, 0, 0, 1, 0, 0]),
    dict(src=GP + "testPose3Utils.cpp:116-123", p1=P3(0, 0, 0, 0, 0, 0), p2=P3(0.1, 0, 0, 0, 0, 0),
         vb=[0, 0, 1, 0, 0, 0], vs=[0, 0, 1, 0, 0, 0]),
    dict(src=GP + "testPose3Utils.cpp:126-133", p1=P3(H, 0, 0, 0, 0, 0), p2=P3(H, 0, 0, 0.1, 0, 0),
         vb=[0, 0, 0, 0, -1, 0], vs=[0, 0, 0, 1, 0, 0]),
    dict(src=GP + "testPose3Utils.cpp:136-143", p1=P3(H, 0, 0, 0, 0, 0), p2=P3(H + 0.1, 0, 0, 0, 0, 0),
         vb=[0, 0, 1, 0, 0, 0], vs=[0, 0, 1, 0, 0, 0]),
    dict(src=GP + "testPose3Utils.cpp:146-153", p1=P3(H, 0, 0, 1.0, 0, 0), p2=P3(H, 0, 0.1, 1.0, 0, 0),
         vb=[1, 0, 0, 0, 0, 0], vs=[0, 1, 0, 0, 0, 1]),
    dict(src=GP + "testPose3Utils.cpp:156-163", p1=P3(0, 0, 0, 0, -1.0, 0), p2=P3(H, 0, 0, 1.0, 0, 0),
         vb=[0, 0, H * 10, H * 10, 0, 0], vs=[0, 0, H * 10, 0, 0, 0]),
]

# testPose3Utils.cpp:167-286: right Jacobians vs numericalLieRightJacobian (:44-56), dt = 1e-6
lie_jacobians = [
    dict(src=GP + "testPose3Utils.cpp:194-214", group="rot3", x=R3(0, 0, 0), tol=1e-6, tol_inv=1e-6),
    dict(src=GP + "testPose3Utils.cpp:194-214", group="rot3", x=R3(1.0, 2.0, 3.0), tol=1e-6, tol_inv=1e-6),
    dict(src=GP + "testPose3Utils.cpp:255-285", group="pose3", x=P3(0, 0, 0, 0, 0, 0), tol=1e-6, tol_inv=1e-6),
    dict(src=GP + "testPose3Utils.cpp:255-285", group="pose3", x=P3(1e-5, 0, 1e-5, 0, 2e-5, 0), tol=1e-6,
         tol_inv=1e-8),
    dict(src=GP + "testPose3Utils.cpp:255-285", group="pose3", x=P3(1.0, 2.0, 3.0, 4.0, 5.0, 6.0), tol=1e-6,
         tol_inv=1e-5),
]
# testPose3Utils.cpp:289-328 (Anderson15iros eq. 8), dt = 0.01
se3_velocity = [
    dict(src=GP + "testPose3Utils.cpp:297-305", base=P3(0, 0, 0, 0, 0, 0), dlog=[0, 0, 0, 0, 0, 0], tol=1e-6),
    dict(src=GP + "testPose3Utils.cpp:308-316", base=P3(0, 0, 0, 0, 0, 0),
         dlog=[1e-4, 2e-4, -4e-4, 5e-4, 2e-4, 3e-4], tol=1e-6),
    dict(src=GP + "testPose3Utils.cpp:319-327", base=P3(2.4, 1.2, 3.9, 43, -5, 12),
         dlog=[1e-4, 2e-4, -4e-4, 5e-4, 2e-4, 3e-4], tol=1e-4),
]

# 2-state Gauss-Newton problems (GaussNewtonParams defaults), expected fixed points
optimization = [
    dict(src=GP + "testGaussianProcessPriorPose3.cpp:146-195", kind="pose3", dt=1.0, qc=0.01, landmark_dim=0,
         init=dict(pose=[P3(0, 0, 0, 0, 0, 0), P3(0, 0, 0, 1, 0, 0)],
                   vel=[[0, 0, 0, 1, 0, 0], [0.1, 0.2, -0.3, 2.0, -0.5, 0.6]]),
         pose_priors=[dict(idx=0, prior=P3(0, 0, 0, 0, 0, 0), sigma=0.001),
                      dict(idx=1, prior=P3(0, 0, 0, 1, 0, 0), sigma=0.001)],
         expect=dict(pose=[P3(0, 0, 0, 0, 0, 0), P3(0, 0, 0, 1, 0, 0)],
                     vel=[[0, 0, 0, 1, 0, 0], [0, 0, 0, 1, 0, 0]]), tol=1e-6, tol_err=1e-6),
    dict(src=GP + "testGaussianProcessPriorPose2.cpp:146-194", kind="pose2", dt=1.0, qc=0.01, landmark_dim=0,
         init=dict(pose=[[0, 0, 0], [1, 0, 0]], vel=[[1, 0, 0], [2.0, -0.5, 0.6]]),
         pose_priors=[dict(idx=0, prior=[0, 0, 0], sigma=0.001), dict(idx=1, prior=[1, 0, 0], sigma=0.001)],
         expect=dict(pose=[[0, 0, 0], [1, 0, 0]], vel=[[1, 0, 0], [1, 0, 0]]), tol=1e-6, tol_err=1e-6),
    dict(src=GP + "testGaussianProcessPriorRot3.cpp:145-194", kind="rot3", dt=0.1, qc=0.01, landmark_dim=0,
         init=dict(pose=[R3(0, 0, 0), R3(0, 0, 0.1)], vel=[[1, 0, 0], [2.0, -0.5, 0.6]]),
         pose_priors=[dict(idx=0, prior=R3(0, 0, 0), sigma=0.001), dict(idx=1, prior=R3(0, 0, 0.1), sigma=0.001)],
         expect=dict(pose=[R3(0, 0, 0), R3(0, 0, 0.1)], vel=[[1, 0, 0], [1, 0, 0]]), tol=1e-6, tol_err=1e-6),
    dict(src=GP + "testGaussianProcessPriorLinear.cpp:144-202", kind="linear3", dt=0.1, qc=0.01, landmark_dim=0,
         init=dict(pose=[[1, 0, 0], [1.1, 0, 0]], vel=[[1, 0.1, 0.2], [2.1, -1.2, 0.9]]),
         pose_priors=[dict(idx=0, prior=[1, 0, 0], sigma=0.001), dict(idx=1, prior=[1.1, 0, 0], sigma=0.001)],
         expect=dict(pose=[[1, 0, 0], [1.1, 0, 0]], vel=[[1, 0, 0], [1, 0, 0]]), tol=1e-6, tol_err=1e-6),
    # interpolated range, SE(2): 3 ranges at tau = 0.05, 0.25, 0.45 from camera poses on the line
    dict(src=SL + "testGPInterpolatedRangeFactorPose2.cpp:200-283", kind="pose2", dt=0.5, qc=0.01, landmark_dim=2,
         init=dict(pose=[[0.1, 0.1, -0.1], [5.1, -0.1, 0.1]], vel=[[9.8, 0, 0.2], [10.2, 0, -0.1]],
                   land=[[2.3, 3.1]]),
         pose_priors=[dict(idx=0, prior=[0, 0, 0], sigma=0.01), dict(idx=1, prior=[5, 0, 0], sigma=0.01)],
         vel_priors=[dict(idx=0, prior=[10, 0, 0], sigma=0.01), dict(idx=1, prior=[10, 0, 0], sigma=0.01)],
         land_priors=[dict(idx=0, prior=[2.4, 3.2], sigma=0.1)],
         ranges=[dict(tau=0.05, cam=[0.5, 0, 0]), dict(tau=0.25, cam=[2.5, 0, 0]), dict(tau=0.45, cam=[4.5, 0, 0])],
         range_sigma=0.1,
         expect=dict(pose=[[0, 0, 0], [5, 0, 0]], vel=[[10, 0, 0], [10, 0, 0]], land=[[2.4, 3.2]]), tol=1e-4,
         tol_err=1e-4),
    # interpolated range, SE(3): tau = -0.1, 0.05, 0.2 with delta_t = 0.1 (extrapolation on both sides)
    dict(src=SL + "testGPInterpolatedRangeFactorPose3.cpp:177-260", kind="pose3", dt=0.1, qc=0.01, landmark_dim=3,
         init=dict(pose=[P3(0.1, 0.2, 0.4, 0.2, 0.3, -0.2), P3(-0.1, -0.2, -0.4, 1.2, -0.3, 0.2)],
                   vel=[[-0.1, 0, 0, 0.8, 0, 0.2], [0, 0, 0.2, 1.2, 0, -0.1]], land=[[0.3, 1.1, 2.9]]),
         pose_priors=[dict(idx=0, prior=P3(0, 0, 0, 0, 0, 0), sigma=0.01),
                      dict(idx=1, prior=P3(0, 0, 0, 1, 0, 0), sigma=0.01)],
         vel_priors=[dict(idx=0, prior=[0, 0, 0, 10, 0, 0], sigma=0.01),
                     dict(idx=1, prior=[0, 0, 0, 10, 0, 0], sigma=0.01)],
         land_priors=[dict(idx=0, prior=[0.4, 1.2, 3], sigma=0.1)],
         ranges=[dict(tau=-0.1, cam=P3(0, 0, 0, -1, 0, 0)), dict(tau=0.05, cam=P3(0, 0, 0, 0.5, 0, 0)),
                 dict(tau=0.2, cam=P3(0, 0, 0, 2, 0, 0))],
         range_sigma=0.1,
         expect=dict(pose=[P3(0, 0, 0, 0, 0, 0), P3(0, 0, 0, 1, 0, 0)],
                     vel=[[0, 0, 0, 10, 0, 0], [0, 0, 0, 10, 0, 0]], land=[[0.4, 1.2, 3]]), tol=1e-6,
         tol_err=1e-6),
    # interpolated range, 2D linear with theta offset 10*pi (:250)
    dict(src=SL + "testGPInterpolatedRangeFactor2DLinear.cpp:228-317", kind="linear3", dt=0.5, qc=0.01,
         landmark_dim=2,
         init=dict(pose=[[0.1, 0.1, 10 * PI - 0.1], [5.1, -0.1, 10 * PI + 0.1]],
                   vel=[[9.8, 0, 0.2], [10.2, 0, -0.1]], land=[[2.3, 3.1]]),
         pose_priors=[dict(idx=0, prior=[0, 0, 10 * PI], sigma=0.01), dict(idx=1, prior=[5, 0, 10 * PI], sigma=0.01)],
         vel_priors=[dict(idx=0, prior=[10, 0, 0], sigma=0.01), dict(idx=1, prior=[10, 0, 0], sigma=0.01)],
         land_priors=[dict(idx=0, prior=[2.4, 3.2], sigma=0.1)],
         ranges=[dict(tau=0.05, cam=[0.5, 0, 10 * PI]), dict(tau=0.25, cam=[2.5, 0, 10 * PI]),
                 dict(tau=0.45, cam=[4.5, 0, 10 * PI])],
         range_sigma=0.1,
         expect=dict(pose=[[0, 0, 10 * PI], [5, 0, 10 * PI]], vel=[[10, 0, 0], [10, 0, 0]], land=[[2.4, 3.2]]),
         tol=1e-4, tol_err=1e-4),
]

out = dict(
    _about="Inputs/expected values transcribed from gtrll/gpslam's own unit tests; see transcribe_reference_tests.py",
    gp_prior=gp_prior, vw_conversion=vw_conversion, interp_gps=interp_gps, gps_optimization=gps_optimization, interp_gps_vw=interp_gps_vw,
    interp_projection=interp_projection, projection_optimization=projection_optimization,
    gp_prior_vw=gp_prior_vw, interpolator_vw=interpolator_vw, interpolator=interpolator, interp_range=interp_range, range2d=range2d,
    bearing_range2d=bearing_range2d, odometry2d=odometry2d, body_centric_velocity=body_centric_velocity,
    lie_jacobians=lie_jacobians, se3_velocity=se3_velocity, optimization=optimization)

if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
