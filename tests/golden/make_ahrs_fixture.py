#!/usr/bin/env python3
"""Convert the reference's AHRS log (DATA files matlab/data/RAW_IMU_DATA_matlab.txt and MOCAP_POSE_DATA_matlab.txt, loaded
by matlab/GPAHRSexample.m:45-49) into tests/golden/ahrs_imu.npz.  Run in the build container, where /root/reference exists:

    python tests/golden/make_ahrs_fixture.py

Arrays (column meaning documented at matlab/GPAHRSexample.m:43-46):
  IMU    (n, 8)  seq, time [s], gyro x y z [rad/s], acc x y z [m/s^2]
  MOCAP  (m, 9)  seq, time [s], position x y z, orientation quaternion x y z w
Rows up to 52 s are kept: the script processes datasetMaxTime = 50 s.
"""
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/matlab/data"
imu = np.loadtxt(os.path.join(src, "RAW_IMU_DATA_matlab.txt"))
mocap = np.loadtxt(os.path.join(src, "MOCAP_POSE_DATA_matlab.txt"))
imu = imu[imu[:, 1] <= 52.0]
mocap = mocap[mocap[:, 1] <= 52.0]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ahrs_imu.npz")
np.savez_compressed(out, IMU=imu.astype(np.float64), MOCAP=mocap.astype(np.float64))
print(out, os.path.getsize(out), "bytes", imu.shape, mocap.shape)
