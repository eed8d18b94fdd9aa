"""gpslam_amd -- MI355X-native GP-SLAM Gauss-Newton / Levenberg-Marquardt inner loop.

Host-side Python mirror of the C ABI in include/gpslam_hip.h (the C++ mirror with the gpslam/GTSAM class
names lives in gpslam_amd/host/).  All compute happens in gpslam_amd/lib/libgpslam_hip.so (hand-written HIP
for gfx950); importing this package never falls back to a CPU implementation.
"""
from .chain import (ChainSolver, GpslamHipError, LINEAR2, LINEAR3, POSE2, POSE3, ROT3, ROT3_BIAS, CHART_EXPMAP,
                    CHART_FIRST_ORDER, FP32, FP64, POSE_DIM, TANGENT_DIM, Params, Stats, load_library, PLAN_UNFUSED_LEVEL0,
                    PLAN_COLUMN_LEVEL0, PLAN_LEVELS_OF_FOUR, PLAN_FS_TWO_LAUNCHES, PLAN_GP_ROWS, PLAN_GENERIC_QC, PLAN_MEAS_ROWS, PLAN_SEPARATE_RETRACT)

__all__ = ["ChainSolver", "GpslamHipError", "LINEAR2", "LINEAR3", "POSE2", "POSE3", "ROT3", "ROT3_BIAS", "CHART_EXPMAP",
           "CHART_FIRST_ORDER", "FP32", "FP64", "POSE_DIM", "TANGENT_DIM", "Params", "Stats", "load_library", "PLAN_UNFUSED_LEVEL0",
           "PLAN_COLUMN_LEVEL0", "PLAN_LEVELS_OF_FOUR", "PLAN_FS_TWO_LAUNCHES", "PLAN_GP_ROWS", "PLAN_GENERIC_QC", "PLAN_MEAS_ROWS", "PLAN_SEPARATE_RETRACT"]
