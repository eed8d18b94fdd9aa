"""Shared helpers for the parity tests (oracle side = checker)."""
import numpy as np

from oracle import oracle as O

KIND = {"linear2": O.LINEAR2, "linear3": O.LINEAR3, "pose2": O.POSE2, "pose3": O.POSE3, "rot3": O.ROT3}


def dec_pose(kind, p):
    """golden pose encoding -> flat oracle/ABI layout"""
    k = KIND[kind] if isinstance(kind, str) else kind
    if k == O.POSE3:
        return O.pose3(p["ypr"], p["t"])
    if k == O.ROT3:
        return O.rot3_ypr(*p["ypr"])
    return O.A(p)


def numdiff_manifold(kind, f, x, h):
    """gtsam::numericalDerivative11 for a manifold argument: central difference along retract(x, h e_i)."""
    d = O.TANGENT_DIM[kind]
    cols = []
    for i in range(d):
        dx = np.zeros(d)
        dx[i] = h
        cols.append((np.atleast_1d(f(O.retract(kind, x, dx))) - np.atleast_1d(f(O.retract(kind, x, -dx)))) / (2 * h))
    return np.stack(cols, 1)


def numdiff_vector(f, x, h):
    x = np.asarray(x, dtype=np.float64)
    cols = []
    for i in range(len(x)):
        dx = np.zeros(len(x))
        dx[i] = h
        cols.append((np.atleast_1d(f(x + dx)) - np.atleast_1d(f(x - dx))) / (2 * h))
    return np.stack(cols, 1)


def pose_close(kind, a, b, tol):
    """assert_equal for poses: compare through local coordinates."""
    return float(np.abs(O.local(kind, a, b)).max()) <= tol


def wrap_pi(a):
    return np.arctan2(np.sin(a), np.cos(a))
