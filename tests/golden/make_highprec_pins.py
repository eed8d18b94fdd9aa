#!/usr/bin/env python3
"""Independent high-precision pins (mpmath, 50 digits) for what the reference's own tests do not pin (SURVEY.md section 7
step 1, section 8(c)): SE(3) / SO(3) / SE(2) Exp, Log and their right Jacobians, the derivative d(Jr^-1(xi) x)/d xi that
the reference approximates by an h = 1e-6 finite difference, the Unit3 tangent basis + attitudeError arithmetic, and the
whitening factor R = chol_upper(Q^-1(dt)) for a non-diagonal Qc.

Nothing here uses the oracle or the library: group elements come from the matrix exponential of the Lie algebra element,
Jacobians from the defining power series  Jr(xi) = sum_k (-ad_xi)^k / (k + 1)!  (GTSAM's right Jacobians, tangent order
(omega, v) for SE(3), (vx, vy, omega) for SE(2)), inverses by matrix inversion.  Output: tests/golden/highprec_pins.json.

    python tests/golden/make_highprec_pins.py
"""
import json
import os

import mpmath as mp

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def skew(w):
    return mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def series_jr(ad, terms=80):
    """sum_k (-ad)^k / (k + 1)!"""
    n = ad.rows
    J, P = mp.eye(n), mp.eye(n)
    for k in range(1, terms):
        P = P * (-ad) / (k + 1)
        J += P
    return J


def fl(M):
    if isinstance(M, mp.matrix):
        return [[float(M[i, j]) for j in range(M.cols)] for i in range(M.rows)]
    return [float(x) for x in M]


def so3_case(w):
    w = [mp.mpf(x) for x in w]
    W = skew(w)
    R = mp.expm(W)
    ad = W                                       # ad of so(3) = the skew matrix itself
    Jr = series_jr(ad)
    return dict(w=fl(w), R=fl(R), Jr=fl(Jr), Jrinv=fl(Jr ** -1))


def se3_hat(xi):
    w, v = xi[:3], xi[3:]
    M = mp.zeros(4, 4)
    W = skew(w)
    for i in range(3):
        for j in range(3):
            M[i, j] = W[i, j]
        M[i, 3] = v[i]
    return M


def se3_ad(xi):
    w, v = xi[:3], xi[3:]
    A = mp.zeros(6, 6)
    W, V = skew(w), skew(v)
    for i in range(3):
        for j in range(3):
            A[i, j] = W[i, j]
            A[3 + i, j] = V[i, j]
            A[3 + i, 3 + j] = W[i, j]
    return A


def se3_jrinv_times(xi, x):
    return (series_jr(se3_ad(xi)) ** -1) * mp.matrix(x)


def se3_case(xi, x):
    xi = [mp.mpf(v) for v in xi]
    x = [mp.mpf(v) for v in x]
    T = mp.expm(se3_hat(xi))
    Jr = series_jr(se3_ad(xi))
    Jrinv = Jr ** -1
    # exact derivative of Jr^-1(xi) x with respect to xi: central difference at h = 1e-20 in 50-digit arithmetic
    h = mp.mpf(10) ** -20
    D = mp.zeros(6, 6)
    for k in range(6):
        xp, xm = list(xi), list(xi)
        xp[k] += h
        xm[k] -= h
        col = (se3_jrinv_times(xp, x) - se3_jrinv_times(xm, x)) / (2 * h)
        for i in range(6):
            D[i, k] = col[i]
    R = [[T[i, j] for j in range(3)] for i in range(3)]
    return dict(xi=fl(xi), x=fl(x), pose=[float(v) for row in R for v in row] + [float(T[i, 3]) for i in range(3)],
                Jr=fl(Jr), Jrinv=fl(Jrinv), dJrinv_x=fl(D))


def se2_case(xi):
    vx, vy, om = [mp.mpf(v) for v in xi]
    M = mp.matrix([[0, -om, vx], [om, 0, vy], [0, 0, 0]])
    T = mp.expm(M)
    theta = mp.atan2(T[1, 0], T[0, 0])
    ad = mp.matrix([[0, -om, vy], [om, 0, -vx], [0, 0, 0]])
    Jr = series_jr(ad)
    return dict(xi=fl([vx, vy, om]), pose=[float(T[0, 2]), float(T[1, 2]), float(theta)], dexp=fl(Jr), dlog=fl(Jr ** -1))


def unit3_basis(n):
    """GTSAM Unit3::basis as SURVEY Appendix A recalls it: cross the direction with the axis of its smallest |component|,
    normalise -> b1; b2 = n x b1.  (The RULE is unpinned by the reference; this pins the arithmetic that follows from it.)"""
    n = [mp.mpf(v) for v in n]
    nn = mp.sqrt(sum(v * v for v in n))
    n = [v / nn for v in n]
    k = min(range(3), key=lambda i: abs(n[i]))
    axis = [mp.mpf(1) if i == k else mp.mpf(0) for i in range(3)]
    cr = lambda a, b: [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
    b1 = cr(n, axis)
    l = mp.sqrt(sum(v * v for v in b1))
    b1 = [v / l for v in b1]
    b2 = cr(n, b1)
    return n, b1, b2


def attitude_case(w, nZ, bRef):
    R = mp.expm(skew([mp.mpf(v) for v in w]))
    nz, z1, z2 = unit3_basis(nZ)
    bn, _, _ = unit3_basis(bRef)
    q = R * mp.matrix(bn)
    qn = mp.sqrt(sum(q[i] * q[i] for i in range(3)))
    q = [q[i] / qn for i in range(3)]
    e = [sum(z1[i] * q[i] for i in range(3)), sum(z2[i] * q[i] for i in range(3))]
    return dict(w=fl([mp.mpf(v) for v in w]), R=fl(R), nZ=fl(nz), bRef=fl(bn), basis=[fl(z1), fl(z2)], error=fl(e))


def whitening_case(Qc, dt):
    d = len(Qc)
    Qc = mp.matrix(Qc)
    dt = mp.mpf(dt)
    Q = mp.zeros(2 * d, 2 * d)
    for i in range(d):
        for j in range(d):
            Q[i, j] = dt ** 3 / 3 * Qc[i, j]
            Q[i, d + j] = Q[d + i, j] = dt ** 2 / 2 * Qc[i, j]
            Q[d + i, d + j] = dt * Qc[i, j]
    L = mp.cholesky(Q ** -1)                       # lower: Q^-1 = L L^T  ->  R = L^T upper, R^T R = Q^-1
    return dict(Qc=fl(Qc), dt=float(dt), R=fl(L.T))


def so3_log(R):
    th = mp.acos((R[0, 0] + R[1, 1] + R[2, 2] - 1) / 2)
    k = th / (2 * mp.sin(th)) if th != 0 else mp.mpf(1) / 2
    return [k * (R[2, 1] - R[1, 2]), k * (R[0, 2] - R[2, 0]), k * (R[1, 0] - R[0, 1])]


def ahrs_case(wi, wj, bias, bias_hat, samples, gyro_cov, coriolis):
    """gtsam::PreintegratedAhrsMeasurements::integrateMeasurement + AHRSFactor::evaluateError (GTSAM 4.0, published
    algorithm; see oracle/orc_factors.c), value in 50 digits, Jacobians by central differences at h = 1e-20 under the
    right perturbations Ri Exp(d), Rj Exp(d), bias + d -- independent of the closed-form Jacobian chain."""
    M = lambda v: [mp.mpf(x) for x in v]
    wi, wj, bias, bias_hat, coriolis = M(wi), M(wj), M(bias), M(bias_hat), M(coriolis)
    gc = mp.matrix(gyro_cov)
    dR, D, dtij, cov = mp.eye(3), mp.zeros(3, 3), mp.mpf(0), mp.zeros(3, 3)
    for om, dt in samples:
        dt = mp.mpf(dt)
        th = [(mp.mpf(om[i]) - bias_hat[i]) * dt for i in range(3)]
        incr = mp.expm(skew(th))
        dtij += dt
        dR = dR * incr
        D = incr.T * D - series_jr(skew(th)) * dt
        cov = incr.T * cov * incr + gc * dt

    def fR(di, dj, db):
        Ri = mp.expm(skew(wi)) * mp.expm(skew(di))
        Rj = mp.expm(skew(wj)) * mp.expm(skew(dj))
        binc = mp.matrix([bias[i] + db[i] - bias_hat[i] for i in range(3)])
        bio = D * binc
        om = so3_log(dR * mp.expm(skew([bio[i] for i in range(3)])))
        cor = Ri.T * mp.matrix(coriolis) * dtij
        com = [om[i] - cor[i] for i in range(3)]
        return so3_log(mp.expm(skew(com)).T * Ri.T * Rj)

    z = [mp.mpf(0)] * 3
    e = fR(z, z, z)
    h = mp.mpf(10) ** -20
    H = []
    for which in range(3):
        Hm = mp.zeros(3, 3)
        for k in range(3):
            dp = [h if i == k else mp.mpf(0) for i in range(3)]
            dm = [-x for x in dp]
            args_p = [z, z, z]
            args_m = [z, z, z]
            args_p[which], args_m[which] = dp, dm
            fp, fm = fR(*args_p), fR(*args_m)
            for i in range(3):
                Hm[i, k] = (fp[i] - fm[i]) / (2 * h)
        H.append(Hm)
    return dict(Ri=fl(mp.expm(skew(wi))), Rj=fl(mp.expm(skew(wj))), bias=fl(bias), bias_hat=fl(bias_hat),
                samples=[[[float(x) for x in om], float(dt)] for om, dt in samples], gyro_cov=gyro_cov, coriolis=fl(coriolis),
                delta_R=fl(dR), dR_dbias=fl(D), delta_tij=float(dtij), cov=fl(cov), e=fl(e), H1=fl(H[0]), H2=fl(H[1]), H3=fl(H[2]))


def main():
    pins = dict(
        note="mpmath %s, %d digits; generated by tests/golden/make_highprec_pins.py" % (mp.__version__, mp.mp.dps),
        so3=[so3_case(w) for w in ([1e-9, -2e-9, 3e-9], [1e-4, 2e-4, -1e-4], [0.01, -0.02, 0.03], [0.3, -0.2, 0.5],
                                    [1.0, 0.5, -0.8], [2.0, -1.5, 1.2], [1.8, 1.8, 1.7])],
        se3=[se3_case(xi, x) for xi, x in (
            ([1e-7, -2e-7, 1e-7, 0.1, 0.2, -0.3], [0.1, -0.2, 0.3, 1.0, 0.5, -0.4]),
            ([0.01, -0.02, 0.03, 0.1, 0.0, 0.05], [0.3, 0.1, -0.2, 1.0, -0.3, 0.2]),
            ([0.03, 0.01, 0.09, 0.1, 0.02, -0.01], [0.0, 0.0, 0.3, 1.0, 0.0, 0.0]),
            ([0.3, -0.2, 0.5, 1.0, -2.0, 0.5], [0.5, -0.3, 0.2, -1.0, 0.4, 2.0]),
            ([1.0, 0.5, -0.8, -0.5, 1.5, 2.0], [-0.2, 0.7, 0.1, 0.3, -0.6, 0.9]),
            ([2.0, -1.5, 1.2, 0.3, 0.6, -0.7], [0.4, 0.4, -0.4, 1.2, 0.2, -0.1]))],
        se2=[se2_case(xi) for xi in ([0.1, -0.2, 1e-12], [0.1, -0.2, 1e-6], [1.0, 0.5, 0.01], [1.0, 0.5, 0.7], [-2.0, 1.0, 2.5], [0.3, 0.3, -3.0])],
        attitude=[attitude_case(w, nz, b) for w, nz, b in (
            ([0.1, -0.2, 0.3], [0.0, 0.0, 1.0], [0.0, 0.0, 1.0]),
            ([0.5, 0.4, -0.3], [0.1, -0.2, 0.97], [0.0, 0.1, 1.0]),
            ([1.0, -1.0, 0.5], [0.7, 0.1, -0.7], [1.0, 0.0, 0.0]))],
        ahrs=[
            ahrs_case([0.1, -0.2, 0.3], [0.12, -0.18, 0.33], [0.002, -0.001, 0.0005], [0, 0, 0],
                      [([0.5, 0.6, 1.0], 0.006), ([0.45, 0.7, 0.9], 0.0065)], [[1e-3, 0, 0], [0, 1e-3, 0], [0, 0, 1e-3]], [0, 0, 0]),
            ahrs_case([1.0, 0.5, -0.8], [1.1, 0.3, -0.9], [0.05, 0.02, -0.03], [0.01, -0.01, 0.02],
                      [([1.0, -2.0, 0.5], 0.01), ([1.2, -1.8, 0.4], 0.012), ([0.9, -2.2, 0.7], 0.009)],
                      [[2e-3, 1e-4, 0], [1e-4, 1e-3, -2e-4], [0, -2e-4, 3e-3]], [0.01, -0.02, 0.03]),
            ahrs_case([0.0, 0.0, 0.0], [1e-4, -2e-4, 1e-4], [0, 0, 0], [0, 0, 0],
                      [([0.02, -0.03, 0.02], 0.005)], [[1e-3, 0, 0], [0, 1e-3, 0], [0, 0, 1e-3]], [0, 0, 7.29e-5])],
        whitening=[whitening_case([[0.01, 0.003, 0.0], [0.003, 0.02, -0.001], [0.0, -0.001, 0.015]], 0.1),
                   whitening_case([[0.01, 0.0], [0.0, 0.01]], 0.1),
                   whitening_case([[1.0, 0.2, 0.1, 0, 0, 0], [0.2, 2.0, 0.3, 0, 0, 0], [0.1, 0.3, 1.5, 0, 0, 0.1],
                                   [0, 0, 0, 0.5, 0.05, 0], [0, 0, 0, 0.05, 0.7, 0], [0, 0, 0.1, 0, 0, 0.9]], 0.25)],
    )
    with open(os.path.join(HERE, "highprec_pins.json"), "w") as f:
        json.dump(pins, f, indent=0)
    print("wrote", os.path.join(HERE, "highprec_pins.json"))


if __name__ == "__main__":
    main()
