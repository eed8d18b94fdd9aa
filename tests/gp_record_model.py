"""numpy model of the structured GP-prior record of round 4 (gpslam_amd/csrc/kernels.hpp, kGps*) and of what the assembly wave
of k_fused_level0 does with it, lane by lane: record -> the 24 columns of the whitened 12 x 24 Jacobian [L | R] of one
GaussianProcessPriorPose3 (gpslam/gp/GaussianProcessPriorPose3.h:60-98).  The index expressions mirror the kernel's
(`oX1`, `oJ2`, ..., `fmac_mat<N, M0, S>`), so that a slip in either shows up in tests/test_gp_record_model.py on the CPU."""
import numpy as np

LEN, XA, XC, JA, JC, FA, FC, FD, Z, E, S = 80, 0, 9, 18, 27, 36, 45, 54, 63, 64, 76


def make_record(X, J, F, ew, dt):
    """What K1 (gp_pose3_record) stores: X = Jinv, J = -Jinv Ad(h^-1), F = the finite-difference block, ew = whitened error."""
    sq = np.sqrt(dt)
    sa, sb, sc = np.sqrt(12.0) / (dt * sq), -np.sqrt(3.0) / sq, 1.0 / sq
    rec = np.zeros(LEN)
    rec[XA:XA + 9] = X[:3, :3].ravel(); rec[XC:XC + 9] = X[3:, :3].ravel()
    rec[JA:JA + 9] = J[:3, :3].ravel(); rec[JC:JC + 9] = J[3:, :3].ravel()
    rec[FA:FA + 9] = F[:3, :3].ravel(); rec[FC:FC + 9] = F[3:, :3].ravel(); rec[FD:FD + 9] = F[3:, 3:].ravel()
    rec[E:E + 12] = ew
    rec[S:S + 4] = [-(sa * dt + sb), sb, sc, sa]
    return rec


def fmac_mat(d, d0, Mr, n, m0, stride, m):
    """dpp.hpp fmac_mat<N, M0, S>: d[d0 + i] += M(m0 + i * stride) * m, matrix element e in lane (e & 15) of register Mr[e >> 4]"""
    for i in range(n):
        e = m0 + i * stride
        d[d0 + i] += Mr[e >> 4][e & 15] * m


def lane_columns(rec, U):
    """(L, R): 12 x 12 each; column c as lane c of the DPP row computes it."""
    Ud = np.zeros(48); Ud[:36] = U.ravel()
    # one register per lane: lane l of Ur[k] holds Ud[min(16 k + l, 35)]
    Ur = [[Ud[min(16 * k + l, 35)] for l in range(16)] for k in range(3)]
    fa = [[rec[FA + min(l, 8)] for l in range(16)]]
    fc = [[rec[FC + min(l, 8)] for l in range(16)]]
    fd = [[rec[FD + min(l, 8)] for l in range(16)]]
    L = np.zeros((12, 12)); R = np.zeros((12, 12)); new = np.zeros(12)
    for r in range(12):
        r6 = r if r < 6 else r - 6
        velc, hi3 = r >= 6, r6 >= 3
        j3 = r6 - 3 if hi3 else r6
        oX1, oX2 = XA + j3, (XA if hi3 else XC) + j3
        oJ1, oJ2 = JA + j3, (JA if hi3 else JC) + j3
        oaL, oaR = (S + 0 if velc else S + 3), (S + 1 if velc else S + 3)
        ob, oc, od = (Z if velc else S + 1), (Z if velc else S + 2), (S + 2 if velc else Z)
        raw = np.zeros(21)
        for k in range(3):
            raw[k] = rec[oX1 + 3 * k]; raw[3 + k] = rec[oX2 + 3 * k]; raw[6 + k] = rec[oJ1 + 3 * k]; raw[9 + k] = rec[oJ2 + 3 * k]
        raw[15] = rec[E + min(r, 11)]
        raw[16], raw[17], raw[18], raw[19], raw[20] = rec[oaL], rec[oaR], rec[ob], rec[oc], rec[od]
        tcol, vcol = (3 <= r < 6) or r >= 9, r >= 6
        X6 = np.zeros(6); J6 = np.zeros(6)
        for k in range(3):
            X6[k] = 0.0 if tcol else raw[k]; X6[3 + k] = raw[3 + k]
            J6[k] = (1.0 if r == 6 + k else 0.0) if vcol else (0.0 if tcol else raw[6 + k])
            J6[3 + k] = (1.0 if r == 9 + k else 0.0) if vcol else raw[9 + k]
        P3 = np.zeros(6); P1 = np.zeros(6)
        for j in range(3):
            fmac_mat(P3, 0, fa, 3, j, 3, X6[j]); fmac_mat(P3, 3, fc, 3, j, 3, X6[j]); fmac_mat(P3, 3, fd, 3, j, 3, X6[3 + j])
            fmac_mat(P1, 0, fa, 3, j, 3, J6[j]); fmac_mat(P1, 3, fc, 3, j, 3, J6[j]); fmac_mat(P1, 3, fd, 3, j, 3, J6[3 + j])
        aL, aR, bb, cc, dR = raw[16], raw[17], raw[18], raw[19], raw[20]
        Zs = np.zeros((4, 6))
        for k in range(6):
            Zs[0, k] = aL * J6[k] + bb * P1[k]
            Zs[1, k] = cc * P1[k] - dR * J6[k]
            Zs[2, k] = aR * X6[k] + bb * P3[k]
            Zs[3, k] = cc * P3[k] + dR * X6[k]
        Lc = np.zeros(12); Rc = np.zeros(12)
        for k in range(6):
            fmac_mat(Lc, 0, Ur, k + 1, k, 6, Zs[0, k]); fmac_mat(Lc, 6, Ur, k + 1, k, 6, Zs[1, k])
            fmac_mat(Rc, 0, Ur, k + 1, k, 6, Zs[2, k]); fmac_mat(Rc, 6, Ur, k + 1, k, 6, Zs[3, k])
        L[:, r] = Lc; R[:, r] = Rc; new[r] = -raw[15]
    return L, R, new


def reference_rows(X, J, F, U, dt):
    """The whitened Jacobian from its definition: R_w [H1 H2 | H3 H4], R_w = [[sa U, sb U], [0, sc U]]."""
    I6, O6 = np.eye(6), np.zeros((6, 6))
    H = np.block([[J, -dt * I6, X, O6], [F @ J, -I6, F @ X, X]])
    sq = np.sqrt(dt)
    sa, sb, sc = np.sqrt(12.0) / (dt * sq), -np.sqrt(3.0) / sq, 1.0 / sq
    Rw = np.block([[sa * U, sb * U], [O6, sc * U]])
    W = Rw @ H
    return W[:, :12], W[:, 12:]


# ---- BetweenFactor<Pose3> record (kBtw*): [RA RC LA LC | 1 / sigma | whitened error]
BLEN, BRA, BRC, BLA, BLC, BW, BE = 48, 0, 9, 18, 27, 36, 42


def make_between_record(H1, H2, w, e):
    rec = np.zeros(BLEN)
    rec[BRA:BRA + 9] = H2[:3, :3].ravel(); rec[BRC:BRC + 9] = H2[3:, :3].ravel()
    rec[BLA:BLA + 9] = H1[:3, :3].ravel(); rec[BLC:BLC + 9] = H1[3:, :3].ravel()
    rec[BW:BW + 6] = w; rec[BE:BE + 6] = w * e
    return rec


def between_lane_columns(rec):
    """(L, R): 6 x 6 each, column c as lane c < 6 of the DPP row computes it (k_fused_level0, the record's six compact rows)."""
    L = np.zeros((6, 6)); R = np.zeros((6, 6))
    for r in range(6):
        hi3 = r >= 3
        j3 = r - 3 if hi3 else r
        oX1, oX2 = XA + j3, (XA if hi3 else XC) + j3          # the GP record's column walk, rebased
        braw = np.zeros(14)
        for k in range(3):
            braw[k] = rec[BRA - XA + oX1 + 3 * k]; braw[3 + k] = rec[BRA - XA + oX2 + 3 * k]
            braw[6 + k] = rec[BLA - XA + oX1 + 3 * k]; braw[9 + k] = rec[BLA - XA + oX2 + 3 * k]
        tcol = r >= 3
        Rc6 = np.zeros(6); Lc6 = np.zeros(6)
        for k in range(3):
            Rc6[k] = 0.0 if tcol else braw[k]; Rc6[3 + k] = braw[3 + k]
            Lc6[k] = 0.0 if tcol else braw[6 + k]; Lc6[3 + k] = braw[9 + k]
        for i in range(6):
            wi = rec[BW + i]                                   # row_bcast<i> of the lanes' weights
            L[i, r] = wi * Lc6[i]; R[i, r] = wi * Rc6[i]
    return L, R


# ---- d = 3 manifolds (SE(2), SO(3), 3-D linear chains): the 32-double record kGp3* and the six rows its consumers form from it
def whitening_scalars(dt):
    """chol_upper(T^-1), T^-1 = [[12 / dt^3, -6 / dt^2], [-6 / dt^2, 4 / dt]] = [[sa, sb], [0, sc]]"""
    sq = np.sqrt(dt)
    return 3.4641016151377545870548926830117 / (dt * sq), -1.7320508075688772935274463415059 / sq, 1.0 / sq


def make_record3(J1, J3, h2t, h2b, h4b, U, e, dt):
    """H1 = [J1; 0], H2 = [h2t I; h2b I], H3 = [J3; 0], H4 = [0; h4b I]; e = the factor's 6-vector error (unwhitened)."""
    sa, sb, sc = whitening_scalars(dt)
    rec = np.zeros(32)
    rec[0:9] = (U @ (sa * J1)).ravel()
    rec[9:18] = (U @ (sa * J3)).ravel()
    rec[18:21] = U @ (sa * e[:3] + sb * e[3:])
    rec[21:24] = sc * (U @ e[3:])
    rec[24:28] = [sa * h2t + sb * h2b, sb * h4b, sc * h2b, sc * h4b]
    return rec


def rows_from_record3(rec, U):
    """The six whitened rows [L | R] (6 x 6 each) as k_assemble_ghost<6> / k_fused_level0<1, double, 6> form them, lane c < 6 holding
    column c: a pose lane (c < 3) its column of A1 / A3, a velocity lane column c - 3 of U scaled by the record's coefficients."""
    A1, A3 = rec[0:9].reshape(3, 3), rec[9:18].reshape(3, 3)
    kLt, kRt, kLb, kRb = rec[24:28]
    L, R = np.zeros((6, 6)), np.zeros((6, 6))
    for c in range(6):
        pc = c < 3
        c3 = c if pc else c - 3
        colL = A1[:, c3] if pc else U[:, c3]
        colR = A3[:, c3] if pc else U[:, c3]
        mLt, mRt, mLb, mRb = (1.0, 1.0, 0.0, 0.0) if pc else (kLt, kRt, kLb, kRb)
        for i in range(6):
            L[i, c] = (mLt if i < 3 else mLb) * colL[i % 3]
            R[i, c] = (mRt if i < 3 else mRb) * colR[i % 3]
    return L, R, rec[18:24].copy()


def reference_rows3(J1, J3, h2t, h2b, h4b, U, e, dt):
    """R_w [H1 H2 | H3 H4] and R_w e with R_w = chol_upper(Q^-1) = [[sa U, sb U], [0, sc U]]"""
    sa, sb, sc = whitening_scalars(dt)
    Z, I = np.zeros((3, 3)), np.eye(3)
    Rw = np.block([[sa * U, sb * U], [Z, sc * U]])
    Hl = np.block([[J1, h2t * I], [Z, h2b * I]])
    Hr = np.block([[J3, Z], [Z, h4b * I]])
    return Rw @ Hl, Rw @ Hr, Rw @ e
