"""numpy model of the two phases of the segment-sharded Gauss-Newton iteration (test infrastructure).

It produces and consumes interface records in exactly the layout of the HIP library
([D | C | G | RD | Rg], see gpslam_amd/csrc/kernels.hpp k_iface_build), using the CPU oracle for the local
normal equations, so that (a) the host orchestration in gpslam_amd/sharded.py can be exercised over gloo with
world_size 2 on a machine without GPUs and (b) the HIP records can be compared against it on the GPU box.
"""
import numpy as np
import torch

from oracle import oracle as O


class Stats:
    def __init__(self, eb, ea, d):
        self.error_before, self.error_after, self.delta_inf_norm = eb, ea, d


class SegmentModel:
    def __init__(self, kind, rank, nranks, chart=O.CHART_EXPMAP):
        self.kind, self.rank, self.P, self.chart = kind, rank, nranks, chart
        self.d, self.pd = O.TANGENT_DIM[kind], O.POSE_DIM[kind]
        self.b = 2 * self.d
        self.has_right = rank < nranks - 1
        self.halo_pose = self.halo_vel = None
        self.fac = []
        b = self.b
        self.BS, self.AS = 2 * b * b + b, b * b + b
        self.send = torch.zeros(self.BS + self.AS, dtype=torch.float64)
        self.recv = torch.zeros(nranks * (self.BS + self.AS), dtype=torch.float64)

    # ---- ChainSolver-like construction surface
    def set_qc(self, Qc):
        self.Qc = np.array(Qc, dtype=np.float64)

    def set_states(self, pose, vel):
        self.pose, self.vel = np.array(pose, dtype=np.float64), np.array(vel, dtype=np.float64)
        self.N = len(self.pose)

    def set_halo_state(self, pose, vel):
        self.halo_pose, self.halo_vel = np.array(pose, dtype=np.float64), np.array(vel, dtype=np.float64)

    def add_gp_priors(self, left, dt):
        self.fac.append(("add_gp_priors", (np.array(left), np.array(dt))))

    def add_pose_priors(self, idx, prior, sig):
        self.fac.append(("add_pose_priors", (np.array(idx), np.array(prior), np.array(sig))))

    def add_vel_priors(self, idx, prior, sig):
        self.fac.append(("add_vel_priors", (np.array(idx), np.array(prior), np.array(sig))))

    def add_between(self, left, meas, sig):
        self.fac.append(("add_between", (np.array(left), np.array(meas), np.array(sig))))

    def compile(self):
        return 0

    def get_states(self):
        return self.pose.copy(), self.vel.copy()

    # ---- the oracle chain of the segment plus its halo state
    def _chain(self):
        ch = O.Chain(self.kind, self.chart)
        ch.set_qc(self.Qc)
        if self.has_right:
            ch.set_states(np.vstack([self.pose, self.halo_pose[None]]), np.vstack([self.vel, self.halo_vel[None]]))
        else:
            ch.set_states(self.pose, self.vel)
        for name, args in self.fac:
            getattr(ch, name)(*args)
        return ch

    def _linearise(self):
        """the segment's normal equations at the current values: H0 (undamped), rhs, error"""
        b, N = self.b, self.N
        ch = self._chain()
        self._err_before = ch.error()
        D, Ocp, g, _, _, _ = ch.normal_equations()
        n = N * b
        H = np.zeros((n + b, n + b))
        rhs = np.zeros(n + b)
        for i in range(len(D)):
            H[i * b:(i + 1) * b, i * b:(i + 1) * b] = D[i]
            rhs[i * b:(i + 1) * b] = g[i]
            if i + 1 < len(D):
                H[(i + 1) * b:(i + 2) * b, i * b:(i + 1) * b] = Ocp[i]
                H[i * b:(i + 1) * b, (i + 1) * b:(i + 2) * b] = Ocp[i].T
        self._H0, self._rhs = H, rhs

    def _reduce(self, lam):
        """eliminate the interior of the (damped) segment -> interface record"""
        b, N = self.b, self.N
        n = N * b
        H = self._H0.copy()
        rhs = self._rhs
        H[:n, :n] += lam * np.eye(n)
        I = slice(b, n)               # interior
        S = [slice(0, b), slice(n, n + b)]   # separator, halo
        HII = H[I, I]
        self._HII, self._H, self._I, self._S = HII, H, I, S
        if n > b:
            sol = np.linalg.solve(HII, np.column_stack([H[I, S[0]], H[I, S[1]], rhs[I]]))
            W0, W1, y = sol[:, :b], sol[:, b:2 * b], sol[:, 2 * b]
        else:
            W0 = W1 = np.zeros((0, b))
            y = np.zeros(0)
        Drec = H[S[0], S[0]] - H[S[0], I] @ W0
        Grec = rhs[S[0]] - H[S[0], I] @ y
        C = H[S[1], S[0]] - H[S[1], I] @ W0          # row: next separator, column: this separator
        RD = H[S[1], S[1]] - H[S[1], I] @ W1
        Rg = rhs[S[1]] - H[S[1], I] @ y
        if not self.has_right:
            C[:] = 0
            RD[:] = 0
            Rg[:] = 0
        rec = np.concatenate([Drec.ravel(), C.ravel(), Grec, RD.ravel(), Rg])
        self.send.copy_(torch.from_numpy(rec))

    def iterate_phase1(self, lam=0.0):
        self._linearise()
        self._reduce(lam)

    # ---- Levenberg-Marquardt trial steps (the caller owns the loop: gpslam_amd/sharded.py, ShardedSolver.iterate_lm; the library's
    # own steps are gpslam_hip_lm_begin / lm_trial_phase1 / iterate_phase2a / lm_trial_phase2 / lm_reject)
    def lm_begin(self):
        self._linearise()
        self._backup = (self.pose.copy(), self.vel.copy(), None if self.halo_pose is None else self.halo_pose.copy(),
                        None if self.halo_vel is None else self.halo_vel.copy())

    def lm_trial_phase1(self, lam):
        self._reduce(lam)

    def iterate_phase2a(self):
        return 0

    def lm_trial_phase2(self):
        """[error at the linearisation point, trial error, |delta|_inf, delta . g, |delta|^2, indefinite flag] of this rank"""
        x, xh = self._solve_and_retract()
        n = self.N * self.b
        dg = float(x.ravel() @ self._rhs[:n]) + (float(xh @ self._rhs[n:]) if self.has_right else 0.0)
        return np.array([self._err_before, self._chain().error(), float(np.abs(x).max()), dg, float(x.ravel() @ x.ravel()), 0.0])

    def lm_reject(self):
        self.pose, self.vel = self._backup[0].copy(), self._backup[1].copy()
        if self._backup[2] is not None:
            self.halo_pose, self.halo_vel = self._backup[2].copy(), self._backup[3].copy()

    def _solve_and_retract(self):
        b, N, P = self.b, self.N, self.P
        RS = self.BS + self.AS
        rec = self.recv.numpy().reshape(P, RS)
        T = np.zeros((P * b, P * b))
        tr = np.zeros(P * b)
        for r in range(P):
            Dm = rec[r, :b * b].reshape(b, b).copy()
            Cm = rec[r, b * b:2 * b * b].reshape(b, b)
            Gm = rec[r, 2 * b * b:2 * b * b + b].copy()
            if r > 0:
                Dm += rec[r - 1, self.BS:self.BS + b * b].reshape(b, b)
                Gm += rec[r - 1, self.BS + b * b:]
            T[r * b:(r + 1) * b, r * b:(r + 1) * b] = Dm
            tr[r * b:(r + 1) * b] = Gm
            if r + 1 < P:
                T[(r + 1) * b:(r + 2) * b, r * b:(r + 1) * b] = Cm
                T[r * b:(r + 1) * b, (r + 1) * b:(r + 2) * b] = Cm.T
        xt = np.linalg.solve(T, tr).reshape(P, b)
        x0 = xt[self.rank]
        xh = xt[self.rank + 1] if self.has_right else np.zeros(b)
        I, S = self._I, self._S
        n = N * b
        x = np.zeros(n)
        x[:b] = x0
        if n > b:
            x[b:] = np.linalg.solve(self._HII, self._rhs[I] - self._H[I, S[0]] @ x0 - self._H[I, S[1]] @ xh)
        x = x.reshape(N, b)
        for i in range(N):
            self.pose[i] = O.retract(self.kind, self.pose[i], x[i, :self.d], self.chart)
            self.vel[i] += x[i, self.d:]
        if self.has_right:
            self.halo_pose = O.retract(self.kind, self.halo_pose, xh[:self.d], self.chart)
            self.halo_vel = self.halo_vel + xh[self.d:]
        return x, xh

    def iterate_phase2(self, want_stats=True):
        x, _xh = self._solve_and_retract()
        err_after = self._chain().error()
        return Stats(self._err_before, err_after, float(np.abs(x).max()))
