"""numpy model of one piece of a chain with locally visible landmarks split across GPUs (test infrastructure).

It speaks the ChainSolver surface gpslam_amd/sharded.py's SplitSolver uses (fs_set_split, fs_split_info, fs_set_top,
fs_interface, fs_phase1, fs_phase2) and produces / consumes interface records in exactly the layout of the HIP library
([Dff | H(last, first) | Dll | g_first | g_last] in blocks of nb_top, unit diagonal on the padding; include/gpslam_hip.h,
gpslam_hip_fs_set_split), with the CPU oracle for the piece's normal equations and dense numpy for the algebra: the host
orchestration runs over gloo with world_size 2 on a machine without GPUs.
"""
import numpy as np
import torch

from oracle import oracle as O


class Stats:
    def __init__(self, eb, ea, d):
        self.error_before, self.error_after, self.delta_inf_norm = eb, ea, d


class SplitPieceModel:
    def __init__(self, kind, chart=O.CHART_EXPMAP, landmark_dim=2):
        self.kind, self.chart, self.ld = kind, chart, landmark_dim
        self.d = O.TANGENT_DIM[kind]
        self.b = 2 * self.d
        self.fac = []
        self.rank, self.P, self.first_lm, self.last_lm = 0, 1, [], []
        self.nb_top = 0

    # ---- construction surface
    def set_qc(self, Qc):
        self.Qc = np.array(Qc, dtype=np.float64)

    def set_states(self, pose, vel):
        self.pose, self.vel = np.array(pose, dtype=np.float64), np.array(vel, dtype=np.float64)
        self.N = len(self.pose)

    def set_landmarks(self, pts):
        self.lmk = np.array(pts, dtype=np.float64)

    def __getattr__(self, name):
        if name.startswith("add_"):
            return lambda *args: self.fac.append((name, args))
        raise AttributeError(name)

    def compile(self):
        return 0

    def get_states(self):
        return self.pose.copy(), self.vel.copy()

    def get_landmarks(self):
        return self.lmk.copy()

    def fs_set_split(self, rank, nranks, first_lm=(), last_lm=()):
        self.rank, self.P, self.first_lm, self.last_lm = rank, nranks, list(first_lm), list(last_lm)

    def fs_split_info(self):
        nb = self.b + self.ld * max(len(self.first_lm), len(self.last_lm))
        return dict(fat_block=(nb + 3) & ~3, fat_blocks=2, segment_length=self.N, nb_top=self.nb_top)

    def fs_set_top(self, nb_top):
        assert nb_top >= self.fs_split_info()["fat_block"]
        self.nb_top = nb_top
        RS = 3 * nb_top * nb_top + 2 * nb_top
        self.send = torch.zeros(RS, dtype=torch.float64)
        self.recv = torch.zeros(self.P * RS, dtype=torch.float64)

    def fs_interface(self):
        return self.send, self.recv

    # ---- the oracle chain of the piece
    def _chain(self):
        ch = O.Chain(self.kind, self.chart, landmark_dim=self.ld)
        ch.set_qc(self.Qc)
        ch.set_states(self.pose, self.vel)
        ch.set_landmarks(self.lmk)
        for name, args in self.fac:
            getattr(ch, name)(*args)
        return ch

    def _ends(self):
        b, N, ld = self.b, self.N, self.ld
        first = list(range(b)) + [N * b + l * ld + q for l in self.first_lm for q in range(ld)]
        last = list(range((N - 1) * b, N * b)) + [N * b + l * ld + q for l in self.last_lm for q in range(ld)]
        return first, last

    def fs_phase1(self, lam=0.0):
        b, N, nl, NT = self.b, self.N, len(self.lmk) * self.ld, self.nb_top
        ch = self._chain()
        self._err_before = ch.error()
        D, Ocp, g, B, HLL, gL = ch.normal_equations()
        n = N * b + nl
        H = np.zeros((n, n))
        rhs = np.zeros(n)
        for i in range(N):
            H[i * b:(i + 1) * b, i * b:(i + 1) * b] = D[i]
            rhs[i * b:(i + 1) * b] = g[i]
            if i + 1 < N:
                H[(i + 1) * b:(i + 2) * b, i * b:(i + 1) * b] = Ocp[i]
                H[i * b:(i + 1) * b, (i + 1) * b:(i + 2) * b] = Ocp[i].T
        H[:N * b, N * b:] = B.reshape(N * b, nl)
        H[N * b:, :N * b] = B.reshape(N * b, nl).T
        H[N * b:, N * b:] = HLL
        rhs[N * b:] = gL
        first, last = self._ends()
        damp = np.full(n, lam)
        if self.rank < self.P - 1:
            damp[last] = 0.0                       # the neighbour on the right damps the shared block
        H += np.diag(damp)
        ends = first + last
        es = set(ends)
        inner = [i for i in range(n) if i not in es]
        Hie = H[np.ix_(inner, ends)]
        Hii = H[np.ix_(inner, inner)]
        sol = np.linalg.solve(Hii, np.column_stack([Hie, rhs[inner]]))
        S = H[np.ix_(ends, ends)] - Hie.T @ sol[:, :-1]
        sr = rhs[ends] - Hie.T @ sol[:, -1]
        self._keep = (H, rhs, inner, ends, Hii)
        nf, nla = len(first), len(last)
        Dff, Hlf, Dll = np.eye(NT), np.zeros((NT, NT)), np.eye(NT)
        Dff[:nf, :nf] = S[:nf, :nf]
        Dll[:nla, :nla] = S[nf:, nf:]
        Hlf[:nla, :nf] = S[nf:, :nf]
        gf, gl = np.zeros(NT), np.zeros(NT)
        gf[:nf], gl[:nla] = sr[:nf], sr[nf:]
        self.send.copy_(torch.from_numpy(np.concatenate([Dff.ravel(), Hlf.ravel(), Dll.ravel(), gf, gl])))

    def _solve_and_update(self):
        P, NT = self.P, self.nb_top
        NT2 = NT * NT
        rec = self.recv.numpy().reshape(P, 3 * NT2 + 2 * NT)
        T = np.zeros(((P + 1) * NT, (P + 1) * NT))
        tr = np.zeros((P + 1) * NT)
        for j in range(P + 1):
            sl = slice(j * NT, (j + 1) * NT)
            if j > 0:
                T[sl, sl] += rec[j - 1, 2 * NT2:3 * NT2].reshape(NT, NT)
                tr[sl] += rec[j - 1, 3 * NT2 + NT:]
            if j < P:
                T[sl, sl] += rec[j, :NT2].reshape(NT, NT)
                tr[sl] += rec[j, 3 * NT2:3 * NT2 + NT]
                nx = slice((j + 1) * NT, (j + 2) * NT)
                T[nx, sl] = rec[j, NT2:2 * NT2].reshape(NT, NT)
                T[sl, nx] = T[nx, sl].T
        xt = np.linalg.solve(T, tr).reshape(P + 1, NT)
        H, rhs, inner, ends, Hii = self._keep
        first, last = self._ends()
        xe = np.concatenate([xt[self.rank][:len(first)], xt[self.rank + 1][:len(last)]])
        x = np.zeros(len(rhs))
        x[ends] = xe
        x[inner] = np.linalg.solve(Hii, rhs[inner] - H[np.ix_(inner, ends)] @ xe)
        b, N, d = self.b, self.N, self.d
        xs = x[:N * b].reshape(N, b)
        for i in range(N):
            self.pose[i] = O.retract(self.kind, self.pose[i], xs[i, :d], self.chart)
            self.vel[i] += xs[i, d:]
        self.lmk += x[N * b:].reshape(self.lmk.shape)
        return x, rhs, last

    def fs_phase2(self, want_stats=True):
        x = self._solve_and_update()[0]
        if not want_stats:
            return None
        return Stats(self._err_before, self._chain().error(), float(np.abs(x).max()))

    # ---- Levenberg-Marquardt trial steps (the caller owns the loop: gpslam_amd/sharded.py, SplitSolver.iterate_lm)
    def lm_begin(self):
        self._saved = (self.pose.copy(), self.vel.copy(), self.lmk.copy())

    def lm_reject(self):
        self.pose, self.vel, self.lmk = (a.copy() for a in self._saved)

    def fs_lm_trial_phase1(self, lam):
        self.fs_phase1(lam)            # the state IS the linearisation point (lm_begin / lm_reject)

    def fs_lm_trial_phase2(self):
        x, rhs, last = self._solve_and_update()
        own = np.ones(len(x), dtype=bool)
        if self.rank < self.P - 1:
            own[last] = False          # a shared unknown enters |delta|^2 on the piece to its right
        return np.array([self._err_before, self._chain().error(), np.abs(x).max(), x @ rhs, x[own] @ x[own], 0.0])
