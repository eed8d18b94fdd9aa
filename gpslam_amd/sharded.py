"""Contiguous-segment sharding of a GP chain across the GPUs of one node.

Rank r owns states [lo_r, hi_r) and every factor whose LEFT state lies in that range; it keeps a read-only copy
(the halo) of the first state of rank r + 1.  One Gauss-Newton iteration is

    phase 1 (local, HIP):  linearise + assemble + eliminate the segment down to its separator (its first state)
                           -> one interface record [D | C | G | RD | Rg] per rank (3.6 KB for Pose3)
    exchange (RCCL):       ONE all-gather of the records over xGMI -- the only collective on the data path
    phase 2 (local, HIP):  every rank solves the tiny P-block reduced system redundantly, back-substitutes,
                           retracts its states and its halo copy (so the halo never needs another exchange)
    with landmarks (replicated on every rank; each rank owns the measurement factors of its segment) phase 2 is
    split around ONE all-reduce of the landmark Schur complement [S | gL] ((L ld)(L ld + 2) doubles: 80 for Plaza)

Scalars (error, |delta|_inf) are reduced only when the caller asks for them.

The orchestration below is backend-agnostic: `backend` is a gpslam_amd.ChainSolver created with nranks > 1 on the
GPU, and the CPU tests drive the very same class over gloo with a numpy model of the two phases.
"""
import numpy as np


def partition(N, P):
    """Contiguous, nearly equal segments: returns the P + 1 boundaries."""
    base, rem = divmod(N, P)
    b = [0]
    for r in range(P):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


def local_problem(problem, rank, nranks):
    """Cut rank `rank`'s segment out of a global problem description (gpslam_amd.synthetic format)."""
    p = problem
    N = p["N"]
    b = partition(N, nranks)
    lo, hi = b[rank], b[rank + 1]
    out = dict(kind=p["kind"], name=p.get("name", ""), N=hi - lo, qc=p["qc"], lo=lo, hi=hi,
               pose=p["pose"][lo:hi].copy(), vel=p["vel"][lo:hi].copy())
    if rank < nranks - 1:
        out["halo_pose"], out["halo_vel"] = p["pose"][hi].copy(), p["vel"][hi].copy()

    def cut(idx_key, keys):
        if idx_key not in p:
            return
        idx = np.asarray(p[idx_key])
        m = (idx >= lo) & (idx < hi)
        out[idx_key] = (idx[m] - lo).astype(np.int32)
        for k in keys:
            out[k] = np.asarray(p[k])[m]

    cut("gp_left", ["gp_dt"])
    cut("prior_idx", ["prior_pose", "prior_sig"])
    cut("vprior_idx", ["vprior", "vprior_sig"])
    cut("between_left", ["between_meas", "between_sig"])
    cut("range_left", ["range_lm", "range_z", "range_sigma", "range_dt", "range_tau"])
    if "landmarks" in p:                      # landmarks are replicated; their priors live on rank 0
        out["landmarks"] = np.asarray(p["landmarks"]).copy()
        out["linear"] = p.get("linear", False)
        if rank == 0:
            for k in ("lprior_idx", "lprior", "lprior_sig"):
                if k in p:
                    out[k] = np.asarray(p[k]).copy()
    return out


def apply_local(lp, solver):
    """Feed a local problem to a sharded ChainSolver-like backend (states first, then the halo, then factors)."""
    solver.set_qc(lp["qc"])
    solver.set_states(lp["pose"], lp["vel"])
    if "halo_pose" in lp:
        solver.set_halo_state(lp["halo_pose"], lp["halo_vel"])
    if "landmarks" in lp:
        solver.set_landmarks(lp["landmarks"])
        if "lprior_idx" in lp:
            solver.add_landmark_priors(lp["lprior_idx"], lp["lprior"], lp["lprior_sig"])
        if "range_left" in lp and len(lp["range_left"]):
            solver.add_interp_range(lp["range_left"], lp["range_lm"], lp["range_z"], lp["range_sigma"], lp["range_dt"],
                                    lp["range_tau"])
    if "gp_left" in lp and len(lp["gp_left"]):
        solver.add_gp_priors(lp["gp_left"], lp["gp_dt"])
    if "prior_idx" in lp and len(lp["prior_idx"]):
        solver.add_pose_priors(lp["prior_idx"], lp["prior_pose"], lp["prior_sig"])
    if "vprior_idx" in lp and len(lp["vprior_idx"]):
        solver.add_vel_priors(lp["vprior_idx"], lp["vprior"], lp["vprior_sig"])
    if "between_left" in lp and len(lp["between_left"]):
        if lp.get("linear", False):
            solver.add_odometry2d(lp["between_left"], lp["between_meas"], lp["between_sig"])
        else:
            solver.add_between(lp["between_left"], lp["between_meas"], lp["between_sig"])
    solver.compile()
    return solver


class GraphRecorder:
    """Records the ChainSolver calls that describe a graph, so that the same description can be replayed onto the
    unsharded solver, the oracle, or -- cut at the segment boundaries -- onto the ranks of a sharded solve.

    Every factor call of the ChainSolver surface starts with the index array of the factors' (left) states; the other
    array arguments with one entry per factor are cut with it, everything else (body_P_sensor, calibration) is shared."""

    FACTOR_CALLS = ("add_gp_priors", "add_pose_priors", "add_vel_priors", "add_between", "add_interp_range", "add_range",
                    "add_interp_attitude", "add_interp_gps", "add_odometry2d", "add_bearing_range", "add_interp_projection")
    # positions of the arguments that are NOT one-per-factor: body_P_sensor, camera calibration
    SHARED_ARGS = {"add_interp_range": (6,), "add_interp_gps": (5,), "add_interp_projection": (6, 7)}

    def __init__(self):
        self.calls = []          # (method name, args)
        self.qc = self.pose = self.vel = self.landmarks = None
        self.lm_priors = []

    def set_qc(self, Qc):
        self.qc = np.array(Qc, dtype=np.float64)

    def set_states(self, pose, vel):
        self.pose, self.vel = np.array(pose, dtype=np.float64), np.array(vel, dtype=np.float64)

    def set_landmarks(self, pts):
        self.landmarks = np.array(pts, dtype=np.float64)

    def add_landmark_priors(self, idx, prior, sigmas):
        self.lm_priors.append((np.asarray(idx), np.asarray(prior), np.asarray(sigmas)))

    def compile(self):
        pass

    def __getattr__(self, name):
        if name in GraphRecorder.FACTOR_CALLS:
            return lambda *args: self.calls.append((name, args))
        raise AttributeError(name)

    def replay(self, solver, rank=0, nranks=1):
        """Feed rank `rank`'s segment of the recorded graph to `solver` (a whole-chain solver for nranks = 1)."""
        N = len(self.pose)
        b = partition(N, nranks)
        lo, hi = b[rank], b[rank + 1]
        solver.set_qc(self.qc)
        solver.set_states(self.pose[lo:hi], self.vel[lo:hi])
        if rank < nranks - 1:
            solver.set_halo_state(self.pose[hi], self.vel[hi])
        if self.landmarks is not None:
            solver.set_landmarks(self.landmarks)
            if rank == 0:                       # replicated landmarks: their priors are counted once
                for idx, prior, sig in self.lm_priors:
                    solver.add_landmark_priors(idx, prior, sig)
        for name, args in self.calls:
            idx = np.asarray(args[0])
            keep = (idx >= lo) & (idx < hi)
            if not keep.any():
                continue
            cut = [(idx[keep] - lo).astype(np.int32)]
            for pos, a in enumerate(args[1:], start=1):
                shared = a is None or pos in GraphRecorder.SHARED_ARGS.get(name, ())
                cut.append(a if shared else np.asarray(a)[keep])
            getattr(solver, name)(*cut)
        solver.compile()
        return solver


class _DevView:
    """Wrap a raw device pointer as something torch.as_tensor understands (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def torch_collectives(dist, group, world):
    """(all_gather, all_reduce_sum) for ChainSolver.set_collectives over torch.distributed (backend "nccl" = RCCL): the
    library's raw device pointers are wrapped as tensors and the collective is enqueued ON THE STREAM THE LIBRARY NAMES (the handle's
    stream: torch.cuda.ExternalStream made current around the call -- ADVICE r5: the argument used to be ignored, which was only
    right while torch's current stream happened to be the handle's)."""
    import torch
    streams = {}

    def on(hip_stream):
        if not hip_stream:                       # the null stream: torch's default stream
            return torch.cuda.stream(torch.cuda.default_stream())
        if hip_stream not in streams:
            streams[hip_stream] = torch.cuda.ExternalStream(int(hip_stream))
        return torch.cuda.stream(streams[hip_stream])

    def all_gather(send, recv, nbytes, hip_stream):
        with on(hip_stream):
            s = torch.as_tensor(_DevView(send, nbytes), device="cuda")
            r = torch.as_tensor(_DevView(recv, nbytes * world), device="cuda")
            if hasattr(dist, "all_gather_into_tensor"):
                dist.all_gather_into_tensor(r, s, group=group)
            else:
                dist.all_gather(list(r.view(world, -1).unbind(0)), s, group=group)

    def all_reduce_sum(buf, n, hip_stream):
        with on(hip_stream):
            dist.all_reduce(torch.as_tensor(_DevView(buf, n * 8), device="cuda"), group=group)

    return all_gather, all_reduce_sum


def device_tensors(solver):
    """(send, recv) torch views of a GPU backend's interface buffers."""
    import torch
    sp, sb, rp, rb = solver.interface_buffers()
    return (torch.as_tensor(_DevView(sp, sb), device="cuda"), torch.as_tensor(_DevView(rp, rb), device="cuda"))


def landmark_tensor(solver):
    """torch view of a GPU backend's landmark Schur-complement buffer [S | gL], or None without landmarks."""
    import torch
    ptr, nb = solver.landmark_reduce_buffer()
    return torch.as_tensor(_DevView(ptr, nb), device="cuda") if nb else None


class ShardedSolver:
    """One rank of a segment-sharded Gauss-Newton solve.

    backend : object with iterate_phase1(lam), iterate_phase2(want_stats) -> stats, get_states()
    send/recv : torch tensors (device for the HIP backend, CPU for the numpy model) of the interface record
                and of the gathered records
    group : torch.distributed process group handle or None for the default group
    """

    def __init__(self, backend, send, recv, rank, nranks, dist=None, group=None, landmark_buf=None):
        self.backend, self.send, self.recv = backend, send, recv
        self.rank, self.nranks, self.dist, self.group = rank, nranks, dist, group
        self.landmark_buf = landmark_buf      # [S | gL] of this rank (torch tensor) when the chain has landmarks
        # Stream ordering: the handle's kernels and torch.distributed's collectives must be ordered against
        # one another.  A handle creates its own NON-BLOCKING stream, which torch knows nothing about, so a GPU backend
        # is moved onto torch's current stream here: RCCL collectives launched through torch.distributed are ordered
        # against that stream (they wait for work already enqueued on it, and later work on it waits for them), which
        # makes phase1 -> all_gather -> phase2a -> all_reduce -> phase2b a correctly ordered sequence with no host sync.
        self.abi = False
        if getattr(send, "is_cuda", False) and hasattr(backend, "set_stream"):
            import torch
            backend.set_stream(torch.cuda.current_stream().cuda_stream)
            # the host's collectives handed to the library (round 5): the optimiser LOOPS of the C ABI -- gpslam_hip_iterate_lm,
            # _optimize, _iterate_gn with the whole chain's statistics -- then run on this rank's handle
            if dist is not None and hasattr(backend, "set_collectives"):
                backend.set_collectives(*torch_collectives(dist, group, nranks))
                self.abi = True

    def optimize(self, params=None):
        """NonlinearOptimizer::optimize() of the whole chain through the C ABI (gpslam_hip_optimize on this rank's handle; every rank
        makes the same collective calls and gets the same statistics).  GPU backends with a process group."""
        if not self.abi:
            raise RuntimeError("optimize() through the C ABI needs a GPU backend and a process group")
        return self.backend.optimize(params)

    def exchange(self):
        if self.dist is None:
            self.recv.copy_(self.send)
            return
        if getattr(self.send, "is_cuda", False) and hasattr(self.dist, "all_gather_into_tensor"):
            # one contiguous output: RCCL gathers straight into recv (a list of views makes torch gather into a temporary and
            # copy it out chunk by chunk -- extra launches on a collective whose payload is a few KB)
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            return
        chunks = list(self.recv.view(self.nranks, -1).unbind(0))
        self.dist.all_gather(chunks, self.send, group=self.group)

    def iterate(self, lam=0.0, want_stats=True):
        self.backend.iterate_phase1(lam)
        self.exchange()
        if self.landmark_buf is None:
            st = self.backend.iterate_phase2(want_stats)
        else:
            self.backend.iterate_phase2a()
            if self.dist is not None:
                self.dist.all_reduce(self.landmark_buf, group=self.group)     # sum of the ranks' Schur complements
            st = self.backend.iterate_phase2b(want_stats)
        if not want_stats:
            return None
        vals = np.array([st.error_before, st.error_after, st.delta_inf_norm], dtype=np.float64)
        if self.dist is not None:
            import torch
            t = torch.from_numpy(vals.copy())
            if self.send.is_cuda:
                t = t.cuda()
            allv = [torch.empty_like(t) for _ in range(self.nranks)]
            self.dist.all_gather(allv, t, group=self.group)
            m = torch.stack(allv).cpu().numpy()
            vals = np.array([m[:, 0].sum(), m[:, 1].sum(), m[:, 2].max()])
        return dict(error_before=float(vals[0]), error_after=float(vals[1]), delta_inf_norm=float(vals[2]))

    def _reduce_lm_scalars(self, loc):
        """sum of (error, trial error, delta.g, |delta|^2), max of (|delta|_inf, indefinite flag) over the ranks"""
        if self.dist is None:
            return loc
        import torch
        t = torch.from_numpy(np.asarray(loc, dtype=np.float64).copy())
        if self.send.is_cuda:
            t = t.cuda()
        allv = [torch.empty_like(t) for _ in range(self.nranks)]
        self.dist.all_gather(allv, t, group=self.group)
        m = torch.stack(allv).cpu().numpy()
        return np.array([m[:, 0].sum(), m[:, 1].sum(), m[:, 2].max(), m[:, 3].sum(), m[:, 4].sum(), m[:, 5].max()])

    def iterate_lm(self, lam, lambda_factor=10.0, lambda_upper_bound=1e5, lambda_lower_bound=0.0, min_model_fidelity=1e-3,
                   relative_error_tol=1e-5):
        """LevenbergMarquardtOptimizer::iterate (GTSAM 4.0 defaults) across the ranks: the loop of gpslam_hip_iterate_lm
        with its three global sums made collective.  Every rank takes the same decisions (identical reduced scalars).
        Returns (stats dict, new lambda).  With the collectives registered on the handle (GPU backend + process group) this IS
        gpslam_hip_iterate_lm: the loop runs inside the library; otherwise (the numpy model over gloo, the single-process tests)
        the same loop runs here around the library's trial phases -- both take their branches in gpslam_hip_lm_decide."""
        be = self.backend
        if self.abi:
            return _lm_through_abi(be, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity, relative_error_tol)

        def trial(lam_):
            be.lm_trial_phase1(lam_)
            self.exchange()
            be.iterate_phase2a()
            if self.landmark_buf is not None and self.dist is not None:
                self.dist.all_reduce(self.landmark_buf, group=self.group)
            return self._reduce_lm_scalars(be.lm_trial_phase2())

        be.lm_begin()
        return lm_loop(trial, be.lm_reject, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity,
                       relative_error_tol)

    def run(self, iters, lam=0.0):
        """`iters` iterations back to back; statistics only for the last one."""
        out = None
        for k in range(iters):
            out = self.iterate(lam, want_stats=(k == iters - 1))
        return out


# ------------------------------------------------------------------------------------------------------------------
# Chains with many locally visible landmarks (BASELINE config 4) across GPUs: pieces joined at shared cut states.
# The dense-border scheme above replicates every landmark on every rank; with 5e4 landmarks that is neither possible
# nor needed -- a landmark is seen from a window of a few hundred states, so it belongs to ONE piece, or to the fat
# separator between two neighbouring pieces (include/gpslam_hip.h, gpslam_hip_fs_set_split).

def split_boundaries(N, P):
    """States s_0 = 0 < s_1 < ... < s_P = N - 1: piece r holds the states [s_r, s_(r+1)], both ends included."""
    b = partition(N - 1, P)
    b[-1] = N - 1
    return b


def split_local_problem(problem, rank, nranks):
    """Cut piece `rank` out of a global problem with range landmarks (gpslam_amd.synthetic format).  Every factor goes to
    the piece its left (or only) state lies in -- the final state counts to the last piece; a landmark goes to every piece
    that holds one of its range factors (at most two neighbours, else the pieces are shorter than its window of
    visibility), its prior to the right-most of them.  Returns the local problem; `lm_global` maps its landmarks back,
    `first_lm` / `last_lm` are the landmarks shared with the left / right neighbour, `own_lm` marks the landmarks whose
    values this piece reports."""
    p = problem
    N, P = p["N"], nranks
    s = split_boundaries(N, P)
    lo, hi = s[rank], s[rank + 1]
    last = rank == P - 1
    out = dict(kind=p["kind"], name=p.get("name", ""), N=hi - lo + 1, qc=p["qc"], lo=lo, hi=hi,
               pose=p["pose"][lo:hi + 1].copy(), vel=p["vel"][lo:hi + 1].copy(), linear=p.get("linear", False))

    def mine(idx):
        idx = np.asarray(idx)
        return (idx >= lo) & ((idx < hi) | (last & (idx == hi)))

    def cut(idx_key, keys):
        if idx_key not in p:
            return
        idx = np.asarray(p[idx_key])
        m = mine(idx)
        out[idx_key] = (idx[m] - lo).astype(np.int32)
        for k in keys:
            out[k] = np.asarray(p[k])[m]

    cut("gp_left", ["gp_dt"])
    cut("prior_idx", ["prior_pose", "prior_sig"])
    cut("vprior_idx", ["vprior", "vprior_sig"])
    cut("between_left", ["between_meas", "between_sig"])
    # landmarks: the piece of every range factor, then the pieces of every landmark
    rl = np.asarray(p["range_left"])
    rlm = np.asarray(p["range_lm"])
    L = len(p["landmarks"])
    piece_of = np.minimum(np.searchsorted(np.asarray(s), rl, side="right") - 1, P - 1)
    pmin = np.full(L, P, dtype=np.int64)
    pmax = np.full(L, -1, dtype=np.int64)
    np.minimum.at(pmin, rlm, piece_of)
    np.maximum.at(pmax, rlm, piece_of)
    seen = pmax >= 0
    pmin[~seen], pmax[~seen] = 0, 0                       # a landmark nobody measures: piece 0 keeps it (and its prior)
    if (pmax - pmin > 1).any():
        raise ValueError("a landmark is seen from more than two neighbouring pieces: use fewer, longer pieces")
    local = np.nonzero((pmin <= rank) & (pmax >= rank))[0]
    g2l = np.full(L, -1, dtype=np.int64)
    g2l[local] = np.arange(len(local))
    out["lm_global"] = local
    out["landmarks"] = np.asarray(p["landmarks"])[local].copy()
    out["first_lm"] = g2l[local[(pmin[local] < rank)]].astype(np.int32)
    out["last_lm"] = g2l[local[(pmax[local] > rank)]].astype(np.int32)
    out["own_lm"] = pmax[local] == rank
    m = mine(rl)
    out["range_left"] = (rl[m] - lo).astype(np.int32)
    out["range_lm"] = g2l[rlm[m]].astype(np.int32)
    for k in ("range_z", "range_sigma", "range_dt", "range_tau"):
        out[k] = np.asarray(p[k])[m]
    if "lprior_idx" in p:
        li = np.asarray(p["lprior_idx"])
        mk = pmax[li] == rank
        out["lprior_idx"] = g2l[li[mk]].astype(np.int32)
        out["lprior"] = np.asarray(p["lprior"])[mk]
        out["lprior_sig"] = np.asarray(p["lprior_sig"])[mk]
    return out


def apply_split(lp, solver, rank, nranks):
    """Feed a piece (split_local_problem) to a ChainSolver created with nranks = 1; fs_set_top still has to follow."""
    from . import synthetic
    solver.fs_set_split(rank, nranks, lp["first_lm"], lp["last_lm"])
    return synthetic.apply(lp, solver)


class SplitSolver:
    """One piece of a chain with locally visible landmarks, one process per GPU (or, for tests, P handles in one process
    with dist = None and `peers`).  iterate(): fs_phase1 -> ONE all-gather of the interface records -> fs_phase2."""

    def __init__(self, backend, rank, nranks, dist=None, group=None, device="cuda"):
        import torch
        self.backend, self.rank, self.nranks, self.dist, self.group = backend, rank, nranks, dist, group
        self.device = device                      # "cpu": the numpy model of the two phases (tests/split_model.py) over gloo
        self.abi = False
        if device == "cuda":
            backend.set_stream(torch.cuda.current_stream().cuda_stream)     # collectives are ordered against this stream
            if dist is not None and hasattr(backend, "set_collectives"):     # the optimiser loops of the C ABI on this piece
                backend.set_collectives(*torch_collectives(dist, group, nranks))
                self.abi = True
        nb = backend.fs_split_info()["fat_block"]
        self.nb_local = nb
        self.send = self.recv = None
        if dist is not None:                      # the pieces agree on the block size of the interface record
            t = torch.tensor([nb], dtype=torch.int32, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            self.set_top(int(t.item()))

    def set_top(self, nb_top):
        import torch
        self.backend.fs_set_top(nb_top)
        if self.device == "cuda":
            sp, sb, rp, rb = self.backend.fs_interface()
            self.send = torch.as_tensor(_DevView(sp, sb), device="cuda")
            self.recv = torch.as_tensor(_DevView(rp, rb), device="cuda")
        else:
            self.send, self.recv = self.backend.fs_interface()

    def exchange(self):
        if self.dist is None:
            if self.nranks > 1:
                raise RuntimeError("SplitSolver without a process group holds one piece of %d: drive the pieces of one process "
                                   "with iterate_pieces / iterate_pieces_lm" % self.nranks)
            self.recv.copy_(self.send)
            return
        if getattr(self.send, "is_cuda", False) and hasattr(self.dist, "all_gather_into_tensor"):
            # one contiguous output: RCCL gathers straight into recv (a list of views makes torch gather into a temporary and
            # copy it out chunk by chunk -- extra launches on a collective whose payload is a few KB)
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            return
        chunks = list(self.recv.view(self.nranks, -1).unbind(0))
        self.dist.all_gather(chunks, self.send, group=self.group)

    def iterate(self, lam=0.0, want_stats=True):
        self.backend.fs_phase1(lam)
        self.exchange()
        st = self.backend.fs_phase2(want_stats)
        if not want_stats:
            return None
        vals = np.array([st.error_before, st.error_after, st.delta_inf_norm], dtype=np.float64)
        if self.dist is not None:
            import torch
            t = torch.from_numpy(vals.copy()).to(self.device)
            allv = [torch.empty_like(t) for _ in range(self.nranks)]
            self.dist.all_gather(allv, t, group=self.group)
            m = torch.stack(allv).cpu().numpy()
            vals = np.array([m[:, 0].sum(), m[:, 1].sum(), m[:, 2].max()])
        return dict(error_before=float(vals[0]), error_after=float(vals[1]), delta_inf_norm=float(vals[2]))


    def iterate_lm(self, lam, lambda_factor=10.0, lambda_upper_bound=1e5, lambda_lower_bound=0.0, min_model_fidelity=1e-3,
                   relative_error_tol=1e-5):
        """LevenbergMarquardtOptimizer::iterate (GTSAM 4.0 defaults) across the pieces: ShardedSolver.iterate_lm with the
        split chain's trial steps.  Returns (stats dict, new lambda).  gpslam_hip_iterate_lm itself once the collectives are
        registered on the handle (GPU backend + process group)."""
        be = self.backend
        if self.abi:
            return _lm_through_abi(be, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity, relative_error_tol)

        def trial(lam_):
            be.fs_lm_trial_phase1(lam_)
            self.exchange()
            return self._reduce_lm_scalars(be.fs_lm_trial_phase2())

        be.lm_begin()
        return lm_loop(trial, be.lm_reject, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity,
                       relative_error_tol)

    def _reduce_lm_scalars(self, loc):
        if self.dist is None:
            return np.asarray(loc, dtype=np.float64)
        import torch
        t = torch.from_numpy(np.asarray(loc, dtype=np.float64).copy()).to(self.device)
        allv = [torch.empty_like(t) for _ in range(self.nranks)]
        self.dist.all_gather(allv, t, group=self.group)
        return reduce_lm_scalars(torch.stack(allv).cpu().numpy())


def reduce_lm_scalars(m):
    """rows = pieces: sum of (error, trial error, delta.g, |delta|^2), max of (|delta|_inf, indefinite flag)"""
    return np.array([m[:, 0].sum(), m[:, 1].sum(), m[:, 2].max(), m[:, 3].sum(), m[:, 4].sum(), m[:, 5].max()])


def _lm_through_abi(be, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity, relative_error_tol):
    """gpslam_hip_iterate_lm on a handle that holds the host's collectives -> (stats dict of the whole chain, new lambda)"""
    p = be.default_params(use_lm=1)
    p.lambda_factor, p.lambda_upper_bound, p.lambda_lower_bound = lambda_factor, lambda_upper_bound, lambda_lower_bound
    p.min_model_fidelity, p.relative_error_tol = min_model_fidelity, relative_error_tol
    _rc, st, lam = be.iterate_lm(lam, p)[:3]
    return dict(error_before=st.error_before, error_after=st.error_after, delta_inf_norm=st.delta_inf_norm,
                accepted=bool(st.accepted), trials=int(st.trials), last_trial_error=st.last_trial_error), lam


def lm_loop(trial, reject, lam, lambda_factor=10.0, lambda_upper_bound=1e5, lambda_lower_bound=0.0, min_model_fidelity=1e-3,
            relative_error_tol=1e-5):
    """The lambda search of one LevenbergMarquardtOptimizer::iterate() around `trial(lambda) -> reduced scalars`
    (error, trial error, |delta|_inf, delta . g, |delta|^2, indefinite flag) and `reject()` (restore the linearisation
    point), for callers that own the loop because the scalars are sums over ranks / pieces.  Every branch is taken by
    gpslam_hip_lm_decide -- the function gpslam_hip_iterate_lm itself uses (include/gpslam_hip.h) -- so all ranks, which hold
    identical reduced scalars, follow the lambda schedule of the unsharded chain.  Returns (stats dict, new lambda)."""
    from .chain import lm_decide
    accepted, err0, new_err, dinf, trials, last = False, 0.0, 0.0, 0.0, 0, 0.0
    while True:
        s = trial(lam)
        trials += 1
        err0 = float(s[0])
        last = float(s[1]) if s[5] == 0.0 else err0
        accepted, done, lam = lm_decide(s, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity,
                                        relative_error_tol)
        if accepted:
            new_err, dinf = float(s[1]), float(s[2])
            break
        reject()
        if done:                       # small cost change (lambda untouched), or lambda at its upper bound
            break
    return dict(error_before=err0, error_after=new_err if accepted else err0, delta_inf_norm=dinf if accepted else 0.0,
                accepted=accepted, trials=trials, last_trial_error=last), lam


def iterate_pieces_lm(pieces, lam, lambda_factor=10.0, lambda_upper_bound=1e5, lambda_lower_bound=0.0, min_model_fidelity=1e-3,
                      relative_error_tol=1e-5):
    """iterate_lm for P SplitSolvers living in ONE process (tests)."""
    P = len(pieces)

    def trial(lam_):
        for sv in pieces:
            sv.backend.fs_lm_trial_phase1(lam_)
        for sv in pieces:
            rv = sv.recv.view(P, -1)
            for k in range(P):
                rv[k].copy_(pieces[k].send)
        return reduce_lm_scalars(np.stack([sv.backend.fs_lm_trial_phase2() for sv in pieces]))

    def reject():
        for sv in pieces:
            sv.backend.lm_reject()

    for sv in pieces:
        sv.backend.lm_begin()
    return lm_loop(trial, reject, lam, lambda_factor, lambda_upper_bound, lambda_lower_bound, min_model_fidelity,
                   relative_error_tol)


def iterate_pieces(pieces, lam=0.0):
    """P SplitSolvers living in ONE process (tests, single-GPU timing): the all-gather is P^2 device copies."""
    P = len(pieces)
    for sv in pieces:
        sv.backend.fs_phase1(lam)
    for sv in pieces:
        rv = sv.recv.view(P, -1)
        for k in range(P):
            rv[k].copy_(pieces[k].send)
    sts = [sv.backend.fs_phase2(True) for sv in pieces]
    return dict(error_before=sum(st.error_before for st in sts), error_after=sum(st.error_after for st in sts),
                delta_inf_norm=max(st.delta_inf_norm for st in sts))


def merge_pieces(problem, locals_, states, landmarks):
    """Global (pose, vel, landmarks) from the pieces' results: a shared state is reported by the piece on its right, a
    shared landmark by the right-most piece that holds it."""
    N, L = problem["N"], len(problem["landmarks"])
    pose = np.zeros_like(np.asarray(problem["pose"], dtype=np.float64))
    vel = np.zeros_like(np.asarray(problem["vel"], dtype=np.float64))
    lmk = np.array(problem["landmarks"], dtype=np.float64)
    for lp, (ps, vs), lm in zip(locals_, states, landmarks):
        pose[lp["lo"]:lp["hi"] + 1] = ps
        vel[lp["lo"]:lp["hi"] + 1] = vs
        own = lp["own_lm"]
        lmk[lp["lm_global"][own]] = np.asarray(lm)[own]
    return pose, vel, lmk
