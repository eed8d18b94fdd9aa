"""Segment sharding on the GPU: P handles on ONE device play the P ranks (the all-gather is a device copy), so the
whole HIP phase-1 / phase-2 path is exercised without a multi-GPU node; the records are also compared with the
numpy segment model that the CPU gloo test uses."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _setup(problem, P, kind):
    import torch
    import gpslam_amd
    from gpslam_amd import sharded
    stream = torch.cuda.current_stream().cuda_stream
    ranks = []
    for r in range(P):
        lp = sharded.local_problem(problem, r, P)
        s = gpslam_amd.ChainSolver(kind, device=0, rank=r, nranks=P)
        s.set_stream(stream)
        sharded.apply_local(lp, s)
        send, recv = sharded.device_tensors(s)
        ranks.append((s, send, recv, lp))
    return ranks


def _iterate(ranks, lam=0.0):
    P = len(ranks)
    for s, _, _, _ in ranks:
        s.iterate_phase1(lam)
    for s, _, recv, _ in ranks:                 # the "all-gather"
        rv = recv.view(P, -1)
        for k in range(P):
            rv[k].copy_(ranks[k][1])
    sts = [s.iterate_phase2(True) for s, _, _, _ in ranks]
    return (sum(st.error_before for st in sts), sum(st.error_after for st in sts), max(st.delta_inf_norm for st in sts))


@pytest.mark.parametrize("P,N", [(2, 41), (3, 100), (4, 1000), (8, 5003), (2, 2)])
def test_sharded_pose3_matches_unsharded_gpu_and_oracle(P, N):
    import gpslam_amd
    from gpslam_amd import synthetic as S
    problem = S.pose3_chain(max(N, P))
    ranks = _setup(problem, P, gpslam_amd.POSE3)
    single = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
    orc = S.apply(problem, O.Chain(O.POSE3))
    for it in range(5):
        eb, ea, dinf = _iterate(ranks)
        rc, st = single.iterate_gn()
        rc0, s0 = orc.iterate_gn()
        assert abs(eb - st.error_before) <= 1e-8 * max(1.0, st.error_before)
        assert abs(ea - st.error_after) <= 1e-6 * max(1.0, st.error_after)
        assert abs(dinf - st.delta_inf_norm) <= 1e-6 * max(1.0, st.delta_inf_norm) + 1e-10
    pose = np.vstack([r[0].get_states()[0] for r in ranks])
    vel = np.vstack([r[0].get_states()[1] for r in ranks])
    p1, v1 = single.get_states()
    p0, v0 = orc.get_states()
    for (pa, va) in ((p1, v1), (p0, v0)):
        assert np.abs(pose - pa).max() <= 1e-9 * max(1.0, np.abs(pa).max())
        assert np.abs(vel - va).max() <= 1e-9 * max(1.0, np.abs(va).max())


def test_sharded_linear_chain_and_interface_records_match_numpy_model():
    import torch
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    from segment_model import SegmentModel
    P, N = 3, 90
    problem = S.linear_chain(N)
    ranks = _setup(problem, P, gpslam_amd.LINEAR3)
    models = [sharded.apply_local(sharded.local_problem(problem, r, P), SegmentModel(O.LINEAR3, r, P)) for r in range(P)]
    for s, _, _, _ in ranks:
        s.iterate_phase1(0.0)
    for m in models:
        m.iterate_phase1(0.0)
    torch.cuda.synchronize()
    for r in range(P):
        rec_gpu = ranks[r][1].cpu().numpy()
        rec_cpu = models[r].send.numpy()
        assert np.abs(rec_gpu - rec_cpu).max() <= 1e-9 * max(1.0, np.abs(rec_cpu).max()), r
    # finish the iteration on the GPU side and compare with the unsharded oracle
    for s, _, recv, _ in ranks:
        rv = recv.view(P, -1)
        for k in range(P):
            rv[k].copy_(ranks[k][1])
    for s, _, _, _ in ranks:
        s.iterate_phase2(True)
    orc = S.apply(problem, O.Chain(O.LINEAR3))
    orc.iterate_gn()
    pose = np.vstack([r[0].get_states()[0] for r in ranks])
    p0, _ = orc.get_states()
    assert np.abs(pose - p0).max() <= 1e-9 * max(1.0, np.abs(p0).max())


def test_rccl_exchange_plumbing_world_size_one():
    """The real torch.distributed (RCCL) all-gather on the library's device buffers, world_size 1 on one GPU:
    a forced-sharded single segment must reproduce the unsharded solver."""
    import os
    import torch
    import torch.distributed as dist
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        problem = S.pose3_chain(3000)
        s = gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=0, rank=0, nranks=1, force_sharded=True)
        s.set_stream(torch.cuda.current_stream().cuda_stream)
        sharded.apply_local(sharded.local_problem(problem, 0, 1), s)
        send, recv = sharded.device_tensors(s)
        sv = sharded.ShardedSolver(s, send, recv, 0, 1, dist=dist)
        single = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
        for _ in range(4):
            st = sv.iterate()
            rc, s1 = single.iterate_gn()
            assert abs(st["error_after"] - s1.error_after) <= 1e-6 * max(1.0, s1.error_after)
        sv.run(3)
        single.run_gn(3)
        p0, v0 = single.get_states()
        p1, v1 = s.get_states()
        assert np.abs(p0 - p1).max() <= 1e-9 * max(1.0, np.abs(p0).max())
        # landmarks + Levenberg-Marquardt through the same orchestration class: all-gather of the records, all-reduce
        # of the landmark Schur complement, all-gather of the decision scalars -- the collectives really run (RCCL)
        from gpslam_amd import plaza
        data = plaza.load(os.path.join(os.path.dirname(__file__), "golden", "plaza2.npz"))
        problem = plaza.build_problem(data)
        kind, chart = gpslam_amd.POSE2, gpslam_amd.CHART_FIRST_ORDER
        sp = gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2, device=0, rank=0, nranks=1, force_sharded=True)
        sp.set_stream(torch.cuda.current_stream().cuda_stream)
        sharded.apply_local(sharded.local_problem(problem, 0, 1), sp)
        send, recv = sharded.device_tensors(sp)
        svp = sharded.ShardedSolver(sp, send, recv, 0, 1, dist=dist, landmark_buf=sharded.landmark_tensor(sp))
        ref = plaza.apply(problem, gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2))
        import lm_lockstep
        assert svp.abi            # (round 5) svp.iterate_lm IS gpslam_hip_iterate_lm: torch's RCCL collectives behind gpslam_hip_set_collectives
        lm_lockstep.run(ref, svp, 1e-5, 7, err_tol=1e-6)
        m = plaza.metrics(problem, sp.get_states()[0], sp.get_landmarks())
        assert m["position_m"] < 0.25
        # ... and gpslam_hip_optimize on the sharded handle: GTSAM's loop, the iteration count of the unsharded handle
        for s_ in (ref, sp):
            s_.set_states(problem["pose"], problem["vel"])
            s_.set_landmarks(problem["landmarks"])
        _rc, so_ref = ref.optimize(ref.default_params(use_lm=1))
        _rc, so = svp.optimize(sp.default_params(use_lm=1))
        assert so.iterations == so_ref.iterations and abs(so.error_after - so_ref.error_after) <= 1e-6 * so_ref.error_after
    finally:
        dist.destroy_process_group()


def test_sharded_chain_with_landmarks_matches_unsharded():
    """SURVEY.md section 8(e) collective (3): the Plaza2 graph (4 landmarks, 1816 interpolated ranges) cut into P
    segments; the landmark Schur complement is summed over the ranks between phase 2a and 2b (here: on the host,
    the P handles share one GPU).  Gauss-Newton from the ground-truth initialisation, against the unsharded solver."""
    import os
    import torch
    import gpslam_amd
    from gpslam_amd import plaza, sharded
    data = plaza.load(os.path.join(os.path.dirname(__file__), "golden", "plaza2.npz"))
    problem = plaza.build_problem(data, init_ground_truth=True)
    kind, chart = gpslam_amd.POSE2, gpslam_amd.CHART_FIRST_ORDER
    single = plaza.apply(problem, gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2))
    for P in (2, 3, 5):      # P = 3 puts range measurements into the intervals that straddle the cuts
        stream = torch.cuda.current_stream().cuda_stream
        ranks = []
        for r in range(P):
            s = gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2, device=0, rank=r, nranks=P)
            s.set_stream(stream)
            sharded.apply_local(sharded.local_problem(problem, r, P), s)
            send, recv = sharded.device_tensors(s)
            ranks.append((s, send, recv, sharded.landmark_tensor(s)))
        ref = plaza.apply(problem, gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2))
        for it in range(4):
            for s, _, _, _ in ranks:
                s.iterate_phase1(0.0)
            for s, _, recv, _ in ranks:
                rv = recv.view(P, -1)
                for k in range(P):
                    rv[k].copy_(ranks[k][1])
            for s, _, _, _ in ranks:
                s.iterate_phase2a()
            total = sum(r[3].clone() for r in ranks)            # the all-reduce
            for r in ranks:
                r[3].copy_(total)
            sts = [s.iterate_phase2b(True) for s, _, _, _ in ranks]
            rc, st = ref.iterate_gn()
            ea = sum(x.error_after for x in sts)
            assert abs(ea - st.error_after) <= 1e-6 * max(1.0, st.error_after), (P, it)
        pose = np.vstack([r[0].get_states()[0] for r in ranks])
        p1, _ = ref.get_states()
        # Plaza fixes the global frame only through 1 m priors (cond(H) ~ 1e8): a different partition = a different
        # summation order moves the whole solution by a few 1e-8 m along that weak direction
        assert np.abs(pose - p1).max() <= 5e-7
        lms = [r[0].get_landmarks() for r in ranks]
        for lm in lms:                                           # replicated landmarks stay bit-identical
            assert np.array_equal(lm, lms[0])
        assert np.abs(lms[0] - ref.get_landmarks()).max() <= 5e-7
        for r in ranks:
            r[0].close()
    single.close()


def _lm_iterate_emulated(ranks, lam):
    """ShardedSolver.iterate_lm with the collectives done by hand across P handles that share one GPU: the caller-owned loop
    of include/gpslam_hip.h around gpslam_hip_lm_decide (gpslam_amd/sharded.py: lm_loop)."""
    from gpslam_amd import sharded
    P = len(ranks)

    def trial(lam_):
        for r in ranks:
            r[0].lm_trial_phase1(lam_)
        for r in ranks:
            rv = r[2].view(P, -1)
            for k in range(P):
                rv[k].copy_(ranks[k][1])
        for r in ranks:
            r[0].iterate_phase2a()
        if ranks[0][3] is not None:
            total = sum(r[3].clone() for r in ranks)
            for r in ranks:
                r[3].copy_(total)
        return sharded.reduce_lm_scalars(np.stack([r[0].lm_trial_phase2() for r in ranks]))

    def reject():
        for r in ranks:
            r[0].lm_reject()

    for r in ranks:
        r[0].lm_begin()
    return sharded.lm_loop(trial, reject, lam)


@pytest.mark.parametrize("P", [2, 3])
def test_sharded_levenberg_marquardt_matches_unsharded(P):
    """LevenbergMarquardtOptimizer::iterate across ranks on the Plaza2 graph from the dead-reckoned initial values (the
    reference's own scenario, matlab/PlazaPose2.m:208-228): same lambda schedule and errors as the unsharded solver."""
    import os
    import torch
    import gpslam_amd
    from gpslam_amd import plaza, sharded
    data = plaza.load(os.path.join(os.path.dirname(__file__), "golden", "plaza2.npz"))
    problem = plaza.build_problem(data)
    kind, chart = gpslam_amd.POSE2, gpslam_amd.CHART_FIRST_ORDER
    ref = plaza.apply(problem, gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2))
    stream = torch.cuda.current_stream().cuda_stream
    ranks = []
    for r in range(P):
        s = gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2, device=0, rank=r, nranks=P)
        s.set_stream(stream)
        sharded.apply_local(sharded.local_problem(problem, r, P), s)
        send, recv = sharded.device_tensors(s)
        ranks.append((s, send, recv, sharded.landmark_tensor(s)))
    import lm_lockstep
    lm_lockstep.run(ref, lambda lam_: _lm_iterate_emulated(ranks, lam_), 1e-5, 6, err_tol=1e-6)
    pose = np.vstack([r[0].get_states()[0] for r in ranks])
    assert np.abs(pose - ref.get_states()[0]).max() <= 1e-5
    for r in ranks:
        r[0].close()


@pytest.mark.parametrize("kind,sensor,P", [(O.POSE2, True, 3), (O.POSE3, True, 2), (O.POSE3, False, 5), (O.LINEAR3, False, 4),
                                           (O.ROT3, False, 3)],
                         ids=["pose2+sensor/3", "pose3+sensor/2", "pose3/5", "linear3/4", "rot3-attitude/3"])
def test_every_measurement_mix_sharded(kind, sensor, P):
    """Every measurement factor kind (interpolated range / attitude / GPS, range, odometry, bearing-range; with and
    without body_P_sensor; with a landmark border) on a chain cut into P segments: the recorded graph is replayed per
    rank (GraphRecorder) and iterated next to the unsharded solver.  Several factors sit in the intervals that straddle
    the cuts (two measurements per interval)."""
    import torch
    import gpslam_amd
    from gpslam_amd import sharded
    from test_gpu_measurements import build_meas_pair, LD
    orc, ref, c, (rec,) = build_meas_pair(kind, N=61, seed=7, sensor=sensor, extra_makers=(sharded.GraphRecorder,))
    chart = O.CHART_FIRST_ORDER if kind == O.POSE2 else O.CHART_EXPMAP
    stream = torch.cuda.current_stream().cuda_stream
    ranks = []
    for r in range(P):
        s = gpslam_amd.ChainSolver(kind, chart, LD[kind], device=0, rank=r, nranks=P)
        s.set_stream(stream)
        rec.replay(s, r, P)
        send, recv = sharded.device_tensors(s)
        ranks.append((s, send, recv, sharded.landmark_tensor(s)))
    for it in range(4):
        for r in ranks:
            r[0].iterate_phase1(0.0)
        for r in ranks:
            rv = r[2].view(P, -1)
            for k in range(P):
                rv[k].copy_(ranks[k][1])
        for r in ranks:
            r[0].iterate_phase2a()
        if ranks[0][3] is not None:
            total = sum(r[3].clone() for r in ranks)
            for r in ranks:
                r[3].copy_(total)
        sts = [r[0].iterate_phase2b(True) for r in ranks]
        rc, st = ref.iterate_gn()
        assert rc == 0
        eb, ea = sum(x.error_before for x in sts), sum(x.error_after for x in sts)
        assert abs(eb - st.error_before) <= 1e-8 * max(1.0, st.error_before), it
        assert abs(ea - st.error_after) <= 1e-6 * max(1.0, st.error_after), it
    pose = np.vstack([r[0].get_states()[0] for r in ranks])
    vel = np.vstack([r[0].get_states()[1] for r in ranks])
    p1, v1 = ref.get_states()
    assert np.abs(pose - p1).max() <= 1e-8 * max(1.0, np.abs(p1).max())
    assert np.abs(vel - v1).max() <= 1e-8 * max(1.0, np.abs(v1).max())
    if LD[kind]:
        assert np.abs(ranks[0][0].get_landmarks() - ref.get_landmarks()).max() <= 1e-8
    for r in ranks:
        r[0].close()


def test_two_process_rccl_when_two_gpus_are_visible():
    """One process per GPU over RCCL / xGMI (the deployment shape; tests/rccl_worker.py): runs whenever the box shows at
    least two devices, on up to eight of them; a single-GPU box (the build farm) skips -- the in-process forms above
    and tests/cpp/sharded_rccl_test.cpp cover the code path there."""
    import os
    import subprocess
    import sys
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs >= 2 visible GPUs (found %d)" % ndev)
    world = min(ndev, 8)
    port = 29900 + os.getpid() % 90
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(4000 * world)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "RCCL_WORKERS_OK" in outs[0]


@pytest.mark.parametrize("P", [2, 3])
def test_c_abi_optimiser_loops_on_sharded_handles(P):
    """gpslam_hip_iterate_lm / gpslam_hip_optimize / gpslam_hip_iterate_gn / gpslam_hip_error THEMSELVES on the P handles of a sharded
    chain with a landmark border (round 5: gpslam_hip_set_collectives; the loop used to be the caller's): the Plaza2 graph from
    the dead-reckoned initial values (matlab/PlazaPose2.m:208-228) -- every rank returns the lambda schedule, the accept flags,
    the trial counts and the whole-chain errors of the unsharded handle."""
    import os
    import gpslam_amd
    from gpslam_amd import plaza, sharded
    data = plaza.load(os.path.join(os.path.dirname(__file__), "golden", "plaza2.npz"))
    problem = plaza.build_problem(data)
    kind, chart = gpslam_amd.POSE2, gpslam_amd.CHART_FIRST_ORDER
    ref = plaza.apply(problem, gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2))
    from thread_ranks import ThreadRanks
    tr = ThreadRanks(P)
    hs, lps = [], []
    for r in range(P):
        s = gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2, device=0, rank=r, nranks=P)
        lps.append(sharded.local_problem(problem, r, P))
        sharded.apply_local(lps[-1], s)
        s.set_collectives(*tr.collectives(r))
        hs.append(s)

    def restart():
        ref.set_states(problem["pose"], problem["vel"])
        ref.set_landmarks(problem["landmarks"])
        for s, lp in zip(hs, lps):
            s.set_states(lp["pose"], lp["vel"])
            if "halo_pose" in lp:
                s.set_halo_state(lp["halo_pose"], lp["halo_vel"])
            s.set_landmarks(problem["landmarks"])
    e_ref = ref.error()
    errs = tr.run(lambda r: hs[r].error())
    assert all(e == errs[0] for e in errs) and abs(errs[0] - e_ref) <= 1e-9 * e_ref     # NonlinearFactorGraph::error of the whole chain

    def lm_run(r):
        lam, hist = 1e-5, []
        for _ in range(6):
            _rc, st, lam = hs[r].iterate_lm(lam)[:3]
            hist.append((lam, int(st.accepted), int(st.trials), st.error_before, st.error_after))
        return hist
    hists = tr.run(lm_run)
    lam = 1e-5
    for it in range(6):
        _rc, st, lam = ref.iterate_lm(lam)[:3]
        for r in range(P):
            h = hists[r][it]
            assert h == hists[0][it]                                    # every rank: bit-identical reduced numbers
            assert h[:3] == (lam, int(st.accepted), int(st.trials)), (it, r, h, lam)
            assert abs(h[3] - st.error_before) <= 1e-7 * st.error_before and abs(h[4] - st.error_after) <= 1e-6 * st.error_after
    pose = np.vstack([s.get_states()[0] for s in hs])
    assert np.abs(pose - ref.get_states()[0]).max() <= 1e-5
    # GaussNewtonOptimizer::iterate and NonlinearOptimizer::optimize (LM), each from the initial values on both sides
    restart()
    _rc, st_ref = ref.iterate_gn()
    sts = tr.run(lambda r: hs[r].iterate_gn()[1])
    for st in sts:
        assert abs(st.error_before - st_ref.error_before) <= 1e-9 * st_ref.error_before
        assert abs(st.error_after - st_ref.error_after) <= 1e-6 * st_ref.error_after
        assert abs(st.delta_inf_norm - st_ref.delta_inf_norm) <= 1e-6 * st_ref.delta_inf_norm
    restart()
    _rc, so_ref = ref.optimize(ref.default_params(use_lm=1))
    sos = tr.run(lambda r: hs[r].optimize(hs[r].default_params(use_lm=1))[1])
    for so in sos:
        assert so.iterations == so_ref.iterations and abs(so.error_after - so_ref.error_after) <= 1e-7 * so_ref.error_after
    for s in hs:
        s.close()
    ref.close()


def test_sharded_handle_without_collectives_still_refuses_the_whole_chain_loops():
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S
    problem = S.pose3_chain(400)
    s = gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=0, rank=0, nranks=2)
    sharded.apply_local(sharded.local_problem(problem, 0, 2), s)
    for call in (s.iterate_gn, lambda: s.iterate_lm(1e-5), s.optimize, lambda: s.run_gn(2)):
        with pytest.raises(gpslam_amd.GpslamHipError, match="collectives"):
            call()
    s.close()


def test_indefinite_pivot_inside_a_sharded_run_is_reported_by_every_rank_with_or_without_statistics():
    """gpslam_hip_run_gn on a sharded handle looped iterate_gn(h, NULL) for every iteration but the last: phase 1 cleared the
    non-positive-pivot flag, nothing read it, and a run WITHOUT a statistics struct returned 0 on every rank whatever happened.  Round 6:
    the flag is cleared once and stays up through the run, one 64-byte gather at the end carries it (and the first failure of any
    rank) to everybody.  A chain with no absolute information at all (GP priors only pin differences: the reduced system is singular,
    GTSAM's IndeterminantLinearSystemException) through run_gn(3) with statistics, through the raw entry point with st = NULL, and
    through a single iterate_gn(h, NULL)."""
    import ctypes as C
    import gpslam_amd
    from gpslam_amd import sharded
    from test_gpu_parity import random_chain
    from thread_ranks import ThreadRanks
    P, N, kind = 2, 96, O.LINEAR3
    c = random_chain(kind, N, 9)
    problem = dict(kind=kind, N=N, qc=np.eye(3) * 0.01, pose=c["pose"], vel=c["vel"], gp_left=np.arange(N - 1, dtype=np.int32), gp_dt=c["dt"])
    tr = ThreadRanks(P)
    hs = []
    for r in range(P):
        s = gpslam_amd.ChainSolver(kind, device=0, rank=r, nranks=P)
        sharded.apply_local(sharded.local_problem(problem, r, P), s)
        s.set_collectives(*tr.collectives(r))
        hs.append(s)

    def with_stats(r):
        try:
            hs[r].run_gn(3)
        except gpslam_amd.GpslamHipError as ex:
            return str(ex)
        return "no error"
    for msg in tr.run(with_stats):
        assert "(-3)" in msg and "non-positive pivot" in msg, msg
    lib = hs[0].lib
    assert tr.run(lambda r: lib.gpslam_hip_run_gn(hs[r]._h, 3, None, None)) == [-3] * P          # (round 5: 0 on every rank)
    assert tr.run(lambda r: lib.gpslam_hip_iterate_gn(hs[r]._h, None)) == [-3] * P
    for s in hs:
        s.close()


@pytest.mark.parametrize("loop", ["iterate_gn", "run_gn", "iterate_lm", "optimize"])
def test_a_failure_on_one_rank_ends_the_call_on_every_rank_with_the_same_code(loop):
    """a rank that failed locally returned at once and left its peers blocked in the next collective.  Round 6: it goes on
    making every collective call and its code travels with the scalars.  Here rank 1's all_reduce_sum (the landmark Schur complement
    of the Plaza graph) completes -- so that its peer is not stuck inside THAT collective -- and then reports failure: rank 1 sees
    GPSLAM_E_COMM from its own callback; the test is that rank 0 returns GPSLAM_E_COMM as well instead of waiting forever in the gather of
    the scalars (the round-5 library hangs here; ThreadRanks gives up after its join timeout)."""
    import os
    import gpslam_amd
    from gpslam_amd import plaza, sharded
    from thread_ranks import ThreadRanks
    data = plaza.load(os.path.join(os.path.dirname(__file__), "golden", "plaza2.npz"))
    problem = plaza.build_problem(data)
    kind, chart, P = gpslam_amd.POSE2, gpslam_amd.CHART_FIRST_ORDER, 2
    tr = ThreadRanks(P)
    hs = []
    calls = [0]
    for r in range(P):
        s = gpslam_amd.ChainSolver(kind, chart=chart, landmark_dim=2, device=0, rank=r, nranks=P)
        sharded.apply_local(sharded.local_problem(problem, r, P), s)
        gather, reduce_ = tr.collectives(r)
        if r == 1:
            good = reduce_

            def reduce_(buf, n, stream, good=good):
                good(buf, n, stream)
                calls[0] += 1
                if calls[0] == 2:
                    raise RuntimeError("injected: rank 1's all_reduce_sum reports failure on its second call")
        s.set_collectives(gather, reduce_)
        hs.append(s)

    def body(r):
        s = hs[r]
        try:
            if loop == "iterate_gn":
                s.iterate_gn(); s.iterate_gn()
            elif loop == "run_gn":
                s.run_gn(3)
            elif loop == "iterate_lm":
                lam = 1e-5
                for _ in range(3):
                    _rc, _st, lam = s.iterate_lm(lam)[:3]
            else:
                s.optimize(s.default_params(use_lm=1))
        except gpslam_amd.GpslamHipError as ex:
            return str(ex)
        return "no error"
    msgs = tr.run(body)
    assert all("(-7)" in m for m in msgs), msgs
    for s in hs:
        s.close()
