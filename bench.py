#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations over the BASELINE config-3 chain (Pose3 GP prior + synthetic odometry,
1e5 states per GPU, fp64) on N GPUs of one node.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one full Gauss-Newton iteration of the hot path: linearise every factor (evaluateError + Jacobians),
assemble the block-tridiagonal normal equations, solve, retract, re-evaluate the error -- exactly what
matlab/PlazaPose2.m:224-226 brackets with tic/toc around optimizer.iterate().  Inputs are resident in HBM before
the timed region.  With N > 1 the ONE chain of N x 1e5 states is cut into N contiguous segments (weak scaling) and
every iteration performs one RCCL all-gather of the 3.6 KB interface records (gpslam_amd/sharded.py).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# rocprofv3 kernel names of the kernels bench.py times itself (time_kernel order); the per-launch HBM traffic of the
# same kernels on the same workload comes from the committed PMC passes (scripts/collect_profiles.sh ->
# profiles/<round>_pmc.json; explicit-size L2->fabric request counters TCC_EA0_RDREQ_{32B,64B,128B}, WRREQ{,_64B}).
PMC_FILE = os.path.join(ROOT, "profiles", "latest_pmc.json")
PMC_KEYS = [("k_gp<double, 3, 0>",), ("k_assemble",), ("k_fused_level0", "k_chunk_forward_rows", "k_chunk_forward<double, 12, true>"),
            ("k_chunk_backward<double, 12>",), ("k_retract<double, 3>",)]


def pmc_traffic(which, n_states):
    """HBM bytes per launch of kernel `which` from the committed counter passes, or None if they do not cover it.
    The solver kernels run once per hierarchy level; level 0 is the launch with the largest grid."""
    try:
        kern = json.load(open(PMC_FILE))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    best = None
    for k, c in kern.items():
        if k.startswith(PMC_KEYS[which]) and "ea_read_bytes" in c and c.get("states", n_states) == n_states:
            grid = int(k.rsplit("@", 1)[1])
            if best is None or grid > best[0]:
                best = (grid, float(c["ea_read_bytes"] + c["ea_write_bytes"]))
    return best[1] if best else None


def cpu_baseline(problem, iters=3):
    """The oracle (CPU restatement of the reference algorithm, 1 thread) timed on the same workload."""
    from oracle import oracle as O
    from gpslam_amd import synthetic as S
    ch = S.apply(problem, O.Chain(problem["kind"]))
    t0 = time.perf_counter()
    for _ in range(iters):
        rc, _st = ch.iterate_gn()
        assert rc == 0
    dt = time.perf_counter() - t0
    return dict(value=problem["N"] * iters / dt, unit="state-iterations/s", cores=1, kind="port",
                sample="%d Gauss-Newton iterations of the full %d-state workload, oracle/liboracle.so (gcc -O2), "
                       "%.1f s" % (iters, problem["N"], dt), seconds_per_iteration=dt / iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--states", type=int, default=100000, help="states per GPU (BASELINE config 3: 100k poses)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import gpslam_amd
    from gpslam_amd import sharded, synthetic as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    N = args.states
    total_states = N * world

    if world == 1:
        problem = S.pose3_chain(N)
        solver = S.apply(problem, gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=local_rank))

        def reset():
            solver.set_states(problem["pose"], problem["vel"])

        def one_with_stats():
            _rc, st = solver.iterate_gn()
            return st.error_after, st.delta_inf_norm

        def run(k):
            solver.run_gn(k)
    else:
        problem = S.pose3_chain(total_states)           # ONE chain, cut into `world` contiguous segments
        lp = sharded.local_problem(problem, rank, world)
        solver = gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=local_rank, rank=rank, nranks=world)
        solver.set_stream(torch.cuda.current_stream().cuda_stream)   # order kernels against the RCCL collective
        sharded.apply_local(lp, solver)
        send, recv = sharded.device_tensors(solver)
        sv = sharded.ShardedSolver(solver, send, recv, rank, world, dist=dist)

        def reset():
            solver.set_states(lp["pose"], lp["vel"])
            if "halo_pose" in lp:
                solver.set_halo_state(lp["halo_pose"], lp["halo_vel"])

        def one_with_stats():
            st = sv.iterate()
            return st["error_after"], st["delta_inf_norm"]

        def run(k):
            for _ in range(k):
                sv.iterate(want_stats=False)

    # convergence run (outside the contract clock): iterations until |delta|_inf < 1e-6
    conv_iters, conv_delta, final_error = 0, float("inf"), float("nan")
    for _ in range(25):
        final_error, conv_delta = one_with_stats()
        conv_iters += 1
        if conv_delta < 1e-6:
            break

    # timed region: restart from the initial values so the steps do real Newton work
    reset()
    if args.warmup > 0:
        run(args.warmup)
    reset()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # per-kernel device time (hipEvents on the handle's stream) of the per-GPU workload, for the roofline
        if world == 1:
            probe, pproblem = solver, problem
        else:
            pproblem = S.pose3_chain(N)
            probe = S.apply(pproblem, gpslam_amd.ChainSolver(gpslam_amd.POSE3, device=local_rank))
        names = ["k_gp (K1 linearise GP priors)", "k_assemble (K3)", "k_chunk_forward level 0 (K4)",
                 "k_chunk_backward level 0 (K4)", "k_retract (K6)"]
        kms = [probe.time_kernel(w, reps=5) for w in range(5)]
        ab = S.algorithmic_bytes_per_state(S.POSE3)
        blocks = ab["linearize"] - (18 * 8 + 8)
        fused = (kms[1] == 0.0)                    # the assembly runs inside the level-0 elimination (k_fused_level0)
        if fused:
            names[1] = "k_assemble (K3): fused into the level-0 elimination, no launch"
            names[2] = "k_fused_level0 (K3 + K4 level 0: assembly + elimination)"
            kms[1] = 1e-9
        alg = [ab["linearize"] * (N - 1),          # K1: read state + dt, write e + H1..H4 (whitened rows)
               (blocks + blocks) * N,              # K3: read rows, write blocks
               # K4 forward: SURVEY 8(d) single-pass solve figure (conservative); fused: read the rows, write the factors
               (blocks + blocks if fused else ab["solve"]) * N,
               (blocks + 12 * 8) * N,              # back-substitution: read factors, write delta
               ab["retract"] * N]
        dom = int(np.argmax(kms))
        achieved = alg[dom] / (kms[dom] * 1e-3) / 1e9
        _st, phase = probe.run_gn(3, timed=True)
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "GN state-iterations/sec (states x Gauss-Newton iters/sec), Pose3 GP chain",
            "value": total_states * args.steps / elapsed,
            "unit": "state-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Pose3/SE(3) GP prior + synthetic odometry, "
                                   "%d poses per GPU, fp64, Gauss-Newton" % N,
                       "states_per_gpu": N, "total_states": total_states,
                       "factors": "N-1 GaussianProcessPriorPose3 + N-1 BetweenFactor<Pose3> + 1 PriorFactor<Pose3>",
                       "parallelism": "1 chain in %d contiguous segments, 1 all-gather of interface records per "
                                      "iteration" % world if world > 1 else "single GPU"},
            "gn_iters_per_sec": args.steps / elapsed,
            "iters_to_convergence": conv_iters,
            "delta_inf_at_convergence": conv_delta,
            "final_error": final_error,
            "states_to_convergence_per_sec": total_states / (conv_iters * ms_per_step * 1e-3),
            "phase_ms_per_iter_1gpu": {k: float(v) / 3 for k, v in
                                       zip(["linearize", "assemble", "solve", "retract+error", "total"], phase)},
            "kernel_ms": {n: float(v) for n, v in zip(names, kms)},
            # every hot kernel against the same roof: algorithmic GB/s (SURVEY 8(d) bytes per unit) and, where the
            # committed counter passes cover it, the HBM bytes it really moved per launch
            "kernel_roofline": {n: {"algorithmic_GBps": alg[i] / (kms[i] * 1e-3) / 1e9,
                                    "frac_of_peak": alg[i] / (kms[i] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "moved_GBps": (pmc_traffic(i, N) / (kms[i] * 1e-3) / 1e9) if pmc_traffic(i, N) else None}
                                for i, n in enumerate(names)},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, N),
                         "traffic_source": "rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B}_sum / TCC_EA0_WRREQ{,_64B}_sum, "
                                           "profiles/latest_pmc.json (bytes per launch, same workload)",
                         "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kms[dom],
                         # what actually limits the kernel the HBM fraction is quoted for
                         "note": ("VALU-issue bound, not HBM bound: two-wave workgroups (assembly + elimination), ~3200 fp64 "
                                  "VALU instructions per workgroup block step, one fat wave of each kind per SIMD"
                                  if fused and dom == 2 else None)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(problem)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
