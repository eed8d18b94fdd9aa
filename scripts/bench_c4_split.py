"""BASELINE config 4 split into P pieces (gpslam_amd/sharded.py: SplitSolver) -- what ONE rank of a P-GPU run does, timed on
one GPU: all P pieces live on this device and run one after the other; a rank's iteration is fs_phase1 -> all-gather of the
interface records (here: device copies, not timed as a collective) -> fs_phase2.  Weak scaling: N states PER PIECE.
python scripts/bench_c4_split.py [N per piece] [P] [iterations]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpslam_amd
from gpslam_amd import sharded, synthetic as S

n_piece = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = n_piece * P
problem = S.pose2_local_landmarks_chain(N, window=200)
kw = dict(chart=gpslam_amd.CHART_FIRST_ORDER, landmark_dim=2)
pieces, locals_ = [], []
for r in range(P):
    lp = sharded.split_local_problem(problem, r, P)
    s = gpslam_amd.ChainSolver(problem["kind"], **kw)
    sharded.apply_split(lp, s, r, P)
    locals_.append(lp)
    pieces.append(sharded.SplitSolver(s, r, P))
nb_top = max(sv.nb_local for sv in pieces)
for sv in pieces:
    sv.set_top(nb_top)
rec_bytes = pieces[0].send.numel() * 8


def gather():
    for sv in pieces:
        rv = sv.recv.view(P, -1)
        for k in range(P):
            rv[k].copy_(pieces[k].send)


sharded.iterate_pieces(pieces)                       # warm-up (first-touch of the buffers)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t1 = t2 = 0.0
for it in range(iters):
    ev[0].record()
    for sv in pieces:
        sv.backend.fs_phase1(0.0)
    ev[1].record()
    gather()
    ev[2].record()
    for sv in pieces:
        sv.backend.fs_phase2(False)
    ev[3].record()
    torch.cuda.synchronize()
    t1 += ev[0].elapsed_time(ev[1])
    t2 += ev[2].elapsed_time(ev[3])
per_rank = (t1 + t2) / iters / P
hist = sharded.iterate_pieces(pieces)
# the same number of states as ONE piece, unsplit, on this GPU
one = S.apply(S.pose2_local_landmarks_chain(n_piece, window=200), gpslam_amd.ChainSolver(problem["kind"], **kw))
one.run_gn(1)
_, ph = one.run_gn(iters, timed=True)
print("C4 split: %d pieces x %d states (nb_top %d, record %.1f KB): %.3f ms per rank and iteration (phase 1 %.3f + phase 2 %.3f); "
      "an unsplit chain of %d states: %.3f ms (that figure includes the error pass after the update, which a split run skips inside a "
      "fixed-count loop); error after %d iterations %.6g, |delta|_inf %.3g"
      % (P, n_piece, nb_top, rec_bytes / 1024, per_rank, t1 / iters / P, t2 / iters / P, n_piece, ph[4] / iters, iters + 2,
         hist["error_after"], hist["delta_inf_norm"]))
