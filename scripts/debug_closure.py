import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import oracle as O
from gpslam_amd import synthetic as S
import gpslam_amd as G
def strip(p):
    return {k: v for k, v in p.items() if not (k.startswith("range_") or k.startswith("lprior") or k.startswith("landmark"))}
N = 500
p = S.add_loop_closures(strip(S.pose2_range_chain(N, seed=1)), [[12, 471], [300, 40]], seed=3)
orc = S.apply(p, O.Chain(O.POSE2)); dev = S.apply(p, G.ChainSolver(O.POSE2))
for it in range(12):
    rc, s0 = orc.iterate_gn(); rc, s1 = dev.iterate_gn()
    xo, vo = orc.get_states(); xd, vd = dev.get_states()
    print(it, "%.12g %.12g  d %.3e %.3e  |x diff| %.3e" % (s0.error_after, s1.error_after, s0.delta_inf_norm, s1.delta_inf_norm, np.abs(xo - xd).max()))
