import sys; sys.path.insert(0,'.')
import gpslam_amd
from gpslam_amd import synthetic as S
for N in (2000, 10000, 50000, 100000, 400000):
    p=S.pose3_chain(N)
    s=S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
    s.run_gn(1)
    t=[s.time_kernel(w,10) for w in (0,1,2,3)]
    print(N, 'us per 1e5 states: gp %.1f asm %.1f fwd %.1f bwd %.1f' % tuple(x*1e3*1e5/N for x in t), ' raw ms', ['%.4f'%x for x in t])
    s.close()
