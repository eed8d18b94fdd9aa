import sys; sys.path.insert(0,'.')
import numpy as np
import gpslam_amd
from gpslam_amd import synthetic as S
N=100000
p=S.pose3_chain(N)
cfgs=[(16,8,32),(16,4,16),(8,4,16)] if len(sys.argv)<2 else [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for m0,m1,top in cfgs:
    s=S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3, chunk=m0, upper_chunk=m1, top_blocks=top))
    s.run_gn(2)
    s.set_states(p['pose'],p['vel'])
    st,ph=s.run_gn(5,timed=True)
    k=[s.time_kernel(w,3) for w in (0,1,2,3)]
    print(m0,m1,top,'phases/iter',np.round(ph/5,4),'gp %.3f asm %.3f fwdL0 %.3f bwdL0 %.3f'%tuple(k), 'dinf',st.delta_inf_norm)
    s.close()
