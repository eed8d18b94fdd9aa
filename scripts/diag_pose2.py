import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import oracle as O
import test_gpu_parity as T
kind=O.POSE2
orc,dev,c=T.build_pair(kind,300,seed=31+kind)
for it in range(8):
    D0,O0,g0,_,_,_=orc.normal_equations(); D1,O1,g1,_=dev.normal_equations()
    x0=O.block_tridiag_solve(D0,O0,g0); x1=dev.block_tridiag_solve(D1,O1,g1); x01=dev.block_tridiag_solve(D0,O0,g0)
    print(it,'neq diffs',np.abs(D0-D1).max(),np.abs(O0-O1).max(),np.abs(g0-g1).max(),'|g|',np.abs(g0).max(),
          'delta diff (own neq)',np.abs(x0-x1).max(),'delta diff (same neq)',np.abs(x0-x01).max(),'|x|',np.abs(x0).max())
    # condition estimate via dense
    if it==0:
        n=300*6; H=np.zeros((n,n))
        for i in range(300):
            H[6*i:6*i+6,6*i:6*i+6]=D0[i]
            if i<299:
                H[6*i+6:6*i+12,6*i:6*i+6]=O0[i]; H[6*i:6*i+6,6*i+6:6*i+12]=O0[i].T
        w=np.linalg.eigvalsh(H); print('cond',w[-1]/w[0], w[0], w[-1])
    orc.iterate_gn(); dev.iterate_gn()
    p0,v0=orc.get_states(); p1,v1=dev.get_states()
    print('   state diff pose',np.abs(p0-p1).max(),'vel',np.abs(v0-v1).max())
