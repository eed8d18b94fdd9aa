import sys, os; sys.path.insert(0,'.')
import gpslam_amd
from gpslam_amd import synthetic as S
N=100000
p=S.pose3_chain(N)
s=S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
s.run_gn(1)
print(os.environ.get('GPSLAM_ASM_WPB',''), 'asm %.4f'%s.time_kernel(1,10))
