import sys; sys.path.insert(0,'.')
import gpslam_amd
from gpslam_amd import synthetic as S
p=S.pose3_chain(100000)
s=S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
print('asm ms', s.time_kernel(1,5))
