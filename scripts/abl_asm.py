import sys, os; sys.path.insert(0,'.')
import gpslam_amd
from gpslam_amd import synthetic as S
p=S.pose3_chain(100000)
s=S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3))
s.run_gn(1)
for m in (0,1,2,3,4,0):
    os.environ['GPSLAM_ASM_DBG']=str(m)
    s.time_kernel(1,3)
    print('mode',m,'asm ms %.4f'%s.time_kernel(1,20))
