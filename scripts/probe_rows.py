import sys, ctypes; sys.path.insert(0,'.')
import numpy as np
import gpslam_amd
from gpslam_amd import synthetic as S
from gpslam_amd import chain
p=S.pose3_chain(100000)
for m0 in (13, 25):
    s=S.apply(p, gpslam_amd.ChainSolver(gpslam_amd.POSE3, chunk=m0))
    s.run_gn(2)
    print('m0',m0,'fwd ms', s.time_kernel(2,5))
    out=(ctypes.c_longlong*32)()
    lib=chain.load_library()
    lib.gpslam_hip_debug_probe(out)
    print('ticks: GJ %d  rec-out %d  stage-in %d  products %d  rollover/out %d'%tuple(out[:5]))
    s.close()
