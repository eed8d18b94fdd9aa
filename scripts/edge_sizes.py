import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
import test_gpu_parity as T
for kind in (O.POSE3, O.POSE2, O.LINEAR2):
    for N in (1, 2, 3, 16, 17, 26, 51, 401):
        orc, dev, c = T.build_pair(kind, N, seed=N)
        for _ in range(4):
            rc0, s0 = orc.iterate_gn(); rc1, s1 = dev.iterate_gn()
            assert rc0 == 0 and rc1 == 0
        (x0, v0), (x1, v1) = orc.get_states(), dev.get_states()
        T.states_close(kind, x0, v0, x1, v1, 1e-9)
        print(kind, N, 'ok', s1.error_after)
