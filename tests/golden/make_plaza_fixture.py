#!/usr/bin/env python3
"""Convert the reference's Plaza2 range-only SLAM log (a DATA file, matlab/data/Plaza2.mat, loaded by
matlab/PlazaPose2.m:12-13) into tests/golden/plaza2.npz.  Run in the build container, where /root/reference exists:

    python tests/golden/make_plaza_fixture.py

Arrays (format documented at matlab/PlazaPose2.m:2, columns as the script indexes them):
  GT  (4091, 4)  time, x, y, heading            ground truth           (PlazaPose2.m:15, :84-86)
  DR  (4090, 3)  time, distance, delta heading  odometry increments    (:113)
  TD  (1816, 4)  time, sender, landmark id, range                       (:148-157)
  TL  (4, 3)     landmark id, x, y                                      (:17-18, :53)
  init_heading_offset  scalar                                           (:85)
(DRp, the dead-reckoned path kept in the .mat for plotting only, is not needed.)
"""
import os
import sys

import numpy as np
import scipy.io as sio

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/matlab/data/Plaza2.mat"
m = sio.loadmat(src)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plaza2.npz")
np.savez_compressed(out, GT=m["GT"].astype(np.float64), DR=m["DR"].astype(np.float64), TD=m["TD"].astype(np.float64),
                    TL=m["TL"].astype(np.float64), init_heading_offset=np.float64(m["init_heading_offset"].ravel()[0]))
print(out, os.path.getsize(out), "bytes")
