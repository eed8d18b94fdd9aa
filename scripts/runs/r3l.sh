cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp32.py tests/test_gpu_boundary.py -q -m gpu -x > $O/t1.log 2>&1; echo "t1 rc=$?" >> $O/t1.log
tail -3 $O/t1.log
cat > /tmp/c5.py <<'PY'
import sys; sys.path.insert(0,'.')
import gpslam_amd
from gpslam_amd import synthetic as S
for name, make, kind in (("rot3+att", lambda: S.rot3_attitude_chain(1000000, refs=2), gpslam_amd.ROT3), ("pose3+gps", lambda: S.pose3_gps_chain(1000000, keep_odometry=True), gpslam_amd.POSE3), ("pose3 c3 1e6", lambda: S.pose3_chain(1000000), gpslam_amd.POSE3)):
    p = make()
    for prec in (0, 1):
        s = S.apply(p, gpslam_amd.ChainSolver(kind, precision=prec))
        s.run_gn(2); s.set_states(p["pose"], p["vel"])
        st, ph = s.run_gn(3, timed=True)
        print(name, "fp32" if prec else "fp64", [round(float(x)/3,3) for x in ph], s.plan_info()["fused"])
        s.close()
PY
timeout 600 python /tmp/c5.py > $O/c5.log 2>&1; cat $O/c5.log | tail -8
