#!/bin/bash
# smoke(), the whole -m gpu suite and the default bench.py line on the GPU box:  gpurun -- bash scripts/verify_gpu.sh
mkdir -p gpurun_out/verify
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q > gpurun_out/verify/tall.log 2>&1; echo "tall rc=$?"; grep -E "passed|failed" gpurun_out/verify/tall.log | tail -2
python bench.py > gpurun_out/verify/bench.json 2> gpurun_out/verify/bench.err; python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/verify/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["phase_ms_per_iter_1gpu"], d["roofline"]["frac"], d["k1_batched_jacobian"]["frac_sec8d"], d["k1_batched_jacobian"]["frac_moved"])
e=d["extras"]; print(e["north_star_1e6_pose3_1gpu"]["ms_per_iteration_device"], e["config4_pose2_1e6_landmarks_5e4_1gpu"]["ms_per_iteration_device"], e["config2_linear3_1e5"]["ms_per_iteration_device"])
PY
