#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
rm -rf /tmp/vg; (cd /tmp && GPSLAM_FUSE_B6=1 PYTHONPATH=$ROOT rocprofv3 --kernel-trace --output-format csv -d /tmp/vg -o t -- python $ROOT/scripts/bench_configs.py 1000000 > /tmp/vg.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/vg/**/t_kernel_trace.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void gps::", "").split("(")[0][:60]
    seen.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    seen[k + "|meta"] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Grid_Size_X"], r["Workgroup_Size_X"])
for k, v in seen.items():
    if k.endswith("|meta") or "fused" not in k: continue
    print("%-52s n=%3d avg %8.2f us  vgpr/agpr/sgpr/scratch/lds/grid/wg %s" % (k, len(v), sum(v) / len(v) / 1e3, seen[k + "|meta"]))
PY
