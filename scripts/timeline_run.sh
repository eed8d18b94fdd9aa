#!/bin/bash
# bash scripts/timeline_run.sh <c2|c3|c5|c5b|c4> [N]: kernel timeline of iteration 8 of an untimed run_gn(12)
export TMPDIR=/tmp; D=/tmp/kt_run_$1; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python scripts/timeline_run.py "$@" > $D/log.txt 2>&1
python - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_lin" in r["Kernel_Name"]]
i0, i1 = idx[-5], idx[-4]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
for r in rows[i0:i1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-50s grid %8s start %8.1f dur %7.1f gap %5.1f" % (r["Kernel_Name"].replace("void gps::", "").split("(")[0][:50], r["Grid_Size_X"], (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3))
    prev = max(prev, en)
print("iteration span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
