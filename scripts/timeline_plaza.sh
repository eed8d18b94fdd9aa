#!/bin/bash
# kernel timeline of one steady Gauss-Newton iteration of BASELINE config 1 (Plaza2) inside run_gn: bash scripts/timeline_plaza.sh
export TMPDIR=/tmp; rm -rf /tmp/kt_pl; mkdir -p /tmp/kt_pl
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_pl -o t -- python scripts/time_plaza.py > /tmp/kt_pl/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt_pl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_lin" in r["Kernel_Name"]]
i0, i1 = idx[30], idx[31]
t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
for r in rows[i0:i1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-46s grid %7s start %7.1f dur %6.1f gap %5.1f" % (r["Kernel_Name"].replace("void gps::", "").split("(")[0][:46], r["Grid_Size_X"], (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3))
    prev = max(prev, en)
print("iteration span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
