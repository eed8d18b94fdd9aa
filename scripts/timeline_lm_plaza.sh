#!/bin/bash
# kernels and memory copies of the last iterate_lm calls of scripts/time_plaza.py (BASELINE config 1): bash scripts/timeline_lm_plaza.sh
export TMPDIR=/tmp; D=/tmp/kt_lm; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -o t -- python scripts/time_plaza.py > $D/log.txt 2>&1
python - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void gps::", "").split("(")[0][:44], r["Grid_Size_X"]) for r in rows]
g = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)
if g:
    for r in csv.DictReader(open(g[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", ""), r.get("Bytes", "")))
ev.sort()
# the LM calls are at the end: take the last ~45 events
tail = ev[-48:]
t0 = tail[0][0]; prev = t0
for st, en, name, grid in tail:
    print("%-46s %8s start %8.1f dur %6.1f gap %6.1f" % (name, grid, (st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3))
    prev = max(prev, en)
PY
