#!/bin/bash
# per-kernel average durations of a command under rocprofv3 --kernel-trace:  bash scripts/kernel_times.sh <tag> <command...>
TAG=$1; shift
ROOT=$(pwd)
OUT=/tmp/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd $ROOT && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- "$@" > $OUT/log.txt 2>&1)
tail -1 $OUT/log.txt
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no kernel trace"); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].replace("void gps::", "").split("(")[0][:48]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:9]:
    print("  %-48s calls %4d avg %9.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
PY
