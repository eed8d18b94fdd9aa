#!/bin/bash
# mean PMC counters per launch of the kernels matching a pattern:  bash scripts/pmc_kernel.sh <pattern> "<counters>" <command...>
PAT=$1; CNT=$2; shift 2
ROOT=$(pwd); OUT=/tmp/pmc_$$; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
# always under a timeout: a pass with seven TCC_* counters at once did not come back within ten minutes on this pool
(cd $ROOT && timeout ${PMC_TIMEOUT:-120} rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1)
python - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counters", open(sys.argv[1] + "/log.txt").read()[-400:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"].replace("void gps::", "").split("(")[0][:40] + "@" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
