#!/bin/bash
# rocprofv3 kernel trace + stats of the config-4 run (GPU box, repo root): bash scripts/prof_c4.sh <tag> [N] [window] [seglen]
set -u
TAG=${1:-c4}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/scripts/bench_c4.py ${2:-1000000} ${3:-200} ${4:-0} > $OUT/run.log 2>&1
cd $ROOT
tail -2 $OUT/run.log
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/t_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
meta = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void gps::", "").split("(")[0][:60]
    agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
tot = sum(sum(v) for v in agg.values())
lines = ["| kernel | calls | total ms | avg us | % | vgpr | agpr | scratch | lds |", "|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    lines.append("| %s | %d | %.3f | %.2f | %.1f | %s | %s | %s | %s |" % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot, *meta[k]))
open("$OUT/kernel_stats.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:24]))
PY
