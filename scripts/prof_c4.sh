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
# ---- MFMA utilisation of the Schur-complement kernel: its own counter pass (counters never share a pass with trace
# domains other than --kernel-trace)
D=$OUT/pmc_mfma
(cd $ROOT && rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o p -- python scripts/bench_c4.py ${2:-1000000} ${3:-200} ${4:-0} > $D.log 2>&1)
python - <<PY
import csv, glob, collections
f = glob.glob("$D/**/p_counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("void gps::", "").split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for kt in glob.glob("$D/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(kt)):
        dur[r["Kernel_Name"].replace("void gps::", "").split("(")[0][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["", "## MFMA counters (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES ..., own pass)", "",
         "| kernel | launches | fp64 MFMA GFLOP / launch (MOPS x 512) | avg launch ms (this pass) | TFLOP/s | frac of 78.6 TF fp64 MFMA peak | MFMA busy cycles / launch / 1024 SIMDs / (duration x 2.4 GHz) |", "|---|---|---|---|---|---|---|"]
for k, c in agg.items():
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) <= 0:
        continue
    n = cnt[(k, "SQ_INSTS_VALU_MFMA_MOPS_F64")]
    mops, busy, gui = c["SQ_INSTS_VALU_MFMA_MOPS_F64"], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    ms = sum(dur[k]) / max(len(dur[k]), 1) / 1e6
    gf = mops * 512 / n / 1e9
    lines.append("| %s | %d | %.2f | %.3f | %.1f | %.2f | %.2f |" % (k, n, gf, ms, gf / ms if ms else 0.0, gf / ms / 78.6 if ms else 0.0,
                                                               busy / n / 1024 / (ms * 1e-3 * 2.4e9) if ms else 0.0))
open("$OUT/kernel_stats.md", "a").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
