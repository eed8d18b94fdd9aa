#!/bin/bash
# where do the 40 us per block of k_fat_elim (the LDS kernel: GPSLAM_FAT_ELIM_ROWS=0) go -- library built with -DGPS_FSY_DBG,
# bits 8: no factorisation, 16: no products; results are wrong on purpose
ROOT=$(pwd)
export TMPDIR=/tmp
for d in 0 8 16 24; do
  OUT=$ROOT/gpurun_out/r3w/d$d; mkdir -p $OUT
  (cd /tmp && GPSLAM_FAT_ELIM_ROWS=0 GPSLAM_FSY_DBG=$d PYTHONPATH=$ROOT rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOT/scripts/bench_c4.py 200000 > $OUT/run.log 2>&1)
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/t_kernel_trace.csv", recursive=True)
if not f: print("dbg=$d: no trace"); raise SystemExit
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_fat_elim" in r["Kernel_Name"]:
        agg[int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("dbg=$d", {k: round(sum(v) / len(v), 1) for k, v in sorted(agg.items())})
PY
done
