#!/bin/bash
# rocprofv3 kernel timeline of the last Gauss-Newton iteration of scripts/profile_iter.py (run on the GPU box from the repo root)
OUT=$(pwd)/gpurun_out/trace_iter
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python scripts/profile_iter.py "$@" > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/trace_iter/**/t_kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_retract' in r['Kernel_Name']]
e, s0 = idx[-1], idx[-2] + 1
t0 = int(rows[s0]['Start_Timestamp'])
for r in rows[s0:e + 1]:
    st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%-44s grid %8s  start %8.1f  dur %7.1f' % (r['Kernel_Name'][:44], r.get('Grid_Size', r.get('Grid_Size_X', '?')), st / 1e3, (en - st) / 1e3))
PY
