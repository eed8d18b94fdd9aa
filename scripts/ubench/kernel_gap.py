"""Median idle time before each launch of scripts/ubench/kernel_gap, by (previous kernel, its grid):  python scripts/ubench/kernel_gap.py <rocprofv3 output dir>"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name(a), int(a["Grid_Size_X"]), name(b))
    acc[key].append(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3))
print("| kernel in front | its grid | its duration us | kernel behind | idle time between them us (median) |\n|---|---|---|---|---|")
for (ka, g, kb), v in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    gaps = sorted(x[0] for x in v); durs = sorted(x[1] for x in v)
    print("| %s | %d | %.1f | %s | %.2f |" % (ka, g, durs[len(durs) // 2], kb, gaps[len(gaps) // 2]))
