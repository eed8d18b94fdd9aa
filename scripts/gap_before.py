"""Idle time before each launch of a kernel in a rocprofv3 --kernel-trace csv:  python scripts/gap_before.py <csv> <name-substring>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
gaps, durs = [], []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if sys.argv[2] in r["Kernel_Name"] and prev_end is not None:
        gaps.append((s - prev_end) / 1e3); durs.append((e - s) / 1e3)
    prev_end = e
print(sys.argv[2], "launches", len(gaps), "median gap us %.2f" % sorted(gaps)[len(gaps) // 2], "median dur us %.2f" % sorted(durs)[len(durs) // 2])
