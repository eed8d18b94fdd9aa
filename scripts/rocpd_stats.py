#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (markdown)."""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
rows = con.execute("""
  select s.display_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
         max(d.grid_size_x), max(d.workgroup_size_x), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count),
         max(d.private_segment_size), max(d.group_segment_size)
  from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
  group by s.display_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | vgpr | agpr | sgpr | scratch B | lds B |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = r[0][:90]
    print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %d | %d | %d | %d | %d | %d | %d |" % (
        name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
