#!/usr/bin/env python3
"""Turn gpurun_out/<tag> (scripts/collect_profiles.sh) into profiles/<name>_kernel_stats.md and profiles/<name>_pmc.json.

  python scripts/summarise_profiles.py gpurun_out/r1v2 round1_v2
  python scripts/summarise_profiles.py gpurun_out/r3_1e6 round3_1e6 1000000     (scripts/collect_profiles_1e6.sh: the north-star size;
                                                                                 profiles/latest_pmc.json -- bench.py's traffic source -- is left alone)
"""
import collections
import csv
import glob
import json
import os
import sys

src, name = sys.argv[1], sys.argv[2]
states = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
what = sys.argv[4] if len(sys.argv) > 4 else None     # other targets (scripts/collect_profiles_cfg.sh): the command that was profiled
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void gps::", "").replace("gps::", "")
    return k.split("(")[0][:70]


def find(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


out = ["# %s -- rocprofv3 --kernel-trace --stats of %s" % (name, ("`%s`" % what) if what else "`python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline`" if states == 100000 else "`python scripts/profile_iter.py %d` (Pose3 chain of %d states, 2 + 5 Gauss-Newton iterations)" % (states, states)), ""]
kt = find("trace/**/t_kernel_trace.csv")
per = collections.OrderedDict()
if kt:
    rows = list(csv.DictReader(open(kt)))
    agg = collections.defaultdict(list)
    meta = {}
    for r in rows:
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]))
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[key] = (r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
    tot = sum(sum(v) for v in agg.values())
    out += ["Per (kernel, grid) -- the solver kernels run once per hierarchy level, the grid tells the level apart.", "",
            "| kernel | grid | calls | total ms | avg us | min us | max us | % | wg | vgpr | agpr | sgpr | scratch | lds |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        m = meta[key]
        out.append("| %s | %d | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s | %s | %s |" % (
            key[0], key[1], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot, *m))
        per["%s@%d" % key] = dict(calls=len(v), avg_us=sum(v) / len(v) / 1e3)
    # one steady-state iteration as a timeline (the last k_retract to the one before it)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_retract" in r["Kernel_Name"]]
    if len(idx) > 3:
        a, b = idx[-3] + 1, idx[-2] + 1
        t0 = int(rows[a]["Start_Timestamp"])
        out += ["", "## One steady-state Gauss-Newton iteration (timeline)", "", "| # | kernel | grid | start us | dur us | gap before us |", "|---|---|---|---|---|---|"]
        prev_end = None
        busy = 0
        for n, r in enumerate(rows[a:b]):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (s - prev_end) / 1e3 if prev_end else 0.0
            out.append("| %d | %s | %s | %.2f | %.2f | %.2f |" % (n, short(r["Kernel_Name"]), r["Grid_Size_X"], (s - t0) / 1e3, (e - s) / 1e3, gap))
            prev_end = e
            busy += e - s
        span = int(rows[b - 1]["End_Timestamp"]) - t0
        out += ["", "span %.1f us, kernels busy %.1f us, gaps %.1f us, %d launches" % (span / 1e3, busy / 1e3, (span - busy) / 1e3, b - a)]

pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        key = "%s@%s" % (short(r["Kernel_Name"]), r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        for c, v in cs.items():
            pmc.setdefault(key, {})[c] = sum(v) / len(v)
    # durations of the same pass: counter collection serialises the dispatches, so kernels that overlap on two streams in the
    # plain trace are timed alone here (the counter pass adds a little to every launch)
    kt2 = glob.glob(os.path.join(d, "**", "p_kernel_trace.csv"), recursive=True)
    if kt2:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(kt2[0])):
            dur["%s@%s" % (short(r["Kernel_Name"]), r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for key, v in dur.items():
            if key in pmc:
                pmc[key].setdefault("serialised_us", []).append(sum(v) / len(v) / 1e3)
for c in pmc.values():
    if "serialised_us" in c:
        c["serialised_us"] = min(c["serialised_us"])
if pmc:
    out += ["", "## PMC counters (separate rocprofv3 --pmc passes over scripts/profile_iter.py, mean per launch)", "",
            "FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them; on gfx950 FETCH_SIZE under-reports wide coalesced",
            "reads by 2x (MI355X_MICROARCH.md, HBM section), so `hbm_read_MB` below = 2 x FETCH_SIZE.", "",
            "`ea_read_MB` = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B and `ea_write_MB` = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B)",
            "are the L2 -> fabric request counters sized explicitly (no correction needed); they are what bench.py reports as `traffic`.", "",
            "| kernel@grid | FETCH_SIZE KB | WRITE_SIZE KB | 2 x FETCH MB | WRITE MB | ea_read_MB | ea_write_MB | VALU insts/wave | SALU insts/wave | active quad-cycles/wave | wave quad-cycles/wave | wait_any % | us alone (counter pass) | (read + write) TB/s alone |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    def _traffic(c):
        return (32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0) +
                64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0)))
    for key, c in sorted(pmc.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) - kv[1].get("WRITE_SIZE", 0) - _traffic(kv[1]) / 1024):
        w = c.get("SQ_WAVES", 0) or 1
        wc = c.get("SQ_WAVE_CYCLES", 0)
        rd = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
        wr = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (c.get("TCC_EA0_WRREQ_sum", 0) - c.get("TCC_EA0_WRREQ_64B_sum", 0))
        c["ea_read_bytes"], c["ea_write_bytes"] = rd, wr
        us = c.get("serialised_us", 0.0)
        out.append("| %s | %.0f | %.0f | %.1f | %.1f | %.1f | %.1f | %.0f | %.0f | %.0f | %.0f | %.0f | %.1f | %.2f |" % (
            key, c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0), 2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6, rd / 1e6, wr / 1e6,
            c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w, wc / w,
            100.0 * c.get("SQ_WAIT_ANY", 0) / wc if wc else 0, us, (rd + wr) / us / 1e6 if us else 0.0))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
open(os.path.join(root, "profiles", name + "_kernel_stats.md"), "w").write("\n".join(out) + "\n")
for c in pmc.values():
    c["states"] = states          # scripts/profile_iter.py workload (default: BASELINE config 3, 1e5 Pose3 states)
blob = dict(source="rocprofv3 --pmc passes over scripts/profile_iter.py (scripts/collect_profiles.sh), mean per launch",
            fetch_size_correction=2.0, unit="FETCH_SIZE/WRITE_SIZE in KB, ea_*_bytes in bytes", kernels=pmc, kernel_trace=per)
for fn in ((name + "_pmc.json", "latest_pmc.json") if (states == 100000 and not what) else (name + "_pmc.json",)):
    json.dump(blob, open(os.path.join(root, "profiles", fn), "w"), indent=1, sort_keys=True)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    lines = [l for l in open(bj) if l.startswith("{")]
    if lines:
        open(os.path.join(root, "profiles", name + "_bench.json"), "w").write(lines[-1])
print("\n".join(out[:60]))
