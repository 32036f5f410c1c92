#!/usr/bin/env python
"""Developer aid: per-kernel table from rocprofv3 --pmc counter_collection CSVs (one or several passes): dispatches, and per
dispatch the averages of the counters found; with GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES the share of the 1024 SIMDs' matrix
pipes that was busy (busy cycles / (1024 x GUI_ACTIVE / 8 XCDs)), with FETCH_SIZE / WRITE_SIZE (KB; gfx950 reports wide coalesced
reads at half their size: MI355X_MICROARCH.md -> 2 x FETCH) the HBM-side bytes per dispatch.
    python tools/pmc_table.py <csv> [<csv> ...]"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("npw::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    return name[:78]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    n = max(len(v) for v in c.values())
    avg = {name: sum(v) / len(v) for name, v in c.items()}
    busy = None
    if avg.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        busy = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0)
    hbm = None
    if "FETCH_SIZE" in avg or "WRITE_SIZE" in avg:
        hbm = (2.0 * avg.get("FETCH_SIZE", 0.0) + avg.get("WRITE_SIZE", 0.0)) * 1024.0
    rows.append((avg.get("GRBM_GUI_ACTIVE", 0.0) * n, k, n, busy, hbm, avg))
rows.sort(reverse=True)
print("%-78s %6s %10s %14s" % ("kernel", "disp.", "MFMA busy", "HBM B / disp."))
for _, k, n, busy, hbm, avg in rows[:24]:
    print("%-78s %6d %10s %14s" % (k, n, "%.3f" % busy if busy is not None else "-", "%.3e" % hbm if hbm is not None else "-"))
