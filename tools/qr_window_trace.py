#!/usr/bin/env python
"""Developer aid: every kernel of every queue inside a time window of the LAST batched QR call of a rocprofv3 kernel-trace CSV
(window given in ms from the call's first panel kernel): what runs beside the panel chain there.
    python tools/qr_window_trace.py <kernel_trace.csv> <from_ms> <to_ms> [panels_per_call=128]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = float(sys.argv[2]), float(sys.argv[3])
npan = int(sys.argv[4]) if len(sys.argv) > 4 else 128
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pan = [i for i, r in enumerate(rows) if "qr_panel" in r["Kernel_Name"]]
t0 = int(rows[pan[-npan]]["Start_Timestamp"])


def short(n):
    n = n.replace("npw::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:52]


for r in rows[pan[-npan]:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e < lo or s > hi:
        continue
    g = "%dx%sx%s" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("Grid_Size_Y", "1"), r.get("Grid_Size_Z", "1"))
    print("q%-2s %9.3f .. %9.3f (%7.1f us)  %-52s grid %s" % (r["Queue_Id"], s, e, (e - s) * 1e3, short(r["Kernel_Name"]), g))
