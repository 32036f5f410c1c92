#!/usr/bin/env python
"""Developer aid: task-level timeline of the last `span_ms` of a rocprofv3 kernel-trace CSV.  Kernels longer than
`min_us` get a line each; runs of shorter kernels with the same name on the same queue are folded into one line
(first start .. last end, count).    python tools/trace_timeline.py <csv> [span_ms] [min_us]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tend = max(int(r["End_Timestamp"]) for r in rows)
seg = [r for r in rows if int(r["Start_Timestamp"]) >= tend - span * 1e6]
t0 = int(seg[0]["Start_Timestamp"])


def short(n):
    n = n.replace("npw::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:56]


runs = {}   # (queue, name) -> [start, end, count]
out = []


def flush(key):
    s, e, c = runs.pop(key)
    out.append((s, e, key[0], "%s x%d" % (key[1], c)))


for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = short(r["Kernel_Name"])
    if e - s >= min_us * 1e3:
        out.append((s, e, q, "%s grid %d" % (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))))
        continue
    key = (q, name)
    if key in runs and s - runs[key][1] < 200e3:
        runs[key][1] = e
        runs[key][2] += 1
    else:
        if key in runs:
            flush(key)
        runs[key] = [s, e, 1]
for key in list(runs):
    flush(key)
out.sort()
for s, e, q, what in out:
    if e - s < 20e3:
        continue
    print("%9.3f .. %9.3f  (%7.3f ms)  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, what))
