#!/usr/bin/env python
"""Developer aid: per-launch timeline of the last npw_dpotrf_lower call in a rocprofv3 kernel trace CSV.
usage: potrf_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "tril_copy" in r["Kernel_Name"]]
i0 = starts[-1]
seq = []
for r in rows[i0:]:
    name = r["Kernel_Name"]
    if seq and not any(k in name for k in ("gemm", "potrf_diag", "potrf_panel", "splitk", "fillBuffer")):
        break
    seq.append(r)
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
tot = {}
for k, r in enumerate(seq):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = "diag" if ("potrf_diag" in name or "potrf_panel" in name) else ("tril" if "tril" in name else ("fill" if "fillBuffer" in name else "gemm"))
    g = f'{r.get("Grid_Size_X", r.get("Grid_Size", "?"))}/{r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))}'
    d = tot.setdefault(short, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    d[2] += (s - prev_end) / 1e3
    if k < 16 or k > len(seq) - 8:
        print(f"{k:4d} {short:5s} grid {g:>12s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f} us")
    prev_end = e
print("total span %.1f us" % ((prev_end - t0) / 1e3))
for k, (n, dur, gap) in tot.items():
    print(f"{k}: {n} launches, busy {dur:.1f} us, gaps before {gap:.1f} us")
