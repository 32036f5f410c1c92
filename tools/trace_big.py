#!/usr/bin/env python
"""Developer aid: the kernels longer than `min_us` in the last `span_ms` of a rocprofv3 kernel-trace CSV, in time order.
    python tools/trace_big.py <csv> [span_ms] [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 500.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tend = int(rows[-1]["End_Timestamp"])
seg = [r for r in rows if int(r["Start_Timestamp"]) >= tend - span * 1e6]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s < min_us * 1e3:
        continue
    print("%8.2f dur %7.2f ms grid %5d %3s %3s stream %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
          r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("Stream_Id", "?"), r["Kernel_Name"].replace("npw::(anonymous namespace)::", "").replace("void ", "")[:64]))
