#!/usr/bin/env python
"""Developer aid: the trailing-update kernel (gemm_kernel<double,128,128,16,true,true,false,1>) in a rocprofv3
kernel-trace CSV, grouped by launch form: tiles per launch (grid z) and queue.  A launch processes `z` 4096^2 tile
updates; per-tile duration = launch duration / z.     python tools/trace_syrk_table.py <kernel_trace.csv> [tile]"""
import collections
import csv
import sys

tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_kernel<double, 128, 128, 16, true, true, false, 1>" in r["Kernel_Name"]]
groups = collections.defaultdict(list)
for r in rows:
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    z = int(r["Grid_Size_Z"])
    groups[(r.get("Queue_Id", "?"), wgs, z)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
flop = 2.0 * tile ** 3
print("| queue | workgroups x tiles per launch | launches | tile updates | avg launch ms | ms per tile | TFLOP/s |")
print("|---|---|---:|---:|---:|---:|---:|")
tot_t = tot_n = 0
for (q, wgs, z), d in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][2])):
    avg = sum(d) / len(d)
    print("| %s | %d x %d | %d | %d | %.4f | %.4f | %.2f |" % (q, wgs, z, len(d), len(d) * z, avg, avg / z, flop * z / (avg * 1e-3) / 1e12))
    tot_t += sum(d)
    tot_n += len(d) * z
print("all forms: %d tile updates, %.4f ms per tile, %.2f TFLOP/s" % (tot_n, tot_t / tot_n, flop / (tot_t / tot_n * 1e-3) / 1e12))
