#!/usr/bin/env python
"""Developer aid: timeline of the last npw_dtrsm_rltn call in a rocprofv3 kernel-trace CSV of tools/devcheck."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "trtri_diag_kernel" in r["Kernel_Name"]]
i0 = idx[-1]
seq = []
for r in rows[i0 - 1:]:
    n = r["Kernel_Name"]
    if seq and not any(k in n for k in ("gemm", "trtri", "fillBuffer")):
        break
    seq.append(r)
t0 = int(seq[0]["Start_Timestamp"])
prev = t0
for k, r in enumerate(seq):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("npw::(anonymous namespace)::", "").replace("void ", "")[:58]
    print(f"{k:3d} {n:58s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}x{r.get('Grid_Size_Z', '1'):>3s} "
          f"start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:5.1f}")
    prev = e
print("span %.1f us" % ((prev - t0) / 1e3))
