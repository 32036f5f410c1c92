#!/usr/bin/env python
"""Developer aid: run npw_dpotrf_lower on one 4096^2 tile a few times (for rocprofv3 --kernel-trace) and, given the
trace CSV, print the per-launch timeline of the last call.
    python tools/chol_trace.py run [n]          # the workload
    python tools/chol_trace.py show <kernel_trace.csv>"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import time
    from numpywren_amd.device import get_backend
    be = get_backend()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    G = be.fill_random((n, 256), seed=5)
    A = be.add_diag(be.gemm(G, G, False, True), float(n))
    for rep in range(4):
        be.synchronize()
        t0 = time.time()
        L, info = be.chol(A)
        be.synchronize()
        print(f"chol({n}) {1e3 * (time.time() - t0):.3f} ms info {be.read_flag(info)}")
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "tril_copy" in r["Kernel_Name"]]
    seq = rows[starts[-1]:]
    t0 = int(seq[0]["Start_Timestamp"])
    prev = t0
    for k, r in enumerate(seq):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("npw::(anonymous namespace)::", "")[:28]
        print(f"{k:3d} {name:28s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:5.1f}")
        prev = e
        if "complete" in name or k > 60:
            break
    print("span us", (prev - t0) / 1e3)
