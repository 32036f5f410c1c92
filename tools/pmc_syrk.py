#!/usr/bin/env python
"""Developer aid: average PMC counter values of the tagged syrk kernel from a rocprofv3 counter_collection CSV."""
import collections
import csv
import re
import sys

agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        # TAG 1 = the tile-level trailing update; later template arguments (waves along n, ...) may follow the tag
        if re.search(r"gemm_kernel<double, 128, 128, 16, true, true, false, 1[,>]", r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if not agg:
    raise SystemExit("pmc_syrk.py: no row of the trailing-update kernel in " + " ".join(sys.argv[1:]) + " (kernel renamed?)")
for c, v in sorted(agg.items()):
    print(f"{c:32s} n={len(v):3d} avg={sum(v) / len(v):.4e}")
