#!/usr/bin/env python
"""Developer aid: average PMC counter values of the tagged syrk kernel from a rocprofv3 counter_collection CSV."""
import collections
import csv
import sys

agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "gemm_kernel<double, 128, 128, 16, true, true, false, 1>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:32s} n={len(v):3d} avg={sum(v) / len(v):.4e}")
