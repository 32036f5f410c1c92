#!/usr/bin/env python
"""Developer aid: aggregate a rocprofv3 kernel-trace CSV over its last `span_ms` milliseconds: per-kernel totals and
the fraction of wall time with at least one kernel running.   python tools/trace_agg.py <csv> [span_ms]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
             int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])) for r in rows)
tend = ks[-1][1]
seg = [k for k in ks if k[0] >= tend - span * 1e6]
t0 = seg[0][0]


def short(n):
    n = n.replace("npw::(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n[:70]


agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, gx, gy, gz in seg:
    agg[short(n)][0] += 1
    agg[short(n)][1] += e - s
ev = sorted([(s, 1) for s, *_ in seg] + [(e, -1) for _, e, *_ in seg])
run = busy = 0
last = None
for t, d in ev:
    if run > 0:
        busy += t - last
    run += d
    last = t
print(f"window {(tend - t0) / 1e6:.3f} ms, {len(seg)} kernels, GPU busy {busy / 1e6:.3f} ms, sum of kernel times {sum(v[1] for v in agg.values()) / 1e6:.3f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {n:72s} n={c:6d} total={t / 1e6:9.3f} ms avg={t / c / 1e3:9.1f} us")
