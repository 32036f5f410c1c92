#!/usr/bin/env python
"""Developer aid: per-kernel totals of the LAST npw_dgeqrt call in a rocprofv3 kernel-trace CSV (devcheck run)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "qr_panel" in r["Kernel_Name"]]
last = idx[-1]
# walk back to the start of that factorisation: the first panel kernel after a gap without panel kernels > 5 ms
start = last
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 3_000_000 and \
        any(k in rows[start - 1]["Kernel_Name"] for k in ("qr_panel", "gemm", "splitk", "fillBuffer", "copyBuffer")):
    start -= 1
seg = rows[start:last + 40]
t0 = int(seg[0]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"].replace("npw::(anonymous namespace)::", "").split("(")[0][:60]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("span %.2f ms, %d kernels" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e6, len(seg)))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{n:62s} n={c:5d} total={t / 1e6:8.3f} ms avg={t / c / 1e3:8.1f} us")
