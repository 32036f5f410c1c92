#!/usr/bin/env python
"""Developer aid: where the wall time of the LAST batched QR call of a rocprofv3 kernel-trace CSV goes.
The panel chain's queue (the one the panel kernels run on) is split into kernel time by name and idle gaps; the other
queues are listed by kernel name with their busy time.  Also prints the chain per outer block (4 panels each).
    python tools/qr_chain_trace.py <kernel_trace.csv> [panels_per_call=128]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
npan = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("npw::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]


pan = [i for i, r in enumerate(rows) if "qr_panel" in r["Kernel_Name"]]
first = pan[-npan]
chain_q = rows[first]["Queue_Id"]
t0 = int(rows[first]["Start_Timestamp"])
# the call ends with the last kernel that starts before the next long idle period after the last panel
end_i = pan[-1]
while end_i + 1 < len(rows) and int(rows[end_i + 1]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in rows[first:end_i + 1]) < 2_000_000 \
        and not any(k in rows[end_i + 1]["Kernel_Name"] for k in ("fill_random", "sumsq")):
    end_i += 1
seg = rows[first:end_i + 1]
t1 = max(int(r["End_Timestamp"]) for r in seg)
print("call span %.2f ms, %d kernels, chain queue %s" % ((t1 - t0) / 1e6, len(seg), chain_q))
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: kv[0] != chain_q):
    agg = collections.defaultdict(lambda: [0, 0])
    busy, gaps, last_end = 0, 0, None
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        agg[short(r["Kernel_Name"])][0] += 1
        agg[short(r["Kernel_Name"])][1] += e - s
        busy += e - s
        if last_end is not None and s > last_end:
            gaps += s - last_end
        last_end = max(e, last_end or e)
    print("queue %s%s: busy %.2f ms, idle between its kernels %.2f ms" % (q, " (chain)" if q == chain_q else "", busy / 1e6, gaps / 1e6))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("    %-62s n=%5d total=%8.3f ms avg=%8.1f us" % (n, c, t / 1e6, t / c / 1e3))
# per outer block of 4 panels: time from the block's first panel start to the next block's first panel start
ps = [rows[i] for i in pan[-npan:]]
print("per outer block (4 panels): block span | panel kernel time | rows of first panel grid")
for b in range(0, npan, 4):
    s = int(ps[b]["Start_Timestamp"])
    e = int(ps[b + 4]["Start_Timestamp"]) if b + 4 < npan else t1
    pk = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ps[b:b + 4])
    if b % 16 == 0 or b + 4 >= npan:
        print("  block %3d: %8.3f ms | %7.3f ms | grid %s x %s" % (b // 4, (e - s) / 1e6, pk / 1e6,
              int(ps[b]["Grid_Size_X"]) // int(ps[b]["Workgroup_Size_X"]), ps[b].get("Grid_Size_Y", "?")))
