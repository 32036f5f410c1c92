"""Summarise a rocprofv3 kernel-trace CSV: per-kernel totals and a coarse timeline of the last step."""
import csv, re, sys, collections
path = sys.argv[1]
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = list(csv.DictReader(open(path)))
def short(n):
    n = n.replace('npw::(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'gemm_kernel<(\w+), (\d+), (\d+), (\d+), (\w+), (\w+), (\w+)>', n)
    if m:
        return f"gemm<{m.group(1)[0]},{m.group(2)}x{m.group(3)}x{m.group(4)},{'KC' if m.group(5)=='true' else 'MC'},{'KC' if m.group(6)=='true' else 'MC'}{',edge' if m.group(7)=='true' else ''}>"
    return re.sub(r'\(.*', '', n)[:40]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Stream_Id']),
             int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows)
tend = ks[-1][1]
seg = [k for k in ks if k[0] >= tend - span_ms * 1e6]
t0 = seg[0][0]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, st, g in seg:
    agg[(n, st)][0] += 1
    agg[(n, st)][1] += e - s
print(f"last {span_ms} ms: {len(seg)} kernels")
for (n, st), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  stream {st:2d} {n:40s} n={v[0]:5d} total={v[1]/1e6:8.3f} ms avg={v[1]/v[0]/1e3:8.1f} us")
# busy union
ev = sorted([(s, 1) for s, e, *_ in seg] + [(e, -1) for s, e, *_ in seg])
run = 0; busy = 0; last = None
for t, d in ev:
    if run > 0: busy += t - last
    run += d; last = t
print(f"  GPU busy (any kernel) {busy/1e6:.3f} ms of {(seg[-1][1]-t0)/1e6:.3f} ms")
print("  big kernels (>0.3 ms):")
for s, e, n, st, g in seg:
    if e - s > 300e3:
        print(f"    {(s-t0)/1e6:8.3f} -> {(e-t0)/1e6:8.3f}  {(e-s)/1e6:6.3f} ms stream {st} grid {g:5d} {n}")
# idle gaps > 15 us inside the window
prev_end = None
gaps = []
for s, e, n, st, g in seg:
    if prev_end is not None and s - prev_end > 15e3:
        gaps.append((s - prev_end, (prev_end - t0) / 1e6, n))
    prev_end = max(prev_end or e, e)
gaps.sort(reverse=True)
print(f"  idle gaps > 15 us: {len(gaps)}, total {sum(g[0] for g in gaps)/1e6:.3f} ms")
for d, at, n in gaps[:12]:
    print(f"    {d/1e3:8.1f} us before {n} at {at:8.3f} ms")
