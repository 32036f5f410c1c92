#!/usr/bin/env python
"""Developer aid: how busy the host link and the kernels are during a budgeted run -- from a rocprofv3
--kernel-trace --memory-copy-trace directory.  Prints, for the last `span_ms`, the union busy time of kernels, of
host-to-device and of device-to-host copies, their pairwise overlaps, and a coarse utilisation strip per 10 ms.
    python tools/spill_timeline.py <dir> [span_ms]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0


def load(pattern):
    rows = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in load("*kernel_trace.csv")]
cop = load("*memory_copy_trace.csv")
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cop if "HOST_TO_DEVICE" in r.get("Direction", "")]
d2h = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cop if "DEVICE_TO_HOST" in r.get("Direction", "")]
tend = max(e for _, e in kern + h2d + d2h)
t0 = tend - int(span * 1e6)


def clip(iv):
    return sorted((max(s, t0), e) for s, e in iv if e > t0)


def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(iv):
    return sum(e - s for s, e in iv) / 1e6


def inter(a, b):
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


K, H, D = union(clip(kern)), union(clip(h2d)), union(clip(d2h))
print("last %.0f ms: kernels busy %.1f ms, H2D busy %.1f ms (%d copies), D2H busy %.1f ms (%d copies)"
      % (span, total(K), total(H), len(clip(h2d)), total(D), len(clip(d2h))))
print("overlap: kernels&H2D %.1f, kernels&D2H %.1f, H2D&D2H %.1f, anything busy %.1f ms"
      % (total(inter(K, H)), total(inter(K, D)), total(inter(H, D)), total(union(K + H + D))))
big = [(e - s) / 1e6 for s, e in clip(h2d) if e - s > 1e6]
if big:
    print("H2D copies > 1 ms: n=%d, mean %.2f ms" % (len(big), sum(big) / len(big)))
big = [(e - s) / 1e6 for s, e in clip(d2h) if e - s > 1e6]
if big:
    print("D2H copies > 1 ms: n=%d, mean %.2f ms" % (len(big), sum(big) / len(big)))
step = 10e6
n = int(span * 1e6 / step)
for name, iv in (("kern", K), ("h2d ", H), ("d2h ", D)):
    line = ""
    for b in range(n):
        lo, hi = t0 + b * step, t0 + (b + 1) * step
        f = total(inter(iv, [[lo, hi]])) / (step / 1e6)
        line += " .:-=+*#%@"[min(9, int(f * 9.999))]
    print(name, line)
