#!/usr/bin/env python
"""Developer aid: repeat npw_dgeqrt_batched and check every result (R^T R = A^T A per tile) and every wall time --
catches lost hand-offs of the panel kernel (a timed-out spin shows as a multi-second call and a wrong R).
    python tools/qr_soak.py [count] [reps] [m] [n]          ($QR_SOAK_NO_T=1: the R-only form, T == NULL)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend

be = get_backend()
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
WANT_T = os.environ.get("QR_SOAK_NO_T", "0") != "1"
A = [be.fill_random((m, n), i + 1) for i in range(cnt)]
G = [be.gemm(a, a, True, False) for a in A]
gn = [be.sumsq(g) for g in G]
times, worst = [], 0.0
for rep in range(reps):
    be.synchronize()
    t0 = time.time()
    out = be.geqrt_batched(A, want_t=WANT_T) if cnt > 1 else [be.geqrt(A[0], want_t=WANT_T)]
    be.synchronize()
    times.append(time.time() - t0)
    for (V, T, R), g, g2 in zip(out, G, gn):
        err = np.sqrt(be.sumsq(be.gemm(R, R, True, False, alpha=1.0, beta=-1.0, C=g)) / g2)
        worst = max(worst, err)
    del out
ts = np.array(times) * 1e3
print(f"geqrt x{cnt} {m}x{n}{'' if WANT_T else ' (R only)'}, {reps} reps: median {np.median(ts):.2f} ms ({np.median(ts) / cnt:.2f} per tile), min {ts.min():.2f}, "
      f"max {ts.max():.2f}; worst |R^T R - A^T A| / |A^T A| = {worst:.2e}")
print("all:", " ".join(f"{t:.1f}" for t in ts))
print("expired hand-off waits:", be.qr_handoff_timeouts())
