#!/usr/bin/env python
"""Developer aid: two batched QR calls of `half` tiles each on two streams, the second one enqueued `delay` ms after the
first (its far-update-heavy early blocks then run beside the first call's latency-bound late blocks), against one call
of 2 * half tiles.      python tools/qr_pingpong_probe.py [half=16] [delays_ms=0,10,20,30]     ($QR_SOAK_NO_T=1: R only)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend

be = get_backend()
half = int(sys.argv[1]) if len(sys.argv) > 1 else 16
delays = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,10,20,30").split(",")]
want_t = os.environ.get("QR_SOAK_NO_T", "0") != "1"
n = 4096
A = [be.fill_random((n, n), i + 1) for i in range(2 * half)]
G = [be.gemm(a, a, True, False) for a in A]
gn = [be.sumsq(g) for g in G]
s1, s2 = be.streams[0], be.streams[1]


def check(outs):
    worst = 0.0
    for (V, T, R), g, g2 in zip(outs, G, gn):
        worst = max(worst, np.sqrt(be.sumsq(be.gemm(R, R, True, False, alpha=1.0, beta=-1.0, C=g)) / g2))
    return worst


for rep in range(3):
    be.synchronize()
    t0 = time.time()
    out = be.geqrt_batched(A, stream=s1, want_t=want_t)
    be.synchronize()
    dt = time.time() - t0
print(f"one call x{2 * half}: {dt * 1e3:.2f} ms, worst residual {check(out):.2e}")
del out
for rep in range(2):
    be.synchronize()
    t0 = time.time()
    out = be.geqrt_batched(A[:half], stream=s1, want_t=want_t)
    be.synchronize()
    dt = time.time() - t0
print(f"one call x{half}: {dt * 1e3:.2f} ms")
del out
for d in delays:
    for rep in range(3):
        be.synchronize()
        t0 = time.time()
        o1 = be.geqrt_batched(A[:half], stream=s1, want_t=want_t)
        t1 = time.time()
        while time.time() - t0 < d * 1e-3:
            pass
        o2 = be.geqrt_batched(A[half:], stream=s2, want_t=want_t)
        t2 = time.time()
        be.synchronize()
        dt = time.time() - t0
    print(f"two calls x{half}, second {d:.0f} ms later: {dt * 1e3:.2f} ms  (host enqueue {1e3 * (t1 - t0):.1f} + {1e3 * (t2 - t1):.1f} ms), "
          f"worst residual {check(o1 + o2):.2e}")
    del o1, o2
