#!/usr/bin/env python
"""Developer aid: repeat the latency-bound kernels (potrf on many sizes and on several streams at once, big potrf + trsm
back to back, batched QR) and check results against NumPy / for bitwise repeatability.  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NUMPYWREN_AMD_STREAMS", "4")
import numpy as np
from numpywren_amd.device import get_backend
be = get_backend()
rng = np.random.default_rng(1)
bad = 0
t0 = time.time()
# potrf on many sizes, on several streams at once, against numpy
sizes = [128, 129, 200, 256, 300, 384, 511, 512, 513, 640, 1000, 1024, 1500, 2048]
mats = {}
for n in sizes:
    x = rng.standard_normal((n, n)); a = x @ x.T + n * np.eye(n)
    mats[n] = (be.to_device(a), np.linalg.cholesky(a))
for rep in range(30):
    outs = []
    for i, n in enumerate(sizes):
        L, info = be.chol(mats[n][0], be.streams[i % 4])
        outs.append((n, L, info))
    for n, L, info in outs:
        err = np.abs(be.to_host(L) - mats[n][1]).max()
        if err > 1e-10 * n or be.read_flag(info) != 0:
            bad += 1
            print("BAD potrf n=%d err=%.3e rep=%d" % (n, err, rep))
# big tiles back to back + trsm consumers
A = be.gemm(be.fill_random((4096, 4096), 3), be.fill_random((4096, 4096), 3), False, True)
A = be.add_diag(A, 4096.0 * 4)
ref = None
for rep in range(40):
    L, info = be.chol(A, be.streams[rep % 4])
    X = be.trsm(L, A, be.streams[(rep + 1) % 4])
    if rep % 10 == 0:   # (sumsq itself accumulates with atomics: compare the tiles, not their checksums)
        cur = (be.to_host(L), be.to_host(X))
        if ref is None:
            ref = cur
        elif not (np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1])):
            bad += 1
            print("BAD nondeterministic big potrf/trsm", rep)
# QR batched / tpqrt repeatedly
Q = [be.fill_random((1024, 512), 10 + i) for i in range(6)]
base = [[be.to_host(t) for t in tr] for tr in be.geqrt_batched(Q)]
for rep in range(20):
    got = be.geqrt_batched(Q, be.streams[rep % 4])
    for tr, b3 in zip(got, base):
        for t, b in zip(tr, b3):
            if not np.array_equal(be.to_host(t), b):
                bad += 1
                print("BAD nondeterministic batched qr", rep)
print("soak done in %.1f s, bad = %d" % (time.time() - t0, bad))
