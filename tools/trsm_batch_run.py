#!/usr/bin/env python
"""Developer aid: HipBackend.trsm one by one vs trsm_batched for `count` right-hand sides (4096^2 tiles)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend

be = get_backend()
n = 4096
G = be.fill_random((n, 256), seed=5)
A = be.add_diag(be.gemm(G, G, False, True), float(n))
L, info = be.chol(A)
for count in (2, 3, 7, 15):
    Ys = [be.fill_random((n, n), seed=10 + i) for i in range(count)]
    for rep in range(3):
        be.synchronize(); t0 = time.time()
        X = [be.trsm(L, y, exact_zero=False) for y in Ys]
        be.synchronize(); t1 = time.time()
        Xb = be.trsm_batched(L, Ys, exact_zero=False)
        be.synchronize(); t2 = time.time()
    print(f"count {count}: one by one {1e3 * (t1 - t0) / count:.3f} ms each, batched {1e3 * (t2 - t1) / count:.3f} ms each")
