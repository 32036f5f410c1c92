#!/usr/bin/env python
"""Developer aid: HipBackend.geqrt one by one vs geqrt_batched vs tpqrt_batched -- timing and agreement."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from numpywren_amd.device import get_backend
be = get_backend()
for (m, n, cnt) in ((256, 128, 3), (200, 67, 4), (1024, 512, 5), (4096, 4096, 8), (8192, 4096, 8), (4096, 4096, 16)):
    A = [be.fill_random((m, n), i + 1) for i in range(cnt)]
    single = [be.geqrt(a) for a in A]
    be.synchronize()
    t0 = time.time(); single = [be.geqrt(a) for a in A]; be.synchronize(); t1 = time.time()
    bat = be.geqrt_batched(A); be.synchronize()
    t2 = time.time(); bat = be.geqrt_batched(A); be.synchronize(); t3 = time.time()
    worst = 0.0
    if m <= 1024:
        for s3, b3 in zip(single, bat):
            for x, y in zip(s3, b3):
                worst = max(worst, float(np.abs(be.to_host(x) - be.to_host(y)).max()))
    else:
        for s3, b3 in zip(single, bat):
            for x, y in zip(s3, b3):
                d = be.axpby(1.0, x, -1.0, y)
                worst = max(worst, float(np.sqrt(be.sumsq(d))))
    print("m=%d n=%d count=%d: single %.2f ms/QR, batched %.2f ms/QR, max diff %.3g" % (m, n, cnt, (t1 - t0) * 1e3 / cnt, (t3 - t2) * 1e3 / cnt, worst))
for (n, cnt) in ((4096, 1), (4096, 8), (4096, 16)):
    P = [(be.geqrt(be.fill_random((n, n), 2 * i + 1))[2], be.geqrt(be.fill_random((n, n), 2 * i + 2))[2]) for i in range(cnt)]
    r = be.tpqrt_batched(P); be.synchronize()
    t2 = time.time(); r = be.tpqrt_batched(P); be.synchronize(); t3 = time.time()
    dense = be.geqrt(be.vstack(list(P[0])))
    d = max(float(np.sqrt(be.sumsq(be.axpby(1.0, x, -1.0, y)))) for x, y in zip(r[0], dense))
    print("stacked triangles n=%d count=%d: %.2f ms per node, max diff vs dense %.3g" % (n, cnt, (t3 - t2) * 1e3 / cnt, d))
