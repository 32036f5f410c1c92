"""Developer aid: 4096^2 trailing updates one by one vs HipBackend.syrk_batched (one launch per <= 16 tiles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
for count in (2, 4, 8, 16):
    P = [(be.fill_random((n, n), 3 * i), be.fill_random((n, n), 3 * i + 1), be.fill_random((n, n), 3 * i + 2)) for i in range(count)]
    for rep in range(3):
        be.synchronize(); t0 = time.time()
        o1 = [be.syrk(*p, exact_zero=False) for p in P]
        be.synchronize(); t1 = time.time()
        o2 = be.syrk_batched(P, exact_zero=False)
        be.synchronize(); t2 = time.time()
        del o1, o2
    print("count %2d: one by one %.4f ms each (%.2f TFLOP/s), batched %.4f ms each (%.2f TFLOP/s)" % (
        count, 1e3 * (t1 - t0) / count, 2 * n ** 3 * count / (t1 - t0) / 1e12, 1e3 * (t2 - t1) / count, 2 * n ** 3 * count / (t2 - t1) / 1e12))
