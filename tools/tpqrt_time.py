"""Developer aid: HipBackend.tpqrt_batched (stacked-triangle QR, the TSQR tree nodes) for several batch sizes (4096^2 triangles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
R = [be.geqrt(be.fill_random((n, n), i + 1))[2] for i in range(8)]
be.synchronize()
for cnt in (1, 2, 4, 8, 16, 32):
    P = [(R[(2 * i) % 8], R[(2 * i + 1) % 8]) for i in range(cnt)]
    ts = []
    for rep in range(int(os.environ.get("TPQRT_REPS", "2"))):
        be.synchronize(); t0 = time.time()
        r = be.tpqrt_batched(P, want_t=os.environ.get("QR_SOAK_NO_T", "0") != "1")
        be.synchronize(); dt = time.time() - t0
        ts.append(dt * 1e3)
        del r
    print("tpqrt x%d: %.2f ms = %.2f ms per node   (all: %s)" % (cnt, dt * 1e3, dt * 1e3 / cnt, " ".join("%.1f" % t for t in ts)))
