"""Developer aid: 4096^3 NT product with and without the C operand (beta = 1 vs 0), and for K = 4096 / 8192: what the
epilogue's C read costs per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
S = be.fill_random((n, n), 1)
for K in (4096, 8192):
    X = be.fill_random((n, K), 2); Y = be.fill_random((n, K), 3)
    for name, kw in (("beta=1", dict(alpha=-1.0, beta=1.0, C=S)), ("beta=0", dict(alpha=-1.0))):
        for rep in range(3):
            be.synchronize(); t0 = time.time()
            for i in range(10):
                D = be.gemm(X, Y, False, True, **kw)
            be.synchronize(); dt = (time.time() - t0) / 10
        print("K %5d %s  %.4f ms  %.2f TFLOP/s" % (K, name, dt * 1e3, 2.0 * n * n * K / dt / 1e12))
