"""Developer aid: fixed cost per launch of the 128x128 trailing-update tiling: D = C - A B^T for A, B (4096 x K) at several K;
T(K) = a + b K  ->  a = prologue + epilogue + launch, b = the k-loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
S = be.fill_random((n, n), 1)
res = {}
for K in (1024, 2048, 4096, 8192):
    X = be.fill_random((n, K), 2); Y = be.fill_random((n, K), 3)
    for rep in range(3):
        be.synchronize(); t0 = time.time()
        for i in range(10):
            D = be.gemm(X, Y, False, True, alpha=-1.0, beta=1.0, C=S)
        be.synchronize(); dt = (time.time() - t0) / 10
    res[K] = dt * 1e3
    print("K %5d  %.4f ms  %.2f TFLOP/s" % (K, dt * 1e3, 2.0 * n * n * K / dt / 1e12))
b = (res[8192] - res[4096]) / 4096
print("slope %.4f ms per 4096 k  (%.2f TFLOP/s asymptotic), fixed cost %.1f us" % (b * 4096, 2.0 * n * n * 4096 / (b * 4096 * 1e-3) / 1e12, (res[4096] - b * 4096) * 1e3))
