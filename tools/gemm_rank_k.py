"""Developer aid: the N / T 128 x 128 tiling on rank-k updates C (M x N) -= A (M x k) B (N x k)^T for short k -- what a tile costs
beyond its k loop (prologue, epilogue, workgroup turnover).  TFLOP/s by k with and without the C operand."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
M, N = int(os.environ.get("M", 32768)), int(os.environ.get("N", 4096))
C = be.fill_random((M, N), 1)
for K in (64, 128, 256, 512, 1024, 4096):
    A = be.fill_random((M, K), 2); B = be.fill_random((N, K), 3)
    for name, kw in (("beta=1", dict(alpha=-1.0, beta=1.0, C=C, out=C)), ("beta=0", dict(alpha=-1.0))):
        best = 1e9
        for rep in range(3):
            be.synchronize(); t0 = time.time()
            for i in range(5):
                D = be.gemm(A, B, False, True, **kw)
            be.synchronize(); best = min(best, (time.time() - t0) / 5)
        tiles = (M // 128) * (N // 128)
        print("K %5d %s  %.3f ms  %.2f TFLOP/s   %.2f us per tile-slot (512 slots)" % (K, name, best * 1e3, 2.0 * M * N * K / best / 1e12, best * 1e6 / (tiles / 512)))
    del A, B
