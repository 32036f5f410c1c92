import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
G = be.fill_random((n, 256), seed=5)
A = be.add_diag(be.gemm(G, G, False, True), float(n))
L, info = be.chol(A)
Y = be.fill_random((n, n), seed=6)
for rep in range(4):
    be.synchronize(); t0 = time.time()
    X = be.trsm(L, Y)
    be.synchronize(); print("trsm ms", 1e3 * (time.time() - t0))
# the batched form (the right-hand sides of one block column: 3 at step 0 of configs[1])
Ys = [be.fill_random((n, n), seed=20 + i) for i in range(3)]
for rep in range(4):
    be.synchronize(); t0 = time.time()
    Xs = be.trsm_batched(L, Ys, exact_zero=False)
    be.synchronize(); print("trsm_batched x3 ms", 1e3 * (time.time() - t0))
