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
