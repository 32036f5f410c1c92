import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("NUMPYWREN_AMD_STREAMS", "4")
import numpy as np
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
G = be.fill_random((n, 256), seed=5)
A = be.add_diag(be.gemm(G, G, False, True), float(n))
S = be.fill_random((n, n), 1); X = be.fill_random((n, n), 2)
L0, info = be.chol(A); ref = be.to_host(L0)
bad = 0
t0 = time.time()
for rep in range(10):
    outs = [be.chol(A, be.streams[i % 4]) for i in range(8)]
    # plus chip-filling work on the other streams at the same time
    junk = [be.syrk(S, X, X, be.streams[(i + 1) % 4], exact_zero=False) for i in range(4)]
    for L, info in outs:
        code = be.read_flag(info)
        if code != 0 or not np.array_equal(be.to_host(L), ref):
            bad += 1; print("BAD rep", rep, "info", code)
print("stress done %.1f s bad %d" % (time.time() - t0, bad))
