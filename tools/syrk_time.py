import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
S = be.fill_random((n, n), 1); X = be.fill_random((n, n), 2); Y = be.fill_random((n, n), 3)
for rep in range(2):
    be.synchronize(); t0 = time.time()
    for i in range(10): D = be.syrk(S, X, Y, exact_zero=False)
    be.synchronize(); dt = (time.time() - t0) / 10
    print(os.environ.get("NPW_GEMM_EXP", "base"), "syrk ms %.4f TFLOP/s %.2f" % (dt * 1e3, 2 * n ** 3 / dt / 1e12))
