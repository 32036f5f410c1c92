"""Developer aid: element-wise deviation of kernels.qr_factor (4096^2 tile; stacked 8192 x 4096 pair) from the oracle's LAPACK
DGEQRT -- what the tolerances of tests/test_tile4096_gpu.py::test_qr_factor_4096_vs_dgeqrt are set from."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import npw_oracle as oracle
from numpywren_amd import kernels
B = 4096
for stack in (False, True):
    rng = np.random.default_rng(41 + stack)
    a = rng.standard_normal((B, B))
    args = (a, rng.standard_normal((B, B))) if stack else (a,)
    V, T, R = kernels.qr_factor(*args)
    Vr, Tr, Rr = oracle.qr_factor(*args)
    print("stack", stack, "R", np.abs(R - Rr).max() / np.abs(Rr).max(), "V", np.abs(V - Vr).max(), "T", np.abs(T - Tr).max() / np.abs(Tr).max(),
          "cond(A1)", np.linalg.cond(np.vstack(args)))
