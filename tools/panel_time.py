"""Developer aid: one panel (m x 32) factorisations in several batch shapes -- the panel kernel alone (run under rocprofv3
--kernel-trace for its launch times; profiles/r03_qr_tsqr.md cites it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend
be = get_backend()
for cnt, m, n in [(256, 256, 32), (512, 256, 32), (32, 512, 32), (32, 4096, 32), (1, 4096, 32), (1, 256, 32), (1,512,32)]:
    A = [be.fill_random((m, n), i + 1) for i in range(cnt)]
    for rep in range(3):
        be.synchronize(); t0 = time.time()
        out = be.geqrt_batched(A) if cnt > 1 else [be.geqrt(A[0])]
        be.synchronize(); dt = time.time() - t0
    print(f"geqrt x{cnt} {m}x{n}: {1e6*dt:.0f} us")
