#!/usr/bin/env python
"""Developer aid: npw_dgeqrt / npw_dgeqrt_batched timing loop (for rocprofv3 --kernel-trace).
    python tools/qr_run.py [count] [m] [n]            ($QR_RUN_NO_T=1: the R-only form; $QR_RUN_REPS: calls, default 3)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd.device import get_backend

be = get_backend()
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
A = [be.fill_random((m, n), i + 1) for i in range(cnt)]
WANT_T = os.environ.get("QR_RUN_NO_T", "0") != "1"
for rep in range(int(os.environ.get("QR_RUN_REPS", "3"))):
    be.synchronize()
    t0 = time.time()
    out = be.geqrt_batched(A, want_t=WANT_T) if cnt > 1 else [be.geqrt(A[0], want_t=WANT_T)]
    be.synchronize()
    dt = time.time() - t0
    flops = cnt * (2.0 * m * n * n - 2.0 * n ** 3 / 3)
    print(f"geqrt x{cnt} {m}x{n}: {1e3 * dt:.2f} ms = {1e3 * dt / cnt:.2f} ms each, {flops / dt / 1e12:.2f} TFLOP/s")
