#!/usr/bin/env python
"""Developer aid: the fp32 4096^3 tile product in its four transpose forms (HipBackend.gemm, no re-use transposes), TFLOP/s.
    NPW_SGEMM_BK32=0|1 python tools/sgemm_time.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd import _ffi
from numpywren_amd.device import get_backend

be = get_backend()
n = 4096
A = be.convert(be.fill_random((n, n), 1), np.float32)
B = be.convert(be.fill_random((n, n), 2), np.float32)
C = be.empty((n, n), np.float32)
be.synchronize()
for ta, tb in ((b"N", b"T"), (b"N", b"N"), (b"T", b"N"), (b"T", b"T")):
    ts = []
    for rep in range(3):
        be.synchronize()
        t0 = time.time()
        for _ in range(20):
            _ffi.check(be.lib.npw_sgemm(ta, tb, n, n, n, ctypes.c_float(1.0), A.ptr, n, B.ptr, n, ctypes.c_float(0.0), None, n, C.ptr, n, None,
                                        be.default_stream.handle), "sgemm")
        be.synchronize()
        ts.append((time.time() - t0) / 20)
    print("sgemm %s%s 4096^3: %.3f ms = %.1f TFLOP/s" % (ta.decode(), tb.decode(), 1e3 * min(ts), 2 * n ** 3 / min(ts) / 1e12))
