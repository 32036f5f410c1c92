"""The C-ABI boundary without a GPU: the in-tree libnpw_hip.so loads, exports every symbol that
include/npw_hip.h declares, and the product path refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from numpywren_amd import _ffi


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "npw_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(npw_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 45
    cdll = ctypes.CDLL(_ffi.library_path())
    missing = [n for n in names if not hasattr(cdll, n)]
    assert not missing, missing
    # and the ctypes prototype table covers exactly the header
    assert sorted(_ffi.PROTOTYPES) == names


def test_version_and_error_string():
    lib = _ffi.lib()
    assert lib.npw_version() >= 100
    assert lib.npw_last_error() is not None
    # argument validation happens before any HIP call, so it is testable on a CPU-only box
    rc = lib.npw_dgemm(b"X", b"N", 4, 4, 4, 1.0, None, 4, None, 4, 0.0, None, 4, None, 4, None, None)
    assert rc == _ffi.NPW_ERR_ARG and b"transA" in lib.npw_last_error()
    # inverse cache: one 1024 x 1024 group (+ a 512 x 512 scratch) per 1024 columns, opaque to the caller
    winv = 4 * (1024 * 1024 + 512 * 512) * 8
    assert lib.npw_dtrtri_diag_bytes(4096) == winv
    assert lib.npw_dtrtri_diag_bytes(0) == 0
    assert lib.npw_dtrsm_rltn_workspace_bytes(4096, 4096) == winv + 4096 * 4096 * 8
    assert lib.npw_dpotrf_lower_workspace_bytes(100) == winv // 4
    assert lib.npw_dpotrf_lower_workspace_bytes(4096) == winv
    assert lib.npw_dgeqrt_workspace_bytes(8192, 4096) > 4096 * 4096 * 8
    # compute units the panel chain of an n x n factorisation has to be resident on (one workgroup per 64 rows below
    # the 128-wide diagonal block, + 1): what a CU-masked stream must offer
    assert [lib.npw_dpotrf_lower_resident_cus(n) for n in (0, 100, 128, 129, 1024, 4096, 8192)] == [0, 1, 1, 2, 15, 63, 127]
    # profiler ranges (roctx, loaded on demand; without a profiler attached they cost a call): push returns the nesting depth
    d0 = lib.npw_range_push(b"outer")
    d1 = lib.npw_range_push(b"inner")
    assert d0 >= 0 and d1 >= d0 and lib.npw_range_pop() >= 0 and lib.npw_range_pop() >= 0


def test_no_cpu_fallback():
    """Without a HIP device the product must fail loudly, never compute on the host."""
    from numpywren_amd import device, kernels
    from numpywren_amd.exceptions import HipExtensionError
    import numpy as np
    if device.hip_available():
        pytest.skip("a GPU is present")
    with pytest.raises(HipExtensionError):
        device.get_backend()
    with pytest.raises(HipExtensionError):
        kernels.gemm(np.eye(4), np.eye(4))
    with pytest.raises(HipExtensionError):
        kernels.chol(np.eye(4))
