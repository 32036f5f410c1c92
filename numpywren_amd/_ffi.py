"""ctypes binding of libnpw_hip.so (the C-ABI declared in include/npw_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C numpywren_amd/csrc` into
numpywren_amd/lib/libnpw_hip.so.  There is NO fallback: if the shared object is missing or
cannot be loaded, `lib()` raises `HipExtensionError` -- the product path never computes on
the CPU.
"""
import ctypes
import os
import threading
from ctypes import POINTER, c_char, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

from .exceptions import HipExtensionError, NpwHipError

# ($NUMPYWREN_AMD_LIB: another build of the same library, for A/B timing runs of the developer tools)
_LIB_PATH = os.environ.get("NUMPYWREN_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnpw_hip.so")
_lib = None
_lock = threading.Lock()

NPW_OK = 0
NPW_ERR_HIP = -1
NPW_ERR_ARG = -2
NPW_ERR_NOT_PD = -3
NPW_ERR_UNSUPPORTED = -4

# name -> (restype, argtypes).  Every symbol include/npw_hip.h declares appears here; the CPU
# test-suite checks that the built library exports each of them.
_vp, _i64, _sz = c_void_p, c_int64, c_size_t
PROTOTYPES = {
    "npw_version": (c_int, []),
    "npw_range_push": (c_int, [c_char_p]),
    "npw_range_pop": (c_int, []),
    "npw_last_error": (c_char_p, []),
    "npw_device_count": (c_int, [POINTER(c_int)]),
    "npw_set_device": (c_int, [c_int]),
    "npw_get_device": (c_int, [POINTER(c_int)]),
    "npw_device_info": (c_int, [c_int, c_char_p, _sz, POINTER(_sz), POINTER(c_int), POINTER(c_int)]),
    "npw_mem_info": (c_int, [POINTER(_sz), POINTER(_sz)]),
    "npw_device_pci_bus_id": (c_int, [c_int, c_char_p, _sz]),
    "npw_malloc": (c_int, [POINTER(_vp), _sz]),
    "npw_free": (c_int, [_vp]),
    "npw_host_alloc": (c_int, [POINTER(_vp), _sz]),
    "npw_host_free": (c_int, [_vp]),
    "npw_memcpy_h2d_async": (c_int, [_vp, _vp, _sz, _vp]),
    "npw_memcpy_d2h_async": (c_int, [_vp, _vp, _sz, _vp]),
    "npw_memcpy_d2d_async": (c_int, [_vp, _vp, _sz, _vp]),
    "npw_memset_async": (c_int, [_vp, c_int, _sz, _vp]),
    "npw_memcpy2d_h2d_async": (c_int, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "npw_memcpy2d_d2h_async": (c_int, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "npw_memcpy2d_d2d_async": (c_int, [_vp, _sz, _vp, _sz, _sz, _sz, _vp]),
    "npw_stream_create": (c_int, [POINTER(_vp), c_int]),
    "npw_stream_create_masked": (c_int, [POINTER(_vp), POINTER(ctypes.c_uint32), c_int]),
    "npw_stream_destroy": (c_int, [_vp]),
    "npw_stream_cu_count": (c_int, [_vp, POINTER(c_int), POINTER(c_int)]),
    "npw_stream_synchronize": (c_int, [_vp]),
    "npw_stream_query": (c_int, [_vp, POINTER(c_int)]),
    "npw_device_synchronize": (c_int, []),
    "npw_event_create": (c_int, [POINTER(_vp), c_int]),
    "npw_event_destroy": (c_int, [_vp]),
    "npw_event_record": (c_int, [_vp, _vp]),
    "npw_event_synchronize": (c_int, [_vp]),
    "npw_event_query": (c_int, [_vp, POINTER(c_int)]),
    "npw_stream_wait_event": (c_int, [_vp, _vp]),
    "npw_event_elapsed_ms": (c_int, [_vp, _vp, POINTER(c_float)]),
    "npw_dgemm": (c_int, [c_char, c_char, _i64, _i64, _i64, c_double, _vp, _i64, _vp, _i64, c_double, _vp, _i64, _vp,
                          _i64, _vp, _vp]),
    "npw_sgemm": (c_int, [c_char, c_char, _i64, _i64, _i64, c_float, _vp, _i64, _vp, _i64, c_float, _vp, _i64, _vp,
                          _i64, _vp, _vp]),
    "npw_dgemm_nt_sub_workspace_bytes": (c_size_t, [_i64, _i64, _i64]),
    "npw_dgemm_nt_sub": (c_int, [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "npw_dgemm_batched": (c_int, [c_int, ctypes.c_char, ctypes.c_char, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npw_sgemm_batched": (c_int, [c_int, ctypes.c_char, ctypes.c_char, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npw_dgemm_nt_sub_batched_workspace_bytes": (c_size_t, [c_int, _i64, _i64, _i64]),
    "npw_dgemm_nt_sub_batched": (c_int, [c_int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "npw_dtrsm_rltn_workspace_bytes": (_sz, [_i64, _i64]),
    "npw_dtrsm_rltn": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "npw_dtrtri_diag_bytes": (_sz, [_i64]),
    "npw_dtrtri_diag": (c_int, [_i64, _vp, _i64, _vp, _vp]),
    "npw_dtrsm_rltn_inv_workspace_bytes": (_sz, [_i64, _i64]),
    "npw_dtrsm_rltn_inv": (c_int, [_i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "npw_dtrsm_rltn_inv_batched": (c_int, [c_int, _i64, _i64, _vp, _i64, _vp, POINTER(_vp), _i64, POINTER(_vp), _i64, _vp, _vp, _vp]),
    "npw_dpotrf_lower_workspace_bytes": (_sz, [_i64]),
    "npw_dpotrf_lower": (c_int, [_i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "npw_dpotrf_lower_resident_cus": (c_int, [_i64]),
    "npw_dpotrf_lower_blocks": (c_int, [_i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "npw_dtrtri_complete": (c_int, [_i64, _vp, _i64, _vp, _vp]),
    "npw_dgeqrt_workspace_bytes": (_sz, [_i64, _i64]),
    "npw_dgeqrt_handoff_timeouts": (c_int, [POINTER(c_int), c_int]),
    "npw_dgeqrt": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "npw_dtpqrt_batched_workspace_bytes": (_sz, [c_int, _i64]),
    "npw_dtpqrt_batched": (c_int, [c_int, _i64, POINTER(_vp), POINTER(_vp), _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64,
                                   _i64, _vp, _vp]),
    "npw_dgeqrt_batched_workspace_bytes": (_sz, [c_int, _i64, _i64]),
    "npw_dgeqrt_batched": (c_int, [c_int, _i64, _i64, POINTER(_vp), _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                   _vp, _vp]),
    "npw_add_n": (c_int, [c_int, POINTER(_vp), POINTER(_i64), POINTER(c_int32), _i64, _i64, _vp, _i64, _vp]),
    "npw_add_diag": (c_int, [_vp, _i64, _i64, _i64, c_double, _vp]),
    "npw_is_zero": (c_int, [_vp, _i64, _i64, _i64, c_double, _vp, _vp]),
    "npw_is_zero_batched": (c_int, [c_int, _vp, _i64, _i64, _i64, c_double, _vp, _vp]),
    "npw_zero_if": (c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "npw_daxpby": (c_int, [_i64, _i64, c_double, _vp, _i64, c_double, _vp, _i64, _vp, _i64, _vp]),
    "npw_dmul": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npw_dflip": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, c_int, c_int, _vp]),
    "npw_dtranspose": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npw_stranspose": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npw_dtri_keep": (c_int, [c_char, c_int, _i64, _i64, _vp, _i64, _vp]),
    "npw_dblockdiag_rows": (c_int, [_i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "npw_convert": (c_int, [_i64, _i64, _vp, _i64, c_int, _vp, _i64, c_int, _vp]),
    "npw_fill_outer": (c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, c_double, _vp]),
    "npw_fill_random": (c_int, [_vp, _i64, _i64, _i64, c_uint64, _i64, _i64, _vp]),
    "npw_dsumsq": (c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "npw_dgebd2_workspace_bytes": (_sz, [_i64]),
    "npw_dgebd2": (c_int, [_i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "npw_comm_unique_id": (c_int, [_vp, _sz]),
    "npw_comm_init": (c_int, [POINTER(_vp), c_int, c_int, _vp]),
    "npw_comm_destroy": (c_int, [_vp]),
    "npw_comm_abort": (c_int, [_vp]),
    "npw_comm_info": (c_int, [_vp, POINTER(c_int), POINTER(c_int), POINTER(_vp)]),
    "npw_comm_group_start": (c_int, [_vp]),
    "npw_comm_group_end": (c_int, [_vp]),
    "npw_send_tile": (c_int, [_vp, _vp, _sz, c_int, _vp]),
    "npw_recv_tile": (c_int, [_vp, _vp, _sz, c_int, _vp]),
    "npw_bcast_tile": (c_int, [_vp, _vp, _sz, c_int, POINTER(c_int), c_int, _vp]),
}
NPW_COMM_ID_BYTES = 128


def library_path():
    return _LIB_PATH


def lib():
    """The loaded CDLL with prototypes applied; raises HipExtensionError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise HipExtensionError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C numpywren_amd/csrc` (hipcc, --offload-arch=gfx950). numpywren_amd has no CPU fallback.")
        try:
            cdll = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # e.g. libamdhip64 missing
            raise HipExtensionError(f"cannot load {_LIB_PATH}: {e}") from e
        missing = []
        for name, (restype, argtypes) in PROTOTYPES.items():
            try:
                fn = getattr(cdll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = restype
            fn.argtypes = argtypes
        if missing:
            raise HipExtensionError(f"{_LIB_PATH} does not export: {', '.join(missing)}")
        _lib = cdll
    return _lib


def check(rc, what=""):
    """Raise NpwHipError carrying npw_last_error() if rc != 0."""
    if rc != 0:
        msg = lib().npw_last_error()
        msg = msg.decode("utf-8", "replace") if msg else ""
        raise NpwHipError(rc, f"{what}: {msg}" if what else msg)
    return rc


def char(c):
    return c.encode("ascii") if isinstance(c, str) else c
