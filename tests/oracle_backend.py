"""A CHECKER backend for the CPU test-suite: the HipBackend interface implemented with NumPy and
the oracle kernels, so that the host logic above the C-ABI (BigMatrix tile paths, the compiler,
LambdaPackProgram, the stream executor, the multi-GPU exchange schedule under gloo) can be
exercised without a GPU.  It lives under tests/ on purpose: the product never falls back to it
(numpywren_amd.device.get_backend raises HipExtensionError without a HIP device).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import npw_oracle as oracle  # noqa: E402

from numpywren_amd.device import DeviceTile, SpilledTile, Stream  # noqa: E402


class _Buf(object):
    def __init__(self):
        self.ptr = 0
        self.nbytes = 0
        self.streams = set()
        self.aux = None       # like DeviceBuffer.aux: {"host_copies": {offset: SpilledTile}} once the tile has a host copy


class HostTile(DeviceTile):
    """DeviceTile look-alike holding an ndarray."""
    __slots__ = ("array",)

    def __init__(self, array):
        array = np.array(array)
        super().__init__(_Buf(), array.shape, array.dtype)
        self.array = array

    def reshaped(self, shape):
        t = HostTile(self.array.reshape(shape))
        t.shared = self.shared
        return t


class _Bytes(object):
    def __init__(self, array):
        self.array = array


class _Flag(object):
    def __init__(self, v):
        self.value = int(v)
        self.streams = set()


class OracleBackend(object):
    def __init__(self, num_streams=4):
        self.streams = [Stream(i + 1, name=f"s{i}") for i in range(num_streams)]
        self.default_stream = self.streams[0]
        self.priority_stream = Stream(99, True, "prio")
        self.device = 0
        self.calls = []
        self.oom_handlers = []

    # plumbing -------------------------------------------------------------------------------
    def bind_thread(self):
        pass

    def synchronize(self):
        pass

    def stream_sync(self, stream=None):
        pass

    def wait_tile(self, tile):
        pass

    # events: the checker runs synchronously, so ordering calls are only recorded
    def record_new(self, stream=None):
        self.calls.append(("record", stream))
        return ("event", stream)

    def wait_event(self, stream, ev):
        self.calls.append(("wait_event", stream, ev))

    def event_sync(self, ev):
        pass

    def recycle_event(self, ev):
        pass

    def flag_stream(self):
        return self.default_stream

    def to_device(self, array, stream=None, dtype=None):
        return HostTile(np.ascontiguousarray(array, dtype=dtype))

    def to_host(self, tile, stream=None, out=None):
        return np.array(tile.array)

    def zeros(self, shape, dtype=np.float64, stream=None):
        return HostTile(np.zeros(shape, dtype=dtype))

    def shared_zeros(self, shape, dtype=np.float64):
        t = HostTile(np.zeros(shape, dtype=dtype))
        t.shared = True
        return t

    def copy(self, tile, stream=None):
        return HostTile(tile.array)

    # host-DRAM tier: "pinned memory" is an ndarray here, the copies are synchronous
    def spill_to_host(self, tile):
        # (like HipBackend.spill_to_host: a tile that still has its host copy -- written through when it was stored, or kept
        #  from an earlier restore -- leaves without a copy)
        kept = (tile.buf.aux.get("host_copies") or {}).get(tile.offset) if isinstance(tile.buf.aux, dict) else None
        if kept is not None and kept.nbytes == tile.nbytes and kept.dtype == tile.dtype:
            self.calls.append(("spill_free", tile.shape))
            return kept if kept.shape == tile.shape else SpilledTile(kept.buf, tile.shape, tile.dtype, kept.ready)
        self.calls.append(("spill", tile.shape))
        return SpilledTile(_Bytes(np.array(tile.array)), tile.shape, tile.dtype)

    def restore_from_host(self, spilled):
        self.calls.append(("restore", spilled.shape))
        t = HostTile(spilled.buf.array.reshape(spilled.shape))
        t.buf.aux = {"host_copies": {0: spilled}}
        return t

    def spilled_to_numpy(self, spilled):
        return np.array(spilled.buf.array).reshape(spilled.shape)

    def read_flag(self, flag, stream=None):
        return flag.value

    def zero_flag(self, tile, stream=None, atol=1e-8):
        return _Flag(np.allclose(tile.array, 0))

    def as_f64(self, tile, stream=None):
        return tile if tile.dtype == np.float64 else HostTile(tile.array.astype(np.float64))

    def convert(self, tile, dtype, stream=None):
        return HostTile(tile.array.astype(dtype))

    # kernels -----------------------------------------------------------------------------------
    def gemm(self, A, B, transpose_A=False, transpose_B=False, stream=None, alpha=1.0, beta=0.0, C=None, out=None,
             skip=None):
        self.calls.append(("gemm", stream))
        r = alpha * oracle.gemm(A.array, B.array, transpose_A=transpose_A, transpose_B=transpose_B)
        if C is not None and beta != 0:
            r = r + beta * C.array
        if out is not None:
            out.array[...] = r.astype(out.array.dtype)
            return out
        return HostTile(r)

    def gemm_batched(self, problems, transpose_A=False, transpose_B=False, stream=None):
        self.calls.append(("gemm_batched", len(problems)))
        return [self.gemm(A, B, transpose_A, transpose_B, stream) for A, B in problems]

    def syrk(self, S, X, Y, stream=None, inplace=False, exact_zero=True):
        self.calls.append(("syrk", stream))
        if exact_zero:
            return HostTile(oracle.syrk(S.array, X.array, Y.array))
        return HostTile(S.array - X.array @ Y.array.T)

    def syrk_batched(self, problems, stream=None, exact_zero=True):
        self.calls.append(("syrk_batched", len(problems)))
        return [self.syrk(S, X, Y, stream, exact_zero=exact_zero) for S, X, Y in problems]

    def trsm_batched(self, L, Ys, stream=None, exact_zero=True):
        self.calls.append(("trsm_batched", len(Ys)))
        return [self.trsm(L, y, stream, exact_zero) for y in Ys]

    def trsm(self, L, Y, stream=None, exact_zero=True):
        self.calls.append(("trsm", stream))
        if exact_zero and np.allclose(Y.array, 0):
            return HostTile(np.zeros(Y.array.shape))
        import scipy.linalg
        return HostTile(scipy.linalg.blas.dtrsm(1.0, L.array.T, Y.array, lower=False, side=1))

    def chol(self, A, stream=None, info_out=None):
        self.calls.append(("chol", stream))
        try:
            return HostTile(oracle.chol(A.array)), _Flag(0)
        except np.linalg.LinAlgError:
            return HostTile(np.full(A.array.shape, np.nan)), _Flag(1)

    def add_n(self, tiles, stream=None):
        self.calls.append(("add_n", stream))
        return HostTile(oracle.add_matrices(*[t.array for t in tiles]))

    def add_diag(self, tile, lam, stream=None):
        a = np.array(tile.array, dtype=np.float64)
        a[np.diag_indices(min(a.shape))] += lam
        return HostTile(a)

    def transpose(self, tile, stream=None):
        return HostTile(tile.array.T)

    def vstack(self, tiles, stream=None):
        return HostTile(np.vstack([t.array for t in tiles]))

    def rows(self, tile, start, stop, stream=None):
        return HostTile(tile.array[start:stop])

    def block(self, tile, r0, r1, c0, c1, stream=None):
        return HostTile(np.ascontiguousarray(tile.array[r0:r1, c0:c1]))

    # want_t=False: T is None, as on the HIP backend (the executor only asks for it for tiles it drops unread)
    def geqrt(self, A, stream=None, want_t=True):
        self.calls.append(("geqrt", stream))
        v, t, r = oracle.fast_qr(A.array)
        rt = HostTile(r)
        rt.upper = r.shape[0] == r.shape[1]
        return HostTile(v), (HostTile(t) if want_t or A.array.shape[0] < A.array.shape[1] else None), rt

    def geqrt_batched(self, As, stream=None, want_t=True, want_v=True):
        self.calls.append(("geqrt_batched", len(As)))
        out = []
        for a in As:
            v, t, r = oracle.fast_qr(a.array)
            rt = HostTile(r)
            rt.upper = r.shape[0] == r.shape[1]
            out.append((HostTile(v) if want_v else None, HostTile(t) if want_t else None, rt))
        return out

    def tpqrt_batched(self, pairs, stream=None, want_t=True, want_v=True):
        self.calls.append(("tpqrt_batched", len(pairs)))
        out = []
        for a, c in pairs:
            assert not np.tril(a.array, -1).any() and not np.tril(c.array, -1).any()   # the caller's promise
            v, t, r = oracle.fast_qr(np.vstack([a.array, c.array]))
            rt = HostTile(r)
            rt.upper = True
            out.append((HostTile(v) if want_v else None, HostTile(t) if want_t else None, rt))
        return out

    def tri(self, tile, uplo, unit_diag=False, stream=None):
        a = np.triu(tile.array) if uplo.upper() == "U" else np.tril(tile.array)
        a = np.array(a, dtype=np.float64)
        if unit_diag:
            a[np.diag_indices(min(a.shape))] = 1.0
        return HostTile(a)

    def blockdiag_rows(self, T, nb, stream=None):
        t = T.array
        n = t.shape[0]
        out = np.zeros((n, n))
        for c0 in range(0, n, nb):
            c1 = min(n, c0 + nb)
            out[:c1 - c0, c0:c1] = t[c0:c1, c0:c1]
        return HostTile(out)

    def axpby(self, alpha, X, beta, Y, stream=None):
        return HostTile(alpha * X.array + beta * Y.array)

    def mul(self, X, Y, stream=None):
        return HostTile(np.asarray(X.array, dtype=np.float64) * np.asarray(Y.array, dtype=np.float64))

    def flip(self, tile, rows=True, cols=True, stream=None):
        a = tile.array
        if rows:
            a = a[::-1]
        if cols:
            a = a[:, ::-1]
        return HostTile(np.ascontiguousarray(a, dtype=np.float64))
