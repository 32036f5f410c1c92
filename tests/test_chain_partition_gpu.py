"""The executor's chain partition (GPU): a `chol` task that is ready together with independent trailing updates runs on
a stream masked to `chain_cus` compute units while those updates run on the other CUs (job_runner.run_chain).

Scheduling must not change results: the same kernels see the same operands, so the factor is BITWISE the one of the
in-order run (reference semantics: any order of ready tasks is a valid execution, lambdapack.py:560-640), and the factor
matches np.linalg.cholesky like the reference's own test (tests/test_alg_correctness.py:31-50)."""
import ctypes

import numpy as np
import pytest

from numpywren_amd import _ffi, alg_wrappers, job_runner
from numpywren_amd import lambdapack as lp
from numpywren_amd.device import Stream, get_backend
from numpywren_amd.exceptions import NpwHipError
from numpywren_amd.matrix import BigMatrix
from numpywren_amd.matrix_init import shard_matrix

pytestmark = pytest.mark.gpu


def _factor(A, b, chain_cus, key, timers=False):
    be = get_backend()
    X = BigMatrix(key, shape=A.shape, shard_sizes=(b, b), write_header=True)
    X.free()
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["chain_cus"] = chain_cus
    if timers:
        be.enable_kernel_timers(("chol", "syrk", "syrk_sym"))
    program.start()
    res = job_runner.lambdapack_run(program, timeout=300)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    times = be.collect_kernel_times() if timers else {}
    L = meta["outputs"][0].numpy()
    program.free()
    X.free()
    return L, res, times


@pytest.mark.parametrize("n,b", [(4096, 1024), (2560, 512)])
def test_chain_partition_is_bitwise_the_in_order_run(n, b, hbm_store):
    rng = np.random.default_rng(n + b)
    G = rng.standard_normal((n, 96))
    A = G @ G.T + n * np.eye(n)
    nb = n // b
    L0, res0, _ = _factor(A, b, 0, f"chainpart_off_{n}")
    L1, res1, times = _factor(A, b, 64, f"chainpart_on_{n}", timers=True)
    assert len(res0["executed_messages"]) == len(res1["executed_messages"]) == nb * (nb + 1) * (nb + 2) // 6
    # every chol but the first (nothing else is ready) and the last (nothing is left) had a trailing update beside it
    assert len(times.get("chol@chain", [])) == nb - 2, {k: len(v) for k, v in times.items()}
    assert len(times.get("syrk@rest", [])) + len(times.get("syrk_sym@rest", [])) >= nb - 2
    assert len(times.get("chol", [])) == 2
    assert np.array_equal(L0, L1)
    np.testing.assert_allclose(L1, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10 * np.sqrt(n))


def test_chain_partition_off_when_the_chain_does_not_fit(hbm_store):
    """A 4096-row tile needs 63 resident workgroups: with a 32-CU chain partition the executor keeps chol on the full chip."""
    be = get_backend()
    assert be.chol_resident_cus(4096) == 63
    n, b = 2048, 1024            # 15 workgroups: fits 16, not 8
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, 64))
    A = G @ G.T + n * np.eye(n)
    for cus, windows in ((8, 0), (16, 0)):   # a 2 x 2 grid has no chol with a ready companion either way
        L, _, times = _factor(A, b, cus, f"chainpart_small_{cus}", timers=True)
        assert len(times.get("chol@chain", [])) == windows
        np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-10, atol=1e-9)


def test_chol_on_a_too_small_partition_fails_loudly():
    """npw_dpotrf_lower on a masked stream with fewer CUs than its panel chain needs: an error, never a hang."""
    be = get_backend()
    words = (be.compute_units + 31) // 32
    mask = (ctypes.c_uint32 * words)(*([0xFFFF] + [0] * (words - 1)))   # 16 CUs
    h = ctypes.c_void_p(0)
    _ffi.check(be.lib.npw_stream_create_masked(ctypes.byref(h), mask, words), "masked")
    small = Stream(h.value, False, "small")
    rng = np.random.default_rng(1)
    G = rng.standard_normal((2048, 32))
    A = be.to_device(G @ G.T + 2048 * np.eye(2048))
    with pytest.raises(NpwHipError, match="resident workgroups"):
        be.chol(A, stream=small)
    # 1024 rows need 15: fine on the same stream, and equal to the full-chip factor bit for bit
    A1 = be.to_device((G @ G.T)[:1024, :1024] + 1024 * np.eye(1024))
    L_small, info = be.chol(A1, stream=small)
    be.stream_sync(small)
    L_full, _ = be.chol(A1)
    assert be.read_flag(info) == 0
    assert np.array_equal(be.to_host(L_small), be.to_host(L_full))


def test_destroyed_stream_takes_its_helper_streams_along():
    """The library keeps helper streams per caller's stream (QR: three of them, created with the caller's CU mask).  Destroying
    the stream retires them and the backend's per-stream state (scratch buffer, pending buffer releases): streams are made,
    used for a QR and destroyed in a loop -- the driver hands the same handle out again -- and every QR has to give the
    default stream's factors.  (Plain streams only: create / destroy cycles of CU-MASKED streams hang inside the HIP runtime
    of this ROCm release about every tenth time, with or without this library's helpers -- tools/README.md; the executor
    makes its masked streams once and keeps them.)"""
    be = get_backend()
    rng = np.random.default_rng(3)
    A = be.to_device(rng.standard_normal((640, 512)))
    V0, T0, R0 = (be.to_host(x) for x in be.geqrt(A))
    handles = set()
    for rep in range(6):
        st = be.create_stream(name="short-lived")
        handles.add(st.handle)
        V, T, R = be.geqrt(A, stream=st)
        be.stream_sync(st)
        for got, want in ((R, R0), (V, V0), (T, T0)):
            assert np.array_equal(be.to_host(got), want)
        be.destroy_stream(st)
        assert st.handle is None
    del V, T, R     # (released after their stream is gone: no event is recorded on a dead handle)
    be.synchronize()


def test_masked_streams_are_parked_not_destroyed():
    """create / destroy cycles of CU-MASKED streams through the C-ABI (they hang inside the HIP runtime of this ROCm release
    about every tenth cycle): npw_stream_destroy parks a masked stream with its helpers, npw_stream_create_masked hands a parked
    stream of the same mask out again -- 40 cycles with a QR on the stream each time, two distinct masks, at most one HIP
    stream per mask ever made, every QR equal to the default stream's."""
    be = get_backend()
    rng = np.random.default_rng(4)
    A = be.to_device(rng.standard_normal((640, 512)))
    V0, T0, R0 = (be.to_host(x) for x in be.geqrt(A))
    words = (be.compute_units + 31) // 32
    masks = [[0xFFFFFFFF] * words, [0xFFFF0000] + [0xFFFFFFFF] * (words - 1)]
    masks[0][0] = 0x0000FFFF
    seen = {0: set(), 1: set()}
    for rep in range(40):
        which = rep % 2
        st = be.stream_from_mask(masks[which], name="cycled")
        seen[which].add(st.handle)
        assert be.stream_cus(st)[0] == be.compute_units - 16
        V, T, R = be.geqrt(A, stream=st)
        be.stream_sync(st)
        assert np.array_equal(be.to_host(R), R0) and np.array_equal(be.to_host(T), T0) and np.array_equal(be.to_host(V), V0)
        be.destroy_stream(st)
    del V, T, R
    be.synchronize()
    assert len(seen[0]) == 1 and len(seen[1]) == 1 and seen[0] != seen[1]
