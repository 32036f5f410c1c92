"""The RCCL transport of the C-ABI (npw_comm_*) on the one GPU of the test box.

A single device cannot hold two ranks of one communicator, so what runs here is everything that does not need a
second GPU: loading librccl on demand, rendezvous id + communicator creation, the transport stream, a grouped
self send/receive of a real tile through ncclSend / ncclRecv (bare and through dist.py's grouped transport), and the
distributed executor over the RCCL transport (`dist.init_process_group()` -> RcclTransport) for a world of one.  The
exchange logic for world > 1 is covered with the host transport in tests/test_dist_gloo.py (CPU, world 2 / 4) and
tests/test_dist_gpu.py (HIP kernels, 2 / 4 ranks sharing this GPU); the driver's 8-GPU run is the first place both
halves meet."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_world_of_one_self_exchange():
    from numpywren_amd import _ffi
    from numpywren_amd.device import Stream, get_backend
    be = get_backend()
    lib = be.lib
    ident = ctypes.create_string_buffer(_ffi.NPW_COMM_ID_BYTES)
    _ffi.check(lib.npw_comm_unique_id(ident, _ffi.NPW_COMM_ID_BYTES), "unique_id")
    assert any(ident.raw)
    h = ctypes.c_void_p(0)
    _ffi.check(lib.npw_comm_init(ctypes.byref(h), 0, 1, ident), "comm_init")
    rank, world, sh = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_void_p(0)
    _ffi.check(lib.npw_comm_info(h, ctypes.byref(rank), ctypes.byref(world), ctypes.byref(sh)))
    assert (rank.value, world.value) == (0, 1) and sh.value
    cs = Stream(sh.value, True, "xgmi")
    rng = np.random.default_rng(3)
    a = rng.standard_normal((512, 384))
    src = be.to_device(a)
    dst = be.zeros(a.shape)
    be._use(cs, src, dst)
    # a tile sent to ourselves: the send and the matching receive in one group (one launch), on the transport stream,
    # ordered behind the H2D copy by the tile's event
    _ffi.check(lib.npw_comm_group_start(h))
    _ffi.check(lib.npw_send_tile(h, src.ptr, src.nbytes, 0, cs.handle), "send")
    _ffi.check(lib.npw_recv_tile(h, dst.ptr, dst.nbytes, 0, cs.handle), "recv")
    _ffi.check(lib.npw_comm_group_end(h))
    be._produced(cs, dst)
    assert np.array_equal(be.to_host(dst), a)
    # the same through dist.py's transport object: a group opened with begin_group() carries a send and its receive,
    # the received tile's event is recorded behind the launch at end_group()
    from numpywren_amd.dist import RcclTransport, TileMeta
    tr = RcclTransport.__new__(RcclTransport)
    tr.be, tr.lib, tr.handle, tr.stream, tr.rank, tr.world, tr._group, tr._held = be, lib, h.value, cs, 0, 1, None, []
    tr.begin_group()
    tr.send(src, [0])
    got = tr.recv(0, TileMeta(a.shape, a.dtype))
    tr.end_group()
    assert tr._group is None and tr._held == [] and np.array_equal(be.to_host(got), a)
    # the physical device's PCI bus id (what the ranks compare to decide between RCCL and host staging)
    bus = ctypes.create_string_buffer(64)
    _ffi.check(lib.npw_device_pci_bus_id(be.device, bus, 64), "pci_bus_id")
    assert bus.value.count(b":") == 2
    members = (ctypes.c_int * 1)(0)
    _ffi.check(lib.npw_bcast_tile(h, src.ptr, src.nbytes, 0, members, 1, cs.handle), "bcast")
    # argument checking
    assert lib.npw_send_tile(h, src.ptr, 8, 5, cs.handle) == _ffi.NPW_ERR_ARG and b"destination" in lib.npw_last_error()
    assert lib.npw_comm_init(ctypes.byref(ctypes.c_void_p(0)), 2, 2, ident) == _ffi.NPW_ERR_ARG
    be.stream_sync(cs)
    _ffi.check(lib.npw_comm_destroy(h))


def test_distributed_executor_over_rccl_world_of_one():
    """torchrun with one rank: init_process_group() picks the RCCL transport (one device per rank), the distributed
    executor runs Cholesky / GEMM / TSQR through it."""
    env = dict(os.environ, DIST_CHECK_N="1024", DIST_CHECK_B="256", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NUMPYWREN_AMD_STORE", None)
    env.pop("NUMPYWREN_AMD_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29651", os.path.join(ROOT, "tools", "dist_check.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "dist_check: PASSED" in text and "backend rccl" in text, text[-3000:]


def test_resident_grid_kernels_beside_rccl_transfers():
    """the kernels whose workgroups wait for one another (a batch of 32 Householder factorisations of
    4096^2 tiles, the Cholesky panel chain of a 4096^2 tile) running WHILE grouped RCCL transfers of 8 tiles at a time are
    in flight on the transport stream: same bits as without the transfers, no expired hand-off, and the library has set
    compute units aside for the transfer kernels since the communicator exists."""
    from numpywren_amd import _ffi
    from numpywren_amd.device import Stream, get_backend
    be = get_backend()
    lib = be.lib
    b, count = 4096, 32
    As = [be.fill_random((b, b), 21, z * b, 0) for z in range(count)]
    X = be.fill_random((b, 128), 22)
    S = be.add_diag(be.gemm(X, X, False, True), float(b))

    def factor():
        out = be.geqrt_batched(As, want_t=False, want_v=False)
        L, _ = be.chol(S)
        return [r for _, _, r in out], L

    before = be.stream_cus()
    be.qr_handoff_timeouts(reset=True)
    R0, L0 = factor()
    be.synchronize()
    ident = ctypes.create_string_buffer(_ffi.NPW_COMM_ID_BYTES)
    _ffi.check(lib.npw_comm_unique_id(ident, _ffi.NPW_COMM_ID_BYTES), "unique_id")
    h = ctypes.c_void_p(0)
    _ffi.check(lib.npw_comm_init(ctypes.byref(h), 0, 1, ident), "comm_init")
    try:
        sh = ctypes.c_void_p(0)
        _ffi.check(lib.npw_comm_info(h, None, None, ctypes.byref(sh)))
        cs = Stream(sh.value, True, "xgmi")
        cus, resident = be.stream_cus()
        assert before == (cus, cus) and 0 < resident < cus          # a live communicator: CUs are left to its kernels
        src = [be.fill_random((b, b), 23, z * b, 0) for z in range(8)]
        dst = [be.empty((b, b)) for _ in range(8)]
        be.synchronize()
        be._use(cs, *src)
        be._use(cs, *dst)
        rounds = 40                                                  # 40 x 8 x 128 MiB: transfers in flight for the whole batch
        def post(n):
            for _ in range(n):
                _ffi.check(lib.npw_comm_group_start(h))
                for s_, d_ in zip(src, dst):
                    _ffi.check(lib.npw_send_tile(h, s_.ptr, s_.nbytes, 0, cs.handle), "send")
                    _ffi.check(lib.npw_recv_tile(h, d_.ptr, d_.nbytes, 0, cs.handle), "recv")
                _ffi.check(lib.npw_comm_group_end(h))
        post(rounds // 2)
        R1, L1 = factor()
        post(rounds - rounds // 2)
        be._produced(cs, *dst)
        be.synchronize()
        assert be.qr_handoff_timeouts(reset=True) == 0
        for r0, r1 in zip(R0, R1):
            assert be.sumsq(be.axpby(1.0, r0, -1.0, r1)) == 0.0     # bit for bit
        assert be.sumsq(be.axpby(1.0, L0, -1.0, L1)) == 0.0
        for s_, d_ in zip(src, dst):
            assert be.sumsq(be.axpby(1.0, s_, -1.0, d_)) == 0.0
    finally:
        be.synchronize()
        _ffi.check(lib.npw_comm_destroy(h))
    assert be.stream_cus() == before                                # the reserve goes with the communicator


def test_one_gpu_executor_beside_a_live_communicator():
    """A plain one-GPU run (job_runner.lambdapack_run with its chain partition) in a process that holds an RCCL communicator:
    the chain stream's 64 CUs no longer guarantee the panel chain its residency (CUs are left to transfer kernels), so the
    executor keeps the factorisations on the full stream instead of letting npw_dpotrf_lower refuse the partition -- same
    factor bit for bit as without a communicator."""
    from numpywren_amd import _ffi, alg_wrappers, job_runner, matrix
    from numpywren_amd import lambdapack as lp
    from numpywren_amd.device import get_backend
    from numpywren_amd.matrix import BigMatrix
    os.environ.pop("NUMPYWREN_AMD_STORE", None)
    matrix.OBJECTS.clear()
    be = get_backend()
    lib = be.lib
    rng = np.random.default_rng(12)
    n, b = 8192, 2048
    G = rng.standard_normal((n, 64))
    A = G @ G.T + n * np.eye(n)

    def run(key):
        X = BigMatrix(key, shape=(n, n), shard_sizes=(b, b), write_header=True)
        for i in range(n // b):
            for j in range(i + 1):
                X.put_block(A[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
        program, meta = alg_wrappers.cholesky(X)
        program.start()
        res = job_runner.lambdapack_run(program, pipeline_width=1)
        program.wait()
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        L = meta["outputs"][0].numpy()
        program.free()
        X.free()
        return L

    L0 = run("comm_live_off")
    ident = ctypes.create_string_buffer(_ffi.NPW_COMM_ID_BYTES)
    _ffi.check(lib.npw_comm_unique_id(ident, _ffi.NPW_COMM_ID_BYTES), "unique_id")
    h = ctypes.c_void_p(0)
    _ffi.check(lib.npw_comm_init(ctypes.byref(h), 0, 1, ident), "comm_init")
    try:
        chain = be.chain_streams(64)[0]
        assert be.stream_cus(chain) == (64, 1)          # the partition's mask, and what it guarantees now
        L1 = run("comm_live_on")
    finally:
        be.synchronize()
        _ffi.check(lib.npw_comm_destroy(h))
    assert np.array_equal(L0, L1)
    np.testing.assert_allclose(np.tril(L1), np.linalg.cholesky(A), rtol=1e-10, atol=1e-9)
    matrix.OBJECTS.clear()
