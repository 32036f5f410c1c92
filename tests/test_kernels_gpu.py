"""PARITY TESTS PROPER (GPU): every numpywren_amd.kernels function, called through the C-ABI of
libnpw_hip.so on the MI355X, against (a) the golden vectors recorded from the reference and
(b) the oracle on seeded inputs, including the reference's edge cases (zero short-circuits, ragged
and non-square tiles, fp32 promotion, non-PD input).

Tolerances (fp64 unless stated): element-wise |got - ref| <= 1e-12 * K-scaled magnitude for GEMM-like
kernels (different summation order than BLAS), 1e-10 relative for the factorizations; fp32 GEMM 1e-4."""
import os

import numpy as np
import pytest

import npw_oracle as oracle
from conftest import GOLDEN, ROOT
from numpywren_amd import kernels

pytestmark = pytest.mark.gpu
KAT = np.load(os.path.join(GOLDEN, "kernels_kat.npz"))

CALLS = {
    "gemm_nn": lambda a, b: kernels.gemm(a, b), "gemm_tn": lambda a, b: kernels.gemm(a, b, transpose_A=True),
    "gemm_nt": lambda a, b: kernels.gemm(a, b, transpose_B=True),
    "gemm_tt": lambda a, b: kernels.gemm(a, b, transpose_A=True, transpose_B=True),
    "gemm_f32": lambda a, b: kernels.gemm(a, b), "gemm_ragged": lambda a, b: kernels.gemm(a, b),
    "syrk": kernels.syrk, "syrk_same": lambda s, x: kernels.syrk(s, x, x), "syrk_xzero": kernels.syrk,
    "syrk_yzero": kernels.syrk, "syrk_ragged": kernels.syrk, "chol": kernels.chol, "trsm": kernels.trsm,
    "trsm_yzero": kernels.trsm, "trsm_ragged": kernels.trsm, "trsm_ragged_yzero": kernels.trsm,
    "add4": kernels.add_matrices, "add_f32": kernels.add_matrices, "identity": kernels.identity,
    "qr_factor": kernels.qr_factor, "qr_factor_stack": kernels.qr_factor, "qr_factor_rr": kernels.qr_factor,
    "qr_factor_tall": kernels.qr_factor, "lq_factor": kernels.lq_factor, "lq_factor_pair": kernels.lq_factor,
    "qr_leaf": kernels.qr_leaf, "lq_leaf": kernels.lq_leaf, "qr_trailing": kernels.qr_trailing_update,
    "lq_trailing": kernels.lq_trailing_update,
}


def _fn_for(case):
    base = case
    while base and base not in CALLS:
        base = base.rsplit("_", 1)[0] if "_" in base else ""
    return CALLS[base]


ALL_CASES = sorted({k.split("/")[0] for k in KAT.files if "/" in k})


@pytest.mark.parametrize("case", ALL_CASES)
def test_golden_kat(case):
    ins = []
    i = 0
    while f"{case}/in{i}" in KAT:
        ins.append(KAT[f"{case}/in{i}"])
        i += 1
    outs = [KAT[f"{case}/out{j}"] for j in range(int(KAT[f"{case}/nout"]))]
    keep = [a.copy() for a in ins]
    got = _fn_for(case)(*ins)
    got = got if isinstance(got, tuple) else (got,)
    assert len(got) == len(outs)
    f32 = outs[0].dtype == np.float32
    for g, o in zip(got, outs):
        assert g.shape == o.shape, (case, g.shape, o.shape)
        assert g.dtype == o.dtype, (case, g.dtype, o.dtype)
        np.testing.assert_allclose(g, o, rtol=1e-4 if f32 else 1e-10, atol=1e-4 if f32 else 1e-11)
    for a, k in zip(ins, keep):
        assert np.array_equal(a, k), f"{case}: kernel modified an input"


@pytest.mark.parametrize("m,n,k", [(64, 64, 64), (128, 128, 128), (256, 192, 160), (100, 37, 19), (1, 1, 1), (130, 257, 33),
                                   (512, 512, 512)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_vs_oracle(m, n, k, ta, tb):
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    A = rng.standard_normal((k, m) if ta else (m, k))
    B = rng.standard_normal((n, k) if tb else (k, n))
    got = kernels.gemm(A, B, transpose_A=ta, transpose_B=tb)
    ref = oracle.gemm(A, B, transpose_A=ta, transpose_B=tb)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * k * 10)
    got32 = kernels.gemm(A.astype(np.float32), B.astype(np.float32), transpose_A=ta, transpose_B=tb)
    assert got32.dtype == np.float32
    np.testing.assert_allclose(got32, ref, rtol=1e-3, atol=1e-5 * k)


def test_gemm_shape_errors():
    with pytest.raises(ValueError, match="not aligned"):
        kernels.gemm(np.ones((4, 5)), np.ones((4, 5)))


@pytest.mark.parametrize("b", [8, 64, 200, 256, 384])
def test_cholesky_chain_vs_oracle(b):
    """chol -> trsm -> syrk on one tile column, the three Cholesky task kinds."""
    rng = np.random.default_rng(b)
    G = rng.standard_normal((b, b))
    A = G @ G.T + b * np.eye(b)
    L = kernels.chol(A)
    Lr = oracle.chol(A)
    np.testing.assert_allclose(L, Lr, rtol=1e-10, atol=1e-10 * np.abs(Lr).max())
    assert not np.triu(L, 1).any()
    Y = rng.standard_normal((b, b))
    X = kernels.trsm(L, Y)
    np.testing.assert_allclose(X, oracle.trsm(Lr, Y), rtol=1e-9, atol=1e-10)
    S = rng.standard_normal((b, b))
    np.testing.assert_allclose(kernels.syrk(S, X, Y), oracle.syrk(S, X, Y), rtol=0, atol=1e-12 * b)


@pytest.mark.parametrize("n,k", [(256, 96), (384, 64), (2048, 128), (2176, 64), (2048, 1024)])
def test_syrk_same_operand_is_bitwise_the_full_product(n, k):
    """x is y (diagonal tiles of the trailing matrix): lower tiles + mirror == the full S - X Y^T, bit for bit."""
    from numpywren_amd.device import get_backend
    be = get_backend()
    rng = np.random.default_rng(n + k)
    Xh = rng.standard_normal((n, k))
    G = rng.standard_normal((n, n))
    Sh = G + G.T
    S, X, Xc = be.to_device(Sh), be.to_device(Xh), be.to_device(Xh)
    sym = be.to_host(be.syrk(S, X, X))
    full = be.to_host(be.syrk(S, X, Xc))          # distinct buffers -> general path
    if n >= 2048 and k >= 1024:
        # the 128 x 128 diagonal blocks are summed k-split (fixed order): same values to rounding, everything
        # else bit for bit
        blk = np.kron(np.eye(n // 128), np.ones((128, 128))).astype(bool)
        assert np.array_equal(sym[~blk], full[~blk])
        np.testing.assert_allclose(sym[blk], full[blk], rtol=0, atol=1e-12 * k)
    else:
        assert np.array_equal(sym, full)
    assert np.array_equal(sym, sym.T)
    np.testing.assert_allclose(sym, oracle.syrk(Sh, Xh, Xh), rtol=0, atol=1e-12 * k)
    z = be.zeros((n, k))                            # allclose(x, 0) short-circuit keeps s
    assert np.array_equal(be.to_host(be.syrk(S, z, z)), Sh)


def test_chol_not_positive_definite():
    A = np.eye(40)
    A[17, 17] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        kernels.chol(A)
    A[17, 17] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        kernels.chol(A)


@pytest.mark.parametrize("n", [129, 192, 256, 257, 300, 384, 511, 513, 640, 1000, 1536])
def test_chol_block_column_handoff(n):
    """Sizes with at least one fused block-column launch (diagonal block + panel rows that follow it one 16-column
    step behind through tagged-slot messages): odd leading dimensions (scalar write-through stores), a single panel
    row, ragged last block columns."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    L = kernels.chol(a)
    ref = np.linalg.cholesky(a)
    np.testing.assert_allclose(L, ref, atol=1e-12 * n, rtol=0)
    assert not np.triu(L, 1).any()
    assert np.linalg.norm(L @ L.T - a) / np.linalg.norm(a) < 1e-14
    # the factor's cached block inverses (one launch for all of them at the end) serve the trsm consumers
    y = rng.standard_normal((n, n))
    np.testing.assert_allclose(kernels.trsm(L, y) @ ref.T, y, atol=1e-9 * n)
    # bitwise repeatable: the hand-off changes when a panel row is computed, never what is computed
    assert np.array_equal(kernels.chol(a), L)


@pytest.mark.parametrize("n,bad", [(300, 5), (300, 140), (300, 299), (256, 3), (256, 200), (640, 128), (640, 400), (1000, 600), (1024, 1023)])
def test_chol_failure_is_reported_from_any_block_column(n, bad):
    """A non-positive pivot in any block column: LinAlgError, and the panel workgroups that were waiting for the
    rest of that block column's messages are released (the call returns instead of spinning)."""
    rng = np.random.default_rng(n + bad)
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    a[bad, bad] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        kernels.chol(a)
    # and the library is in a sane state afterwards
    a[bad, bad] = 4.0 * n
    np.testing.assert_allclose(kernels.chol(a), np.linalg.cholesky(a), atol=1e-11 * n)


def test_zero_short_circuits():
    rng = np.random.default_rng(5)
    s, x = rng.standard_normal((32, 32)), rng.standard_normal((32, 32))
    tiny = np.full((32, 32), 9e-9)
    assert np.array_equal(kernels.syrk(s, tiny, x), s)          # allclose(x, 0) -> s unchanged
    assert np.array_equal(kernels.syrk(s, x, tiny), s)
    assert not np.array_equal(kernels.syrk(s, np.full((32, 32), 2e-8), x), s)   # just above atol
    nanx = tiny.copy()
    nanx[3, 3] = np.nan                                                         # NaN is never allclose to 0
    assert np.isnan(kernels.syrk(s, nanx, x)).any()
    L = np.linalg.cholesky(np.eye(32) * 4 + 1)
    assert not kernels.trsm(L, tiny).any()
    z = kernels.trsm(L[:8, :8], np.zeros((5, 8)))                               # reference's odd zero shape
    assert z.shape == (8, 5) and not z.any()


def test_add_matrices_promotes_to_float64():
    rng = np.random.default_rng(6)
    a = rng.standard_normal((16, 24)).astype(np.float32)
    b = rng.standard_normal((16, 24))
    out = kernels.add_matrices(a, b, a)
    assert out.dtype == np.float64
    assert np.array_equal(out, oracle.add_matrices(a, b, a))                    # bit-exact: same left-to-right order
    many = [rng.standard_normal((8, 8)) for _ in range(11)]
    assert np.array_equal(kernels.add_matrices(*many), oracle.add_matrices(*many))


@pytest.mark.parametrize("dtype,ta,tb", [(np.float64, False, False), (np.float64, False, True), (np.float64, True, False),
                                         (np.float32, False, False), (np.float32, True, True)])
def test_gemm_batched_is_gemm_bit_for_bit(dtype, ta, tb):
    """npw_dgemm_batched / npw_sgemm_batched (the executor's batch of ready gemm tasks): every problem of the one launch equals
    the single call bitwise -- full tiles, a ragged shape (EDGE tiling), 1, 2, 5 and 19 problems (two launches)."""
    from numpywren_amd.device import get_backend
    be = get_backend()
    rng = np.random.default_rng(71)
    for (m, n, k), count in (((256, 128, 192), 5), ((200, 67, 93), 2), ((128, 128, 64), 19), ((64, 64, 64), 1)):
        As = [rng.standard_normal((k, m) if ta else (m, k)).astype(dtype) for _ in range(count)]
        Bs = [rng.standard_normal((n, k) if tb else (k, n)).astype(dtype) for _ in range(count)]
        dA, dB = [be.to_device(a) for a in As], [be.to_device(b) for b in Bs]
        got = be.gemm_batched(list(zip(dA, dB)), ta, tb)
        for a, b, da, db, g in zip(As, Bs, dA, dB, got):
            one = be.to_host(be.gemm(da, db, ta, tb))
            assert g.dtype == dtype and np.array_equal(be.to_host(g), one)
            ref = oracle.gemm(a.astype(np.float64), b.astype(np.float64), transpose_A=ta, transpose_B=tb)
            np.testing.assert_allclose(one, ref, rtol=0, atol=(1e-12 if dtype == np.float64 else 1e-4) * k)
    # through kernels.gemm's batch entry (what the executor calls), mixed with a host-array task
    a, b = rng.standard_normal((64, 32)), rng.standard_normal((32, 48))
    outs = kernels.gemm._npw_batch(be, None, [[be.to_device(a), be.to_device(b)], [be.to_device(a), be.to_device(b)], [a, b]], [{}, {}, {}])
    assert np.array_equal(be.to_host(outs[0]), be.to_host(outs[1])) and np.allclose(outs[2], a @ b)


def test_add_matrices_skips_the_shared_zero_tile_bit_for_bit():
    """The padding operands of the GEMM program's add tree are reads of never-written constant_zeros tiles: the backend's shared
    zero tile, which add_n does not read.  Same bits as the reference's np.zeros(shape) += a, signed zeros included."""
    from numpywren_amd.device import get_backend
    be = get_backend()
    rng = np.random.default_rng(61)
    a = rng.standard_normal((48, 40))
    a[0, :5] = -0.0
    a[1, :5] = 0.0
    b = -a.copy()                       # exact cancellation: +0.0 everywhere in a + b
    z = be.shared_zeros((48, 40))
    zeros = np.zeros((48, 40))
    for ops, ref in (((a, None), (a, zeros)), ((None, a, None, None), (zeros, a, zeros, zeros)), ((a, None, b), (a, zeros, b)),
                     ((None, None), (zeros, zeros))):
        tiles = [z if o is None else be.to_device(o) for o in ops]
        got = be.to_host(kernels.add_matrices(*tiles))
        want = oracle.add_matrices(*ref)
        assert got.dtype == np.float64 and np.array_equal(got, want)
        assert np.array_equal(np.signbit(got), np.signbit(want))
    assert not be.to_host(z).any()      # the shared tile itself is untouched


@pytest.mark.parametrize("m,n", [(8, 8), (16, 8), (64, 64), (96, 32), (128, 128), (200, 67), (256, 128)])
def test_qr_factor_vs_oracle(m, n):
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n))
    V, T, R = kernels.qr_factor(A)
    Vr, Tr, Rr = oracle.qr_factor(A)
    tol = 1e-11 * max(m, n)
    np.testing.assert_allclose(V, Vr, atol=tol)
    np.testing.assert_allclose(T, Tr, atol=tol)
    np.testing.assert_allclose(R, Rr, atol=tol * 10)
    # structure and the defining identity Q = I - V T V^T, Q[:, :n] R = A
    assert not np.triu(V, 1).any() and np.all(np.diag(V) == 1) and not np.tril(T, -1).any() and not np.tril(R, -1).any()
    Q = np.eye(m) - V @ T @ V.T
    np.testing.assert_allclose(Q[:, :n] @ R, A, atol=1e-11 * m)
    np.testing.assert_allclose(Q.T @ Q, np.eye(m), atol=1e-11 * m)


@pytest.mark.parametrize("m,n", [(8, 12), (32, 33), (64, 200), (100, 256), (256, 512)])
def test_qr_factor_wide_vs_oracle(m, n):
    """More columns than rows: the reference's fast_qr hands over to slow_qr (kernels.py:94-95 -> 67-84)."""
    rng = np.random.default_rng(m * n)
    A = rng.standard_normal((m, n))
    V, T, R = kernels.qr_factor(A)
    assert V.shape == (m, m) and T.shape == (m, m) and R.shape == (m, n)
    Vr, Tr, Rr = oracle.qr_factor(A)
    tol = 1e-11 * n
    np.testing.assert_allclose(V, Vr, atol=tol)
    np.testing.assert_allclose(T, Tr, atol=tol)
    np.testing.assert_allclose(R, Rr, atol=tol * 10)
    assert not np.triu(V, 1).any() and np.all(np.diag(V) == 1) and not np.tril(T, -1).any() and not np.tril(R, -1).any()
    Q = np.eye(m) - V @ T @ V.T
    np.testing.assert_allclose(Q @ R, A, atol=1e-11 * n)
    np.testing.assert_allclose(Q.T @ Q, np.eye(m), atol=1e-11 * m)
    for got, ref in zip(kernels.slow_qr(A), (Vr, Tr, Rr)):
        np.testing.assert_allclose(got, ref, atol=tol * 10)
    # lq_factor of a tall block is the same path transposed
    for got, ref in zip(kernels.lq_factor(A.T.copy()), oracle.lq_factor(A.T.copy())):
        np.testing.assert_allclose(got, ref, atol=tol * 10)


@pytest.mark.parametrize("m,n,count", [(64, 64, 2), (200, 67, 5), (256, 128, 3), (512, 300, 4), (1024, 512, 9),
                                       (1536, 1280, 8), (1400, 1100, 8),   # several superblocks: the far updates of round 4
                                       (2100, 96, 36)])     # 36 x 9 slabs > 256 workgroups: two rows per thread
def test_qr_batched_equals_one_by_one(m, n, count):
    """npw_dgeqrt_batched: `count` factorisations in lock step give what npw_dgeqrt gives for each (the same kernels
    on the same data; only the split-k of the long reductions may regroup sums when the batch changes their grid)."""
    be = kernels.get_backend()
    rng = np.random.default_rng(m + n + count)
    As = [rng.standard_normal((m, n)) for _ in range(count)]
    tiles = [be.to_device(a) for a in As]
    got = be.geqrt_batched(tiles)
    assert len(got) == count
    for a, t, (V, T, R) in zip(As, tiles, got):
        V1, T1, R1 = be.geqrt(t)
        for x, y in ((V, V1), (T, T1), (R, R1)):
            np.testing.assert_allclose(be.to_host(x), be.to_host(y), atol=1e-12, rtol=0)
        Vr, Tr, Rr = oracle.qr_factor(a)
        tol = 1e-11 * max(m, n)
        np.testing.assert_allclose(be.to_host(V), Vr, atol=tol)
        np.testing.assert_allclose(be.to_host(T), Tr, atol=tol)
        np.testing.assert_allclose(be.to_host(R), Rr, atol=tol * 10)
    # the outputs of a batch share allocations: tiles must stay valid after their siblings are dropped
    keep = got[-1]
    want = [be.to_host(x) for x in keep]
    del got
    be.synchronize()
    junk = [be.fill_random((m, n), 3) for _ in range(3)]
    for x, w in zip(keep, want):
        assert np.array_equal(be.to_host(x), w)
    # mixed shapes and wide blocks fall back to one call each
    mixed = be.geqrt_batched([tiles[0], be.to_device(rng.standard_normal((m + 8, n)))])
    assert mixed[0][0].shape == (m, n) and mixed[1][0].shape == (m + 8, n)


@pytest.mark.parametrize("m,n,count,tri", [(300, 200, 1, False), (512, 384, 3, False), (1536, 1280, 8, False), (1400, 1100, 9, False),
                                           (96, 96, 1, True), (512, 512, 3, True), (1024, 1024, 9, True)])
def test_qr_r_only_form_gives_the_same_v_and_r(m, n, count, tri):
    """T == NULL through the C-ABI (npw_hip.h: the "R only" request the executor makes for T tiles it would drop unread):
    T is not returned, V and R are bit for bit those of the full call -- the factorisation applies the same diagonal
    blocks of T either way, only their home (a strip of the workspace) and the assembly of the rest differ."""
    be = kernels.get_backend()
    rng = np.random.default_rng(m + 3 * n + count)
    if tri:
        tiles = [(be.to_device(np.triu(rng.standard_normal((n, n)))), be.to_device(np.triu(rng.standard_normal((n, n)))))
                 for _ in range(count)]
        full, lean = be.tpqrt_batched(tiles), be.tpqrt_batched(tiles, want_t=False)
    else:
        tiles = [be.to_device(rng.standard_normal((m, n))) for _ in range(count)]
        full, lean = be.geqrt_batched(tiles), be.geqrt_batched(tiles, want_t=False)
    for (V, T, R), (V2, T2, R2) in zip(full, lean):
        assert T is not None and T2 is None and R2.upper
        assert np.array_equal(be.to_host(V), be.to_host(V2)) and np.array_equal(be.to_host(R), be.to_host(R2))
    # neither V nor T (a batch whose V tiles are dropped unread as well): the working matrix lives in the stream's scratch
    bare = be.tpqrt_batched(tiles, want_t=False, want_v=False) if tri else be.geqrt_batched(tiles, want_t=False, want_v=False)
    for (V, T, R), (V3, T3, R3) in zip(full, bare):
        assert T3 is None and (V3 is None or count == 1) and np.array_equal(be.to_host(R), be.to_host(R3))
    # through the kernel seam nothing changes unless the executor says the tile is dropped on store
    a = rng.standard_normal((96, 64))
    assert kernels.qr_factor(a)[1] is not None
    with kernels.stream_scope(None, None, True, unwanted=[{0, 1}]):
        v, t, r = kernels.qr_factor(a)
    assert t is None and np.array_equal(r, kernels.qr_factor(a)[2])
    assert be.qr_handoff_timeouts() == 0          # no wait for a hand-off slot expired (npw_dgeqrt_handoff_timeouts)


@pytest.mark.parametrize("n,count", [(8, 1), (32, 2), (40, 3), (96, 1), (128, 4), (200, 2), (512, 3), (1024, 2), (1024, 9),
                                     (512, 60)])     # 60 x 5 slabs > 256 workgroups: the panel kernel's two-rows-per-thread form
def test_stacked_triangle_qr_equals_dense(n, count):
    """npw_dtpqrt_batched (the node of a TSQR tree: two stacked R factors) against the dense factorisation of the same
    stack and against the oracle: same V = [I; V2], T, R."""
    be = kernels.get_backend()
    rng = np.random.default_rng(n * 7 + count)
    pairs = [(np.triu(rng.standard_normal((n, n))), np.triu(rng.standard_normal((n, n)))) for _ in range(count)]
    tiles = [(be.to_device(a), be.to_device(c)) for a, c in pairs]
    got = be.tpqrt_batched(tiles)
    tol = 1e-11 * n
    for (a, c), (V, T, R) in zip(pairs, got):
        assert V.shape == (2 * n, n) and T.shape == (n, n) and R.shape == (n, n) and R.upper
        V, T, R = be.to_host(V), be.to_host(T), be.to_host(R)
        Vd, Td, Rd = (be.to_host(x) for x in be.geqrt(be.to_device(np.vstack([a, c]))))
        np.testing.assert_allclose(V, Vd, atol=tol)
        np.testing.assert_allclose(T, Td, atol=tol)
        np.testing.assert_allclose(R, Rd, atol=tol * 10)
        Vr, Tr, Rr = oracle.qr_factor(a, c)
        np.testing.assert_allclose(V, Vr, atol=tol)
        np.testing.assert_allclose(T, Tr, atol=tol)
        np.testing.assert_allclose(R, Rr, atol=tol * 10)
        # structure: identity on top of an upper triangle
        assert np.array_equal(V[:n], np.eye(n)) and not np.tril(V[n:], -1).any()
        assert not np.tril(T, -1).any() and not np.tril(R, -1).any()
    # kernels.qr_factor takes the structured route only for tiles the backend itself flagged as R factors
    Ra = be.geqrt(be.to_device(rng.standard_normal((2 * n, n))))[2]
    Rb = be.geqrt(be.to_device(rng.standard_normal((n, n))))[2]
    assert Ra.upper and Rb.upper
    Vk, Tk, Rk = kernels.qr_factor(Ra, Rb)
    Vo, To, Ro = oracle.qr_factor(be.to_host(Ra), be.to_host(Rb))
    np.testing.assert_allclose(be.to_host(Vk), Vo, atol=tol)
    np.testing.assert_allclose(be.to_host(Rk), Ro, atol=tol * 10)
    plain = be.to_device(be.to_host(Ra))         # same numbers, no flag: dense route, same answer
    assert not plain.upper
    np.testing.assert_allclose(be.to_host(kernels.qr_factor(plain, Rb)[2]), Ro, atol=tol * 10)


def test_qr_family_vs_oracle():
    rng = np.random.default_rng(9)
    b = 48
    A, B, S0, S1 = (rng.standard_normal((b, b)) for _ in range(4))
    for got, ref in zip(kernels.qr_factor(A, B), oracle.qr_factor(A, B)):
        np.testing.assert_allclose(got, ref, atol=1e-10)
    for got, ref in zip(kernels.lq_factor(A, B), oracle.lq_factor(A, B)):
        np.testing.assert_allclose(got, ref, atol=1e-10)
    V, T, R = oracle.qr_factor(np.triu(A), np.triu(B))
    for got, ref in zip(kernels.qr_trailing_update(V, T, S0, S1), oracle.qr_trailing_update(V, T, S0, S1)):
        np.testing.assert_allclose(got, ref, atol=1e-10)
    Vl, Tl, Ll = oracle.lq_factor(np.tril(A), np.tril(B))
    for got, ref in zip(kernels.lq_trailing_update(Vl, Tl, S0, S1), oracle.lq_trailing_update(Vl, Tl, S0, S1)):
        np.testing.assert_allclose(got, ref, atol=1e-10)
    V1, T1, _ = oracle.qr_factor(A)
    np.testing.assert_allclose(kernels.qr_leaf(V1, T1, S0), oracle.qr_leaf(V1, T1, S0), atol=1e-11)
    Vq, Tq, _ = oracle.lq_factor(A)
    np.testing.assert_allclose(kernels.lq_leaf(Vq, Tq, S0), oracle.lq_leaf(Vq, Tq, S0), atol=1e-10)
    a, z = kernels.qr_trailing_update(V1, T1, S0, None)
    np.testing.assert_allclose(a, oracle.qr_leaf(V1, T1, S0), atol=1e-11)
    assert not z.any()


def test_flop_models_match_reference():
    import json
    ref = json.loads(bytes(KAT["flops_json"]).decode())
    a8, a16 = np.zeros((8, 8)), np.zeros((16, 8))
    assert kernels.gemm.flops(a8, a8) == ref["gemm"]
    assert kernels.syrk.flops(a8, a8, a8) == ref["syrk"]
    assert kernels.chol.flops(a8) == ref["chol"]
    assert kernels.qr_factor.flops(a8) == ref["qr_factor"]
    assert kernels.qr_factor.flops(a8, a8) == ref["qr_factor_stack"]
    assert kernels.qr_leaf.flops(a8, a8, a8) == ref["qr_leaf"]
    assert kernels.qr_trailing_update.flops(a16, a8, a8, a8) == ref["qr_trailing_update"]
    assert not hasattr(kernels.trsm, "flops")


def test_tile_size_properties():
    """Size-independent checks at tile sizes the oracle cannot afford in a test: residuals of the
    4096-class kernels (linearity / round trips) computed on the device."""
    from numpywren_amd.device import get_backend
    be = get_backend()
    n = 2048
    G = be.fill_random((n, n), seed=11)
    A = be.gemm(G, G, False, True, alpha=1.0 / n)
    A = be.add_diag(A, 4.0)
    L, info = be.chol(A)
    assert be.read_flag(info) == 0
    # || A - L L^T ||_F / || A ||_F
    Rm = be.gemm(L, L, False, True, alpha=-1.0, beta=1.0, C=A)
    res = np.sqrt(be.sumsq(Rm) / be.sumsq(A))
    assert res < 1e-14, res
    Y = be.fill_random((n, n), seed=12)
    Xs = be.trsm(L, Y)
    Rt = be.gemm(Xs, L, False, True, alpha=1.0, beta=-1.0, C=Y)
    assert np.sqrt(be.sumsq(Rt) / be.sumsq(Y)) < 1e-14
    V, T, R = be.geqrt(Y)
    # R^T R = Y^T Y
    YtY = be.gemm(Y, Y, True, False)
    D = be.gemm(R, R, True, False, alpha=1.0, beta=-1.0, C=YtY)
    assert np.sqrt(be.sumsq(D) / be.sumsq(YtY)) < 1e-13


QRG = np.load(os.path.join(GOLDEN, "qr.npz"))


@pytest.mark.parametrize("tag", ["tri_4", "tri_7", "tri_8", "tri_32", "tri_40", "tri_64", "tri_full_8", "tri_full_40"])
def test_qr_factor_triangular_golden(tag):
    v, t, r = kernels.qr_factor_triangular(QRG[f"{tag}/x0"], QRG[f"{tag}/x1"])
    np.testing.assert_allclose(v, QRG[f"{tag}/v"], atol=1e-13)
    np.testing.assert_allclose(t, QRG[f"{tag}/t"], atol=1e-11)
    np.testing.assert_allclose(r, QRG[f"{tag}/r"], atol=1e-11)


def test_qr_factor_triangular_tile_size():
    """300 x 300 triangles: R^T R == x0^T x0 + x1^T x1 and the blocked T has 32 non-zero rows."""
    rng = np.random.default_rng(31)
    x0, x1 = np.triu(rng.standard_normal((300, 300))), np.triu(rng.standard_normal((300, 300)))
    v, t, r = kernels.qr_factor_triangular(x0, x1)
    G = x0.T @ x0 + x1.T @ x1
    np.testing.assert_allclose(r.T @ r, G, atol=1e-10 * np.abs(G).max())
    assert not np.tril(r, -1).any() and not t[32:].any() and np.array_equal(v, np.eye(300))
    vo, to, ro = oracle.qr_factor_triangular(x0, x1)
    np.testing.assert_allclose(t, to, atol=1e-10)
    np.testing.assert_allclose(r, ro, atol=1e-10)


@pytest.mark.parametrize("log_cond,n", [(6, 1024), (12, 1024), (10, 2048)])
def test_chol_trsm_ill_conditioned(log_cond, n):
    """explicit inverses of 128 / 512 / 1024-wide diagonal blocks must not cost backward stability"""
    import scipy.linalg as sl
    rng = np.random.default_rng(log_cond)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * np.logspace(0, -log_cond, n)) @ Q.T
    A = (A + A.T) / 2
    L = kernels.chol(A)
    assert np.linalg.norm(A - L @ L.T) / np.linalg.norm(A) < 5e-15
    B = rng.standard_normal((300, n))
    X = kernels.trsm(L, B)
    assert np.linalg.norm(X @ L.T - B) / (np.linalg.norm(X) * np.linalg.norm(L)) < 1e-15
    Xr = sl.solve_triangular(np.linalg.cholesky(A), B.T, lower=True).T
    assert np.linalg.norm(X - Xr) / np.linalg.norm(Xr) < 1e-13 * 10.0 ** log_cond


def test_block_copy_and_reshard_down(hbm_store):
    """HipBackend.block (strided device copy) and matrix_init.reshard_down on device tiles."""
    from numpywren_amd import matrix_init
    from numpywren_amd.matrix import BigMatrix
    from numpywren_amd.matrix_init import shard_matrix
    be = kernels.get_backend()
    rng = np.random.default_rng(2)
    a = rng.standard_normal((70, 45))
    t = be.to_device(a)
    assert np.array_equal(be.to_host(be.block(t, 3, 61, 7, 44)), a[3:61, 7:44])
    assert np.array_equal(be.to_host(be.block(t, 0, 70, 0, 45)), a)
    f = a.astype(np.float32)
    assert np.array_equal(be.to_host(be.block(be.to_device(f), 10, 11, 0, 45)), f[10:11])
    Xh = rng.standard_normal((300, 200))
    X = BigMatrix("rs_gpu", shape=Xh.shape, shard_sizes=(128, 64), write_header=True)
    shard_matrix(X, Xh)
    Y = matrix_init.reshard_down(X, [4, 2])
    assert tuple(Y.shard_sizes) == (32, 32) and np.array_equal(Y.numpy(), Xh)


@pytest.mark.parametrize("s,nblk", [(1, 2), (5, 1), (8, 3), (33, 2), (130, 2), (256, 1)])
def test_banded_to_bidiagonal_vs_oracle(s, nblk):
    """kernels.banded_to_bidiagonal (reference kernels.py:43-65) against the oracle's restatement: same (d, e) -- both
    follow DLARFG's signs -- and the singular values of the blocks (the invariant DGBBRD's own output would share)."""
    rng = np.random.default_rng(s * 10 + nblk)
    x = [rng.standard_normal((s, s)) for _ in range(nblk)]
    keep = [b.copy() for b in x]
    d, e = kernels.banded_to_bidiagonal(x)
    do, eo = oracle.banded_to_bidiagonal(x)
    assert d.shape == do.shape and e.shape == eo.shape
    tol = 1e-12 * s * max(1.0, np.abs(do).max())
    np.testing.assert_allclose(d, do, rtol=0, atol=tol)
    np.testing.assert_allclose(e, eo, rtol=0, atol=tol)
    B = np.diag(d) + np.diag(e, 1)
    ref = np.sort(np.concatenate([np.linalg.svd(b, compute_uv=False) for b in x]))
    np.testing.assert_allclose(np.sort(np.linalg.svd(B, compute_uv=False)), ref, atol=1e-12 * s * ref.max())
    for a, k in zip(x, keep):
        assert np.array_equal(a, k)
    # device tiles in, device vectors out
    be = kernels.get_backend()
    dd, ee = kernels.banded_to_bidiagonal([be.to_device(b) for b in x])
    assert np.array_equal(be.to_host(dd), d) and np.array_equal(be.to_host(ee), e)
    with pytest.raises(ValueError):
        kernels.banded_to_bidiagonal([x[0], np.zeros((s + 1, s))])


@pytest.mark.parametrize("n,m,count", [(64, 64, 2), (300, 200, 3), (640, 640, 5), (1024, 512, 16), (2048, 2048, 3)])
def test_trsm_batched_equals_one_by_one(n, m, count):
    """npw_dtrsm_rltn_inv_batched (right-hand sides in separate allocations sharing one factor -- the trsm tasks of a
    block column) against `trsm` on each and against the oracle; one all-zero right-hand side takes the reference's
    short-circuit inside the batch."""
    be = kernels.get_backend()
    rng = np.random.default_rng(n + m + count)
    G = rng.standard_normal((n, n))
    Lh = np.linalg.cholesky(G @ G.T + n * np.eye(n))
    Ys = [rng.standard_normal((m, n)) for _ in range(count)]
    Ys[-1] = np.zeros((m, n))
    L = be.to_device(Lh)
    junk = [be.fill_random((7 + i, 5), i) for i in range(3)]     # perturb the allocator: no constant stride between tiles
    tiles = [be.to_device(y) for y in Ys]
    got = be.trsm_batched(L, tiles)
    assert len(got) == count
    for y, t, x in zip(Ys, tiles, got):
        one = be.to_host(be.trsm(L, t))
        xb = be.to_host(x)
        assert np.array_equal(xb, one) or np.abs(xb - one).max() <= 1e-13 * max(1.0, np.abs(one).max())
        ref = oracle.trsm(Lh, y) if y.any() else np.zeros((m, n))
        np.testing.assert_allclose(xb, ref, rtol=1e-9, atol=1e-10)
    assert not be.to_host(got[-1]).any()
    del junk


def test_trsm_fused_form_matches_the_recursive_one():
    """$NPW_TRSM_FUSED=1 (off by default: profiles/r05_step_level_experiments.md): the solve with a factor made of whole
    1024-wide groups as ONE block-diagonal triangular product (gemm's b_blockdiag, 128 x 128 tiles handed out longest first)
    followed by in-place updates with the premultiplied blocks inv(L_gg) L[g, h] (a_blockdiag) -- the GEMM options and the
    factor's extra blocks stay tested although no default path takes them.  In a subprocess: the switch is read once."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import npw_oracle as oracle
from numpywren_amd import kernels
be = kernels.get_backend()
for n, m, count in ((2048, 1024, 1), (4096, 512, 3), (3072, 256, 2)):
    rng = np.random.default_rng(n + m)
    G = rng.standard_normal((n, 256))
    Lh = np.linalg.cholesky(G @ G.T + n * np.eye(n))
    L, _ = be.chol(be.to_device(Lh @ Lh.T))
    Ys = [rng.standard_normal((m, n)) for _ in range(count)]
    tiles = [be.to_device(y) for y in Ys]
    got = be.trsm_batched(L, tiles) if count > 1 else [be.trsm(L, tiles[0])]
    Ld = np.tril(be.to_host(L))
    for y, x in zip(Ys, got):
        xh = be.to_host(x)
        assert np.abs(xh @ Ld.T - y).max() <= 1e-11 * np.abs(y).max() * n ** 0.5, (n, np.abs(xh @ Ld.T - y).max())
        np.testing.assert_allclose(xh, oracle.trsm(Ld, y), rtol=1e-8, atol=1e-9)
print("fused ok")
""" % (ROOT, os.path.join(ROOT, "oracle"))
    env = dict(os.environ, NPW_TRSM_FUSED="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fused ok" in out.stdout, (out.stdout + out.stderr)[-3000:]


def test_trsm_tasks_of_a_block_column_run_as_one_batch(hbm_store, monkeypatch):
    """The executor issues the trailing updates that enable further trsm tasks first and then hands the block column's
    trsm tasks to kernels.trsm._npw_batch together, which runs them as ONE batched solve from _TRSM_BATCH_MIN right-hand
    sides on (8 by default; lowered here so that a 5 x 5 tile grid exercises it); the factor is what the reference's
    order gives."""
    monkeypatch.setattr(kernels, "_TRSM_BATCH_MIN", 2)
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    from numpywren_amd.matrix import BigMatrix
    from numpywren_amd.matrix_init import shard_matrix
    be = kernels.get_backend()
    rng = np.random.default_rng(77)
    n, b = 1280, 256
    G = rng.standard_normal((n, 64))
    A = G @ G.T + n * np.eye(n)
    X = BigMatrix("trsm_batch_chol", shape=A.shape, shard_sizes=(b, b))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    be.enable_kernel_timers(("trsm_batch", "trsm"))
    program.start()
    res = job_runner.lambdapack_run(program, timeout=300)
    program.wait()
    times = be.collect_kernel_times()
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    nb = n // b
    assert len(res["executed_messages"]) == nb * (nb + 1) * (nb + 2) // 6
    # block columns 0 .. nb-3 have >= 2 trsm tasks each: one batched call per column; the last column's single task alone
    assert len(times["trsm_batch"]) == nb - 2 and len(times["trsm"]) == 1
    L = meta["outputs"][0].numpy()
    ref = oracle.cholesky(A, b)
    np.testing.assert_allclose(L, ref, rtol=1e-10, atol=1e-10 * np.abs(ref).max())
    program.free()


@pytest.mark.parametrize("m,n,k,count", [(256, 256, 256, 3), (384, 256, 128, 5), (1024, 1024, 512, 16), (200, 136, 72, 4)])
def test_syrk_batched_equals_one_by_one(m, n, k, count):
    """npw_dgemm_nt_sub_batched (independent trailing updates in separate allocations, one launch) against `syrk` on
    each -- bit for bit: every problem runs the same tiles in the same order -- and against the oracle; one problem with
    an all-zero x takes the reference's short-circuit (kernels.py:213-214) inside the batch."""
    be = kernels.get_backend()
    rng = np.random.default_rng(m + n + k + count)
    probs = [(rng.standard_normal((m, n)), rng.standard_normal((m, k)), rng.standard_normal((n, k))) for _ in range(count)]
    probs[1] = (probs[1][0], np.zeros((m, k)), probs[1][2])
    junk = [be.fill_random((5 + i, 3), i) for i in range(3)]
    tiles = [tuple(be.to_device(a) for a in pr) for pr in probs]
    got = be.syrk_batched(tiles)
    assert len(got) == count
    for (S, X, Y), t, d in zip(probs, tiles, got):
        one = be.to_host(be.syrk(*t))
        assert np.array_equal(be.to_host(d), one)
        np.testing.assert_allclose(one, oracle.syrk(S, X, Y), rtol=0, atol=1e-12 * k)
    assert np.array_equal(be.to_host(got[1]), probs[1][0])
    del junk


def test_syrk_batched_symmetric_route_equals_one_by_one():
    """Several x-is-y updates of one shape in one batch (strictly-lower tile pairs of all problems in ONE launch, then the
    diagonal blocks): bit for bit `syrk` on each, one of them with an all-zero x (short-circuit: s comes back)."""
    be = kernels.get_backend()
    rng = np.random.default_rng(12)
    n, k, count = 1024, 384, 3
    probs = [(rng.standard_normal((n, n)), rng.standard_normal((n, k))) for _ in range(count)]
    probs[2] = (probs[2][0], np.zeros((n, k)))
    tiles = []
    for S, X in probs:
        dX = be.to_device(X)
        tiles.append((be.to_device(S), dX, dX))
    got = be.syrk_batched(tiles)
    for (S, X), t, d in zip(probs, tiles, got):
        one = be.to_host(be.syrk(*t))
        assert np.array_equal(be.to_host(d), one)
        np.testing.assert_allclose(one, oracle.syrk(S, X, X), rtol=0, atol=1e-12 * k)
    assert np.array_equal(be.to_host(got[2]), probs[2][0])


def test_syrk_batched_mixed_kinds():
    """x is y (the symmetric route), another shape and a general update in one call: grouped by kind, same answers."""
    be = kernels.get_backend()
    rng = np.random.default_rng(3)
    S = rng.standard_normal((1024, 1024)); X = rng.standard_normal((1024, 256)); Y = rng.standard_normal((1024, 256))
    S2 = rng.standard_normal((128, 128)); X2 = rng.standard_normal((128, 64))
    dS, dX, dY, dS2, dX2 = (be.to_device(a) for a in (S, X, Y, S2, X2))
    got = be.syrk_batched([(dS, dX, dY), (dS, dX, dX), (dS2, dX2, dX2), (dS, dY, dX)])
    for d, ref in zip(got, (S - X @ Y.T, S - X @ X.T, S2 - X2 @ X2.T, S - Y @ X.T)):
        np.testing.assert_allclose(be.to_host(d), ref, rtol=0, atol=1e-11)


def test_trailing_updates_of_a_block_column_run_batched(hbm_store):
    """kernels.syrk._npw_batch through the executor: the ready off-diagonal updates of a block column go out as batched
    launches; the factor equals the oracle's tile Cholesky."""
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    from numpywren_amd.matrix import BigMatrix
    from numpywren_amd.matrix_init import shard_matrix
    be = kernels.get_backend()
    rng = np.random.default_rng(78)
    n, b = 1536, 256
    G = rng.standard_normal((n, 64))
    A = G @ G.T + n * np.eye(n)
    X = BigMatrix("syrk_batch_chol", shape=A.shape, shard_sizes=(b, b))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    res = job_runner.lambdapack_run(program, timeout=300)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    nb = n // b
    assert len(res["executed_messages"]) == nb * (nb + 1) * (nb + 2) // 6
    prof = program.get_all_profiling_info()
    batched = [p for p in prof if p.get("kernel") == "syrk" and p.get("batch", 1) > 1]
    assert batched, "no syrk task ran in a batch"
    L = meta["outputs"][0].numpy()
    ref = oracle.cholesky(A, b)
    np.testing.assert_allclose(L, ref, rtol=1e-10, atol=1e-10 * np.abs(ref).max())
    program.free()


@pytest.mark.parametrize("n,m", [(1024, 192), (1536, 64), (640, 130)])
def test_trsm_in_place_through_the_abi(n, m):
    """npw_dtrsm_rltn_inv with X == B (allowed by the header): the first column block is solved in place with the
    128-wide block inverses, the rest with the wide inverse groups; same answer as the out-of-place call."""
    import ctypes
    from numpywren_amd import _ffi
    be = kernels.get_backend()
    lib = be.lib
    rng = np.random.default_rng(n + m)
    G = rng.standard_normal((n, n))
    Lh = np.linalg.cholesky(G @ G.T + n * np.eye(n))
    Bh = rng.standard_normal((m, n))
    L, B = be.to_device(Lh), be.to_device(Bh)
    winv = be.alloc(lib.npw_dtrtri_diag_bytes(n))
    ws = be.alloc(lib.npw_dtrsm_rltn_inv_workspace_bytes(m, n))
    sh = be.default_stream.handle
    _ffi.check(lib.npw_dtrtri_diag(n, L.ptr, n, winv.ptr, sh), "trtri_diag")
    X = be.empty((m, n))
    _ffi.check(lib.npw_dtrsm_rltn_inv(m, n, L.ptr, n, winv.ptr, B.ptr, n, X.ptr, n, None, ws.ptr, sh), "trsm")
    _ffi.check(lib.npw_dtrsm_rltn_inv(m, n, L.ptr, n, winv.ptr, B.ptr, n, B.ptr, n, None, ws.ptr, sh), "trsm in place")
    be.synchronize()
    out, inplace = be.to_host(X), be.to_host(B)
    ref = oracle.trsm(Lh, Bh)
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(inplace, ref, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gemm_reused_operand_takes_the_transposed_copy(dtype):
    """A big k x n operand that is multiplied several times (the B tiles of the GEMM program) is transposed once on its
    second use and the later products run in the NT form: same products in the same order -> bitwise the first call's
    result; a rewritten tile drops the copy."""
    be = kernels.get_backend()
    n = be.GEMM_TRANSPOSE_MIN
    rng = np.random.default_rng(7)
    A = be.to_device(rng.standard_normal((256, n)).astype(dtype))
    Bh = rng.standard_normal((n, n)).astype(dtype)
    B = be.to_device(Bh)
    first = be.to_host(be.gemm(A, B))
    assert B.gemm_bt is None and B.gemm_uses == 1        # (a 256-row product: no temporary copy either)
    second = be.to_host(be.gemm(A, B))
    assert B.gemm_bt is not None and B.gemm_bt.shape == (n, n)
    assert np.array_equal(be.to_host(B.gemm_bt), Bh.T)
    third = be.to_host(be.gemm(A, B, alpha=1.0))
    assert np.array_equal(first, second) and np.array_equal(first, third)
    ref = be.to_host(A).astype(np.float64) @ Bh.astype(np.float64)
    np.testing.assert_allclose(first, ref, rtol=1e-3 if dtype == np.float32 else 1e-12, atol=1e-2 if dtype == np.float32 else 1e-10)
    # small operands and transposed ones are left alone
    small = be.to_device(rng.standard_normal((n, 64)).astype(dtype))
    be.gemm(A, small), be.gemm(A, small)
    assert small.gemm_bt is None
    # the tile is rewritten (here: as the output of a product): the stale copy goes
    be.gemm(be.to_device(np.eye(n, dtype=dtype)), B.gemm_bt, False, True, out=B)
    assert B.gemm_bt is None and B.gemm_uses == 0
