"""End-to-end parity on the GPU: alg_wrappers.{cholesky,gemm,tsqr,bdfac} -> compiled program ->
HIP-stream executor -> HIP tile kernels, compared with the whole-algorithm outputs recorded from the
reference (tests/golden/algos.npz), with the oracle at other sizes, and through size-independent
residual properties at larger sizes.  The call sequences mirror the reference's
tests/test_alg_correctness.py."""
import os

import numpy as np
import pytest

import npw_oracle as oracle
from conftest import GOLDEN
from numpywren_amd import alg_wrappers, job_runner
from numpywren_amd import lambdapack as lp
from numpywren_amd.device import DeviceTile, get_backend
from numpywren_amd.matrix import BigMatrix
from numpywren_amd.matrix_init import shard_matrix

pytestmark = pytest.mark.gpu
ALG = np.load(os.path.join(GOLDEN, "algos.npz"))


def run(program, **kw):
    program.start()
    res = job_runner.lambdapack_run(program, timeout=300, idle_timeout=6, **kw)
    program.wait()
    program.free()
    return res


@pytest.mark.parametrize("tag", ["32_8", "20_8", "24_8_lam", "40_8_t2"])
def test_cholesky_golden(tag, hbm_store):
    A, L = ALG[f"cholesky_{tag}/A"], ALG[f"cholesky_{tag}/L"]
    n, b, lam, trunc, ntasks = ALG[f"cholesky_{tag}/meta"]
    X = BigMatrix(f"chol_in_{tag}", shape=A.shape, shard_sizes=(int(b), int(b)), write_header=True, lambdav=float(lam))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X, truncate=int(trunc))
    res = run(program)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    assert len(res["executed_messages"]) == int(ntasks)
    np.testing.assert_allclose(meta["outputs"][0].numpy(), L, rtol=1e-10, atol=1e-11)


def test_cholesky_like_reference_test(hbm_store):
    """reference tests/test_alg_correctness.py:31-50 (64 x 64, tile 8, allclose to np.linalg.cholesky)."""
    rng = np.random.default_rng(0)
    Xr = rng.standard_normal((64, 64))
    A = Xr.dot(Xr.T) + np.eye(64)
    A_sharded = BigMatrix("cholesky_test_A", shape=A.shape, shard_sizes=(8, 8), write_header=True)
    A_sharded.free()
    shard_matrix(A_sharded, A)
    program, meta = alg_wrappers.cholesky(A_sharded)
    run(program)
    assert np.allclose(meta["outputs"][0].numpy(), np.linalg.cholesky(A))


@pytest.mark.parametrize("n,b,width", [(512, 128, 4), (768, 256, 2), (1000, 300, 3)])
def test_cholesky_vs_oracle_and_residual(n, b, width, hbm_store):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, 64))
    A = G @ G.T + n * np.eye(n)
    X = BigMatrix(f"chol_{n}", shape=A.shape, shard_sizes=(b, b))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    run(program, pipeline_width=width)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    L = meta["outputs"][0].numpy()
    ref = oracle.cholesky(A, b)
    np.testing.assert_allclose(L, ref, rtol=1e-10, atol=1e-10 * np.abs(ref).max())
    assert np.linalg.norm(A - L @ L.T) / np.linalg.norm(A) < 1e-13
    # every tile the executor produced stayed in HBM
    tiles = meta["outputs"][0]._tiles(False)
    assert tiles and all(isinstance(t, DeviceTile) for t in tiles.values())


def test_cholesky_reference_generator_with_lambdav(hbm_store):
    """The experiment's input (reference experiments/cholesky_experiment.py:78-92): x x^T with a huge
    lambdav applied on every read of a diagonal tile."""
    n, b = 512, 128
    np.random.seed(0)
    x = np.random.randn(n, 1)
    A = x @ x.T
    X = BigMatrix("sosp_like", shape=A.shape, shard_sizes=(b, b), lambdav=n * 20e12)
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    run(program)
    assert program.program_status() == lp.PS.SUCCESS
    L = meta["outputs"][0].numpy()
    Afull = A + n * 20e12 * np.eye(n)
    assert np.allclose(L, np.linalg.cholesky(Afull))
    assert np.linalg.norm(Afull - L @ L.T) / np.linalg.norm(Afull) < 1e-14


def test_cholesky_not_positive_definite(hbm_store):
    A = np.eye(256)
    A[200, 200] = -3.0
    X = BigMatrix("chol_bad", shape=A.shape, shard_sizes=(64, 64))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    run(program)
    assert program.program_status() == lp.PS.EXCEPTION
    assert any("positive definite" in str(v) for v in program.exceptions.values())


def test_cholesky_reclaim_and_no_exact_zero(hbm_store):
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("chol_opts", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["reclaim_intermediates"] = True
    program.config["executor"]["exact_zero_shortcircuit"] = False
    run(program)
    np.testing.assert_allclose(meta["outputs"][0].numpy(), ALG["cholesky_32_8/L"], rtol=1e-10, atol=1e-11)
    assert meta["intermediates"][0].block_idxs_exist == []


@pytest.mark.parametrize("tag,b", [("32_8", 8), ("40_8", 8), ("16_8_f32", 8)])
def test_gemm_golden(tag, b, hbm_store):
    A, B, C = ALG[f"gemm_{tag}/A"], ALG[f"gemm_{tag}/B"], ALG[f"gemm_{tag}/C"]
    Ab = BigMatrix(f"gemm_A_{tag}", shape=A.shape, shard_sizes=(b, b), dtype=A.dtype)
    Bb = BigMatrix(f"gemm_B_{tag}", shape=B.shape, shard_sizes=(b, b), dtype=B.dtype)
    shard_matrix(Ab, A)
    shard_matrix(Bb, B)
    program, meta = alg_wrappers.gemm(Ab, Bb)
    run(program, pipeline_width=3)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    got = meta["outputs"][0].numpy()
    assert got.dtype == C.dtype
    tol = 1e-4 if A.dtype == np.float32 else 1e-11
    np.testing.assert_allclose(got, C, rtol=tol, atol=tol)


def test_gemm_like_reference_test(hbm_store):
    """reference tests/test_alg_correctness.py:137-156 (64 x 64, tile 16) + a larger fp32 instance."""
    rng = np.random.default_rng(1)
    A, B = rng.standard_normal((64, 64)), rng.standard_normal((64, 64))
    As = BigMatrix("Gemm_test_A", shape=A.shape, shard_sizes=(16, 16), write_header=True)
    Bs = BigMatrix("Gemm_test_B", shape=A.shape, shard_sizes=(16, 16), write_header=True)
    shard_matrix(As, A)
    shard_matrix(Bs, B)
    program, meta = alg_wrappers.gemm(As, Bs)
    run(program, pipeline_width=3)
    assert np.allclose(meta["outputs"][0].numpy(), A.dot(B))
    n, b = 1024, 256
    A32 = rng.standard_normal((n, n)).astype(np.float32)
    B32 = rng.standard_normal((n, n)).astype(np.float32)
    As = BigMatrix("g32A", shape=A32.shape, shard_sizes=(b, b), dtype=np.float32)
    Bs = BigMatrix("g32B", shape=B32.shape, shard_sizes=(b, b), dtype=np.float32)
    shard_matrix(As, A32)
    shard_matrix(Bs, B32)
    program, meta = alg_wrappers.gemm(As, Bs)
    run(program)
    ref = A32.astype(np.float64) @ B32.astype(np.float64)
    got = meta["outputs"][0].numpy()
    assert got.dtype == np.float64                                   # add_matrices promotes (reference quirk)
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-2 * np.sqrt(n))


@pytest.mark.parametrize("tag,b", [("64_8", 8), ("32_16", 16)])
def test_tsqr_golden(tag, b, hbm_store):
    Xh = ALG[f"tsqr_{tag}/X"]
    X = BigMatrix(f"tsqr_in_{tag}", shape=Xh.shape, shard_sizes=(b, Xh.shape[1]))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.tsqr(X)
    run(program)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    R, V, T = meta["outputs"]
    levels = int(np.log2(Xh.shape[0] // b))
    np.testing.assert_allclose(R.get_block(levels, 0), ALG[f"tsqr_{tag}/R_final"], atol=1e-11)
    np.testing.assert_allclose(R.get_block(0, 0), ALG[f"tsqr_{tag}/R_leaf0"], atol=1e-11)
    np.testing.assert_allclose(V.get_block(0, 0), ALG[f"tsqr_{tag}/V_leaf0"], atol=1e-11)
    np.testing.assert_allclose(T.get_block(0, 0), ALG[f"tsqr_{tag}/T_leaf0"], atol=1e-11)
    np.testing.assert_allclose(V.get_block(levels, 0), ALG[f"tsqr_{tag}/V_top"], atol=1e-11)
    np.testing.assert_allclose(T.get_block(levels, 0), ALG[f"tsqr_{tag}/T_top"], atol=1e-11)


def test_tsqr_like_reference_test(hbm_store):
    """reference tests/test_alg_correctness.py:72-102: 256 x 32, tile 32, R equals numpy's up to row signs;
    plus R^T R = A^T A on a taller instance."""
    np.random.seed(1)
    X = np.random.randn(256, 32)
    Xs = BigMatrix("tsqr_test_X", shape=X.shape, shard_sizes=(32, 32), write_header=True)
    shard_matrix(Xs, X)
    program, meta = alg_wrappers.tsqr(Xs)
    run(program)
    R_npw = meta["outputs"][0].get_block(3, 0)
    R = np.linalg.qr(X)[1]
    R_npw = R_npw * np.where(np.diag(R_npw) <= 0, -1, 1)[:, None]
    R = R * np.where(np.diag(R) <= 0, -1, 1)[:, None]
    assert np.allclose(R_npw, R)
    rng = np.random.default_rng(4)
    X = rng.standard_normal((16 * 128, 128))
    Xs = BigMatrix("tsqr_tall", shape=X.shape, shard_sizes=(128, 128))
    shard_matrix(Xs, X)
    program, meta = alg_wrappers.tsqr(Xs)
    run(program)
    Rf = meta["outputs"][0].get_block(4, 0)
    G = X.T @ X
    assert np.linalg.norm(Rf.T @ Rf - G) / np.linalg.norm(G) < 1e-12


def test_bdfac_golden(hbm_store):
    """As-written parity.  The reference's committed qr_leaf (S0 - V^T S0) makes the intermediate
    matrices rank deficient, so some Householder pivots are ~0 and their reflection SIGN is decided by
    rounding noise (LAPACK vs MFMA summation order): tiles are compared in magnitude, which is also how the
    reference's own test compares its R block (tests/test_alg_correctness.py:272)."""
    Xh = ALG["bdfac_16_4/X"]
    X = BigMatrix("bdfac_in", shape=Xh.shape, shard_sizes=(4, 4))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.bdfac(X)
    res = run(program, pipeline_width=1)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    assert len(res["executed_messages"]) == int(ALG["bdfac_16_4/ntasks"])
    L, R = meta["outputs"]

    def signs_factor(got, ref):
        # stricter than |got| == |ref|: the sign of a reflection flips a whole ROW (a reflection from the left: the QR
        # sweeps) or a whole COLUMN (from the right: the LQ sweeps) of a tile, nothing else, so the sign pattern
        # sign(got * ref) must be an outer product r c^T; a sign error inside a trailing update breaks that.  Entries
        # that are ~0 in the reference carry no sign.
        sig = np.abs(ref) > 1e-7
        M = np.sign(got * ref) * sig
        rows, cols = M.shape
        for i in range(rows):
            for k in range(i + 1, rows):
                both = sig[i] & sig[k]
                prod = M[i, both] * M[k, both]           # = r_i r_k for every shared column
                assert prod.size == 0 or np.all(prod == prod[0]), (got, ref)

    for name in ("R_0_2_0", "R_1_2_1", "R_2_1_2", "R_3_0_3"):
        got = R.get_block(*[int(x) for x in name.split("_")[1:]])
        np.testing.assert_allclose(np.abs(got), np.abs(ALG[f"bdfac_16_4/{name}"]), atol=1e-9)
        signs_factor(got, ALG[f"bdfac_16_4/{name}"])
    for name in ("L_0_2_1", "L_1_1_2", "L_2_0_3"):
        got = L.get_block(*[int(x) for x in name.split("_")[1:]])
        np.testing.assert_allclose(np.abs(got), np.abs(ALG[f"bdfac_16_4/{name}"]), atol=1e-9)
        signs_factor(got, ALG[f"bdfac_16_4/{name}"])


@pytest.mark.parametrize("n,b", [(16, 4), (128, 32), (192, 64)])
def test_bdfac_well_posed(n, b, hbm_store):
    """BDFAC with the WY-form leaf update (the line the reference has commented out): every pivot is
    well conditioned, so tiles must match the oracle running the same program strictly, and the assembled
    block-bidiagonal factor keeps X's singular values (the reference test's intended invariant)."""
    from numpywren_amd import compiler, kernels
    from numpywren_amd.algs import BDFAC
    rng = np.random.default_rng(n)
    Xh = rng.standard_normal((n, n))
    X = BigMatrix(f"bdfac_wp_{n}", shape=Xh.shape, shard_sizes=(b, b))
    shard_matrix(X, Xh)
    _, meta = alg_wrappers.bdfac(X)   # builds the eight work matrices with the reference's names / shapes
    L_LQ, R_QR = meta["outputs"]
    S_LQ, S_QR, T_QR, V_QR, V_LQ, T_LQ = meta["intermediates"]
    nb = X.num_blocks(0)
    p = compiler.lpcompile_for_execution(BDFAC, ["I"], ["R_QR", "L_LQ"], kernels={"qr_leaf": kernels.qr_leaf_wy})(
        X, V_QR, T_QR, S_QR, R_QR, V_LQ, T_LQ, S_LQ, L_LQ, nb, 0)
    program = lp.LambdaPackProgram(p, config={})
    run(program)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    fixed = dict(oracle.KERNELS)
    fixed["qr_leaf"] = lambda V, T, S0, *a, **k: S0 - V @ T.T @ (V.T @ S0)
    R_ref, L_ref = oracle.bdfac(Xh, b, kernels=fixed)
    levels = lambda m: int(np.ceil(np.log2(m))) if m > 1 else 0
    fac = np.zeros((n, n))
    for i in range(nb):
        idx = (i, levels(nb - i), i)
        got = R_QR.get_block(*idx)
        np.testing.assert_allclose(got, R_ref.get(idx), atol=1e-9)
        fac[i * b:(i + 1) * b, i * b:(i + 1) * b] = got
        if i + 1 < nb:
            idx = (i, levels(nb - i - 1), i + 1)
            got = L_LQ.get_block(*idx)
            np.testing.assert_allclose(got, L_ref.get(idx), atol=1e-9)
            fac[i * b:(i + 1) * b, (i + 1) * b:(i + 2) * b] = got
    np.testing.assert_allclose(np.linalg.svd(fac, compute_uv=False), np.linalg.svd(Xh, compute_uv=False), atol=1e-9)


def test_bigmatrix_hbm_roundtrips(hbm_store):
    rng = np.random.default_rng(8)
    X = rng.standard_normal((200, 200))
    m = BigMatrix("hbm_rt", shape=X.shape, shard_sizes=(101, 101), write_header=True)
    shard_matrix(m, X)
    assert all(isinstance(t, DeviceTile) for t in m._tiles(False).values())
    assert np.array_equal(m.numpy(), X)
    assert np.array_equal(m.T.numpy(), X.T)
    assert np.array_equal(BigMatrix("hbm_rt").numpy(), X)
    t = m.get_tile(1, 0)
    assert t.shape == (99, 101)
    assert np.array_equal(get_backend().to_host(m.T.get_tile(0, 1)), X[101:, :101].T)
    lam = BigMatrix("hbm_lam", shape=(64, 64), shard_sizes=(32, 32), lambdav=7.0)
    shard_matrix(lam, X[:64, :64])
    assert np.allclose(get_backend().to_host(lam.get_tile(1, 1)), X[32:64, 32:64] + 7 * np.eye(32))
    assert np.array_equal(get_backend().to_host(lam.get_tile(0, 1)), X[:32, 32:64])
    m.free()
    assert m.block_idxs_exist == []


QRG = np.load(os.path.join(GOLDEN, "qr.npz"))


@pytest.mark.parametrize("tag,b", [("28_7", 7), ("16_8", 8), ("24_8", 8), ("80_40", 40)])
def test_qr_golden(tag, b, hbm_store):
    """alg_wrappers.qr on the GPU against the R tiles of the reference's own run (as written: only the first
    block row is a true QR factor, see kernels.qr_factor_triangular)."""
    Xh = QRG[f"qr_{tag}/X"]
    X = BigMatrix(f"QR_input_{tag}", shape=Xh.shape, shard_sizes=(b, b))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.qr(X)
    res = run(program, pipeline_width=1)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    assert len(res["executed_messages"]) == int(QRG[f"qr_{tag}/meta"][4])
    Rs = meta["outputs"][0]
    nb = Xh.shape[0] // b
    for i in range(nb):
        for k in range(i, nb):
            np.testing.assert_allclose(Rs.get_block(i, k, 0), QRG[f"qr_{tag}/R_{i}_{k}"], rtol=1e-8, atol=1e-8,
                                       err_msg=f"R[{i},{k}]")
    R = np.linalg.qr(Xh)[1]
    np.testing.assert_allclose(np.abs(Rs.get_block(0, 0, 0)), np.abs(R[:b, :b]), atol=1e-10)


def test_two_factorisations_in_flight(hbm_store):
    """enqueue a second program before waiting for the first (job_runner.lambdapack_run(wait=False))"""
    rng = np.random.default_rng(17)
    progs = []
    for t in range(2):
        G = rng.standard_normal((512, 512))
        A = G @ G.T + 512 * np.eye(512)
        X = BigMatrix(f"inflight_{t}", shape=A.shape, shard_sizes=(128, 128))
        shard_matrix(X, A)
        program, meta = alg_wrappers.cholesky(X)
        program.start()
        # the second run is ordered behind the first on the device (after=), both are enqueued before either is settled
        marks = progs[-1][0].completion_marks if progs else None
        job_runner.lambdapack_run(program, wait=False, after=marks, pipeline_width=1 + t)
        assert program.completion_marks
        progs.append((program, meta, A))
    for program, meta, A in progs:
        program.wait()
        assert program.program_status() == lp.PS.SUCCESS
        L = np.tril(meta["outputs"][0].numpy())
        np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)


def test_tsqr_batched_tasks_match_one_by_one(hbm_store):
    """The executor hands ready qr_factor tasks to the GPU in batches (executor.batch_tasks); R, V, T equal the
    one-task-at-a-time run to rounding and the factor satisfies R^T R = X^T X."""
    rng = np.random.default_rng(77)
    b, leaves = 96, 16
    Xh = rng.standard_normal((b * leaves, b))
    res = {}
    for width in (8, 1):
        X = BigMatrix(f"tsqr_gpu_batch_{width}", shape=Xh.shape, shard_sizes=(b, b))
        for j in range(leaves):
            X.put_block(Xh[j * b:(j + 1) * b], j, 0)
        program, meta = alg_wrappers.tsqr(X)
        program.config["executor"]["batch_tasks"] = width
        program.start()
        out = job_runner.lambdapack_run(program)
        program.wait()
        assert program.program_status() == lp.PS.SUCCESS
        assert len(out["executed_messages"]) == 2 * leaves - 1
        R, V, T = meta["outputs"]
        res[width] = (R.get_block(4, 0), V.get_block(4, 0), T.get_block(2, 4), V.get_block(0, 5))
        program.free()
    for a, c in zip(res[8], res[1]):
        np.testing.assert_allclose(a, c, atol=1e-12, rtol=0)
    R = res[8][0]
    np.testing.assert_allclose(R.T @ R, Xh.T @ Xh, atol=1e-10 * np.linalg.norm(Xh) ** 2)


def test_roctx_ranges_option_runs(hbm_store):
    """executor.roctx_ranges: every task's enqueue is bracketed with a named profiler range (npw_range_push / _pop); the run is
    the same run (rocprofv3 --marker-trace output of such a run: profiles/r05_roctx_ranges.txt)."""
    rng = np.random.default_rng(31)
    n, b = 1024, 256
    G = rng.standard_normal((n, n))
    A = G @ G.T + n * np.eye(n)
    X = BigMatrix("roctx_chol", shape=A.shape, shard_sizes=(b, b), write_header=True)
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["roctx_ranges"] = True
    program.start()
    job_runner.lambdapack_run(program, timeout=120)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    np.testing.assert_allclose(np.tril(meta["outputs"][0].numpy()), np.linalg.cholesky(A), rtol=1e-10, atol=1e-9)
