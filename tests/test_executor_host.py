"""The host side of the executor (alg_wrappers -> compiler -> LambdaPackProgram -> job_runner ->
kernels wrappers -> BigMatrix tile paths) on CPU, with the NumPy/oracle CHECKER backend injected
under it (tests/oracle_backend.py).  Results are compared with the whole-algorithm golden outputs
recorded from the reference (tests/golden/algos.npz).  The same scenarios run on the real HIP backend
in test_algorithms_gpu.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from numpywren_amd import alg_wrappers, job_runner
from numpywren_amd import lambdapack as lp
from numpywren_amd.matrix import BigMatrix
from numpywren_amd.matrix_init import shard_matrix

ALG = np.load(os.path.join(GOLDEN, "algos.npz"))


def run(program, **kw):
    program.start()
    res = job_runner.lambdapack_run(program, timeout=60, idle_timeout=6, **kw)
    program.wait()
    program.free()
    return res


@pytest.mark.parametrize("tag", ["32_8", "20_8", "24_8_lam", "40_8_t2"])
def test_cholesky(tag, oracle_backend):
    A, L = ALG[f"cholesky_{tag}/A"], ALG[f"cholesky_{tag}/L"]
    n, b, lam, trunc, ntasks = ALG[f"cholesky_{tag}/meta"]
    X = BigMatrix(f"chol_in_{tag}", shape=A.shape, shard_sizes=(int(b), int(b)), write_header=True, lambdav=float(lam))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X, truncate=int(trunc))
    assert meta["outputs"][0].key == f"Cholesky(chol_in_{tag})"
    assert meta["intermediates"][0].key == f"Cholesky.Intermediate(chol_in_{tag})"
    res = run(program)
    assert program.program_status() == lp.PS.SUCCESS
    assert set(res) == {"up_time", "exec_time", "executed_messages", "operator_refs", "log"}
    assert len(res["executed_messages"]) == int(ntasks)
    np.testing.assert_allclose(meta["outputs"][0].numpy(), L, rtol=1e-12, atol=1e-12)
    assert program.get_progress() == int(ntasks)
    assert program.get_flops() > 0 and program.get_read() > 0 and program.get_write() > 0
    assert program.get_up() == 0


def test_cholesky_streams_and_priority(oracle_backend):
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("chol_streams", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["priority_stream"] = True
    run(program, pipeline_width=3)
    prio = {c[1] for c in oracle_backend.calls if c[0] in ("chol", "trsm")}
    bulk = {c[1] for c in oracle_backend.calls if c[0] == "syrk"}
    assert prio == {oracle_backend.priority_stream}            # panel kernels on the high-priority stream
    assert len(bulk) == 3 and oracle_backend.priority_stream not in bulk


def test_chol_gets_the_device_to_itself_with_several_streams(oracle_backend):
    """the Cholesky panel chain needs whole CUs: its stream waits for the other streams' tails and they wait for it"""
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("chol_excl", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    run(program, pipeline_width=3)
    calls = oracle_backend.calls
    for i, c in enumerate(calls):
        if c[0] != "chol":
            continue
        s = c[1]
        before = [x for x in calls[max(0, i - 8):i] if x[0] == "wait_event" and x[1] == s]
        after = [x for x in calls[i + 1:i + 8] if x[0] == "wait_event" and x[2] == ("event", s)]
        assert len(before) >= 2 and len(after) == 2, (before, after)


def test_cholesky_not_positive_definite(oracle_backend):
    A = -np.eye(16)
    X = BigMatrix("chol_bad", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    run(program)
    assert program.program_status() == lp.PS.EXCEPTION
    assert any("positive definite" in str(v) for v in program.exceptions.values())


def test_reclaim_intermediates(oracle_backend):
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("chol_reclaim", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["reclaim_intermediates"] = True
    run(program)
    np.testing.assert_allclose(meta["outputs"][0].numpy(), ALG["cholesky_32_8/L"], rtol=1e-12, atol=1e-12)
    assert meta["intermediates"][0].block_idxs_exist == []       # every S version died after its last reader
    assert len(X.block_idxs_exist) == 16                           # inputs are never reclaimed


@pytest.mark.parametrize("r_only", [False, True])
def test_reclaim_keeps_v_t_unless_r_only(r_only, oracle_backend):
    """TSQR: V and T of every node are written and never read (only R goes up the tree).  The reference's wrapper
    returns them (alg_wrappers.py:47), so `reclaim_intermediates` alone keeps them; with `drop_unread_outputs` (an
    explicit R-only run) they are dropped as soon as they are stored.  Either way the R factors below the root die
    after their parent and the result stays."""
    Xh = ALG["tsqr_64_8/X"]
    X = BigMatrix("tsqr_reclaim", shape=Xh.shape, shard_sizes=(8, Xh.shape[1]))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.tsqr(X)
    program.config["executor"]["reclaim_intermediates"] = True
    program.config["executor"]["drop_unread_outputs"] = r_only
    run(program)
    assert program.program_status() == lp.PS.SUCCESS
    R, V, T = meta["outputs"]
    if r_only:
        assert V.block_idxs_exist == [] and T.block_idxs_exist == []
    else:
        nodes = [(0, j) for j in range(8)] + [(1, 0), (1, 2), (1, 4), (1, 6), (2, 0), (2, 4), (3, 0)]
        assert all(V.tile_exists(*n) and T.tile_exists(*n) for n in nodes)          # 8 leaves + 7 tree nodes
    np.testing.assert_allclose(np.abs(R.get_block(3, 0)), np.abs(ALG["tsqr_64_8/R_final"]), atol=1e-12)
    assert len(X.block_idxs_exist) == 8                            # inputs are never reclaimed


@pytest.mark.parametrize("tag,b", [("32_8", 8), ("40_8", 8), ("16_8_f32", 8)])
def test_gemm(tag, b, oracle_backend):
    A, B, C = ALG[f"gemm_{tag}/A"], ALG[f"gemm_{tag}/B"], ALG[f"gemm_{tag}/C"]
    Ab = BigMatrix(f"gemm_A_{tag}", shape=A.shape, shard_sizes=(b, b), dtype=A.dtype)
    Bb = BigMatrix(f"gemm_B_{tag}", shape=B.shape, shard_sizes=(b, b), dtype=B.dtype)
    shard_matrix(Ab, A)
    shard_matrix(Bb, B)
    program, meta = alg_wrappers.gemm(Ab, Bb)
    run(program, pipeline_width=3)
    assert program.program_status() == lp.PS.SUCCESS
    got = meta["outputs"][0].numpy()
    tol = 1e-5 if A.dtype == np.float32 else 1e-12
    np.testing.assert_allclose(got, C, rtol=tol, atol=tol)


@pytest.mark.parametrize("tag,b", [("64_8", 8), ("32_16", 16)])
def test_tsqr(tag, b, oracle_backend):
    Xh = ALG[f"tsqr_{tag}/X"]
    X = BigMatrix(f"tsqr_in_{tag}", shape=Xh.shape, shard_sizes=(b, Xh.shape[1]))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.tsqr(X)
    run(program)
    assert program.program_status() == lp.PS.SUCCESS
    R, V, T = meta["outputs"]
    levels = int(np.log2(Xh.shape[0] // b))
    np.testing.assert_allclose(R.get_block(levels, 0), ALG[f"tsqr_{tag}/R_final"], atol=1e-12)
    np.testing.assert_allclose(V.get_block(levels, 0), ALG[f"tsqr_{tag}/V_top"], atol=1e-12)
    np.testing.assert_allclose(T.get_block(0, 0), ALG[f"tsqr_{tag}/T_leaf0"], atol=1e-12)


def test_tsqr_leaves_and_tree_levels_run_as_batches(oracle_backend):
    """Ready qr_factor tasks are grouped (config executor.batch_tasks, default 16) into one batched call; the task
    set, the order constraints and the results are those of the one-by-one run."""
    Xh = ALG["tsqr_64_8/X"]
    outs = {}
    for width in (8, 3, 1):
        matrix_key = f"tsqr_batch_{width}"
        X = BigMatrix(matrix_key, shape=Xh.shape, shard_sizes=(8, Xh.shape[1]))
        shard_matrix(X, Xh)
        program, meta = alg_wrappers.tsqr(X)
        program.config["executor"]["batch_tasks"] = width
        del oracle_backend.calls[:]
        res = run(program)
        assert program.program_status() == lp.PS.SUCCESS
        assert len(res["executed_messages"]) == 15            # 8 leaves + 4 + 2 + 1
        sizes = [c[1] for c in oracle_backend.calls if c[0] == "geqrt_batched"]
        tree = [c[1] for c in oracle_backend.calls if c[0] == "tpqrt_batched"]
        singles = sum(1 for c in oracle_backend.calls if c[0] == "geqrt")
        assert sum(sizes) + sum(tree) + singles == 15
        # the leaves are dense blocks; the tree nodes stack two R factors -> the structured factorisation
        if width == 8:
            assert sizes == [8] and tree == [4, 2, 1] and singles == 0
        elif width == 3:
            assert max(sizes) == 3 and max(tree) <= 3 and sum(tree) == 7
        else:
            assert sizes == [] and tree == [1] * 7 and singles == 8
        R, V, T = meta["outputs"]
        outs[width] = (R.get_block(3, 0), V.get_block(3, 0), T.get_block(0, 0))
        np.testing.assert_allclose(outs[width][0], ALG["tsqr_64_8/R_final"], atol=1e-12)
    for w in (3, 1):
        for a, b in zip(outs[8], outs[w]):
            assert np.array_equal(a, b)


def test_dequeue_matching_takes_best_priority_first(oracle_backend):
    Xh = ALG["tsqr_64_8/X"]
    X = BigMatrix("tsqr_dq", shape=Xh.shape, shard_sizes=(8, Xh.shape[1]))
    shard_matrix(X, Xh)
    program, _ = alg_wrappers.tsqr(X)
    program.start()
    assert program.num_ready() == 8
    first = program.dequeue()
    more = program.dequeue_matching(lambda e, v: e == first[0], 3)
    assert len(more) == 3 and program.num_ready() == 4
    assert program.dequeue_matching(lambda e, v: False, 5) == [] and program.num_ready() == 4
    assert program.dequeue_matching(lambda e, v: True, 0) == []
    rest = program.dequeue_matching(lambda e, v: True, 100)
    assert len(rest) == 4 and program.num_ready() == 0 and program.dequeue() is None


def test_bdfac(oracle_backend):
    Xh = ALG["bdfac_16_4/X"]
    X = BigMatrix("bdfac_in", shape=Xh.shape, shard_sizes=(4, 4))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.bdfac(X)
    res = run(program, pipeline_width=1)
    assert program.program_status() == lp.PS.SUCCESS
    assert len(res["executed_messages"]) == int(ALG["bdfac_16_4/ntasks"])
    L, R = meta["outputs"]
    for name in ("R_0_2_0", "R_1_2_1", "R_2_1_2", "R_3_0_3"):
        np.testing.assert_allclose(R.get_block(*[int(x) for x in name.split("_")[1:]]), ALG[f"bdfac_16_4/{name}"], atol=1e-11)
    for name in ("L_0_2_1", "L_1_1_2", "L_2_0_3"):
        np.testing.assert_allclose(L.get_block(*[int(x) for x in name.split("_")[1:]]), ALG[f"bdfac_16_4/{name}"], atol=1e-11)


def test_user_python_kernel(oracle_backend):
    """Arbitrary callables from the DSL function's scope receive ndarrays, like in the reference."""
    from numpywren_amd import compiler

    def double_it(x):
        return 2 * x

    def prog(A: BigMatrix, B: BigMatrix, N: int):
        for i in range(N):
            B[i, 0] = double_it(A[i, 0])

    Xh = np.arange(32.0).reshape(16, 2)
    A = BigMatrix("uk_A", shape=Xh.shape, shard_sizes=(4, 2))
    B = BigMatrix("uk_B", shape=Xh.shape, shard_sizes=(4, 2))
    shard_matrix(A, Xh)
    p = compiler.lpcompile_for_execution(prog, ["A"], ["B"], kernels={"double_it": double_it})(A, B, 4)
    program = lp.LambdaPackProgram(p, config={})
    run(program)
    assert program.program_status() == lp.PS.SUCCESS
    np.testing.assert_array_equal(B.numpy(), 2 * Xh)


def test_duplicate_and_replayed_tasks_do_not_double_count(oracle_backend):
    """reference tests/test_job_runner.py:120-190: duplicate messages must not corrupt the run."""
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("chol_dup", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    program._enqueue((0, {}))          # a second copy of the starter
    ex = job_runner.LambdaPackExecutor(program)
    import asyncio
    loop = asyncio.new_event_loop()
    loop.run_until_complete(ex.run(0, {}))
    loop.run_until_complete(ex.run(0, {}))   # replay of a finished node: skipped
    loop.close()
    program.post_op(0, {}, lp.PS.SUCCESS, None)  # replayed post_op: edges are sets, nothing double counts
    job_runner.lambdapack_run(program, timeout=60)
    assert program.program_status() == lp.PS.SUCCESS
    np.testing.assert_allclose(meta["outputs"][0].numpy(), ALG["cholesky_32_8/L"], rtol=1e-12, atol=1e-12)


QRG = np.load(os.path.join(GOLDEN, "qr.npz"))


@pytest.mark.parametrize("tag,b", [("28_7", 7), ("16_8", 8), ("24_8", 8), ("80_40", 40)])
def test_qr(tag, b, oracle_backend):
    """alg_wrappers.qr (reference alg_wrappers.py:67-89) end to end; R tiles against the reference's own run."""
    Xh = QRG[f"qr_{tag}/X"]
    X = BigMatrix(f"QR_input_{tag}", shape=Xh.shape, shard_sizes=(b, b))
    shard_matrix(X, Xh)
    program, meta = alg_wrappers.qr(X)
    res = run(program, pipeline_width=1)
    assert program.program_status() == lp.PS.SUCCESS
    assert len(res["executed_messages"]) == int(QRG[f"qr_{tag}/meta"][4])
    Rs = meta["outputs"][0]
    nb = Xh.shape[0] // b
    for i in range(nb):
        for k in range(i, nb):
            np.testing.assert_allclose(Rs.get_block(i, k, 0), QRG[f"qr_{tag}/R_{i}_{k}"], rtol=1e-9, atol=1e-9,
                                       err_msg=f"R[{i},{k}]")


@pytest.mark.parametrize("tag,b", [("32_8", 8), ("40_8", 8), ("16_8_f32", 8)])
def test_gemm_fused_reduction(tag, b, oracle_backend):
    """executor.fuse_gemm_reduction: the K partial products of a C tile accumulate in one buffer, no Temp tile is ever
    stored, the add_matrices tree only hands the buffer on; same result (and output dtype) as the parity mode."""
    from numpywren_amd.job_runner import ReductionFusion
    A, B, C = ALG[f"gemm_{tag}/A"], ALG[f"gemm_{tag}/B"], ALG[f"gemm_{tag}/C"]
    Ab = BigMatrix(f"gemmf_A_{tag}", shape=A.shape, shard_sizes=(b, b), dtype=A.dtype)
    Bb = BigMatrix(f"gemmf_B_{tag}", shape=B.shape, shard_sizes=(b, b), dtype=B.dtype)
    shard_matrix(Ab, A)
    shard_matrix(Bb, B)
    program, meta = alg_wrappers.gemm(Ab, Bb)
    program.config["executor"]["fuse_gemm_reduction"] = True
    fusion = ReductionFusion(program.program)
    nb = A.shape[0] // b
    assert len(fusion.roots) == nb * nb and set(fusion.roots.values()) == {nb}      # one root per C tile, K products each
    run(program)
    assert program.program_status() == lp.PS.SUCCESS
    out, temp = meta["outputs"][0], meta["intermediates"][0]
    got = out.numpy()
    assert got.dtype == np.float64                                   # add_matrices' promotion survives the fusion
    tol = 1e-4 if A.dtype == np.float32 else 1e-12
    np.testing.assert_allclose(got, C, rtol=tol, atol=tol)
    # only the tree's final tile of every C tile was stored
    stored = sum(1 for i in range(nb) for j in range(nb) for k in range(nb) for l in range(4) if temp.tile_exists(i, j, k, l))
    assert stored == nb * nb
    gemms = sum(1 for c in oracle_backend.calls if c[0] == "gemm")
    assert gemms == nb ** 3 and not any(c[0] == "add_n" for c in oracle_backend.calls)


@pytest.mark.parametrize("nb", [1, 2, 3, 4, 5, 6])
def test_gemm_fused_reduction_every_tree_shape(nb, oracle_backend):
    """Square grids of 1 ... 6 blocks: tree depth 0 (K = 1: no add_matrices task at all), 1 (K <= 4) and 2 (K = 5, 6:
    a second level whose other operands are never-written constant_zeros tiles); fused == parity == A @ B."""
    from numpywren_amd.job_runner import ReductionFusion
    rng = np.random.default_rng(nb)
    b = 4
    A, B = rng.standard_normal((nb * b, nb * b)), rng.standard_normal((nb * b, nb * b))
    outs = []
    for fuse in (False, True):
        Ab = BigMatrix(f"gemmt_A_{nb}_{fuse}", shape=A.shape, shard_sizes=(b, b))
        Bb = BigMatrix(f"gemmt_B_{nb}_{fuse}", shape=B.shape, shard_sizes=(b, b))
        shard_matrix(Ab, A)
        shard_matrix(Bb, B)
        program, meta = alg_wrappers.gemm(Ab, Bb)
        program.config["executor"]["fuse_gemm_reduction"] = fuse
        if fuse:
            fusion = ReductionFusion(program.program)
            if nb == 1:
                assert fusion.roots == {}            # K = 1: the tree has no add_matrices task, nothing to fuse
            else:
                assert len(fusion.roots) == nb * nb and set(fusion.roots.values()) == {nb}
        run(program)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        outs.append(meta["outputs"][0].numpy())
    np.testing.assert_allclose(outs[0], A @ B, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(outs[1], outs[0], rtol=1e-13, atol=1e-13)


def test_fused_gemm_survives_a_timed_out_run(oracle_backend):
    """A fused run that stops on its time limit keeps its accumulators with the program; the resuming run finishes the sums."""
    A, B, C = ALG["gemm_32_8/A"], ALG["gemm_32_8/B"], ALG["gemm_32_8/C"]
    Ab = BigMatrix("gemmr_A", shape=A.shape, shard_sizes=(8, 8), dtype=A.dtype)
    Bb = BigMatrix("gemmr_B", shape=B.shape, shard_sizes=(8, 8), dtype=B.dtype)
    shard_matrix(Ab, A)
    shard_matrix(Bb, B)
    program, meta = alg_wrappers.gemm(Ab, Bb)
    program.config["executor"]["fuse_gemm_reduction"] = True
    program.start()
    real = job_runner.time.time
    ticks = iter([0.0] + [0.0] * 30 + [1e9] * 10000)       # the clock jumps after ~30 looks: a handful of products are done
    job_runner.time.time = lambda: next(ticks)
    try:
        res = job_runner.lambdapack_run(program, timeout=10.0)
    finally:
        job_runner.time.time = real
    done = len(res["executed_messages"])
    assert 0 < done < 64 and program.program_status() == lp.PS.RUNNING and program._fusion_acc
    job_runner.lambdapack_run(program)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS
    np.testing.assert_allclose(meta["outputs"][0].numpy(), C, rtol=1e-12, atol=1e-12)


def test_restart_after_an_aborted_fused_run_starts_from_clean_sums(oracle_backend):
    """the fused GEMM's accumulators live with the program; a run that is abandoned half-way and RESTARTED
    (free() + start(), what bench.py does between steps) must not add the new products to the stale partial sums."""
    A, B, C = ALG["gemm_32_8/A"], ALG["gemm_32_8/B"], ALG["gemm_32_8/C"]
    Ab = BigMatrix("gemmx_A", shape=A.shape, shard_sizes=(8, 8), dtype=A.dtype)
    Bb = BigMatrix("gemmx_B", shape=B.shape, shard_sizes=(8, 8), dtype=B.dtype)
    shard_matrix(Ab, A)
    shard_matrix(Bb, B)
    program, meta = alg_wrappers.gemm(Ab, Bb)
    program.config["executor"]["fuse_gemm_reduction"] = True
    program.start()
    real = job_runner.time.time
    ticks = iter([0.0] + [0.0] * 30 + [1e9] * 10000)
    job_runner.time.time = lambda: next(ticks)
    try:
        job_runner.lambdapack_run(program, timeout=10.0)
    finally:
        job_runner.time.time = real
    assert program.program_status() == lp.PS.RUNNING and program._fusion_acc      # stopped with partial sums in place
    program.stop()
    program.free()
    assert not program._fusion_acc                                                # ... which free() drops
    for m in meta["outputs"] + meta["intermediates"]:
        m.free()
    program.start()
    assert not program._fusion_acc and not program._was_up
    job_runner.lambdapack_run(program)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS
    np.testing.assert_allclose(meta["outputs"][0].numpy(), C, rtol=1e-12, atol=1e-12)


def test_expired_qr_handoffs_fail_the_program(oracle_backend):
    """The QR panel kernel bounds its waits for hand-off slots (a lost hand-off must not hang the GPU); libnpw_hip.so counts the
    waits that expired and the executor reads the count when a run with qr_factor tasks settles: non-zero fails the program
    loudly instead of returning undefined factors.  (Here the checker backend plays a device that lost three.)"""
    Xh = ALG["tsqr_64_8/X"]
    X = BigMatrix("tsqr_handoff", shape=Xh.shape, shard_sizes=(8, Xh.shape[1]))
    shard_matrix(X, Xh)
    asked = []
    oracle_backend.qr_handoff_timeouts = lambda reset=True: (asked.append(reset), 3)[1]
    try:
        program, meta = alg_wrappers.tsqr(X)
        program.start()
        job_runner.lambdapack_run(program)
        program.wait()
    finally:
        del oracle_backend.qr_handoff_timeouts
    assert asked == [True] and program.program_status() == lp.PS.EXCEPTION
    assert any("hand-off" in str(v) for v in program.exceptions.values())
    # a program without Householder factorisations never asks
    A = ALG["cholesky_32_8/A"]
    Xc = BigMatrix("handoff_chol", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(Xc, A)
    oracle_backend.qr_handoff_timeouts = lambda reset=True: (asked.append(reset), 3)[1]
    try:
        program, meta = alg_wrappers.cholesky(Xc)
        run(program)
    finally:
        del oracle_backend.qr_handoff_timeouts
    assert asked == [True] and program.program_status() == lp.PS.SUCCESS


def test_fusion_leaves_other_programs_alone(oracle_backend):
    """Nothing fuses in a program without the gemm -> add_matrices pattern; a Temp tile with a second reader would not
    fuse either (the DAG decides, not the program's name)."""
    from numpywren_amd.job_runner import ReductionFusion
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("fusion_chol", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    assert ReductionFusion(program.program).roots == {}
    program.config["executor"]["fuse_gemm_reduction"] = True
    run(program)
    np.testing.assert_allclose(meta["outputs"][0].numpy(), ALG["cholesky_32_8/L"], rtol=1e-12, atol=1e-12)


def test_wait_after_a_timed_out_run_reports_and_the_program_resumes(oracle_backend):
    """A run that leaves its loop on the time limit leaves the program RUNNING with no worker up: wait() raises the
    timeout instead of sleeping for ever, does NOT fail the program, and a second lambdapack_run finishes it.  A wait()
    that starts before any worker came up (the reference's start -> launch workers elsewhere -> wait) keeps waiting."""
    import threading
    from numpywren_amd.exceptions import LambdaPackTimeoutException
    A, L = ALG["cholesky_32_8/A"], ALG["cholesky_32_8/L"]
    X = BigMatrix("timeout_A", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    res = job_runner.lambdapack_run(program, timeout=-1.0)       # expires at the first task
    assert len(res["executed_messages"]) == 0 and program.program_status() == lp.PS.RUNNING
    with pytest.raises(LambdaPackTimeoutException):
        program.wait(sleep_time=0.01)
    assert program.program_status() == lp.PS.RUNNING             # still resumable
    job_runner.lambdapack_run(program)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS
    np.testing.assert_allclose(np.tril(meta["outputs"][0].numpy()), L, atol=1e-12)
    # wait() first, worker later (another thread): no false alarm
    X2 = BigMatrix("timeout_B", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X2, A)
    program2, meta2 = alg_wrappers.cholesky(X2)
    program2.start()
    th = threading.Timer(0.2, lambda: job_runner.lambdapack_run(program2))
    th.start()
    program2.wait(sleep_time=0.02)
    th.join()
    assert program2.program_status() == lp.PS.SUCCESS


def test_run_without_waiting_settles_in_program_wait(oracle_backend):
    """lambdapack_run(wait=False) only enqueues; program.wait() -- next in the reference's call sequence -- settles."""
    A, L = ALG["cholesky_32_8/A"], ALG["cholesky_32_8/L"]
    X = BigMatrix("nowait_A", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    res = job_runner.lambdapack_run(program, wait=False)
    assert len(res["executed_messages"]) == 20
    assert program.program_status() == lp.PS.RUNNING          # not settled yet
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS
    np.testing.assert_allclose(np.tril(meta["outputs"][0].numpy()), L, atol=1e-12)
    program.wait()                                             # idempotent
    bad = -np.eye(16)
    Y = BigMatrix("nowait_bad", shape=bad.shape, shard_sizes=(8, 8))
    shard_matrix(Y, bad)
    program, meta = alg_wrappers.cholesky(Y)
    program.start()
    job_runner.lambdapack_run(program, wait=False)
    program.wait()
    assert program.program_status() == lp.PS.EXCEPTION


def test_profiling_records(oracle_backend):
    """get_profiling_info / get_all_profiling_info (reference lambdapack.py:765-776 read a pickle per node from S3)."""
    Xh = ALG["tsqr_64_8/X"]
    X = BigMatrix("tsqr_prof", shape=Xh.shape, shard_sizes=(8, Xh.shape[1]))
    shard_matrix(X, Xh)
    program, _ = alg_wrappers.tsqr(X)
    program.config["executor"]["batch_tasks"] = 4
    res = run(program)
    infos = program.get_all_profiling_info()
    assert len(infos) == len(res["executed_messages"]) == 15
    e, v = res["executed_messages"][0]
    rec = program.get_profiling_info(e, v)
    assert rec["kernel"] == "qr_factor" and rec["expr_idx"] == e and rec["var_values"] == v
    assert rec["enqueue_end"] >= rec["enqueue_start"] and rec["batch"] in (1, 2, 4)
    assert max(r["batch"] for r in infos) == 4
    import pickle
    assert pickle.loads(program.dump_profiling_info(None, e, v)) == rec


def test_lambdapack_run_async_entry(oracle_backend):
    import asyncio
    A = ALG["cholesky_32_8/A"]
    X = BigMatrix("chol_async_in", shape=A.shape, shard_sizes=(8, 8))
    shard_matrix(X, A)
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    shared = {}
    loop = asyncio.new_event_loop()
    try:
        res = loop.run_until_complete(job_runner.lambdapack_run_async(loop, program, None, None, shared, None))
    finally:
        loop.close()
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS and "running_times" in shared
    assert set(res) >= {"up_time", "exec_time", "executed_messages", "operator_refs", "log"}
    np.testing.assert_allclose(meta["outputs"][0].numpy(), ALG["cholesky_32_8/L"], atol=1e-12)
