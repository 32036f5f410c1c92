"""Checkpoint / resume of a run on disk (numpywren_amd/checkpoint.py): tiles as the reference's objects + the finished nodes;
a program built again in a "new process" (cleared tile table) continues where the first one stopped and ends in the same factor."""
import numpy as np
import pytest

from numpywren_amd import alg_wrappers, checkpoint, job_runner, matrix
from numpywren_amd import lambdapack as lp
from numpywren_amd.matrix import BigMatrix


def _input(a, b, key="ckpt_in"):
    n = a.shape[0]
    X = BigMatrix(key, shape=a.shape, shard_sizes=(b, b), write_header=True)
    for i in range(n // b):
        for j in range(n // b):
            X.put_block(np.ascontiguousarray(a[i * b:(i + 1) * b, j * b:(j + 1) * b]), i, j)
    return X


def _run_some(program, k):
    """the first k tasks of the ready heap, one at a time (what a run that hit its time limit leaves behind)"""
    ex = job_runner.LambdaPackExecutor(program, pipeline_width=1)
    for _ in range(k):
        e, v = program.dequeue()
        program.set_node_status(e, v, lp.NS.RUNNING)
        ex.run_task(e, v)
        program.post_op(e, v, lp.PS.SUCCESS, None)
        program.set_node_status(e, v, lp.NS.FINISHED)


@pytest.mark.parametrize("done_first", [0, 7, 21, 35])
def test_cholesky_resumes_from_a_checkpoint(oracle_backend, tmp_path, done_first):
    rng = np.random.default_rng(11)
    n, b = 40, 8
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    X = _input(a, b)
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["reclaim_intermediates"] = True
    program.start()
    total = len(program.program.tasks)
    assert total == 35
    _run_some(program, done_first)
    saved = checkpoint.save(program, str(tmp_path))
    assert saved["finished"] == done_first and saved["tasks"] == total and saved["tiles"] >= 25
    # "a new process": nothing in the tile table, the program built again from the same matrix description
    matrix.OBJECTS.clear()
    X2 = BigMatrix("ckpt_in", shape=a.shape, shard_sizes=(b, b), write_header=True)
    program2, meta2 = alg_wrappers.cholesky(X2)
    program2.config["executor"]["reclaim_intermediates"] = True
    left = checkpoint.load(program2, str(tmp_path))
    assert left == total - done_first
    res = job_runner.lambdapack_run(program2)
    program2.wait()
    assert program2.program_status() == lp.PS.SUCCESS, program2.exceptions
    assert len(res["executed_messages"]) == left                      # only what was left ran
    L = meta2["outputs"][0].numpy()
    np.testing.assert_allclose(np.tril(L), np.linalg.cholesky(a), rtol=1e-12, atol=1e-12)
    assert meta2["intermediates"][0].block_idxs_exist == []          # reclaim counts only the readers that still ran


def test_tsqr_resumes_and_a_mismatched_program_is_refused(oracle_backend, tmp_path):
    rng = np.random.default_rng(12)
    b, leaves = 8, 8
    xh = rng.standard_normal((b * leaves, b))
    X = BigMatrix("ckpt_tsqr", shape=xh.shape, shard_sizes=(b, b), write_header=True)
    for j in range(leaves):
        X.put_block(np.ascontiguousarray(xh[j * b:(j + 1) * b]), j, 0)
    program, meta = alg_wrappers.tsqr(X)
    program.start()
    _run_some(program, 10)          # the 8 leaves and two nodes of the first level
    checkpoint.save(program, str(tmp_path), name="t")
    matrix.OBJECTS.clear()
    X2 = BigMatrix("ckpt_tsqr", shape=xh.shape, shard_sizes=(b, b), write_header=True)
    program2, meta2 = alg_wrappers.tsqr(X2)
    assert checkpoint.load(program2, str(tmp_path), name="t") == 5
    job_runner.lambdapack_run(program2)
    program2.wait()
    assert program2.program_status() == lp.PS.SUCCESS, program2.exceptions
    R = meta2["outputs"][0].get_block(3, 0)
    np.testing.assert_allclose(np.abs(R), np.abs(np.linalg.qr(xh, mode="r")), rtol=1e-10, atol=1e-10)
    # another program (16 leaves) must not swallow this checkpoint
    X3 = BigMatrix("ckpt_tsqr", shape=(b * 16, b), shard_sizes=(b, b), write_header=True)
    program3, _ = alg_wrappers.tsqr(X3)
    with pytest.raises(ValueError):
        checkpoint.load(program3, str(tmp_path), name="t")
    # a "finished" set that is not closed under parents is refused, too
    program4, _ = alg_wrappers.tsqr(X2)
    with pytest.raises(ValueError):
        program4.resume([(1, {"l": 2, "j": 0})])


@pytest.mark.gpu
def test_checkpoint_on_the_gpu(hbm_store, tmp_path):
    """The same on the HIP backend: tiles leave HBM as the reference's objects and come back; the resumed run ends in the factor
    of the uninterrupted one."""
    from numpywren_amd.device import get_backend
    be = get_backend()
    rng = np.random.default_rng(13)
    n, b = 1536, 256
    g = rng.standard_normal((n, 64))
    a = g @ g.T + n * np.eye(n)
    X = _input(a, b, key="ckpt_gpu")
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    job_runner.lambdapack_run(program)
    program.wait()
    L_ref = meta["outputs"][0].numpy()
    program.free()
    for m in meta["outputs"] + meta["intermediates"]:
        m.free()
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    _run_some(program, 20)
    be.synchronize()
    saved = checkpoint.save(program, str(tmp_path))
    assert saved["finished"] == 20 and saved["tasks"] == 56
    matrix.OBJECTS.clear()
    X2 = BigMatrix("ckpt_gpu", shape=a.shape, shard_sizes=(b, b), write_header=True)
    program2, meta2 = alg_wrappers.cholesky(X2)
    assert checkpoint.load(program2, str(tmp_path)) == 36
    job_runner.lambdapack_run(program2)
    program2.wait()
    assert program2.program_status() == lp.PS.SUCCESS, program2.exceptions
    L = meta2["outputs"][0].numpy()
    np.testing.assert_allclose(L, L_ref, rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(np.tril(L), np.linalg.cholesky(a), rtol=1e-10, atol=1e-9)


def _gemm_program(tag, A, B, b, fuse=True):
    from numpywren_amd.matrix_init import shard_matrix
    Ab = BigMatrix(f"ckg_A_{tag}", shape=A.shape, shard_sizes=(b, b), write_header=True)
    Bb = BigMatrix(f"ckg_B_{tag}", shape=B.shape, shard_sizes=(b, b), write_header=True)
    shard_matrix(Ab, A)
    shard_matrix(Bb, B)
    program, meta = alg_wrappers.gemm(Ab, Bb)
    program.config["executor"]["fuse_gemm_reduction"] = fuse
    return program, meta


@pytest.mark.parametrize("done_first", [3, 10, 40, 64])
def test_fused_gemm_resumes_from_a_checkpoint(oracle_backend, tmp_path, done_first):
    """executor.fuse_gemm_reduction keeps the partial sums of unfinished C tiles outside the tile store: the
    products that went into a still-open accumulator are saved as NOT finished and run again after the resume."""
    rng = np.random.default_rng(21)
    n, b = 32, 8
    A, B = rng.standard_normal((n, n)), rng.standard_normal((n, n))
    program, meta = _gemm_program("x", A, B, b)
    program.start()
    _run_some(program, done_first)
    saved = checkpoint.save(program, str(tmp_path))
    assert saved["finished"] <= done_first                         # open accumulators' products do not count
    if done_first == 3:
        assert saved["finished"] == 0                               # no C tile has all four products yet
    matrix.OBJECTS.clear()
    program2, meta2 = _gemm_program("x", A, B, b)
    left = checkpoint.load(program2, str(tmp_path))
    assert left == saved["tasks"] - saved["finished"]
    job_runner.lambdapack_run(program2)
    program2.wait()
    assert program2.program_status() == lp.PS.SUCCESS, program2.exceptions
    np.testing.assert_allclose(meta2["outputs"][0].numpy(), A @ B, rtol=1e-12, atol=1e-12)


def test_checkpoint_is_atomic_and_lists_its_tiles(oracle_backend, tmp_path):
    """A second save into the same root after intermediates were reclaimed must not bring their old tile objects back; an
    interrupted write (a leftover .tmp) is ignored; a program in EXCEPTION state and a distributed run are refused."""
    import glob
    import os
    rng = np.random.default_rng(14)
    n, b = 32, 8
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    X = _input(a, b, key="ckpt_at")
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["reclaim_intermediates"] = True
    program.start()
    _run_some(program, 6)
    first = checkpoint.save(program, str(tmp_path))
    _run_some(program, 10)
    second = checkpoint.save(program, str(tmp_path))
    assert second["finished"] == 16
    inter = meta["intermediates"][0]
    on_disk = len(glob.glob(os.path.join(str(tmp_path), inter.bucket, inter.key_base, "*_*")))
    assert on_disk > len(inter.block_idxs_exist)                   # stale objects of reclaimed tiles are still on disk ...
    assert not glob.glob(os.path.join(str(tmp_path), "**", "*.tmp"), recursive=True)
    open(os.path.join(str(tmp_path), inter.bucket, inter.key_base, "0_8_8_0_8_8_0_8_8_.tmp"), "wb").write(b"torn")
    matrix.OBJECTS.clear()
    X2 = BigMatrix("ckpt_at", shape=a.shape, shard_sizes=(b, b), write_header=True)
    program2, meta2 = alg_wrappers.cholesky(X2)
    program2.config["executor"]["reclaim_intermediates"] = True
    checkpoint.load(program2, str(tmp_path))
    assert len(meta2["intermediates"][0].block_idxs_exist) == len(inter.block_idxs_exist) < on_disk   # ... and stay there
    job_runner.lambdapack_run(program2)
    program2.wait()
    assert program2.program_status() == lp.PS.SUCCESS, program2.exceptions
    np.testing.assert_allclose(np.tril(meta2["outputs"][0].numpy()), np.linalg.cholesky(a), rtol=1e-12, atol=1e-12)
    assert first["tiles"] > 0
    program2._distributed_world = 2
    with pytest.raises(NotImplementedError):
        checkpoint.save(program2, str(tmp_path), name="d")
    program2._distributed_world = 1
    program2.handle_exception("boom", tb="", expr_idx=0, var_values={})
    with pytest.raises(ValueError):
        checkpoint.save(program2, str(tmp_path), name="e")
