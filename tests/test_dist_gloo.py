"""The N > 1 path on CPU: world_size 2 and 4 over the gloo backend (torch.multiprocessing spawn), with
the NumPy/oracle checker backend doing the tile arithmetic.  What is under test is numpywren_amd/dist.py:
tile ownership, the owner-computes schedule walked identically by every rank, the matched point-to-point
exchange of produced tiles, reclaim accounting, and failure propagation.  On GPUs the same code runs with
the "nccl" (= RCCL) backend and the HIP backend."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scenario, outdir):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "NUMPYWREN_AMD_STORE": "hbm"})
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    from numpywren_amd import alg_wrappers, device, dist
    from numpywren_amd import lambdapack as lp
    from numpywren_amd.matrix import BigMatrix
    from oracle_backend import OracleBackend

    device.set_backend(OracleBackend())
    # (the two-phase form bench.py uses: control group first, the payload transport -- collectively -- later)
    comm = dist.init_process_group("gloo", open_transport=False)
    assert comm.transport is None and comm.backend == "host"
    comm.barrier()
    comm.open_transport()
    assert comm.transport is not None and comm.backend == "host" and comm.open_transport() is comm
    ALG = np.load(os.path.join(GOLDEN, "algos.npz"))
    result = {}

    def scatter_owned(bigm, X, name):
        """each rank only materialises the input tiles it owns"""
        for bidx, blk in zip(bigm.block_idxs, bigm.blocks):
            if comm.owner(name, bidx) == rank:
                bigm.put_block(np.ascontiguousarray(X[tuple(slice(s, e) for s, e in blk)]), *bidx)

    if scenario.startswith("cholesky"):
        tag = scenario.split(":")[1]
        A, L = ALG[f"cholesky_{tag}/A"], ALG[f"cholesky_{tag}/L"]
        n, b, lam, trunc, ntasks = ALG[f"cholesky_{tag}/meta"]
        X = BigMatrix(f"chol_in_{tag}", shape=A.shape, shard_sizes=(int(b), int(b)), lambdav=float(lam))
        scatter_owned(X, A, "I")
        program, meta = alg_wrappers.cholesky(X, truncate=int(trunc))
        program.config["executor"]["reclaim_intermediates"] = True
        program.start()
        res = dist.lambdapack_run_distributed(program, comm)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        O = meta["outputs"][0]
        # every output tile lives on its owner, and only there unless it was pushed to a consumer
        for bidx in O.block_idxs_exist:
            assert comm.owner("O", bidx) == rank or res["bytes_received"] > 0
        got = dist.gather_matrix(O, comm)
        counts = [None] * world
        comm.dist.all_gather_object(counts, len(res["executed_messages"]))
        if rank == 0:
            np.testing.assert_allclose(got, L, rtol=1e-12, atol=1e-12)
            assert sum(counts) == int(ntasks)           # every task ran exactly once, somewhere
            assert min(counts) > 0 or int(ntasks) < world
        assert res["bytes_sent"] > 0 or world == 1
        assert res["headers"] == 0     # every receiver derived shape and dtype from the static plan (TileMetaPlan)
        assert meta["intermediates"][0].block_idxs_exist == []   # reclaim works under sharding
    elif scenario == "grid16":
        # the shape of the driver's 8-GPU run (bench.py --gpus 8: 65536^2 in 4096^2 tiles): a 16 x 16 tile grid on the
        # 2 x 4 process grid, 816 tasks, every panel tile pushed to the owners of its consumers -- with 4 x 4 tiles
        rng = np.random.default_rng(816)
        nb, b = 16, 4
        G = rng.standard_normal((nb * b, nb * b))
        A = G @ G.T + nb * b * np.eye(nb * b)
        X = BigMatrix("chol_grid16", shape=A.shape, shard_sizes=(b, b))
        scatter_owned(X, A, "I")
        program, meta = alg_wrappers.cholesky(X)
        program.config["executor"]["reclaim_intermediates"] = True
        program.config["executor"]["task_timers"] = True       # the diagnostics of bench.py's N > 1 lines
        program.start()
        res = dist.lambdapack_run_distributed(program, comm, pipeline_width=3)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        got = dist.gather_matrix(meta["outputs"][0], comm)
        counts = [None] * world
        comm.dist.all_gather_object(counts, len(res["executed_messages"]))
        if rank == 0:
            np.testing.assert_allclose(got, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)
            assert sum(counts) == nb * (nb + 1) * (nb + 2) // 6 == 816
            assert min(counts) > 0
        assert res["headers"] == 0 and res["bytes_sent"] > 0
        # every rank reports what its walk cost: all 816 positions visited, its own share of them executed, bytes both ways
        d = res["diag"]
        assert d["rank"] == rank and d["positions"] <= 816 and d["tasks_run_here"] == len(res["executed_messages"])
        assert d["bytes_sent"] == res["bytes_sent"] and d["bytes_received"] == res["bytes_received"] and d["transport"] == "host"
        assert d["host_walk_ms"] > 0 and d["host_blocked_ms"] >= 0 and d["drain_ms"] >= 0
        assert d["transfer_wait_ms"] >= 0 and "kernel_busy_ms" in d
        with open(os.path.join(outdir, f"diag_{rank}.json"), "w") as f:
            import json
            json.dump(d, f)
    elif scenario == "tsqr8":
        # one leaf per rank: every tree edge crosses ranks (the 3 cross-GPU levels of bench.py --workload tsqr --gpus 8)
        Xh = ALG["tsqr_64_8/X"]
        X = BigMatrix("tsqr_in_8", shape=Xh.shape, shard_sizes=(8, 8))
        comm.ownership = dist.tsqr_ownership(world, 8)
        scatter_owned(X, Xh, "A")
        assert len(X.block_idxs_exist) == 1
        program, meta = alg_wrappers.tsqr(X)
        program.start()
        res = dist.lambdapack_run_distributed(program, comm, pipeline_width=2)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        counts, sent = [None] * world, [None] * world
        comm.dist.all_gather_object(counts, len(res["executed_messages"]))
        comm.dist.all_gather_object(sent, res["bytes_sent"])
        assert sum(counts) == 15 and sum(sent) == 7 * 8 * 8 * 8     # 8 leaves + 7 nodes; one R factor per tree edge
        R = meta["outputs"][0]
        if R.tile_exists(3, 0):
            np.testing.assert_allclose(R.get_block(3, 0), ALG["tsqr_64_8/R_final"], atol=1e-12)
            result["has_final"] = True
        flags = [None] * world
        comm.dist.all_gather_object(flags, bool(result.get("has_final")))
        assert sum(flags) == 1
        comm.ownership = None
    elif scenario == "tsqr":
        Xh = ALG["tsqr_64_8/X"]
        X = BigMatrix("tsqr_in", shape=Xh.shape, shard_sizes=(8, 8))
        scatter_owned(X, Xh, "A")
        program, meta = alg_wrappers.tsqr(X)
        program.start()
        dist.lambdapack_run_distributed(program, comm)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        R = meta["outputs"][0]
        if R.tile_exists(3, 0):
            np.testing.assert_allclose(R.get_block(3, 0), ALG["tsqr_64_8/R_final"], atol=1e-12)
            result["has_final"] = True
        flags = [None] * world
        comm.dist.all_gather_object(flags, bool(result.get("has_final")))
        assert sum(flags) >= 1
    elif scenario == "tsqr_chunked":
        # algorithm-aware ownership: contiguous leaf chunks, local sub-trees, log2(world) exchanged R factors
        Xh = ALG["tsqr_64_8/X"]
        X = BigMatrix("tsqr_in_c", shape=Xh.shape, shard_sizes=(8, 8))
        comm.ownership = dist.tsqr_ownership(world, 8)
        scatter_owned(X, Xh, "A")
        assert len(X.block_idxs_exist) == 8 // world
        program, meta = alg_wrappers.tsqr(X)
        program.start()
        be = device.get_backend()
        res = dist.lambdapack_run_distributed(program, comm)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        # 8 leaves + 7 tree nodes; rank r runs its 8/world leaves, its local sub-tree, and rank 0 the top
        local_nodes = 8 // world - 1
        top = {2: [1, 0], 4: [2, 0, 1, 0]}[world][rank]
        assert len(res["executed_messages"]) == 8 // world + local_nodes + top
        # the leaves of a rank went to the device as one batch
        sizes = [c[1] for c in be.calls if c[0] == "geqrt_batched"] or [sum(1 for c in be.calls if c[0] == "geqrt")]
        assert sizes and sizes[0] == 8 // world
        # every tree node stacks two R factors -- also those that arrived from another rank (the flag travels in
        # the tile header) -- and takes the structured factorisation
        nodes = local_nodes + top
        assert sum(c[1] for c in be.calls if c[0] == "tpqrt_batched") == nodes
        assert sum(1 for c in be.calls if c[0] == "geqrt") == (8 // world if 8 // world == 1 else 0)
        # only R factors travel: one 8 x 8 tile per tree edge that crosses ranks
        sent = [None] * world
        comm.dist.all_gather_object(sent, res["bytes_sent"])
        assert sum(sent) == (world - 1) * 8 * 8 * 8
        R = meta["outputs"][0]
        assert R.tile_exists(3, 0) == (rank == 0)
        if rank == 0:
            np.testing.assert_allclose(R.get_block(3, 0), ALG["tsqr_64_8/R_final"], atol=1e-12)
        comm.ownership = None
    elif scenario == "gemm":
        A, B, C = ALG["gemm_40_8/A"], ALG["gemm_40_8/B"], ALG["gemm_40_8/C"]
        Ab = BigMatrix("gemm_A", shape=A.shape, shard_sizes=(8, 8))
        Bb = BigMatrix("gemm_B", shape=B.shape, shard_sizes=(8, 8))
        scatter_owned(Ab, A, "A")
        scatter_owned(Bb, B, "B")
        program, meta = alg_wrappers.gemm(Ab, Bb)
        program.start()
        res = dist.lambdapack_run_distributed(program, comm)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        assert res["headers"] == 0 and res["transfers"] > 0     # Temp is safe=False, its tile shapes are still static
        got = dist.gather_matrix(meta["outputs"][0], comm)
        if rank == 0:
            np.testing.assert_allclose(got, C, rtol=1e-12, atol=1e-12)
    elif scenario == "block_sparse":
        A = ALG["cholesky_32_8/A"]
        X = BigMatrix("chol_bs", shape=A.shape, shard_sizes=(8, 8))
        scatter_owned(X, A, "I")
        program, meta = alg_wrappers.cholesky(X)
        program.block_sparse = True
        program.start()
        try:
            dist.lambdapack_run_distributed(program, comm)
            raise AssertionError("block_sparse must be refused")
        except NotImplementedError:
            pass
    elif scenario == "slow_rank":
        # One rank's host is slow and the time limit expires: the decision to stop is taken on the maximum over the
        # ranks at fixed positions of the common task sequence, so every rank leaves the walk at the SAME position (a
        # rank leaving on its own clock would strand its peers inside a transfer), nothing hangs, and the run can be
        # resumed because the program is still RUNNING.
        import time as _time
        rng = np.random.default_rng(5)
        nb, b = 12, 4
        G = rng.standard_normal((nb * b, nb * b))
        A = G @ G.T + nb * b * np.eye(nb * b)
        X = BigMatrix("chol_slow", shape=A.shape, shard_sizes=(b, b))
        scatter_owned(X, A, "I")
        program, meta = alg_wrappers.cholesky(X)
        program.start()
        from numpywren_amd import job_runner
        real, real_batch = job_runner.LambdaPackExecutor.run_task, job_runner.LambdaPackExecutor.run_batch

        def slow(self, e, v, stream=None):
            if rank == 1:
                _time.sleep(0.05)
            return real(self, e, v, stream=stream)

        def slow_batch(self, nodes, stream=None):
            if rank == 1:
                _time.sleep(0.05)
            return real_batch(self, nodes, stream=stream)
        job_runner.LambdaPackExecutor.run_task, job_runner.LambdaPackExecutor.run_batch = slow, slow_batch
        dist.TIMEOUT_CHECK_EVERY = 4
        res = dist.lambdapack_run_distributed(program, comm, timeout=0.2)      # rank 1 is past it after four of its launches
        job_runner.LambdaPackExecutor.run_task, job_runner.LambdaPackExecutor.run_batch = real, real_batch
        steps = [None] * world
        comm.dist.all_gather_object(steps, (res["steps"], res["timed_out"]))
        assert len(set(steps)) == 1 and steps[0][1] is True, steps          # same position everywhere
        assert 0 < res["steps"] and len(res["executed_messages"]) < nb * (nb + 1) * (nb + 2) // 6 // 2
        assert program.program_status() == lp.PS.RUNNING
        # resume without a limit: the program completes and the factor is right
        res2 = dist.lambdapack_run_distributed(program, comm, timeout=None)
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        assert not res2["timed_out"]
        got = dist.gather_matrix(meta["outputs"][0], comm)
        if rank == 0:
            np.testing.assert_allclose(got, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)
    elif scenario == "stalled_rank":
        # a rank that stops making progress (here: rank 1 sleeps 1.5 s inside one task) must leave a
        # report on EVERY rank that is held up by it -- who, at which position of the common sequence, in which task, waiting
        # for which peer -- instead of a silent hang.  Nothing is aborted (abort_s = 0): the run completes once rank 1 wakes up.
        import io
        import time as _time
        os.environ["NUMPYWREN_AMD_DIST_STALL_REPORT_S"] = "0.3"
        # the link calibration first (host transport on this box: blocking pairs over the control group)
        calib = dist.link_calibration(comm, nbytes=1 << 16, repeats=2)
        assert calib["transport"] == "host" and calib["xgmi_GBps"]["min"] > 0 and calib["xgmi_GBps"]["min"] <= calib["xgmi_GBps"]["median"] <= calib["xgmi_GBps"]["max"]
        assert len(calib["samples"]) == world * (world - 1 if world == 2 else 2) and {s["rank"] for s in calib["samples"]} == set(range(world))
        rng = np.random.default_rng(6)
        nb, b = 6, 4
        G = rng.standard_normal((nb * b, nb * b))
        A = G @ G.T + nb * b * np.eye(nb * b)
        X = BigMatrix("chol_stall", shape=A.shape, shard_sizes=(b, b))
        scatter_owned(X, A, "I")
        program, meta = alg_wrappers.cholesky(X)
        program.start()
        from numpywren_amd import job_runner
        real = job_runner.LambdaPackExecutor.run_task
        calls = {"n": 0}

        def stall(self, e, v, stream=None):
            calls["n"] += 1
            if rank == 1 and calls["n"] == 3:
                _time.sleep(1.5)
            return real(self, e, v, stream=stream)
        job_runner.LambdaPackExecutor.run_task = stall
        err, sys.stderr = sys.stderr, io.StringIO()
        try:
            res = dist.lambdapack_run_distributed(program, comm, timeout=None)
            printed = sys.stderr.getvalue()
        finally:
            sys.stderr = err
            job_runner.LambdaPackExecutor.run_task = real
        assert program.program_status() == lp.PS.SUCCESS, program.exceptions
        reports = comm.stall_watch.reports
        counts = [None] * world
        comm.dist.all_gather_object(counts, len(reports))
        assert counts[0] >= 1, counts                     # rank 0 waited for rank 1 and said so
        if rank == 0:
            text = reports[0]
            assert text in printed and res["diag"]["stall_reports"] == len(reports)
            assert "rank 0/2" in text and "position" in text and "task (" in text and "no progress for" in text
            # the peer it was held up by: a receive from rank 1, or (blocking host transport) a send rank 1 has not taken yet
            assert "from rank 1" in text or "to [1]" in text, text
        got = dist.gather_matrix(meta["outputs"][0], comm)
        if rank == 0:
            np.testing.assert_allclose(got, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)
    elif scenario == "calib":
        # the link calibration on more than two ranks of the (blocking) host transport: the shift exchanges at distance 1 and
        # world - 1 are rings of blocking pairs that must not lock up; every rank ends with the same gathered table
        calib = dist.link_calibration(comm, nbytes=1 << 14, repeats=1)
        assert calib["transport"] == "host" and len(calib["samples"]) == 2 * world
        assert {(s_["rank"], s_["d"]) for s_ in calib["samples"]} == {(r, d) for r in range(world) for d in (1, world - 1)}
        tables = [None] * world
        comm.dist.all_gather_object(tables, calib["xgmi_GBps"])
        assert all(t == tables[0] for t in tables) and tables[0]["min"] > 0
    elif scenario == "not_pd":
        A = np.eye(32)
        A[20, 20] = -1.0
        X = BigMatrix("chol_bad", shape=A.shape, shard_sizes=(8, 8))
        scatter_owned(X, A, "I")
        program, meta = alg_wrappers.cholesky(X)
        program.start()
        dist.lambdapack_run_distributed(program, comm)
        assert program.program_status() == lp.PS.EXCEPTION     # on EVERY rank, not only the failing one
    comm.barrier()
    comm.shutdown()
    with open(os.path.join(outdir, f"ok_{rank}"), "w") as f:
        f.write("ok")


def _spawn(world, scenario, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, scenario, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(os.path.join(str(tmp_path), f"ok_{r}"))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("tag", ["32_8", "40_8_t2", "24_8_lam"])
def test_cholesky_sharded(world, tag, tmp_path):
    _spawn(world, f"cholesky:{tag}", tmp_path)


def test_cholesky_16x16_tiles_on_8_ranks(tmp_path):
    _spawn(8, "grid16", tmp_path)


def test_tsqr_and_gemm_on_8_ranks(tmp_path):
    _spawn(8, "tsqr8", tmp_path)
    _spawn(8, "gemm", tmp_path)


def test_tsqr_sharded(tmp_path):
    _spawn(2, "tsqr", tmp_path)


@pytest.mark.parametrize("world", [2, 4])
def test_tsqr_chunked_ownership_and_batches(world, tmp_path):
    _spawn(world, "tsqr_chunked", tmp_path)


def test_gemm_sharded(tmp_path):
    _spawn(2, "gemm", tmp_path)


def test_time_limit_is_a_collective_decision(tmp_path):
    """one slow rank, a short limit -- every rank stops at the same position, then resumes."""
    _spawn(2, "slow_rank", tmp_path)


def test_a_stalled_rank_is_reported_by_the_ranks_it_holds_up(tmp_path):
    """the per-rank stall report (dist.StallWatch) and the link calibration's fields on two gloo ranks."""
    _spawn(2, "stalled_rank", tmp_path)


def test_link_calibration_on_four_host_ranks(tmp_path):
    _spawn(4, "calib", tmp_path)


def test_failure_reaches_every_rank(tmp_path):
    _spawn(2, "not_pd", tmp_path)


def test_block_sparse_is_refused(tmp_path):
    """an owner that skips a zero tile would leave its consumers waiting in recv."""
    _spawn(2, "block_sparse", tmp_path)


def test_tile_meta_plan_matches_what_the_kernels_produce(oracle_backend):
    """The static (shape, dtype) table receivers rely on == the tiles a real run stores, for the Cholesky and GEMM
    programs (fp64 and fp32 inputs, ragged edge blocks)."""
    import numpy as np
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd.dist import TileMetaPlan
    from numpywren_amd.matrix import BigMatrix
    from numpywren_amd.matrix_init import shard_matrix
    rng = np.random.default_rng(0)

    def check(program, meta):
        compiled = program.program
        plan = TileMetaPlan(compiled)
        program.start()
        job_runner.lambdapack_run(program)
        n = 0
        for t in compiled.tasks:
            outs = plan.visit(t, getattr(compiled.kernel(t.expr_idx), "__name__", ""))
            assert outs is not None, t.key
            for (name, idx), m in zip(t.writes, outs):
                got = compiled.matrices[name].get_tile(*idx)
                assert int(np.prod(got.shape)) == int(np.prod(m.shape)) and np.dtype(got.dtype) == m.dtype, (t.key, got, m)
                n += 1
        return n

    G = rng.standard_normal((40, 40))
    A = G @ G.T + 40 * np.eye(40)
    X = BigMatrix("meta_chol", shape=A.shape, shard_sizes=(16, 16))     # 16 + 16 + 8: ragged last block
    shard_matrix(X, A)
    assert check(*alg_wrappers.cholesky(X)) == 10
    for dt in (np.float64, np.float32):
        Ah, Bh = rng.standard_normal((24, 24)).astype(dt), rng.standard_normal((24, 24)).astype(dt)
        Am = BigMatrix(f"meta_gA{np.dtype(dt).itemsize}", shape=Ah.shape, shard_sizes=(8, 8), dtype=dt)
        Bm = BigMatrix(f"meta_gB{np.dtype(dt).itemsize}", shape=Bh.shape, shard_sizes=(8, 8), dtype=dt)
        shard_matrix(Am, Ah)
        shard_matrix(Bm, Bh)
        assert check(*alg_wrappers.gemm(Am, Bm)) > 27


def test_process_grid_and_ownership():
    from numpywren_amd.dist import Comm, process_grid
    assert [process_grid(w) for w in (1, 2, 4, 8, 6)] == [(1, 1), (1, 2), (2, 2), (2, 4), (2, 3)]

    class C(Comm):
        def __init__(self, world):
            self.world, self.grid = world, process_grid(world)

    c = C(8)
    # all SSA versions of a trailing tile and the factor tile share one owner
    assert c.owner("S", (3, 5, 2)) == c.owner("I", (5, 2)) == c.owner("O", (5, 2)) == (5 % 2) * 4 + 2
    owners = {c.owner("O", (j, k)) for j in range(16) for k in range(j + 1)}
    assert owners == set(range(8))
    # TSQR: contiguous leaf chunks, a tree node lives with its left operand
    from numpywren_amd.dist import tsqr_ownership
    c.ownership = tsqr_ownership(8, 256)
    assert [c.owner("A", (j, 0)) for j in (0, 31, 32, 255)] == [0, 0, 1, 7]
    assert c.owner("Rs", (0, 33)) == 1 and c.owner("Rs", (5, 32)) == 1 and c.owner("Rs", (6, 0)) == 0
    assert c.owner("Vs", (8, 0)) == 0 and c.owner("Ts", (7, 128)) == 4
    assert c.owner("O", (5, 2)) == (5 % 2) * 4 + 2       # other matrices keep the block-cyclic map
