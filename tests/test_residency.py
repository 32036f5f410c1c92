"""The store's host-DRAM tier (numpywren_amd/residency.py): LRU policy and bookkeeping on the CPU with the checker
backend, the pinned asynchronous copies and the out-of-memory path on the GPU."""
import numpy as np
import pytest

from numpywren_amd import matrix, residency
from numpywren_amd.device import DeviceTile, SpilledTile
from numpywren_amd.matrix import BigMatrix

TILE = 16 * 16 * 8


def _mat(key, nb=4, b=16):
    m = BigMatrix(key, shape=(nb * b, nb * b), shard_sizes=(b, b), write_header=True)
    rng = np.random.default_rng(5)
    ref = {}
    for i in range(nb):
        for j in range(nb):
            ref[i, j] = rng.standard_normal((b, b))
            m.put_block(ref[i, j], i, j)
    return m, ref


def _kinds(m):
    return {matrix.block_key_to_block(k): type(v) for k, v in m._tiles(False).items()}


def test_parse_bytes():
    assert residency.parse_bytes(None) is None
    assert residency.parse_bytes("none") is None
    assert residency.parse_bytes(1234) == 1234
    assert residency.parse_bytes("1234") == 1234
    assert residency.parse_bytes("2k") == 2048
    assert residency.parse_bytes("200G") == 200 << 30
    assert residency.parse_bytes("1.5 GiB") == 3 << 29
    assert residency.parse_bytes("512MB") == 512 << 20
    with pytest.raises(ValueError):
        residency.parse_bytes("lots")


def test_no_budget_keeps_everything_resident(oracle_backend):
    m, _ = _mat("res_nobudget")
    st = matrix.RESIDENCY.stats()
    assert st["resident_tiles"] == 16 and st["resident_bytes"] == 16 * TILE and st["evictions"] == 0
    m.free()
    assert matrix.RESIDENCY.stats()["resident_bytes"] == 0


def test_budget_evicts_least_recently_used(oracle_backend):
    matrix.RESIDENCY.set_budget(6 * TILE)
    m, ref = _mat("res_lru")
    st = matrix.RESIDENCY.stats()
    assert st["resident_bytes"] == 6 * TILE and st["evictions"] == 10
    kinds = _kinds(m)
    spilled = sorted(k for k, t in kinds.items() if t is SpilledTile)
    assert len(spilled) == 10
    # the ten oldest puts went: rows 0, 1 and (2,0), (2,1)
    assert spilled == sorted(m.blocks[:10])
    # reads return the bytes that left, spilled or not; get_block of a spilled tile does not promote it
    for (i, j), a in ref.items():
        assert np.array_equal(m.get_block(i, j), a)
    assert matrix.RESIDENCY.stats()["restores"] == 0
    # get_tile promotes: the tile is resident again and the least recently used resident tile makes room
    t = m.get_tile(0, 0)
    assert isinstance(t, DeviceTile) and np.array_equal(oracle_backend.to_host(t), ref[0, 0])
    st = matrix.RESIDENCY.stats()
    assert st["restores"] == 1 and st["resident_bytes"] == 6 * TILE
    kinds = _kinds(m)
    assert kinds[m.blocks[0]] is not SpilledTile
    assert kinds[m.blocks[10]] is SpilledTile      # (2,2) was the oldest resident one
    # touching protects: read (2,3), then promote another tile -> (3,0) goes, not (2,3)
    m.get_tile(2, 3)
    m.get_tile(0, 1)
    kinds = _kinds(m)
    assert kinds[m.blocks[11]] is not SpilledTile and kinds[m.blocks[12]] is SpilledTile
    assert np.array_equal(m.numpy(), np.block([[ref[i, j] for j in range(4)] for i in range(4)]))


def test_shared_buffers_move_together_and_deletes_are_accounted(oracle_backend):
    be = oracle_backend
    m = BigMatrix("res_shared", shape=(64, 16), shard_sizes=(16, 16), write_header=True)
    a = np.arange(256.0).reshape(16, 16)
    t = be.to_device(a)
    m.put_tile(t, 0, 0)
    m.put_tile(t, 1, 0)          # same buffer under two keys (what kernels.identity produces)
    assert matrix.RESIDENCY.stats()["resident_bytes"] == TILE
    m.put_block(a + 1, 2, 0)
    assert matrix.RESIDENCY.stats()["resident_bytes"] == 2 * TILE
    matrix.RESIDENCY.set_budget(TILE)
    kinds = _kinds(m)
    assert kinds[m.blocks[0]] is SpilledTile and kinds[m.blocks[1]] is SpilledTile
    assert matrix.RESIDENCY.stats()["evictions"] == 1     # one copy for the shared buffer
    assert np.array_equal(m.get_block(1, 0), a) and np.array_equal(m.get_block(0, 0), a)
    m.delete_block(2, 0)
    assert matrix.RESIDENCY.stats()["resident_bytes"] == 0
    m.put_block(a + 2, 3, 0)
    m.put_block(a + 3, 3, 0)     # overwrite: the old object stops counting
    assert matrix.RESIDENCY.stats()["resident_bytes"] == TILE
    # constant zero tiles are shared read-only objects of the backend: never counted, never evicted
    z = be.shared_zeros((16, 16))
    m.put_tile(z, 2, 0)
    assert matrix.RESIDENCY.stats()["resident_bytes"] == TILE and _kinds(m)[m.blocks[2]] is not SpilledTile


def test_reclaim_is_the_allocators_out_of_memory_handler(oracle_backend):
    m, ref = _mat("res_oom")
    assert matrix.RESIDENCY.reclaim in oracle_backend.oom_handlers
    freed = oracle_backend.oom_handlers[0](3 * TILE)
    assert freed == 3 * TILE and matrix.RESIDENCY.stats()["resident_bytes"] == 13 * TILE
    assert sorted(k for k, t in _kinds(m).items() if t is SpilledTile) == sorted(m.blocks[:3])
    for (i, j), a in ref.items():
        assert np.array_equal(oracle_backend.to_host(m.get_tile(i, j)), a)


def test_cholesky_under_a_tight_budget(oracle_backend):
    """The executor never sees the tier: a program runs to the same result with almost nothing resident."""
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    rng = np.random.default_rng(3)
    n, b = 64, 16
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    A = BigMatrix("res_chol_in", shape=(n, n), shard_sizes=(b, b), write_header=True)
    matrix.RESIDENCY.set_budget(3 * TILE)
    for i in range(4):
        for j in range(4):
            A.put_block(a[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
    program, meta = alg_wrappers.cholesky(A)
    program.start()
    job_runner.lambdapack_run(program)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS
    L = meta["outputs"][0].numpy()
    assert np.allclose(L, np.linalg.cholesky(a))
    st = matrix.RESIDENCY.stats()
    assert st["resident_bytes"] <= 3 * TILE and st["evictions"] > 10 and st["restores"] > 5
    program.free()


def test_spill_plan_next_use_follows_the_issued_tasks():
    """SpillPlan: the next read of a tile is the first not yet issued task of the predicted order that reads it, whatever
    order the tasks are really issued in; `upcoming` lists what the next tasks read, nearest first."""
    class T(object):
        def __init__(self, index, reads):
            self.index, self.reads = index, reads
    order = [T(10, [("A", (0,))]), T(11, [("A", (0,)), ("B", (0,))]), T(12, [("B", (0,))]), T(13, [("A", (0,)), ("C", (0,))])]
    plan = residency.SpillPlan(order, lambda name, idx: (name, idx))
    a, b, c = ("A", (0,)), ("B", (0,)), ("C", (0,))
    assert (plan.next_use(a), plan.next_use(b), plan.next_use(c)) == (0, 1, 3) and plan.next_use(("Z", (0,))) == plan.NEVER
    assert plan.upcoming(2) == [a, b]
    plan.issued(12)                      # out of order (a batch pulled it forward)
    assert plan.next_use(b) == 1 and plan.cursor == 0
    plan.issued(10)
    plan.issued(11)
    assert (plan.next_use(a), plan.next_use(b), plan.cursor) == (3, plan.NEVER, 3) and plan.upcoming(4) == [a, c]
    plan.issued(13)
    assert plan.next_use(a) == plan.NEVER and plan.upcoming(4) == []


def test_predicted_issue_order_is_the_real_one(oracle_backend):
    """job_runner.predicted_issue_order dry-walks lambdapack_run's own loop on a shadow of the program's state: the order it
    records is the order the real run issues (batches, the tasks pulled ahead to complete a batch, ...), for any batch size."""
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd.compiler import node_key
    rng = np.random.default_rng(6)
    nb, b = 6, 8
    n = nb * b
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    for batch in (32, 3, 1):
        A = BigMatrix("res_order_%d" % batch, shape=(n, n), shard_sizes=(b, b), write_header=True)
        for i in range(nb):
            for j in range(i + 1):
                A.put_block(a[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
        program, meta = alg_wrappers.cholesky(A)
        program.config["executor"]["batch_tasks"] = batch
        predicted = [t.key for t in job_runner.predicted_issue_order(program)]
        program.start()
        res = job_runner.lambdapack_run(program)
        assert predicted == [node_key(e, v) for e, v in res["executed_messages"]] and len(predicted) == 56
        program.free()
        A.free()


def test_plan_driven_eviction_moves_fewer_tiles_than_lru(oracle_backend):
    """The same budgeted Cholesky (8 x 8 tiles, 24 tiles of budget: the shape of tools/bench_aux.py spill) with the victims
    chosen by the DAG (farthest next read) and by LRU: same factor bit for bit, a fraction of the copies -- a Cholesky sweeps
    its trailing matrix once per step, the access pattern on which LRU misses everything."""
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    rng = np.random.default_rng(4)
    nb, b = 8, 8
    n = nb * b
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)

    def run(key, plan, budget_tiles=24):
        matrix.RESIDENCY.reset()
        matrix.RESIDENCY.set_budget(budget_tiles * b * b * 8)
        A = BigMatrix(key, shape=(n, n), shard_sizes=(b, b), write_header=True)
        for i in range(nb):
            for j in range(i + 1):
                A.put_block(a[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
        program, meta = alg_wrappers.cholesky(A)
        program.config["executor"]["reclaim_intermediates"] = True
        program.config["executor"]["spill_plan"] = plan
        program.start()
        job_runner.lambdapack_run(program)
        program.wait()
        assert program.program_status() == lp.PS.SUCCESS
        st = matrix.RESIDENCY.stats()
        L = meta["outputs"][0].numpy()
        program.free()
        A.free()
        return L, st

    L_lru, lru = run("res_plan_off", False)
    L_plan, plan = run("res_plan_on", True)
    assert lru["policy"] == "lru" and plan["policy"] == "plan"
    assert np.array_equal(L_lru, L_plan) and np.allclose(np.tril(L_plan), np.linalg.cholesky(a))
    assert plan["restores"] * 2 <= lru["restores"] and plan["evictions"] < lru["evictions"], (plan, lru)
    # with half the budget, tiles that cannot stay until their next read are copied out as they are stored (write-through) and
    # pushing them out later is free: the backend sees "spill_free" calls; same factor again
    del oracle_backend.calls[:]
    L_tight, tight = run("res_plan_tight", True, budget_tiles=12)
    assert np.array_equal(L_tight, L_plan) and tight["written_through"] > 0
    assert [c[0] for c in oracle_backend.calls].count("spill_free") > 0
    matrix.RESIDENCY.reset()


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_spill_and_restore_are_bit_exact_and_asynchronous(hbm_store):
    from numpywren_amd.device import get_backend
    be = get_backend()
    rng = np.random.default_rng(0)
    a = rng.standard_normal((1024, 1024))
    t = be.to_device(a)
    prod = be.gemm(t, t)                      # spilling must wait for the producer on another stream
    sp = be.spill_to_host(prod)
    assert isinstance(sp, SpilledTile) and sp.nbytes == prod.nbytes
    back = be.restore_from_host(sp)
    want = be.to_host(prod)
    assert np.array_equal(be.to_host(back), want)
    assert np.array_equal(be.spilled_to_numpy(sp), want)
    # pinned buffers are pooled
    before = be.pinned_bytes
    del sp, back
    sp2 = be.spill_to_host(prod)
    assert be.pinned_bytes == before
    assert np.array_equal(be.spilled_to_numpy(sp2), want)
    del sp2
    be.trim_pinned()
    assert be.pinned_bytes == 0


@pytest.mark.gpu
def test_budgeted_cholesky_matches_unbudgeted_bitwise(hbm_store):
    """Same program, same kernels, same order: a budget of a few tiles changes where tiles wait, not one bit of L."""
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    from numpywren_amd.device import get_backend
    be = get_backend()
    n, b = 1536, 256
    rng = np.random.default_rng(9)
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)

    def run(key, budget):
        matrix.RESIDENCY.set_budget(budget)
        A = BigMatrix(key, shape=(n, n), shard_sizes=(b, b), write_header=True)
        for i in range(n // b):
            for j in range(n // b):
                A.put_block(a[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
        program, meta = alg_wrappers.cholesky(A)
        program.start()
        job_runner.lambdapack_run(program)
        program.wait()
        assert program.program_status() == lp.PS.SUCCESS
        L = meta["outputs"][0].numpy()
        st = matrix.RESIDENCY.stats()
        program.free()
        A.free()
        return L, st

    L0, st0 = run("res_gpu_free", None)
    assert st0["evictions"] == 0
    L1, st1 = run("res_gpu_tight", 8 * b * b * 8)
    assert st1["evictions"] > 20 and st1["restores"] > 20
    assert np.array_equal(L0, L1)
    assert np.linalg.norm(L1 @ L1.T - a) / np.linalg.norm(a) < 1e-13
    matrix.RESIDENCY.set_budget(None)
    be.trim_pinned()


@pytest.mark.gpu
def test_allocation_failure_pushes_tiles_out(hbm_store, monkeypatch):
    """Drive the out-of-memory path without filling 288 GB: the first npw_malloc of a fresh size is made to fail."""
    import ctypes
    from numpywren_amd.device import get_backend
    be = get_backend()
    m = BigMatrix("res_gpu_oom", shape=(1024, 256), shard_sizes=(256, 256), write_header=True)
    rng = np.random.default_rng(1)
    ref = [rng.standard_normal((256, 256)) for _ in range(4)]
    for i, r in enumerate(ref):
        m.put_block(r, i, 0)
    be.synchronize()
    real = be.lib.npw_malloc
    fails = {"left": 2}

    class Lib(object):
        def __getattr__(self, name):
            return getattr(be_lib, name)

        def npw_malloc(self, p, nbytes):
            if nbytes == 3 * 256 * 256 * 8 + 256 and fails["left"] > 0:
                fails["left"] -= 1
                return -2
            return real(p, nbytes)

    be_lib = be.lib
    monkeypatch.setattr(be, "lib", Lib())
    t = be.empty((3 * 256 * 256 + 32,), np.float64)   # fails twice -> trim, then reclaim, then succeeds
    assert t.nbytes == 3 * 256 * 256 * 8 + 256 and fails["left"] == 0
    monkeypatch.undo()
    st = matrix.RESIDENCY.stats()
    assert st["evictions"] == 4 and st["resident_bytes"] == 0      # 4 tiles of 512 KiB < the 1.5 MiB asked for
    for i, r in enumerate(ref):
        assert np.array_equal(be.to_host(m.get_tile(i, 0)), r)
    assert matrix.RESIDENCY.stats()["restores"] == 4
    m.free()
    be.trim_pinned()


def test_plan_of_a_resumed_run_skips_finished_tasks_and_is_released_at_the_end(oracle_backend):
    """a second lambdapack_run on a program that stopped on its time limit builds its plan from all tasks; the
    FINISHED ones are never issued again, so they must count as issued (their reads are not "upcoming").  When a run ends
    the process-wide tier must not keep the executor's plan or the factory bound to it."""
    from numpywren_amd import alg_wrappers, job_runner
    from numpywren_amd import lambdapack as lp
    rng = np.random.default_rng(5)
    n, b = 64, 16
    x = rng.standard_normal((n, n))
    a = x @ x.T + n * np.eye(n)
    A = BigMatrix("res_resume_in", shape=(n, n), shard_sizes=(b, b), write_header=True)
    matrix.RESIDENCY.set_budget(6 * TILE)
    for i in range(4):
        for j in range(4):
            A.put_block(a[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
    program, meta = alg_wrappers.cholesky(A)
    program.start()
    ex = job_runner.LambdaPackExecutor(program, pipeline_width=1)
    for _ in range(7):
        e, v = program.dequeue()
        program.set_node_status(e, v, lp.NS.RUNNING)
        ex.run_task(e, v)
        program.post_op(e, v, lp.PS.SUCCESS, None)
        program.set_node_status(e, v, lp.NS.FINISHED)
    ex.release_spill_plan()
    ex2 = job_runner.LambdaPackExecutor(program, pipeline_width=1)      # what the second lambdapack_run creates
    plan = ex2.spill_plan
    assert plan is not None and matrix.RESIDENCY.plan is plan
    finished = [t for t in program.program.tasks if program.get_node_status(t.expr_idx, t.vars) == lp.NS.FINISHED]
    assert len(finished) == 7 and all(plan.done[plan.pos[t.index]] for t in finished)
    assert sum(plan.done) == 7
    job_runner.lambdapack_run(program, _executor=ex2)
    program.wait()
    assert program.program_status() == lp.PS.SUCCESS
    assert np.allclose(meta["outputs"][0].numpy(), np.linalg.cholesky(a))
    assert matrix.RESIDENCY.plan is None and matrix.RESIDENCY.plan_factory is None
    program.free()
    # without a budget the plan is only a factory during the run, and gone afterwards
    matrix.RESIDENCY.set_budget(None)
    program, meta = alg_wrappers.cholesky(A)
    program.start()
    job_runner.lambdapack_run(program)
    assert matrix.RESIDENCY.plan is None and matrix.RESIDENCY.plan_factory is None
    program.free()
