"""The worker: drives a LambdaPackProgram's DAG on a pool of HIP streams.

Counterpart of the reference's job_runner.py (numpywren/job_runner.py:78-159, 224-370).  There a
worker is a stateless Lambda / process that pulls one task id at a time from SQS and pushes it
through a 3-stage asyncio pipeline  read (S3 GET) -> compute (NumPy on a thread pool) -> write
(S3 PUT)  of depth `pipeline_width`; many workers run concurrently and synchronise through
Redis.  Here ONE call of `lambdapack_run` executes the whole program on the local MI355X:

  * tiles are already in HBM, so "read" is a table lookup (+ H2D only for host-tier tiles) and
    "write" stores the produced DeviceTile handle -- no copies;
  * "compute" is asynchronous HIP kernels; `pipeline_width` is the number of HIP streams the
    tasks are spread over, plus one high-priority stream for the latency-bound panel kernels
    (chol / trsm / qr_factor) so the critical path overtakes queued trailing updates (lookahead);
  * cross-stream dependencies are HIP events carried by the tiles -- the host never blocks on
    the GPU except to bound how far it runs ahead (`max_inflight`) and at the end of the run;
  * the host walks the DAG in critical-path order from LambdaPackProgram's ready heap, so
    `post_op` bookkeeping (pure Python dict updates, a few microseconds) replaces the
    reference's ~28 ms of sympy + Redis per task.

The returned dict has the reference's keys (up_time, exec_time, executed_messages,
operator_refs, log).
"""
import collections
import pickle
import time
import traceback

import numpy as np

from . import kernels
from . import lambdapack as lp
from .device import DeviceTile, get_backend


class LRUCache(object):
    """Tile cache with the reference's interface (numpywren/job_runner.py:34-64).  Tiles resident in
    HBM make it unnecessary for the device path; it is kept for the host-API instruction blocks."""

    def __init__(self, max_items=10):
        self.cache = collections.OrderedDict()
        self.max_items = max_items

    def __setitem__(self, key, value):
        self.cache[key] = value
        self.cache.move_to_end(key, last=False)
        while len(self.cache) > self.max_items:
            self.cache.popitem(last=True)

    def __getitem__(self, key):
        value = self.cache[key]
        self.cache.move_to_end(key, last=False)
        return value

    def __contains__(self, key):
        return key in self.cache


def calculate_busy_time(rtimes):
    """Union of [start, end] intervals (reference job_runner.py:162-175)."""
    events = sorted([(s, 1) for s, _ in rtimes] + [(e, -1) for _, e in rtimes])
    running, out, cur = 0, [], 0
    for t, d in events:
        if running == 0 and d == 1:
            cur = t
        if running == 1 and d == -1:
            out.append([cur, t])
        running += d
    return out


class ReductionFusion(object):
    """`executor.fuse_gemm_reduction`: the GEMM program's `Temp` + `add_matrices` tree as an accumulation in place.

    The reference computes C[i, j] as K separate tile products Temp[i, j, k, 0] = gemm(A[i, k], B[k, j]) followed by a
    fan-in-4 tree of add_matrices tasks (reference algs.py:251-266, kernels.py:16-20): every partial product is written
    to and read back from the object store.  With tiles resident in HBM the same sum is one buffer per C tile that every
    product accumulates into (beta = 1 on the GEMM's own accumulator type: fp32 products of fp32 tiles accumulate in
    fp32), and the tree's tasks have nothing left to do but hand that buffer on; the last one converts it to float64,
    the dtype add_matrices promotes to, so the output matrix has the reference's dtype either way.  Which tasks fuse is
    read off the compiled DAG, not off the program's name: an add_matrices task fuses when each of its operands is (a)
    the output of a `gemm` task, or of an add_matrices task that fuses itself, which nobody else reads, or (b) a tile no
    task writes in a matrix whose parent_fn is constant_zeros (the tree's padding).  The default -- the parity mode --
    materialises every Temp tile like the reference."""

    def __init__(self, compiled, acc=None):
        self.compiled = compiled
        readers = collections.defaultdict(list)
        for t in compiled.tasks:
            for r in set(t.reads):
                readers[r].append(t)
        name = lambda t: getattr(compiled.kernel(t.expr_idx), "__name__", "")
        fus = {}

        def zero_pad(r):
            m = compiled.matrices[r[0]]
            return (compiled.writer_of(*r) is None and r[0] not in compiled.inputs
                    and getattr(getattr(m, "parent_fn", None), "_npw_zero_shape", None) is not None)

        def fusable(t):
            if t.index in fus:
                return fus[t.index]
            ok = name(t) == "add_matrices" and len(t.writes) == 1 and not t.kwargs
            if ok:
                for r in t.reads:
                    w = compiled.writer_of(*r)
                    if w is None:
                        ok = zero_pad(r)
                    else:
                        only_me = all(x is t for x in readers[r])
                        ok = only_me and len(w.writes) == 1 and ((name(w) == "gemm" and len(w.reads) == 2) or fusable(w))
                    if not ok:
                        break
            fus[t.index] = ok
            return ok

        self.root_of = {}        # task index (gemm leaf or fused add) -> index of the add task that stores the sum
        self.roots = {}          # root add task index -> number of products it sums
        for t in compiled.tasks:
            if not fusable(t):
                continue
            consumers = readers.get(t.writes[0], [])
            if len(consumers) == 1 and fusable(consumers[0]) and t.writes[0] in consumers[0].reads:
                continue         # an inner node of the tree
            stack, leaves = [t], 0
            while stack:
                a = stack.pop()
                self.root_of[a.index] = t.index
                for r in a.reads:
                    w = compiled.writer_of(*r)
                    if w is None:
                        continue
                    if name(w) == "gemm":
                        self.root_of[w.index] = t.index
                        leaves += 1
                    else:
                        stack.append(w)
            self.roots[t.index] = leaves
        # root index -> the accumulating DeviceTile.  Kept by the caller (the program) across runs: a run that stops on
        # its time limit is resumed by a new executor, which must find the sums of the products already done.
        self.acc = acc if acc is not None else {}

    def handles(self, task):
        return task.index in self.root_of

    def run(self, ex, task, compute, stream):
        """The fused counterpart of one task; returns the tile the task stores (None: nothing)."""
        be = ex.be
        root = self.root_of[task.index]
        mats = self.compiled.matrices
        if getattr(compute, "__name__", "") == "gemm":
            tiles = [mats[m].get_tile(*idx, stream=stream) for m, idx in task.reads]
            args = [tiles[j] if kind == "tile" else task.consts[j] for kind, j in task.arg_kinds]
            A, B = [a for a in args if isinstance(a, DeviceTile)][:2]
            acc = self.acc.get(root)
            ta, tb = bool(task.kwargs.get("transpose_A", False)), bool(task.kwargs.get("transpose_B", False))
            if acc is None:
                self.acc[root] = be.gemm(A, B, ta, tb, stream)
            else:
                self.acc[root] = be.gemm(A, B, ta, tb, stream, alpha=1.0, beta=1.0, C=acc, out=acc)
            flops_fn = getattr(compute, "flops", None)
            if flops_fn is not None:
                ex.program.incr_flops(flops_fn(A, B))
            ex.program.incr_read(sum(t.nbytes for t in tiles))
            return None
        if task.index != root:
            return None                      # an inner add: its sum is still travelling in the accumulator
        acc = self.acc.pop(root)
        out = be.as_f64(acc, stream)         # add_matrices promotes to float64 (reference quirk a5)
        (m, idx), = task.writes
        mats[m].put_tile(out, *idx)
        ex.program.incr_write(out.nbytes)
        return out



class LambdaPackExecutor(object):
    """Executes single tasks of a program on the HIP backend."""
    dry = False           # True for the recording stand-in of predicted_issue_order: nothing touches the device

    def __init__(self, program, loop=None, cache=None, read_queue=None, pipeline_width=4, exact_zero=None,
                 is_local=None, send_plan=None):
        self.program = program
        self.cache = cache
        self.is_local = is_local      # multi-GPU: predicate "this rank executes the task"
        self.send_plan = send_plan    # multi-GPU: task -> {output position: remote consumer ranks}
        self.be = get_backend()
        cfg = (program.config or {}).get("executor", {}) if isinstance(program.config, dict) else {}
        self.exact_zero = cfg.get("exact_zero_shortcircuit", True) if exact_zero is None else exact_zero
        self.reclaim = cfg.get("reclaim_intermediates", False)
        # tiles NO task reads (TSQR's V / T factors) are part of what the reference's wrapper returns (alg_wrappers.py:47):
        # they are only dropped on store when the caller asks for it by name (an R-only TSQR)
        self.drop_unread = bool(cfg.get("drop_unread_outputs", False))
        self.batch_tasks = max(1, int(cfg.get("batch_tasks", 32)))
        self.spill_batch = None       # cap on the batches of throughput kernels while the host-DRAM tier is at work (below)
        # diagnostics (off by default: two event records per task cost ~8 us of stream time): every task's kernels are
        # bracketed with timing events on their stream; collect_task_times() adds them up by kernel name
        self.task_timers = bool(cfg.get("task_timers", False)) and hasattr(self.be, "new_event")
        # profiler ranges (roctx through the C-ABI's npw_range_push / _pop): one host-side range per task or batch, named
        # "<kernel>(<node>)", around the calls that enqueue its kernels -- `rocprofv3 --marker-trace` then shows tasks beside
        # kernels.  The counterpart of the reference's per-instruction start / end records (lambdapack.py:210-211, 361, 379).
        self.roctx = bool(cfg.get("roctx_ranges", False)) and hasattr(self.be, "lib") and hasattr(self.be.lib, "npw_range_push")
        self._task_times = program.__dict__.setdefault("_task_times", [])
        pool = getattr(self.be, "bulk_streams", None) or self.be.streams
        n = max(1, min(int(pipeline_width), len(pool)))
        self.streams = pool[:n]
        self.prio_stream = self.be.priority_stream if cfg.get("priority_stream", False) else None
        self._rr = 0
        self.compiled = program.program
        self._readers_left = None
        self._unread = set()
        # chain partition (one in-order stream only): see run_chain
        self.chain_cus = int(cfg.get("chain_cus", 0) or 0) if n == 1 and self.prio_stream is None else 0
        if self.chain_cus and not (hasattr(self.be, "chain_streams") and 0 < self.chain_cus < getattr(self.be, "compute_units", 0)):
            self.chain_cus = 0
        self.chain_runs = 0
        self.chain_stmts = []
        self.fusion = None
        if cfg.get("fuse_gemm_reduction", False) and not program.block_sparse and is_local is None:
            fusion = ReductionFusion(self.compiled, program.__dict__.setdefault("_fusion_acc", {}))
            self.fusion = fusion if fusion.roots else None
        if self.chain_cus:
            self.chain_stmts = [i for i, k in getattr(self.compiled, "_kernels", {}).items()
                                if getattr(k, "_npw_chain_resident_cus", None) is not None]
            if not self.chain_stmts:
                self.chain_cus = 0
        # the host-DRAM tier reads the static DAG (residency.SpillPlan): victims by farthest next read, spilled operands
        # of the next `spill_prefetch_tasks` tasks copied back ahead of them
        self.spill_plan = None
        self.prefetch_tasks = max(0, int(cfg.get("spill_prefetch_tasks", 2)))
        self._pipeline_width = pipeline_width
        self._issued_idx = []          # task indices issued so far (a plan built late is told about them)
        if hasattr(self.be, "restore_from_host"):
            from . import matrix
            res = matrix.RESIDENCY
            # "at work": a byte budget is set, or tiles of THIS program's matrices sit in host memory right now (an earlier
            # eviction somewhere in the process's past does not make every later, fully resident run pay for small batches)
            tier_at_work = matrix._store_tier() != "host" and (res.budget is not None or self._has_spilled_tiles())
            if tier_at_work:
                # (also in the dry walk: the plan must see the batches the real run forms)
                self.spill_batch = max(1, int(cfg.get("spill_batch_tasks", 8)))
            if not self.dry and cfg.get("spill_plan", True) and matrix._store_tier() != "host" and getattr(self.compiled, "tasks", None):
                if tier_at_work:
                    self._install_spill_plan()          # the tier is at work: plan before the first task
                else:
                    # nothing has ever been pushed out: the plan (a dry walk of the scheduling loop, ~0.07 ms per task) is
                    # only made if the allocator's out-of-memory handler asks the tier for memory during this run
                    res.plan = None
                    res.last_policy = "lru"
                    res.plan_factory = self._install_spill_plan
            elif not self.dry:
                # no plan wanted (executor.spill_plan off, the host tier): whatever an earlier run left behind must not steer
                # this one's evictions
                res.plan = res.plan_factory = None
                res.last_policy = "lru"

    def release_spill_plan(self):
        """The run is over: the process-wide residency tier must not keep this executor's plan, nor the bound method that
        would build it (it holds the executor and the program alive, and a later out-of-memory reclaim would dry-walk a
        finished program and install a plan in which every tile's next use is "never")."""
        try:
            from . import matrix
            res = matrix.RESIDENCY
        except Exception:
            return
        if self.spill_plan is not None and getattr(res, "plan", None) is self.spill_plan:
            res.plan = None
        fac = getattr(res, "plan_factory", None)
        if fac is not None and getattr(fac, "__self__", None) is self:
            res.plan_factory = None

    def _has_spilled_tiles(self):
        from .device import SpilledTile
        for m in getattr(self.compiled, "matrices", {}).values():
            tiles = m._tiles(False) if hasattr(m, "_tiles") else None
            if tiles and any(isinstance(t, SpilledTile) for t in list(tiles.values())):
                return True
        return False

    def _install_spill_plan(self):
        from . import matrix
        from .residency import SpillPlan
        matrix.RESIDENCY.plan_factory = None
        tasks = self.compiled.tasks
        order = None
        if self.is_local is None:
            try:
                order = predicted_issue_order(self.program, self._pipeline_width)
            except Exception:
                order = None
        if order is None or len(order) != len(tasks):
            # (several ranks, or a program the dry walk cannot finish: critical-path order, the ready heap's own key)
            prio = self.program._priorities()
            order = sorted(tasks, key=lambda t: (-prio.get(t.key, 0), t.index))
        mats = self.compiled.matrices

        def key_of(name, idx):
            m = mats[name]
            return (m.bucket, m.key_base, m.__shard_idx_to_key__(idx))
        try:
            self.spill_plan = SpillPlan(order, key_of)
            # a resumed program (a second lambdapack_run after a time limit, checkpoint.load + resume): the tasks that are
            # FINISHED will never be issued by this run -- their reads must not count as upcoming (prefetch would bring
            # back dead tiles, the farthest-next-read rule would keep them)
            seen = set(self._issued_idx)
            for t in tasks:
                if t.index not in seen and self.program.get_node_status(t.expr_idx, t.vars) == lp.NS.FINISHED:
                    self.spill_plan.issued(t.index)
            for i in self._issued_idx:
                self.spill_plan.issued(i)
        except Exception:
            self.spill_plan = None     # (a matrix without the tile-key surface: the tier falls back to LRU)
        matrix.RESIDENCY.plan = self.spill_plan
        matrix.RESIDENCY.last_policy = "plan" if self.spill_plan is not None else "lru"
        return self.spill_plan

    def _issued(self, tasks):
        """The reads of `tasks` have been taken (their kernels hold the tiles): tell the plan, then bring back what the
        next tasks need.  Only with a budget / after an eviction is there anything to prefetch."""
        plan = self.spill_plan
        if plan is None:
            if not self.dry:
                self._issued_idx.extend(t.index for t in tasks)
            return
        from . import matrix
        for t in tasks:
            plan.issued(t.index)
        res = matrix.RESIDENCY
        if self.prefetch_tasks and res.evictions and res.plan is plan:   # (nothing ever pushed out: nothing to bring back)
            res.prefetch(self.be, self.prefetch_tasks)

    def _range(self, compute, nodes):
        """Context manager: a named profiler range around the enqueue of `nodes` (no-op unless executor.roctx_ranges)."""
        import contextlib
        if not self.roctx:
            return contextlib.nullcontext()
        lib = self.be.lib
        e, v = nodes[0]
        name = "{0}({1}{2})".format(getattr(compute, "__name__", "task"), self.program._node_str(e, v),
                                    "" if len(nodes) == 1 else " +%d" % (len(nodes) - 1))

        @contextlib.contextmanager
        def scope():
            lib.npw_range_push(name.encode())
            try:
                yield
            finally:
                lib.npw_range_pop()
        return scope()

    # ---- per-task device timing (executor.task_timers) ----
    def _tic(self, stream):
        if not self.task_timers:
            return None
        ev = self.be.new_event(timing=True)
        self.be.record(ev, stream)
        return ev

    def _toc(self, name, stream, ev0, count=1):
        if ev0 is None:
            return
        ev1 = self.be.new_event(timing=True)
        self.be.record(ev1, stream)
        self._task_times.append((name, ev0, ev1, count))

    # ---- stream choice ----
    def pick_stream(self, compute):
        if self.prio_stream is not None and getattr(compute, "_npw_latency_bound", False):
            return self.prio_stream
        s = self.streams[self._rr % len(self.streams)]
        self._rr += 1
        return s

    # ---- reclaim bookkeeping: tiles of non-input / non-output matrices die after their last reader.  A tile of such a
    #      matrix that NO task reads (the V and T factors of the TSQR program: only R goes up the tree) is kept -- the
    #      reference persists it -- unless `drop_unread_outputs` says the caller wants the compiled outputs only ----
    def _init_reclaim(self):
        left = collections.Counter()
        keep = set(self.compiled.inputs) | set(self.compiled.outputs)
        read_by_anyone = set()
        for t in self.compiled.tasks:
            read_by_anyone.update(t.reads)
        self._unread = ({w for t in self.compiled.tasks for w in t.writes if w[0] not in keep and w not in read_by_anyone}
                        if self.drop_unread else set())
        for t in self.compiled.tasks:
            local = self.is_local is None or self.is_local(t)
            if local and self.program.get_node_status(t.expr_idx, t.vars) == lp.NS.FINISHED:
                continue        # (a resumed run -- LambdaPackProgram.resume: this reader has had its turn)
            if local:
                for r in set(t.reads):
                    if r[0] not in keep and self.compiled.writer_of(*r) is not None:
                        left[r] += 1
                if self.send_plan is not None:
                    # a tile that still has to be pushed to another GPU is kept until it has been sent
                    for pos in self.send_plan(t):
                        w = t.writes[pos]
                        if w[0] not in keep:
                            left[w] += 1
        self._readers_left = left

    def sent(self, name, idx):
        """Multi-GPU: the tile has been handed to the transport for all its remote consumers."""
        if not self.reclaim:
            return
        if self._readers_left is None:
            self._init_reclaim()
        w = (name, tuple(idx))
        if w in self._readers_left:
            self._readers_left[w] -= 1
            if self._readers_left[w] == 0:
                self.compiled.matrices[name].delete_block(*idx)

    def _consumed(self, task):
        if not self.reclaim:
            return
        if self._readers_left is None:
            self._init_reclaim()
        for r in set(task.reads):
            if r in self._readers_left:
                self._readers_left[r] -= 1
                if self._readers_left[r] == 0:
                    self.compiled.matrices[r[0]].delete_block(*r[1])
        for w in task.writes:
            if w in self._unread:
                self.compiled.matrices[w[0]].delete_block(*w[1])

    def _unwanted(self, tasks):
        """Per task: the output positions whose tiles `_consumed` will drop as soon as they are stored (nobody reads them
        and the caller asked for `drop_unread_outputs`), or None when nothing is dropped.  A kernel may return None there."""
        if not (self.reclaim and self.drop_unread):
            return None
        if self._readers_left is None:
            self._init_reclaim()
        if not self._unread:
            return None
        return [{pos for pos, w in enumerate(t.writes) if w in self._unread} for t in tasks]

    # ---- the panel chain beside trailing updates ----
    def chain_companions(self, expr_idx, var_values):
        """For a ready task whose kernel needs whole CUs and whose workgroups wait for one another (kernels.chol): the
        ready throughput tasks (removed from the heap, best priority first) to issue beside it, or [] when the task
        should run on the full chip (no chain partition configured, the tile's chain does not fit it, nothing else is
        ready).  Ready tasks are independent of one another, so any of them may run concurrently with the chain."""
        if not self.chain_cus:
            return []
        compute = self.compiled.kernel(expr_idx)
        need = getattr(compute, "_npw_chain_resident_cus", None)
        if need is None:
            return []
        task = self.compiled.task(expr_idx, var_values)
        try:
            m, idx = task.reads[0]
            # rows of the tile: the second to last axis (the Cholesky program's S[version, j, k] has its version axis first --
            # axis 0 made every diagonal tile but the first look one row tall, and the check below always passed)
            ranges = self.compiled.matrices[m].__block_idx_to_real_idx__(idx)
            s0, e0 = ranges[-2] if len(ranges) >= 2 else ranges[0]
            offered = self.chain_cus
            if not self.dry and hasattr(self.be, "stream_cus"):
                # what the chain stream really offers a resident-grid kernel: fewer than its mask while an RCCL
                # communicator is live in this process (compute units are left to its transfer kernels, npw_hip.h) --
                # the factorisation then runs on the full stream instead of failing on the partition
                offered = self.be.stream_cus(self.be.chain_streams(self.chain_cus)[0])[1]
            if need(self.be, int(e0 - s0)) > offered:
                return []
        except Exception:
            return []
        # Fill the window (weight 1.0: the factorisation on its 64 CUs takes as long as one off-diagonal update on the
        # other 192) without running far past it -- the full chip waits for BOTH partitions.  Best: one whole-window
        # update (2.85 ms beside a 2.85 ms chol); then two light ones (x is y: 0.69 each) sharing one batched launch
        # (measured 3.2 ms for the pair there); then a single light one.
        def weigh(e2, v2):
            w = getattr(self.compiled.kernel(e2), "_npw_chain_weight", None)
            return None if w is None else w(self.compiled.task(e2, v2))

        def light(e2, v2):
            w = weigh(e2, v2)
            return w is not None and w < 0.95

        def full(e2, v2):
            w = weigh(e2, v2)
            return w is not None and w >= 0.95

        picked = self.program.dequeue_matching(full, 1)
        if picked:
            return picked
        pair = self.program.dequeue_matching(light, 2) if self.batch_tasks > 1 else []
        if len(pair) == 2 and not (pair[0][0] == pair[1][0] and self.batch_fn(pair[0][0]) is not None):
            self.program._enqueue(pair.pop())
        return pair or self.program.dequeue_matching(light, 1)

    def is_chain_task(self, expr_idx):
        return self.chain_cus > 0 and getattr(self.compiled.kernel(expr_idx), "_npw_chain_resident_cus", None) is not None

    def run_chain(self, node, companions):
        """`node` on the chain stream (chain_cus compute units), `companions` on the stream masked to the other CUs;
        the in-order stream resumes when the chain task is done.  CDNA4 does not preempt and the chain's workgroups
        need a CU's whole LDS, so sharing CUs with ~1 ms GEMM workgroups would stall every one of its launches
        (profiles/r01_overlap_study.md); a static partition for the duration of the task does not.  Data dependencies
        need nothing extra: tiles carry their producers' events across streams."""
        be = self.be
        full = self.streams[0]
        chain, rest = be.chain_streams(self.chain_cus)
        # both partitions start when the full chip has drained and the full chip resumes when both are done: the chain's
        # CUs must stay free of ~1 ms GEMM workgroups, and two chip-filling GEMMs sharing CUs run slower than one after
        # the other (measured: a 1024-tile and a 496-tile syrk side by side 3.65 ms, in sequence 3.0)
        ev = be.record_new(full)
        be.wait_event(chain, ev)
        be.wait_event(rest, ev)
        be.recycle_event(ev)
        out = [self.run_task(node[0], node[1], stream=chain)]
        if len(companions) > 1 and self.batch_fn(companions[0][0]) is not None and all(c[0] == companions[0][0] for c in companions):
            out.append(self.run_batch(companions, stream=rest))
        else:
            for e, v in companions:
                out.append(self.run_task(e, v, stream=rest))
        for part in (chain, rest):
            ev = be.record_new(part)
            be.wait_event(full, ev)
            be.recycle_event(ev)
        self.chain_runs += 1
        return out

    # ---- a kernel whose workgroups wait for one another gets the device to itself ----
    def _fence_in(self, compute, stream):
        """With several streams: `stream` waits for the tails of the others (returned, for _fence_out).  Two kernels of
        this kind side by side -- the Cholesky panel chain, the QR panel kernel of a batch: every workgroup of a launch
        has to be resident -- can each hold the slots the other one is waiting for (workgroups are dealt round-robin
        to the XCDs, so BOTH launches can be partially resident at once); the bounded spins would then time out and
        the results would be wrong, not late."""
        if len(self.streams) <= 1 or not getattr(compute, "_npw_needs_whole_cus", False):
            return []
        others = [s for s in self.streams if s is not stream]
        for o in others:
            ev = self.be.record_new(o)
            self.be.wait_event(stream, ev)
            self.be.recycle_event(ev)   # (a stream wait captures the event's state when it is enqueued)
        return others

    def _fence_out(self, others, stream):
        if others:
            ev = self.be.record_new(stream)
            for o in others:
                self.be.wait_event(o, ev)
            self.be.recycle_event(ev)

    # ---- one task ----
    def run_task(self, expr_idx, var_values, stream=None):
        t_enq = time.time()
        task = self.compiled.task(expr_idx, var_values)
        compute = self.compiled.kernel(expr_idx)
        mats = self.compiled.matrices
        if stream is None:
            stream = self.pick_stream(compute)
        if self.fusion is not None and self.fusion.handles(task):
            last = self.fusion.run(self, task, compute, stream)
            self._issued((task,))
            self._consumed(task)
            self.program.record_profile(expr_idx, var_values, kernel=getattr(compute, "__name__", str(compute)) + "+fused",
                                        stream=getattr(stream, "name", str(stream)), enqueue_start=t_enq,
                                        enqueue_end=time.time(), read_bytes=0, write_bytes=0, batch=1)
            return last
        device_kernel = getattr(compute, "_npw_device_kernel", False)
        # A kernel whose workgroups need a whole CU to themselves (the Cholesky panel chain: 150 KiB of LDS each)
        # starves next to chip-filling GEMMs of other streams -- every launch then waits for ~1 ms workgroups to
        # retire.  With several streams such a task gets the device to itself: its stream first waits for the other
        # streams' tails, and they wait for it afterwards.
        others = self._fence_in(compute, stream)
        tiles = [mats[m].get_tile(*idx, stream=stream) for m, idx in task.reads]
        self._issued((task,))
        read_bytes = sum(t.nbytes for t in tiles)
        if device_kernel:
            args = [tiles[j] if kind == "tile" else task.consts[j] for kind, j in task.arg_kinds]
            tic = self._tic(stream)
            with self._range(compute, [(expr_idx, var_values)]), \
                    kernels.stream_scope(stream, self.program.info_flags_sink(task), self.exact_zero, self._unwanted([task])):
                results = compute(*args, **task.kwargs)
            self._toc(getattr(compute, "__name__", "kernel"), stream, tic)
        else:
            # arbitrary Python callable from the DSL's scope: give it ndarrays, like the reference does
            host = [self.be.to_host(t, stream) for t in tiles]
            args = [host[j] if kind == "tile" else task.consts[j] for kind, j in task.arg_kinds]
            results = compute(*args, **task.kwargs)
        self._fence_out(others, stream)
        flops_fn = getattr(compute, "flops", None)
        if flops_fn is not None:
            try:
                self.program.incr_flops(flops_fn(*[a for a in args if not isinstance(a, (int, float))]))
            except Exception:
                pass
        if isinstance(results, tuple):
            if len(results) != len(task.writes):
                raise Exception("Expected {0} results, got {1}".format(len(task.writes), len(results)))
        else:
            results = (results,)
            if len(task.writes) != 1:
                raise Exception("Expected {0} results, got {1}".format(len(task.writes), 1))
        write_bytes = 0
        last = None
        for (m, idx), r in zip(task.writes, results):
            if r is None and (m, idx) in self._unread:
                continue   # a tile nobody reads, dropped on store anyway: the kernel did not compute it
            if isinstance(r, DeviceTile):
                if self.program.block_sparse and self.be.read_flag(self.be.zero_flag(r, stream), stream):
                    self.program.incr_sparse_write(r.nbytes)
                    continue
                mats[m].put_tile(r, *idx)
                write_bytes += r.nbytes
                last = r
            else:
                r = np.asarray(r)
                if self.program.block_sparse and np.allclose(r, 0):
                    self.program.incr_sparse_write(r.nbytes)
                    continue
                mats[m].put_block(r, *idx)
                write_bytes += r.nbytes
        self.program.incr_read(read_bytes)
        self.program.incr_write(write_bytes)
        self._consumed(task)
        self.program.record_profile(expr_idx, var_values, kernel=getattr(compute, "__name__", str(compute)),
                                    stream=getattr(stream, "name", str(stream)), enqueue_start=t_enq, enqueue_end=time.time(),
                                    read_bytes=read_bytes, write_bytes=write_bytes, batch=1)
        return last

    # ---- several independent tasks of one kind as one batched kernel call ----
    def batch_limit(self, expr_idx):
        """Most tasks of statement `expr_idx` one batched call takes: executor.batch_tasks, or fewer for a throughput kernel
        while the host-DRAM tier is at work (a batched launch waits for the copy-in of all its operands)."""
        if self.spill_batch is not None and not getattr(self.compiled.kernel(expr_idx), "_npw_needs_whole_cus", False):
            return min(self.batch_tasks, self.spill_batch)
        return self.batch_tasks

    def batch_fn(self, expr_idx):
        """The batched implementation of the task's kernel, or None (no batching configured / kernel has none)."""
        if self.batch_tasks <= 1 or self.program.block_sparse:
            return None
        kernel = self.compiled.kernel(expr_idx)
        if self.fusion is not None and getattr(kernel, "__name__", "") == "gemm":
            return None      # fused GEMM reduction: every product goes through ReductionFusion.run, one by one
        return getattr(kernel, "_npw_batch", None)

    def run_batch(self, nodes, stream=None):
        """Run the ready tasks `nodes` (all of the same expr_idx, whose kernel has a `_npw_batch`) with one call.
        Same reads, writes and bookkeeping as run_task for each of them."""
        t_enq = time.time()
        expr_idx = nodes[0][0]
        compute = self.compiled.kernel(expr_idx)
        mats = self.compiled.matrices
        if stream is None:
            stream = self.pick_stream(compute)
        tasks = [self.compiled.task(e, v) for e, v in nodes]
        arg_lists, kwargs_list, read_bytes = [], [], 0
        for task in tasks:
            tiles = [mats[m].get_tile(*idx, stream=stream) for m, idx in task.reads]
            read_bytes += sum(t.nbytes for t in tiles)
            arg_lists.append([tiles[j] if kind == "tile" else task.consts[j] for kind, j in task.arg_kinds])
            kwargs_list.append(task.kwargs)
        self._issued(tasks)
        others = self._fence_in(compute, stream)
        tic = self._tic(stream)
        with self._range(compute, nodes), \
                kernels.stream_scope(stream, self.program.info_flags_sink(tasks[0]), self.exact_zero, self._unwanted(tasks)):
            results = compute._npw_batch(self.be, stream, arg_lists, kwargs_list)
        self._toc(getattr(compute, "__name__", "kernel"), stream, tic, len(tasks))
        self._fence_out(others, stream)
        flops_fn = getattr(compute, "flops", None)
        last, write_bytes = None, 0
        for task, args, res in zip(tasks, arg_lists, results):
            if flops_fn is not None:
                try:
                    self.program.incr_flops(flops_fn(*[a for a in args if not isinstance(a, (int, float))]))
                except Exception:
                    pass
            res = res if isinstance(res, tuple) else (res,)
            if len(res) != len(task.writes):
                raise Exception("Expected {0} results, got {1}".format(len(task.writes), len(res)))
            for (m, idx), r in zip(task.writes, res):
                if r is None and (m, idx) in self._unread:
                    continue   # dropped on store anyway: not computed
                mats[m].put_tile(r, *idx)
                write_bytes += r.nbytes
                last = r
            self._consumed(task)
        self.program.incr_read(read_bytes)
        self.program.incr_write(write_bytes)
        t_end = time.time()
        for (e, v), task in zip(nodes, tasks):
            self.program.record_profile(e, v, kernel=getattr(compute, "__name__", str(compute)),
                                        stream=getattr(stream, "name", str(stream)), enqueue_start=t_enq, enqueue_end=t_end,
                                        read_bytes=None, write_bytes=None, batch=len(nodes))
        return last

    async def run(self, expr_idx, var_values, computer=None, profile=True):
        """Reference-shaped entry point (job_runner.py:78): run one task and its eager successors."""
        refs = [(expr_idx, var_values)]
        done = []
        for e, v in refs:
            status = self.program.get_node_status(e, v)
            if status in (lp.NS.READY, lp.NS.RUNNING):
                self.program.set_node_status(e, v, lp.NS.RUNNING)
                self.run_task(e, v)
                nxt, _ = self.program.post_op(e, v, lp.PS.SUCCESS, None)
                self.program.set_node_status(e, v, lp.NS.FINISHED)
                done.append((e, v))
                if nxt is not None:
                    refs.append(nxt)
            elif status == lp.NS.FINISHED:
                self.program.incr_repeated_finish()
            elif status == lp.NS.NOT_READY:
                self.program.incr_not_ready()
        return [(d, None) for d in done]


def _info_sink(program, task):
    class _Sink(list):
        def append(self_inner, flag):
            program.info_flags.append((flag, task.node))

    return _Sink()


lp.LambdaPackProgram.info_flags_sink = lambda self, task: _info_sink(self, task)


def _drain_task_times(program, synchronise=True):
    """Turn the pending (name, start event, stop event, count) records of executor.task_timers into totals on the program
    and hand the events back to the backend's timing pool -- called when a run settles, so that neither the records nor
    their events pile up over runs nobody collects."""
    from .device import get_backend
    be = get_backend()
    records = program.__dict__.get("_task_times") or []
    totals = program.__dict__.setdefault("_task_time_totals", {})
    if records:
        if synchronise:
            be.synchronize()
        for name, ev0, ev1, count in records:
            slot = totals.setdefault(name, {"tasks": 0, "ms": 0.0})
            slot["tasks"] += count
            slot["ms"] += be.elapsed_ms(ev0, ev1)
            be.recycle_event(ev0)
            be.recycle_event(ev1)
        del records[:]
    return totals


def collect_task_times(program):
    """{kernel name: {"tasks": n, "ms": total device time}} of the tasks run with executor.task_timers since the last call
    (synchronises the device; with several streams the brackets of concurrent tasks overlap, so the sum can exceed the
    wall time)."""
    out = dict(_drain_task_times(program))
    program.__dict__["_task_time_totals"] = {}
    return out


def check_handoffs(program, be):
    """A run with Householder factorisations: did a hand-off wait inside the panel kernel expire (libnpw_hip.so bounds them
    so that a lost hand-off cannot hang the GPU)?  Then some factorisation returned undefined numbers: fail the program.
    The counter is one per PROCESS (a device global): with two wait=False programs in flight, or direct be.geqrt callers
    beside a program, an expired wait is charged to whichever run settles first -- the attribution is per process, not
    per program; what matters is that it is never lost.  Callers evaluate this check whatever the info flags said
    (`settle_checks`), so a failing run does not leave its count behind for the next one."""
    kinds = getattr(program.program, "_kernels", {}) or {}
    if not hasattr(be, "qr_handoff_timeouts") or not any(getattr(k, "_npw_handoff", False) for k in kinds.values()):
        return True
    lost = be.qr_handoff_timeouts(reset=True)
    if lost:
        program.handle_exception(RuntimeError("{0} hand-off waits of the QR panel kernel expired during this run: the "
                                              "factorisations' results are undefined".format(lost)),
                                 tb="", expr_idx=-1, var_values={})
        return False
    return True


def check_info_flags(program, be, stream=None):
    """Deferred np.linalg.LinAlgError: read back the Cholesky info flags of the run (the caller has synchronised with
    their producers; `stream` only carries the copies)."""
    flags, program.info_flags = program.info_flags, []
    if hasattr(be, "read_flags"):
        codes = be.read_flags([f for f, _ in flags], stream)
    else:
        codes = [be.read_flag(f) for f, _ in flags]
    for (flag, node), code in zip(flags, codes):
        if code != 0:
            msg = "Matrix is not positive definite (leading minor of order {0} in task {1})".format(code, node)
            program.handle_exception(np.linalg.LinAlgError(msg), tb="", expr_idx=node[0], var_values=node[1])
            return False
    return True


class _RecordingExecutor(LambdaPackExecutor):
    """The executor's scheduling decisions without its work: run_task / run_batch / run_chain only note which task would
    have been issued.  lambdapack_run driven by this stand-in on a shadow LambdaPackProgram IS the real loop -- the
    batches it forms, the tasks it pulls ahead to complete a batch, the panel factorisations it moves to the chain
    partition -- so the recorded order is the order of the real run, not a model of it."""
    dry = True

    def __init__(self, program, pipeline_width):
        LambdaPackExecutor.__init__(self, program, pipeline_width=pipeline_width)
        self.order = []

    def run_task(self, expr_idx, var_values, stream=None):
        self.order.append(self.compiled.task(expr_idx, var_values))
        return None

    def run_batch(self, nodes, stream=None):
        self.order.extend(self.compiled.task(e, v) for e, v in nodes)
        return None

    def run_chain(self, node, companions):
        self.order.append(self.compiled.task(*node))
        self.order.extend(self.compiled.task(e, v) for e, v in companions)
        self.chain_runs += 1
        return []


def predicted_issue_order(program, pipeline_width=1):
    """The tasks of `program` in the order lambdapack_run will issue them (a dry walk of the same loop on a shadow of the
    program's run state; nothing is enqueued on the device).  The host-DRAM tier's SpillPlan is built from it."""
    shadow = lp.LambdaPackProgram(program.program, config=program.config, block_sparse=program.block_sparse)
    shadow._priority = program._priorities()
    shadow.start()
    ex = _RecordingExecutor(shadow, pipeline_width)
    lambdapack_run(shadow, pipeline_width=pipeline_width, timeout=1e9, _executor=ex)
    return ex.order


def settle_checks(program, be, stream=None):
    """Both deferred checks of a settled run, ALWAYS both (no short-circuit: the hand-off counter must be read and reset
    even when the info flags already failed the program)."""
    flags_ok = check_info_flags(program, be, stream) if stream is not None else check_info_flags(program, be)
    handoffs_ok = check_handoffs(program, be)
    return flags_ok and handoffs_ok


def lambdapack_run(program, pipeline_width=1, msg_vis_timeout=60, cache_size=5, timeout=200, idle_timeout=5,
                   msg_vis_timeout_jitter=15, compute_threads=1, max_inflight=64, wait=True, after=None, _executor=None):
    """Run `program` to completion (or until `timeout` seconds) on the local GPU.

    pipeline_width -> number of HIP streams; the SQS visibility / idle / thread arguments of the
    reference are accepted and ignored (no queue service, no worker fleet).  Returns the reference's
    result dict.

    wait=False returns as soon as every task has been enqueued on the device; `program.wait()` -- the next call of
    the reference's sequence start / lambdapack_run / wait / free -- then synchronises, evaluates the deferred
    LinAlgError flags and settles the status.  A caller that factors one matrix after another can enqueue the next
    program before waiting for the previous one, so the GPU does not idle during the host-side turnaround.
    `after` (a list of events, e.g. the `completion_marks` of an earlier run) makes every stream of this run wait for
    them on the DEVICE first: two runs then never overlap on the GPU although the host has enqueued both."""
    program.incr_up(1)
    t_start = time.time()
    be = get_backend()
    be.bind_thread()
    ex = _executor if _executor is not None else LambdaPackExecutor(
        program, cache=LRUCache(cache_size) if cache_size > 0 else None, pipeline_width=pipeline_width)
    dry = ex.dry           # predicted_issue_order: the loop below decides, nothing is enqueued
    executed, refs, running_times = [], [], []
    inflight = collections.deque()

    def track(last):
        """Bound the host's run-ahead (and the tile references it pins) in EVERY branch of the loop."""
        if last is not None and last.ready is not None:
            inflight.append(last)
        while len(inflight) > max_inflight:
            be.wait_tile(inflight.popleft())

    program._defer_success = True
    if after and not dry:
        chain_pair = list(be.chain_streams(ex.chain_cus)) if ex.chain_cus else []
        for sh in list(ex.streams) + ([ex.prio_stream] if ex.prio_stream is not None else []) + chain_pair:
            for ev in after:
                be.wait_event(sh, ev)
    try:
        while program.program_status() == lp.PS.RUNNING and not program.all_terminators_done():
            node = program.dequeue()
            if node is None:
                break
            if time.time() - t_start > timeout:
                program._enqueue(node)
                break
            e, v = node
            if ex.chain_cus and not ex.is_chain_task(e):
                # with a chain partition a ready panel factorisation goes first: it then runs beside the trailing updates
                # that are ready with it (equal critical-path priority would issue those first, in ready order)
                first = program.dequeue_matching(lambda e2, v2: ex.is_chain_task(e2), 1)
                if first:
                    program._enqueue(node)
                    e, v = first[0]
            t0 = time.time()
            if ex.batch_fn(e) is not None and getattr(ex.compiled.kernel(e), "_npw_batch_gather", False):
                # A batched kind whose siblings become ready one at a time (the trsm tasks of a block column wait for
                # their trailing updates): first issue the ready tasks that are the last missing parent of a sibling,
                # then come back to this task -- by then its siblings are in the heap and join the batch.
                enablers = program.dequeue_enablers(e)
                if enablers:
                    program._enqueue(node)
                    for ne, nv in enablers:
                        program.set_node_status(ne, nv, lp.NS.RUNNING)
                        try:
                            last = ex.run_task(ne, nv)
                        except Exception as exc:
                            program.handle_exception(exc, tb=traceback.format_exc(), expr_idx=ne, var_values=nv)
                            raise
                        program.post_op(ne, nv, lp.PS.SUCCESS, None)
                        program.set_node_status(ne, nv, lp.NS.FINISHED)
                        executed.append([ne, nv])
                        refs.append((ne, nv))
                        track(last)
                    running_times.append((t0, time.time()))
                    continue
            companions = ex.chain_companions(e, v)
            if companions:
                chain_group = [(e, v)] + companions
                for ge, gv in chain_group:
                    program.set_node_status(ge, gv, lp.NS.RUNNING)
                try:
                    lasts = ex.run_chain((e, v), companions)
                except Exception as exc:
                    program.handle_exception(exc, tb=traceback.format_exc(), expr_idx=e, var_values=v)
                    raise
                for ge, gv in chain_group:
                    program.post_op(ge, gv, lp.PS.SUCCESS, None)
                    program.set_node_status(ge, gv, lp.NS.FINISHED)
                    executed.append([ge, gv])
                    refs.append((ge, gv))
                running_times.append((t0, time.time()))
                for last in lasts:
                    track(last)
                continue
            if ex.chain_cus and ex.batch_fn(e) is not None:
                # before a batch swallows the ready tasks of this kind: the one whose completion makes a panel
                # factorisation ready goes first and alone -- the factorisation then finds the others still queued and
                # takes one of them along as its companion
                first = [(e, v)] if any(program.enables(e, v, stmt) for stmt in ex.chain_stmts) else []
                if not first:
                    for stmt in ex.chain_stmts:
                        first = program.dequeue_enablers(stmt, limit=1)
                        if first:
                            program._enqueue((e, v))
                            break
                if first:
                    e, v = first[0]
                    program.set_node_status(e, v, lp.NS.RUNNING)
                    try:
                        last = ex.run_task(e, v)
                    except Exception as exc:
                        program.handle_exception(exc, tb=traceback.format_exc(), expr_idx=e, var_values=v)
                        raise
                    program.post_op(e, v, lp.PS.SUCCESS, None)
                    program.set_node_status(e, v, lp.NS.FINISHED)
                    executed.append([e, v])
                    refs.append((e, v))
                    running_times.append((t0, time.time()))
                    track(last)
                    continue
            # independent ready tasks of the same kind (TSQR leaves, the nodes of a tree level, the trailing updates of a
            # block column) go to the device as ONE batched launch sequence
            group = [(e, v)]
            if ex.batch_fn(e) is not None:
                group += program.dequeue_matching(lambda e2, v2: e2 == e, ex.batch_limit(e) - 1)
            for ge, gv in group:
                program.set_node_status(ge, gv, lp.NS.RUNNING)
            try:
                last = ex.run_batch(group) if len(group) > 1 else ex.run_task(e, v)
            except Exception as exc:
                tb = traceback.format_exc()
                program.handle_exception(exc, tb=tb, expr_idx=e, var_values=v)
                raise
            # stream order + tile events already encode "children run after me" on the device, so
            # the host can release the children immediately
            for ge, gv in group:
                program.post_op(ge, gv, lp.PS.SUCCESS, None)
                program.set_node_status(ge, gv, lp.NS.FINISHED)
                executed.append([ge, gv])
                refs.append((ge, gv))
            running_times.append((t0, time.time()))
            track(last)
        if dry:
            program._defer_success = False
            return {"executed_messages": executed, "operator_refs": refs}
        # completion marks of THIS run: one event per stream it used (a device-wide synchronise would also wait for
        # whatever a pipelining caller has enqueued behind it)
        used = list(ex.streams) + ([ex.prio_stream] if ex.prio_stream is not None else [])
        if ex.chain_runs:
            used += list(be.chain_streams(ex.chain_cus))
        marks = [be.record_new(sh) for sh in used] if hasattr(be, "record_new") and hasattr(be, "event_sync") else None
        program.completion_marks = list(marks) if (marks is not None and not wait) else []

        def finish():
            program._finish = None
            try:
                if marks is None or wait:
                    be.synchronize()
                    ok = settle_checks(program, be)
                    for ev in (marks or []):
                        be.recycle_event(ev)
                else:
                    for ev in marks:
                        be.event_sync(ev)   # (events stay alive: a later run may still be told to wait for them)
                    ok = settle_checks(program, be, be.flag_stream())
                if ex.task_timers:
                    _drain_task_times(program, synchronise=False)   # (their events precede the marks just waited for)
                program._defer_success = False
                if ok and program._success_pending and program.program_status() == lp.PS.RUNNING:
                    program.return_success()
            finally:
                program._defer_success = False

        if wait:
            finish()
        else:
            program._finish = finish
    except BaseException:
        program._defer_success = False
        raise
    finally:
        program.decr_up(1)
        ex.release_spill_plan()
    t_stop = time.time()
    return {"up_time": [t_start, t_stop],
            "exec_time": calculate_busy_time(running_times),
            "executed_messages": executed,
            "operator_refs": refs,
            "log": pickle.dumps({})}


async def lambdapack_run_async(loop, program, computer=None, cache=None, shared_state=None, read_queue=None, pipeline_width=1,
                               msg_vis_timeout=60, timeout=200, msg_vis_timeout_jitter=15):
    """Coroutine form of the worker loop (reference job_runner.py:418-: one of `pipeline_width` SQS pollers sharing an
    event loop).  Here one call drives the whole DAG; `shared_state["running_times"]`, when given, receives the busy
    intervals as the reference's pollers record them."""
    res = lambdapack_run(program, pipeline_width=pipeline_width, timeout=timeout)
    if isinstance(shared_state, dict):
        shared_state.setdefault("running_times", []).extend(res["exec_time"])
    return res


def lambdapack_run_with_failures(failure_key, program, pipeline_width=5, msg_vis_timeout=60, cache_size=5,
                                 timeout=200, idle_timeout=5, msg_vis_timeout_jitter=15):
    """Signature-compatible with the reference's fault-injection driver (job_runner.py:190-221); a
    single-process run has no remote workers to kill, so this is a plain run."""
    return lambdapack_run(program, pipeline_width=pipeline_width, cache_size=cache_size, timeout=timeout)
