"""LambdaPACK runtime objects: the per-task instruction IR and the program state machine.

Same class names and call surface as the reference (numpywren/lambdapack.py:199-776) so drivers
written against it keep working:

    program = LambdaPackProgram(compiled_program, config)
    program.start(); job_runner.lambdapack_run(program, ...); program.wait(); program.free()

What changed underneath: the reference keeps node/edge state in Redis (atomic
`conditional_increment`, lambdapack.py:154-196) and hands ready tasks to workers through SQS;
here both are in-process -- edge sets and counters behind a lock, a priority heap of ready
tasks ordered by critical-path length -- and a single `lambdapack_run` call drives the whole
DAG on HIP streams.  Exceptions are kept on the program object (status EXCEPTION + traceback)
instead of being written to S3.
"""
import heapq
import itertools
import threading
import time
import traceback
from enum import Enum

import numpy as np

from .exceptions import LambdaPackTimeoutException


class RemoteInstructionOpCodes(Enum):
    S3_LOAD = 0
    S3_WRITE = 1
    GENERIC = 3
    RET = 4


class NodeStatus(Enum):
    NOT_READY = 0
    READY = 1
    RUNNING = 2
    POST_OP = 3
    FINISHED = 4


class EdgeStatus(Enum):
    NOT_READY = 0
    READY = 1


class ProgramStatus(Enum):
    SUCCESS = 0
    RUNNING = 1
    EXCEPTION = 2
    NOT_STARTED = 3


OC = RemoteInstructionOpCodes
NS = NodeStatus
ES = EdgeStatus
PS = ProgramStatus


# ------------------------------------------------------------------------------------------------
# instruction IR (reference lambdapack.py:199-474)
# ------------------------------------------------------------------------------------------------
class RemoteInstruction(object):
    def __init__(self, i_id):
        self.id = i_id
        self.ret_code = -1
        self.start_time = None
        self.end_time = None
        self.type = None
        self.executor = None
        self.cache = None
        self.run = False
        self.read_size = 0
        self.write_size = 0

    def get_flops(self):
        return 0

    def clear(self):
        self.result = None


def _tile_bytes(matrix):
    return int(np.prod(matrix.shard_sizes)) * np.dtype(matrix.dtype).itemsize


class RemoteRead(RemoteInstruction):
    """Fetch tile `bidxs` of `matrix` (the reference's S3_LOAD).  Awaiting / calling it yields the
    ndarray exactly as BigMatrix.get_block returns it."""

    def __init__(self, i_id, matrix, *bidxs):
        super().__init__(i_id)
        self.i_code = OC.S3_LOAD
        self.matrix = matrix
        self.bidxs = bidxs
        self.result = None
        self.cache_hit = False
        self.read_size = _tile_bytes(matrix)

    def execute(self):
        self.start_time = time.time()
        if self.result is None:
            key = (self.matrix.key, self.matrix.bucket, self.matrix.true_block_idx(*self.bidxs))
            if self.cache is not None and key in self.cache:
                self.result = self.cache[key]
                self.cache_hit = True
            else:
                self.result = self.matrix.get_block(*self.bidxs)
                if self.cache is not None:
                    self.cache[key] = self.result
        self.end_time = time.time()
        return self.result

    async def __call__(self):
        return self.execute()

    def clear(self):
        self.result = None

    def __str__(self):
        return "{0} = S3_LOAD {1} {2} {3}".format(self.id, self.matrix, len(self.bidxs),
                                                  " ".join(str(x) for x in self.bidxs))


class RemoteWrite(RemoteInstruction):
    def __init__(self, i_id, matrix, data_loc, data_idx, *bidxs):
        super().__init__(i_id)
        self.i_code = OC.S3_WRITE
        self.matrix = matrix
        self.bidxs = bidxs
        self.data_loc = data_loc
        self.data_idx = data_idx
        self.result = None
        self.sparse_write = False
        self.write_size = _tile_bytes(matrix)

    def execute(self, skip_empty=False):
        self.start_time = time.time()
        if self.result is None:
            data = self.data_loc[self.data_idx]
            if self.cache is not None:
                key = (self.matrix.key, self.matrix.bucket, self.matrix.true_block_idx(*self.bidxs))
                self.cache[key] = data
            if skip_empty and np.allclose(data, 0):
                self.sparse_write = True
            else:
                self.result = self.matrix.put_block(data, *self.bidxs)
            self.ret_code = 0
        self.end_time = time.time()
        return self.result

    async def __call__(self, skip_empty=False):
        return self.execute(skip_empty)

    def clear(self):
        self.result = None
        self.data_loc = None

    def __str__(self):
        return "{0} = S3_WRITE {1} {2} {3} {4}".format(self.id, self.matrix, len(self.bidxs),
                                                      " ".join(str(x) for x in self.bidxs), self.data_idx)


class RemoteCall(RemoteInstruction):
    def __init__(self, i_id, compute, argv_instr, num_outputs, symbols, **kwargs):
        super().__init__(i_id)
        self.i_code = OC.GENERIC
        self.results = [None for _ in range(num_outputs)]
        self.kwargs = kwargs
        self.compute = compute
        self.symbols = symbols
        self.argv_instr = argv_instr

    def _pyargs(self):
        out = []
        for arg in self.argv_instr:
            if isinstance(arg, RemoteRead):
                out.append(arg.result)
            elif isinstance(arg, (float, int)):
                out.append(arg)
        return out

    def execute(self):
        self.start_time = time.time()
        results = self.compute(*self._pyargs(), **self.kwargs)
        if isinstance(results, tuple) and len(results) != len(self.results):
            raise Exception("Expected {0} results, got {1}".format(len(self.results), len(results)))
        elif isinstance(results, tuple):
            for i, r in enumerate(results):
                self.results[i] = r
        else:
            self.results[0] = results
        self.ret_code = 0
        self.end_time = time.time()
        return self.results

    async def __call__(self, prev=None):
        if prev is not None:
            await prev
        return self.execute()

    def clear(self):
        self.results = [None for _ in self.results]
        self.argv_instr = [None for _ in self.argv_instr]

    def get_flops(self):
        flops = getattr(self.compute, "flops", None)
        return flops(*self._pyargs()) if flops is not None else 0

    def __str__(self):
        outs = ",".join(str(i + len(self.symbols)) for i in range(len(self.results)))
        return "{1} = {0}({2}, **kwargs)".format(self.compute, outs, ",".join(self.symbols))


class RemoteReturn(RemoteInstruction):
    def __init__(self, i_id):
        super().__init__(i_id)
        self.i_code = OC.RET
        self.result = None

    def __str__(self):
        return "RET"


class InstructionBlock(object):
    block_count = 0

    def __init__(self, instrs, label=None, priority=0):
        self.instrs = instrs
        self.label = label
        self.priority = priority
        if self.label is None:
            self.label = "%{0}".format(InstructionBlock.block_count)
        InstructionBlock.block_count += 1

    def execute(self, skip_empty=False):
        """Run reads, call and writes sequentially on the host API (ndarray path)."""
        out = []
        for ins in self.instrs:
            out.append(ins.execute(skip_empty) if isinstance(ins, RemoteWrite) else ins.execute())
        return out

    def __str__(self):
        return self.label + "\n" + "".join("\t" + str(i) + "\n" for i in self.instrs)

    def clear(self):
        [x.clear() for x in self.instrs]

    def total_flops(self):
        return sum(getattr(x, "flops", 0) for x in self.instrs)

    def total_io(self):
        return sum(getattr(x, "size", 0) for x in self.instrs)

    def __copy__(self):
        return InstructionBlock(self.instrs.copy(), self.label)


# ------------------------------------------------------------------------------------------------
# program state machine (reference lambdapack.py:477-776)
# ------------------------------------------------------------------------------------------------
_hash_counter = itertools.count()


class LambdaPackProgram(object):
    """Global state of one run of a compiled program: node status, dependency counts, the ready
    queue, progress / flop / byte counters and the final status."""

    def __init__(self, program, config=None, num_priorities=1, eager=False, block_sparse=False):
        self.config = config if config is not None else {}
        self.program = program
        self.block_sparse = block_sparse
        self.max_priority = num_priorities - 1
        self.eager = eager
        self.bucket = "hbm"
        self.hash = "{0}_{1}".format(int(time.time()), next(_hash_counter))
        self.up = 'up' + self.hash
        self._lock = threading.RLock()
        self._status = PS.NOT_STARTED
        self._node_status = {}
        self._edges = {}            # child key -> set of finished parent keys
        self._ready = []            # heap of (-priority, seq, (expr_idx, vars))
        self._seq = itertools.count()
        self._counters = {}
        self._finished_terminators = set()
        self._priority = None
        self.exceptions = {}
        self.profiles = {}          # node -> per-task record (what the reference pickles to S3 after every task)
        self.info_flags = []        # (device int32 flag, node) pairs checked at completion (Cholesky info)
        self._defer_success = False  # the async runner reports SUCCESS only after the GPU has drained
        self._success_pending = False
        self.set_up(0)

    # ---- keys (same string forms as the reference, lambdapack.py:507-519) ----
    def _node_str(self, expr_idx, var_values):
        # (memoised: the walk asks for a node's string several times per task -- a quarter of its host time.  The key
        #  keeps the dict's own item order, so two orders of one node are two entries with the same string.)
        ck = (expr_idx, tuple(var_values.items()))
        cache = self.__dict__.setdefault("_nstr_cache", {})
        out = cache.get(ck)
        if out is None:
            var_strs = sorted(["{0}:{1}".format(key, value) for key, value in var_values.items()])
            out = cache[ck] = "{0}_({1})".format(expr_idx, "-".join(var_strs))
        return out

    def _node_key(self, expr_idx, var_values):
        return "{0}_{1}".format(self.hash, self._node_str(expr_idx, var_values))

    def _node_edge_sum_key(self, expr_idx, var_values):
        return "{0}_{1}_edgesum".format(self.hash, self._node_str(expr_idx, var_values))

    def _edge_key(self, expr_idx1, var_values1, expr_idx2, var_values2):
        return "{0}_{1}_{2}".format(self.hash, self._node_str(expr_idx1, var_values1),
                                    self._node_str(expr_idx2, var_values2))

    # ---- node status ----
    def get_node_status(self, expr_idx, var_values):
        with self._lock:
            return self._node_status.get(self._node_str(expr_idx, var_values), NS.NOT_READY)

    def set_node_status(self, expr_id, var_values, status):
        with self._lock:
            self._node_status[self._node_str(expr_id, var_values)] = status
        return status

    # ---- scheduling priority: length of the longest path to a sink ----
    def _priorities(self):
        if self._priority is None:
            prio = {}
            tasks = getattr(self.program, "tasks", None)
            if tasks is not None:
                for t in reversed(tasks):  # program order is a topological order (SSA, writers first)
                    best = 0
                    for c in t.children:
                        best = max(best, prio.get(c.key, 0))
                    prio[t.key] = best + 1
            self._priority = prio
        return self._priority

    def _enqueue(self, node):
        from .compiler import node_key
        prio = self._priorities().get(node_key(node[0], node[1]), 0)
        with self._lock:
            heapq.heappush(self._ready, (-prio, next(self._seq), (int(node[0]), dict(node[1]))))

    def dequeue(self):
        """Highest-priority ready task or None."""
        with self._lock:
            if not self._ready:
                return None
            return heapq.heappop(self._ready)[2]

    def dequeue_matching(self, pred, limit):
        """Up to `limit` further ready tasks with pred(expr_idx, vars) true, best priority first (the executor uses
        this to run independent tasks of one kind as a single batched launch)."""
        if limit <= 0:
            return []
        with self._lock:
            taken, kept = [], []
            for item in sorted(self._ready):
                if len(taken) < limit and pred(item[2][0], item[2][1]):
                    taken.append(item[2])
                else:
                    kept.append(item)
            if taken:
                self._ready = kept
                heapq.heapify(self._ready)
            return taken

    def enables(self, e, v, expr_idx):
        """True when task (e, v) is the LAST missing parent of a task of statement `expr_idx`."""
        if e == expr_idx:
            return False
        with self._lock:
            me = self._node_str(e, v)
            for child in self.program.find_children(e, v):
                if child[0] != expr_idx:
                    continue
                edges = self._edges.get(self._node_str(*child), ())
                if me not in edges and len(edges) == len(self.program.find_parents(child[0], child[1])) - 1:
                    return True
        return False

    def dequeue_enablers(self, expr_idx, limit=64):
        """Ready tasks (removed from the heap, best priority first) whose completion makes a task of statement
        `expr_idx` ready -- i.e. they are the LAST missing parent of such a task.  The executor runs them before a
        batched task of that statement so that its siblings join the batch (the trsm tasks of one block column of the
        Cholesky DAG become ready one by one as their trailing updates are issued).  Any order of ready tasks is a valid
        schedule; this one only changes which of them goes first."""
        with self._lock:
            items = sorted(self._ready)
            taken, kept = [], []
            for item in items:
                e, v = item[2]
                ok = len(taken) < limit and self.enables(e, v, expr_idx)
                (taken if ok else kept).append(item)
            if taken:
                self._ready = kept
                heapq.heapify(self._ready)
        return [t[2] for t in taken]

    def num_ready(self):
        with self._lock:
            return len(self._ready)

    # ---- run control ----
    def start(self, parallel=False):
        with self._lock:
            self._status = PS.RUNNING
            # a fresh run, whatever an earlier one on this object left behind (a program that is merely to be RESUMED
            # after a timed-out lambdapack_run is handed to lambdapack_run again, not to start())
            self._node_status = {}
            self._edges = {}
            self._ready = []
            self._finished_terminators = set()
            self._success_pending = False
            self.info_flags = []       # (deferred LinAlgError flags of an abandoned run must not fail this one)
        # no worker has been up yet (wait() tells "not started" from "worker gone" by this), and partial sums of an
        # aborted fused-GEMM run (job_runner.ReductionFusion) must not be accumulated into
        self._was_up = False
        self._drop_fusion_accumulators()
        seeds = self.program.starters
        tasks = getattr(self.program, "tasks", None)
        if tasks is not None:
            # an instance is runnable once all tiles it reads exist: seed the nodes without parents
            # (the reference seeds every `starters` entry; both coincide for its four algorithms)
            seeds = [t.node for t in tasks if not t.parents]
        for s in seeds:
            self.set_node_status(s[0], s[1], NS.READY)
            self._enqueue(s)
        if not seeds and self.program.num_terminators == 0:
            self.return_success()
        return 0

    def resume(self, finished):
        """A fresh run state in which the tasks `finished` ([(expr_idx, var_values)]: e.g. checkpoint.finished_nodes of an
        interrupted run whose tiles are back in the store) count as done: start(), then their dependency accounting is
        replayed -- each one is taken off the ready heap and post_op'ed, which releases its children -- without running them.
        What is left on the heap is what lambdapack_run continues with.  (In the reference this state IS durable -- Redis
        edge counters, lambdapack.py:545-639 -- and nothing has to be replayed.)"""
        from .compiler import node_key
        done = {node_key(e, v) for e, v in finished}
        self.start()
        held, replayed = [], 0
        while True:
            node = self.dequeue()
            if node is None:
                break
            e, v = node
            if node_key(e, v) in done:
                self.set_node_status(e, v, NS.RUNNING)
                self.post_op(e, v, PS.SUCCESS, None)
                self.set_node_status(e, v, NS.FINISHED)
                replayed += 1
            else:
                held.append(node)
        for node in held:
            self._enqueue(node)
        if replayed != len(done):
            raise ValueError("resume: {0} of the {1} finished tasks are not reachable through finished parents".format(
                len(done) - replayed, len(done)))
        return replayed

    def post_op(self, expr_idx, var_values, ret_code, inst_block, tb=None):
        """Dependency accounting after a task: mark its out-edges, enqueue children whose parents
        are all done, count terminators (reference lambdapack.py:545-639)."""
        try:
            post_op_start = time.time()
            task_of = getattr(self.program, "task", None)
            if task_of is not None:     # the expanded DAG: children and their parent counts without further look-ups
                kids = task_of(expr_idx, var_values).children
                children = [c.node for c in kids]
                nparents = [len(c.parents) for c in kids]
            else:
                children = self.program.find_children(expr_idx, var_values)
                nparents = None
            self.set_node_status(expr_idx, var_values, NS.POST_OP)
            if ret_code == PS.EXCEPTION and tb is not None:
                self.handle_exception(" EXCEPTION", tb=tb, expr_idx=expr_idx, var_values=var_values)
            me = self._node_str(expr_idx, var_values)
            ready_children = []
            for ci, child in enumerate(children):
                ckey = self._node_str(*child)
                with self._lock:
                    edges = self._edges.setdefault(ckey, set())
                    edges.add(me)  # idempotent: a replayed task never double counts
                    val = len(edges)
                num_child_parents = nparents[ci] if nparents is not None else len(self.program.find_parents(child[0], child[1]))
                if val == num_child_parents and self.get_node_status(*child) not in (NS.FINISHED, NS.READY,
                                                                                      NS.RUNNING, NS.POST_OP):
                    self.set_node_status(child[0], child[1], NS.READY)
                    ready_children.append(child)
            next_operator = None
            if self.eager and ready_children:
                next_operator = ready_children.pop()
            for child in ready_children:
                self._enqueue(child)
            if inst_block is not None:
                inst_block.end_time = time.time()
                inst_block.clear()
                inst_block.post_op_start = post_op_start
                inst_block.post_op_end = time.time()
                inst_block.expr_idx = expr_idx
                inst_block.var_values = var_values
            self.incr_progress()
            if self.program.is_terminator(expr_idx):
                with self._lock:
                    self._finished_terminators.add(me)
                    done = len(self._finished_terminators)
                if done == self.program.num_terminators and self.program_status() == PS.RUNNING:
                    self.return_success()
            return next_operator, None
        except Exception:
            tb = traceback.format_exc()
            self.handle_exception("POST OP EXCEPTION", tb=tb, expr_idx=expr_idx, var_values=var_values)
            raise

    def stop(self):
        self.exceptions["DRIVER.CANCELLED"] = "cancelled by driver"
        with self._lock:
            self._status = PS.EXCEPTION

    def return_success(self):
        with self._lock:
            if self._defer_success:
                self._success_pending = True
            else:
                self._status = PS.SUCCESS

    def all_terminators_done(self):
        with self._lock:
            return self._success_pending or self._status == PS.SUCCESS

    def handle_exception(self, error, tb, expr_idx, var_values):
        self.exceptions[self._node_str(expr_idx, var_values)] = (tb or "") + str(error)
        with self._lock:
            self._status = PS.EXCEPTION

    def program_status(self):
        with self._lock:
            return self._status

    def wait(self, sleep_time=1):
        finish = getattr(self, "_finish", None)
        if finish is not None:      # job_runner.lambdapack_run(..., wait=False): settle the run here
            finish()
        status = self.program_status()
        while status == PS.RUNNING:
            if self.get_up() == 0 and getattr(self, "_was_up", False):
                # A worker drove this program and left it unfinished (lambdapack_run left its loop on a timeout or on a
                # stalled DAG) and, unlike the reference's fleet, no other worker will pick it up: say so instead of
                # sleeping for ever.  The status stays RUNNING -- calling lambdapack_run again resumes the program --
                # and a wait() that merely runs before the worker of another thread has come up keeps waiting.
                with self._lock:
                    pending = len(self._ready)
                raise LambdaPackTimeoutException(
                    "program is still RUNNING but no worker is up ({0} ready tasks left): lambdapack_run timed out or the "
                    "DAG stalled; call lambdapack_run again to resume".format(pending))
            time.sleep(sleep_time)
            status = self.program_status()

    def _drop_fusion_accumulators(self):
        acc = self.__dict__.get("_fusion_acc")
        if acc:
            acc.clear()   # (in place: a live executor's ReductionFusion holds the same dict)

    def free(self):
        with self._lock:
            self._ready = []
            self._edges = {}
        self._drop_fusion_accumulators()
        marks, self.completion_marks = getattr(self, "completion_marks", None), []
        if marks:   # events of a lambdapack_run(wait=False): nobody may be told to wait for them after free()
            from .device import get_backend
            be = get_backend()
            for ev in marks:
                be.recycle_event(ev)

    # ---- counters (reference lambdapack.py:683-752; Redis keys become dict entries) ----
    def _incr(self, name, amount=1):
        with self._lock:
            self._counters[name] = self._counters.get(name, 0) + amount
            return self._counters[name]

    def _get(self, name):
        with self._lock:
            return self._counters.get(name, 0)

    def incr_up(self, amount):
        self._was_up = True
        self._incr(self.up, amount)

    def decr_up(self, amount):
        self._incr(self.up, -amount)

    def get_up(self):
        return self._get(self.up)

    def set_up(self, value):
        with self._lock:
            self._counters[self.up] = value

    def incr_repeated_compute(self, amount=1):
        self._incr("repeated_compute", amount)

    def incr_repeated_post_op(self, amount=1):
        self._incr("repeated_post_op", amount)

    def incr_repeated_finish(self, amount=1):
        self._incr("repeated_finish", amount)

    def incr_not_ready(self, amount=1):
        self._incr("not_ready", amount)

    def incr_progress(self):
        self._incr("progress")

    # ---- per-task profiling records (reference lambdapack.py:531-533, 765-776: one pickled instruction block per
    # node under s3://<bucket>/lambdapack/<hash>/<expr>_<vars>) ----
    def record_profile(self, expr_idx, var_values, **info):
        with self._lock:
            self.profiles[self._node_str(expr_idx, var_values)] = dict(info, expr_idx=int(expr_idx), var_values=dict(var_values))

    def get_profiling_info(self, expr_idx, var_values):
        """The record of one executed task: kernel name, stream, host enqueue window (the GPU runs asynchronously: use
        rocprofv3 for device times), bytes read / written, flops, size of the batch it ran in."""
        return self.profiles[self._node_str(expr_idx, var_values)]

    def get_all_profiling_info(self):
        with self._lock:
            return list(self.profiles.values())

    def dump_profiling_info(self, inst_block, expr_idx, var_values):
        import pickle
        return pickle.dumps(self.profiles.get(self._node_str(expr_idx, var_values)))

    async def begin_write(self, loop=None):
        return None

    async def begin_read(self, loop=None):
        return None

    def incr_flops(self, amount):
        if amount > 0:
            self._incr("flops", amount)

    def incr_read(self, amount):
        if amount > 0:
            self._incr("read", amount)

    def incr_sparse_read(self, amount):
        if amount > 0:
            self._incr("sparse_read", amount)

    def incr_write(self, amount):
        if amount > 0:
            self._incr("write", amount)

    def incr_sparse_write(self, amount):
        if amount > 0:
            self._incr("write_sparse", amount)

    def decr_flops(self, amount):
        if amount > 0:
            self._incr("flops", -amount)

    def decr_read(self, amount):
        if amount > 0:
            self._incr("read", -amount)

    def decr_write(self, amount):
        if amount > 0:
            self._incr("write", -amount)

    def get_flops(self):
        return self._get("flops")

    def get_read(self):
        return self._get("read")

    def get_write(self):
        return self._get("write")

    def get_progress(self):
        return self._get("progress")
