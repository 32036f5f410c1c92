"""Multi-GPU execution of a LambdaPACK program on one node: one process per GPU, tiles sharded
2-D block-cyclically, panel tiles exchanged point-to-point over RCCL / xGMI.

The reference has no worker-to-worker communication at all -- every tile goes through S3 and every
dependency through Redis ("Parallelism strategies").  The MI355X-native
counterpart keeps the tiles where they are produced and moves only what another GPU needs:

  * ownership:   tile (matrix, idx) belongs to rank  (idx[-2] mod Pr) * Pc + (idx[-1] mod Pc)
                 on a Pr x Pc process grid (1x2, 2x2, 2x4 for 2, 4, 8 GPUs); all SSA versions of a
                 trailing tile (S[v, j, k]) therefore live on the same GPU as the input tile (j, k) and
                 the factor tile O[j, k] -- a trailing update only ever fetches its two panel tiles.
  * owner computes: a task runs on the owner of its first output tile.
  * schedule:    the DAG is static, so every rank walks the SAME global task sequence (critical-path
                 order from LambdaPackProgram's ready heap).  A rank executes its own tasks and, right
                 after any task, the producer pushes each output tile to the ranks that own a consumer
                 task.  Producer and consumer reach that point of the sequence independently and post the
                 matching send / recv there, so the per-pair order of point-to-point operations is
                 identical on both sides (the NCCL/RCCL requirement) and no collective is needed on the
                 data path.  xGMI is point-to-point: a panel tile needed by k GPUs is k independent
                 128 MiB sends on k links rather than a ring broadcast.
  * overlap:     payload sends are posted as soon as the tile is produced (HIP event -> torch stream
                 -> RCCL), payload receives are asynchronous (consumers wait on the tile's event on
                 the device).  A 48-byte header (shape, dtype) travels on the CPU side channel (gloo)
                 because `safe=False` matrices hold tiles whose shape only the producer knows.

The same code runs on CPU under gloo with host tiles (tests/test_dist_gloo.py) and on GPUs under the
"nccl" backend, which is RCCL on ROCm.
"""
import collections
import os
import pickle
import time
import traceback

import numpy as np

from . import job_runner
from . import lambdapack as lp
from .device import DeviceBuffer, DeviceTile, get_backend

_DTYPES = [np.dtype(np.float64), np.dtype(np.float32), np.dtype(np.int32), np.dtype(np.int64)]


def process_grid(world):
    """Pr x Pc with Pr <= Pc and Pr * Pc == world (1x1, 1x2, 2x2, 2x4, ...)."""
    pr = int(np.floor(np.sqrt(world)))
    while world % pr:
        pr -= 1
    return pr, world // pr


class Comm(object):
    """Tile exchange over torch.distributed (payload: RCCL for device tensors, gloo on CPU)."""
    ownership = None  # optional algorithm-aware map (matrix_name, idx) -> rank or None (see tsqr_ownership)

    def __init__(self, rank, world, backend, device_tensors):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.backend = rank, world, backend
        self.device_tensors = device_tensors
        self.grid = process_grid(world)
        self.bytes_sent = 0
        self.bytes_received = 0
        self.transfers = 0
        self._pending = []  # (work handle, keep-alive objects)

    # ---- ownership ----
    def owner(self, matrix_name, idx):
        if self.ownership is not None:
            r = self.ownership(matrix_name, tuple(idx))
            if r is not None:
                return r
        pr, pc = self.grid
        if len(idx) >= 2:
            return (idx[-2] % pr) * pc + (idx[-1] % pc)
        if len(idx) == 1:
            return idx[0] % self.world
        return 0

    def owner_fn(self, nb=None):
        return self.owner

    # ---- control-plane collectives ----
    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        if self.device_tensors and "gloo" not in self.backend:
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        self.flush()
        try:
            self.dist.destroy_process_group()
        except Exception:
            pass

    def flush(self):
        for work, _ in self._pending:
            work.wait()
        self._pending = []

    def _trim(self):
        if len(self._pending) > 32:
            self._pending = [(w, a) for w, a in self._pending if not w.is_completed()]

    # ---- tile transport ----
    def _send_header(self, tile, dst):
        hdr = np.zeros(6, dtype=np.int64)
        hdr[0] = len(tile.shape)
        hdr[1:1 + len(tile.shape)] = tile.shape
        # dtype code + what the producer knows about the tile's structure (an R factor stays one on arrival)
        hdr[5] = _DTYPES.index(np.dtype(tile.dtype)) + (16 if getattr(tile, "upper", False) else 0)
        self.dist.send(self.torch.from_numpy(hdr), dst)

    def _recv_header(self, src):
        hdr = self.torch.zeros(6, dtype=self.torch.int64)
        self.dist.recv(hdr, src)
        h = hdr.numpy()
        return tuple(int(x) for x in h[1:1 + int(h[0])]), _DTYPES[int(h[5]) & 15], bool(int(h[5]) & 16)

    def send_tile(self, tile, dst):
        if len(tile.shape) > 4:
            raise ValueError("tiles with more than 4 dimensions cannot be exchanged")
        torch = self.torch
        be = get_backend()
        self._send_header(tile, dst)
        if self.device_tensors:
            # stage into a torch-owned tensor on torch's current stream, ordered after the producer
            ts = torch.cuda.current_stream().cuda_stream
            staging = torch.empty(max(tile.nbytes, 1), dtype=torch.uint8, device="cuda")
            if tile.ready is not None and tile.ready[1] != ts:
                be.wait_event(ts, tile.ready[0])
            tile.buf.streams.add(ts)
            be.lib.npw_memcpy_d2d_async(staging.data_ptr(), tile.ptr, tile.nbytes, ts)
            work = self.dist.isend(staging, dst)
            self._pending.append((work, (staging, tile)))
        else:
            arr = np.ascontiguousarray(be.to_host(tile))
            self.dist.send(torch.from_numpy(arr.reshape(-1).view(np.uint8).copy()), dst)
        self.bytes_sent += tile.nbytes
        self.transfers += 1
        self._trim()

    def recv_tile(self, src):
        torch = self.torch
        be = get_backend()
        shape, dtype, upper = self._recv_header(src)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        self.bytes_received += nbytes
        if self.device_tensors:
            buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
            work = self.dist.irecv(buf, src)
            work.wait()  # stream-level: torch's current stream now waits for the transfer
            ts = torch.cuda.current_stream().cuda_stream
            dbuf = DeviceBuffer(None, buf.data_ptr(), nbytes)
            dbuf.aux = {"keepalive": buf}
            dbuf.streams.add(ts)
            tile = DeviceTile(dbuf, shape, dtype)
            tile.ready = (be.record_new(ts), ts)
            tile.upper = upper
            return tile
        flat = torch.empty(max(nbytes, 1), dtype=torch.uint8)
        self.dist.recv(flat, src)
        arr = flat.numpy()[:nbytes].view(dtype).reshape(shape)
        tile = be.to_device(arr)
        tile.upper = upper
        return tile


def init_process_group(backend=None):
    """torch.distributed process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    backend: None -> "cpu:gloo,cuda:nccl" when a GPU is visible (nccl is RCCL on ROCm), else "gloo"."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    gpu = torch.cuda.is_available()
    if backend is None:
        # NUMPYWREN_AMD_DIST_BACKEND=gloo stages payloads through the host: lets several ranks share one GPU
        # (RCCL refuses two ranks on one device), which is how the GPU-side logic is tested on a 1-GPU box
        backend = os.environ.get("NUMPYWREN_AMD_DIST_BACKEND") or ("cpu:gloo,cuda:nccl" if gpu else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if "nccl" in backend:
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if not dist.is_initialized():
        import datetime
        # a lost peer must end the run with an error, not hang it: collectives / p2p time out
        limit = datetime.timedelta(seconds=int(os.environ.get("NUMPYWREN_AMD_DIST_TIMEOUT", "900")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=limit)
    return Comm(rank, world, backend, device_tensors=("nccl" in backend))


def _consumer_ranks(comm, task):
    """{output position: sorted ranks other than the producer that own a task reading that tile}."""
    if not task.writes:
        return {}
    me = comm.owner(*task.writes[0])
    out = {}
    for pos, w in enumerate(task.writes):
        ranks = set()
        for c in task.children:
            if c.writes and w in c.reads:
                r = comm.owner(*c.writes[0])
                if r != me:
                    ranks.add(r)
        if ranks:
            out[pos] = sorted(ranks)
    return out


def tsqr_ownership(world, num_leaves, input_name="A"):
    """Ownership for the TSQR program (reference algs.py:30-36): the leaves are dealt out in contiguous chunks
    (num_leaves / world each) and a tree node lives where its left operand lives.  The first log2(chunk) levels of the
    tree are then local to a GPU (and independent across its leaves: the executor batches them), and only the last
    log2(world) levels move one R factor per pair of GPUs -- the tree all-reduce of SURVEY 8(e).  The generic
    block-cyclic map would scatter every level.  Install with `comm.ownership = tsqr_ownership(...)`."""
    def chunk(j):
        return min(world - 1, int(j) * world // max(1, int(num_leaves)))

    def own(name, idx):
        if name == input_name:
            return chunk(idx[0])              # A[j, 0]
        if name in ("Vs", "Ts", "Rs") and len(idx) == 2:
            return chunk(idx[1])              # X[level, j]
        return None
    return own


def lambdapack_run_distributed(program, comm, pipeline_width=1, timeout=3600, max_inflight=64):
    """Distributed counterpart of job_runner.lambdapack_run: every rank calls it with the same program."""
    program.incr_up(1)
    t_start = time.time()
    be = get_backend()
    be.bind_thread()
    compiled = program.program
    rank = comm.rank
    ex = job_runner.LambdaPackExecutor(
        program, pipeline_width=pipeline_width,
        is_local=lambda t: (comm.owner(*t.writes[0]) if t.writes else 0) == rank,
        send_plan=lambda t: _consumer_ranks(comm, t))
    mats = compiled.matrices
    inputs = set(compiled.inputs)
    executed = []
    inflight = collections.deque()
    program._defer_success = True
    try:
        # prologue: input tiles read by tasks that live on another rank than the tile itself
        moved = set()
        for t in compiled.tasks:
            if not t.writes:
                continue
            consumer = comm.owner(*t.writes[0])
            for r in dict.fromkeys(t.reads):
                if r[0] in inputs and compiled.writer_of(*r) is None:
                    home = comm.owner(*r)
                    if home != consumer and (r, consumer) not in moved:
                        moved.add((r, consumer))
                        if rank == home:
                            comm.send_tile(mats[r[0]].get_tile(*r[1]), consumer)
                        elif rank == consumer:
                            mats[r[0]].put_tile(comm.recv_tile(home), *r[1])
        while program.program_status() == lp.PS.RUNNING and not program.all_terminators_done():
            node = program.dequeue()
            if node is None:
                break
            if time.time() - t_start > timeout:
                program._enqueue(node)
                break
            e, v = node
            # every rank forms the same group of ready tasks of one batchable kind and runs its own members of it as
            # one batched launch sequence; the exchange plan below is then walked in the common order
            group = [(e, v)]
            if ex.batch_fn(e) is not None:
                group += program.dequeue_matching(lambda e2, v2: e2 == e, ex.batch_tasks * comm.world - 1)
            tasks = [compiled.task(ge, gv) for ge, gv in group]
            owners = [comm.owner(*t.writes[0]) if t.writes else 0 for t in tasks]
            mine = [g for g, o in zip(group, owners) if o == rank]
            for ge, gv in group:
                program.set_node_status(ge, gv, lp.NS.RUNNING)
            if mine:
                try:
                    last = ex.run_batch(mine) if len(mine) > 1 else ex.run_task(*mine[0])
                except Exception as exc:
                    program.handle_exception(exc, tb=traceback.format_exc(), expr_idx=e, var_values=v)
                    raise
                executed.extend([ge, gv] for ge, gv in mine)
                if last is not None and last.ready is not None:
                    inflight.append(last)
                    if len(inflight) > max_inflight:
                        be.wait_tile(inflight.popleft())
            # push the outputs to the remote consumers: both sides evaluate the same static plan here
            for (ge, gv), task, owner in zip(group, tasks, owners):
                for pos, ranks in _consumer_ranks(comm, task).items():
                    name, idx = task.writes[pos]
                    if rank == owner:
                        if mats[name].tile_exists(*idx):
                            tile = mats[name].get_tile(*idx)
                            for dst in ranks:
                                comm.send_tile(tile, dst)
                            ex.sent(name, idx)
                    elif rank in ranks:
                        mats[name].put_tile(comm.recv_tile(owner), *idx)
                program.post_op(ge, gv, lp.PS.SUCCESS, None)
                program.set_node_status(ge, gv, lp.NS.FINISHED)
        comm.flush()
        be.synchronize()
        ok = job_runner.check_info_flags(program, be)
        # a failure on any rank fails the program everywhere
        bad = comm.max_over_ranks(0.0 if ok and program.program_status() != lp.PS.EXCEPTION else 1.0)
        program._defer_success = False
        if bad > 0:
            if program.program_status() != lp.PS.EXCEPTION:
                program.handle_exception("a task failed on another rank", tb="", expr_idx=-1, var_values={})
        elif program._success_pending and program.program_status() == lp.PS.RUNNING:
            program.return_success()
    finally:
        program._defer_success = False
        program.decr_up(1)
    return {"up_time": [t_start, time.time()], "exec_time": [], "executed_messages": executed,
            "operator_refs": [tuple(x) for x in executed], "log": pickle.dumps({}),
            "bytes_sent": comm.bytes_sent, "bytes_received": comm.bytes_received, "transfers": comm.transfers}


def gather_matrix(bigm, comm, root=0):
    """Collect the tiles of `bigm` that exist on any rank onto `root` (verification / output)."""
    local = {}
    for bidx in bigm.block_idxs_exist:
        local[tuple(bidx)] = bigm.get_block(*bidx)
    gathered = [None] * comm.world
    comm.dist.all_gather_object(gathered, local)
    if comm.rank != root:
        return None
    out = np.zeros(tuple(bigm.shape), dtype=bigm.dtype)
    for part in gathered:
        for bidx, blk in part.items():
            sl = tuple(slice(s, e) for s, e in bigm.__block_idx_to_real_idx__(bidx))
            out[sl] = blk.reshape(out[sl].shape)
    return out
