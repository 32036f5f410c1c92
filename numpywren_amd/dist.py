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
  * transport:   libnpw_hip.so's npw_comm_* entry points (include/npw_hip.h: grouped ncclSend / ncclRecv on a
                 dedicated high-priority HIP stream per rank).  A send is posted when the producing task is
                 enqueued and waits on the device for the producer's event only; a receive lands in a tile of
                 the backend's own allocator whose event the consumers wait for.  No torch tensors on the data
                 path, no staging copies, no host synchronisation.
  * metadata:    receivers derive shape and dtype of a tile from the static plan (TileMetaPlan: one rule per
                 Cholesky / GEMM kernel); only tiles of kernels without a rule (the QR family, whose `safe=False`
                 matrices hold tiles of data-dependent shapes) send a 48-byte header over the control group.
  * control:     torch.distributed over gloo -- rendezvous (the RCCL unique id), barriers, max-over-ranks.

The same code runs on CPU with the host transport (tests/test_dist_gloo.py).
"""
import collections
import ctypes
import os
import pickle
import sys
import time
import traceback

import numpy as np

from . import _ffi
from . import job_runner
from . import lambdapack as lp
from .device import DeviceTile, Stream, get_backend

_DTYPES = [np.dtype(np.float64), np.dtype(np.float32), np.dtype(np.int32), np.dtype(np.int64)]


def process_grid(world):
    """Pr x Pc with Pr <= Pc and Pr * Pc == world (1x1, 1x2, 2x2, 2x4, ...)."""
    pr = int(np.floor(np.sqrt(world)))
    while world % pr:
        pr -= 1
    return pr, world // pr


class TileMeta(tuple):
    """(shape, dtype, upper) of a tile: what a receiver must know before it can post the matching receive."""

    def __new__(cls, shape, dtype, upper=False):
        return tuple.__new__(cls, (tuple(int(x) for x in shape), np.dtype(dtype), bool(upper)))

    shape = property(lambda self: self[0])
    dtype = property(lambda self: self[1])
    upper = property(lambda self: self[2])

    @property
    def nbytes(self):
        return int(np.prod(self[0], dtype=np.int64)) * self[1].itemsize


class RcclTransport(object):
    """Payload path for one-GPU-per-rank runs: libnpw_hip.so's npw_comm_* entry points (RCCL point-to-point over xGMI).
    A tile is sent straight out of the buffer its producer wrote and received straight into a buffer of the backend's
    allocator (so the store's HBM budget and out-of-memory hook see it); both run on the communicator's transport
    stream, ordered behind the producer by the tile's event and ahead of the consumers by the received tile's event.
    No torch tensors, no staging copies, no host synchronisation."""
    name = "rccl"
    diag = False                          # True: every exchange is bracketed with timing events on the transport stream
    _open = None                          # start event of the exchange being posted

    def __init__(self, rank, world, control):
        self.be = get_backend()          # binds this process to its device (LOCAL_RANK) before the communicator exists
        self.lib = self.be.lib
        ident = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(_ffi.NPW_COMM_ID_BYTES)
            _ffi.check(self.lib.npw_comm_unique_id(buf, _ffi.NPW_COMM_ID_BYTES), "npw_comm_unique_id")
            ident[0] = buf.raw
        if world > 1:
            control.broadcast_object_list(ident, src=0)   # the rendezvous id travels over the control group
        h = ctypes.c_void_p(0)
        idbuf = ctypes.create_string_buffer(ident[0], _ffi.NPW_COMM_ID_BYTES)
        _ffi.check(self.lib.npw_comm_init(ctypes.byref(h), rank, world, idbuf), "npw_comm_init")
        self.handle = h.value
        sh = ctypes.c_void_p(0)
        _ffi.check(self.lib.npw_comm_info(self.handle, None, None, ctypes.byref(sh)), "npw_comm_info")
        self.stream = Stream(sh.value, True, "xgmi")
        self.rank, self.world = rank, world
        self._group = None                # open group: received tiles whose `ready` event is recorded at group end
        self._held = []                   # ... and the tiles being sent: alive until the launch has been enqueued
        self._spans = []                  # diag: (start event, end event) of every exchange

    def begin_group(self):
        """Everything posted until end_group() leaves as ONE RCCL launch (ncclGroupStart / ncclGroupEnd): the sends and
        receives of a group progress side by side on their links instead of one after the other on the transport
        stream.  Every rank opens and closes its groups at the same points of the common task sequence."""
        if self._group is None:
            self._span_begin()
            _ffi.check(self.lib.npw_comm_group_start(self.handle), "npw_comm_group_start")
            self._group = []

    # ---- diagnostics: how long the transport stream spends inside its exchanges (waiting for the local producers and
    #      for the peers included -- that IS the time a consumer of these tiles cannot start) ----
    def _span_begin(self):
        if self.diag and self._open is None:
            ev = self.be.new_event(timing=True)
            self.be.record(ev, self.stream)
            self._open = ev

    def _span_end(self):
        if self._open is not None:
            ev = self.be.new_event(timing=True)
            self.be.record(ev, self.stream)
            if not hasattr(self, "_spans"):
                self._spans = []
            self._spans.append((self._open, ev))
            self._open = None

    def exchange_ms(self):
        """Milliseconds of transport-stream time inside exchanges since the last call (synchronises the stream)."""
        self._span_end()
        total = 0.0
        if getattr(self, "_spans", None):
            self.be.stream_sync(self.stream)
            for a, b in self._spans:
                total += self.be.elapsed_ms(a, b)
                self.be.recycle_event(a)
                self.be.recycle_event(b)
            self._spans = []
        return total

    def end_group(self):
        if self._group is not None:
            pending, self._group = self._group, None
            try:
                _ffi.check(self.lib.npw_comm_group_end(self.handle), "npw_comm_group_end")   # the launch happens here
            finally:
                # Only now may a sent tile's last reference go: inside the group RCCL has merely noted the transfer, and a
                # buffer released before the launch is enqueued could be recycled behind an event that does not cover it.
                self._held = []
            if pending:
                self.be._produced(self.stream, *pending)   # one event for the whole group
            self._span_end()

    def abort_group(self):
        """An exception inside an open group (e.g. the owner's check of a tile against the static plan): launching the
        part of the group that was posted would pair transfers with the wrong ones on the peers, so the communicator is
        aborted instead (ncclCommAbort) and this transport is dead.  The peers' matching receives never complete; they
        leave through the collective time limit / the control group's timeout -- the failure is loud on every rank."""
        self._group = None
        self._held = []
        if self.handle:
            self.lib.npw_comm_abort(self.handle)

    def send(self, tile, dsts):
        be, cs = self.be, self.stream
        be._use(cs, tile)                 # after the producer; the buffer is not recycled before the send has left
        single = self._group is None
        if single:
            self._span_begin()
        if self._group is not None:
            self._held.append(tile)
            for d in dsts:
                _ffi.check(self.lib.npw_send_tile(self.handle, tile.ptr, tile.nbytes, int(d), cs.handle), "npw_send_tile")
        elif len(dsts) == 1:
            _ffi.check(self.lib.npw_send_tile(self.handle, tile.ptr, tile.nbytes, int(dsts[0]), cs.handle), "npw_send_tile")
        else:
            arr = (ctypes.c_int * len(dsts))(*[int(d) for d in dsts])
            _ffi.check(self.lib.npw_bcast_tile(self.handle, tile.ptr, tile.nbytes, self.rank, arr, len(dsts), cs.handle),
                       "npw_bcast_tile")
        if single:
            self._span_end()

    def recv(self, src, meta):
        be, cs = self.be, self.stream
        tile = be.empty(meta.shape, meta.dtype)
        be._use(cs, tile)
        if self._group is None:
            self._span_begin()
        _ffi.check(self.lib.npw_recv_tile(self.handle, tile.ptr, tile.nbytes, int(src), cs.handle), "npw_recv_tile")
        if self._group is not None:
            self._group.append(tile)      # the kernel is launched by end_group(): the event goes behind it
        else:
            be._produced(cs, tile)
            self._span_end()
        tile.upper = meta.upper
        return tile

    def flush(self):
        self.end_group()
        self.be.stream_sync(self.stream)

    def close(self):
        if self.handle:
            self.lib.npw_comm_destroy(self.handle)
            self.handle = None


class HostTransport(object):
    """Payload path over the gloo control group, staged through host memory: the CPU tests (checker backend) and
    several ranks sharing ONE GPU (RCCL refuses two ranks on a device) -- never a production path."""
    name = "host"

    def __init__(self, rank, world, control):
        import torch
        self.torch, self.control = torch, control
        self.rank, self.world = rank, world
        self.groups = 0
        self.diag = False
        self._host_s = 0.0    # host time inside the blocking sends / receives

    def exchange_ms(self):
        ms, self._host_s = 1e3 * self._host_s, 0.0
        return ms

    def begin_group(self):
        self.groups += 1      # blocking pairwise operations in the common order: a group changes nothing here

    def end_group(self):
        pass

    def abort_group(self):
        pass

    def send(self, tile, dsts):
        arr = np.ascontiguousarray(get_backend().to_host(tile))
        flat = self.torch.from_numpy(arr.reshape(-1).view(np.uint8).copy())
        t0 = time.time()
        for d in dsts:
            self.control.send(flat, int(d))
        self._host_s += time.time() - t0

    def recv(self, src, meta):
        flat = self.torch.empty(max(meta.nbytes, 1), dtype=self.torch.uint8)
        t0 = time.time()
        self.control.recv(flat, int(src))
        self._host_s += time.time() - t0
        arr = flat.numpy()[:meta.nbytes].view(meta.dtype).reshape(meta.shape)
        tile = get_backend().to_device(arr)
        tile.upper = meta.upper
        return tile

    def flush(self):
        pass

    def close(self):
        pass


class Comm(object):
    """One rank's view of the job: ownership map, the control group (torch.distributed over gloo: barriers, a handful
    of scalars, tile headers for dynamically shaped tiles) and the payload transport."""
    ownership = None  # optional algorithm-aware map (matrix_name, idx) -> rank or None (see tsqr_ownership)

    def __init__(self, rank, world, transport, control):
        self.dist = control
        self.rank, self.world = rank, world
        self.transport = transport
        self.backend = transport.name if transport is not None else "none"
        self.grid = process_grid(world)
        self.bytes_sent = 0
        self.bytes_received = 0
        self.transfers = 0
        self.headers = 0     # tiles whose (shape, dtype) had to travel over the control group

    def open_transport(self):
        """Make the payload transport now (init_process_group(open_transport=False)); every rank calls it at the same point."""
        if self.transport is None:
            self.transport = self._make_transport()
            self.backend = self.transport.name
        return self

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

    # ---- control-plane collectives (host side, tiny) ----
    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.transport is not None:
            self.flush()
            self.transport.close()
        try:
            if self.dist.is_initialized():
                self.dist.destroy_process_group()
        except Exception:
            pass

    def flush(self):
        self.transport.flush()

    # ---- tile transport ----
    def _send_header(self, tile, dst):
        import torch
        hdr = np.zeros(6, dtype=np.int64)
        hdr[0] = len(tile.shape)
        hdr[1:1 + len(tile.shape)] = tile.shape
        # dtype code + what the producer knows about the tile's structure (an R factor stays one on arrival)
        hdr[5] = _DTYPES.index(np.dtype(tile.dtype)) + (16 if getattr(tile, "upper", False) else 0)
        self.dist.send(torch.from_numpy(hdr), dst)

    def _recv_header(self, src):
        import torch
        hdr = torch.zeros(6, dtype=torch.int64)
        self.dist.recv(hdr, src)
        h = hdr.numpy()
        return TileMeta(tuple(int(x) for x in h[1:1 + int(h[0])]), _DTYPES[int(h[5]) & 15], bool(int(h[5]) & 16))

    def send_tile(self, tile, dsts, known=False, meta=None):
        """Push `tile` to the ranks `dsts` (one grouped launch on the transport stream).  known=True: the receivers
        derived the tile's shape and dtype from the static plan (TileMetaPlan), nothing but the payload travels --
        `meta` is that plan entry and the tile is checked against it HERE, on the owner: the receiver has posted a
        receive of meta.nbytes, and a payload of another size would hang or truncate inside RCCL instead of failing.
        Otherwise a 48-byte header goes ahead of the payload over the control group."""
        if isinstance(dsts, int):
            dsts = [dsts]
        if len(tile.shape) > 4:
            raise ValueError("tiles with more than 4 dimensions cannot be exchanged")
        if known and meta is not None:
            if (int(tile.nbytes) != meta.nbytes or np.dtype(tile.dtype) != meta.dtype
                    or int(np.prod(tile.shape, dtype=np.int64)) != int(np.prod(meta.shape, dtype=np.int64))):
                raise RuntimeError("tile to be sent to ranks {0} is {1} {2} ({3} bytes) but the static plan promised its "
                                   "receivers {4} {5} ({6} bytes)".format(list(dsts), tuple(tile.shape), np.dtype(tile.dtype),
                                                                          tile.nbytes, meta.shape, meta.dtype, meta.nbytes))
        if not known:
            for d in dsts:
                self._send_header(tile, d)
            self.headers += len(dsts)
        self.transport.send(tile, dsts)
        self.bytes_sent += tile.nbytes * len(dsts)
        self.transfers += len(dsts)

    def recv_tile(self, src, meta=None, key=None):
        """`key` = (matrix name, tile index) of the tile: the transports move bytes and ignore it; the measuring
        stand-in of tools/dist_host_split.py (one rank of a pretended larger job) looks the tile up by it."""
        if meta is None:
            meta = self._recv_header(src)
            self.headers += 1
        tile = self.transport.recv(src, meta, key) if getattr(self.transport, "wants_key", False) else self.transport.recv(src, meta)
        self.bytes_received += meta.nbytes
        return tile


def init_process_group(backend=None, open_transport=True):
    """Join the job described by the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).

    open_transport=False: only the control group and the decision which payload transport to use; the transport itself (the
    RCCL communicator) is made by `comm.open_transport()` -- collectively, later.  bench.py times rank 0's one-GPU anchor in
    between: while a communicator is live the library leaves compute units to its transfer kernels, which a one-GPU run
    beside an idle communicator would pay for nothing.

    The control group is torch.distributed over gloo (host side: rendezvous, barriers, a few scalars).  The PAYLOAD
    transport is libnpw_hip.so's RCCL layer (`npw_comm_*`, one communicator per rank on its own GPU) unless
    backend / $NUMPYWREN_AMD_DIST_BACKEND says "gloo" or the ranks have to share a device -- then tiles are staged
    through the host over the control group (CPU tests; several ranks on the one GPU of a test box)."""
    import datetime
    import torch.distributed as dist
    # (dmabuf IPC for RCCL between processes: only effective if no HIP call has been made in this process yet; bench.py sets it first thing)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if not dist.is_initialized():
        # a lost peer must end the run with an error, not hang it: control-plane operations time out
        limit = datetime.timedelta(seconds=int(os.environ.get("NUMPYWREN_AMD_DIST_TIMEOUT", "900")))
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=limit)
    want = backend or os.environ.get("NUMPYWREN_AMD_DIST_BACKEND") or "auto"
    use_rccl = False
    if "gloo" not in want:
        from .device import hip_available
        # RCCL needs every rank of the communicator on its OWN physical GPU.  Device counts do not tell: under torchrun
        # with per-rank HIP_VISIBLE_DEVICES every rank sees one device and they are all different; on a one-GPU test
        # box every rank sees one device and it is the same.  So the ranks compare the PCI bus id of the device each
        # of them will bind (LOCAL_RANK mod visible devices, like HipBackend) over the control group.
        mine = None
        if hip_available():
            n = ctypes.c_int(0)
            _ffi.lib().npw_device_count(ctypes.byref(n))
            if n.value > 0:
                buf = ctypes.create_string_buffer(64)
                dev = int(os.environ.get("LOCAL_RANK", "0")) % n.value
                if _ffi.lib().npw_device_pci_bus_id(dev, buf, 64) == 0:
                    # (bus ids are only unique within one host: the pair is what has to differ between ranks)
                    import socket
                    mine = (socket.gethostname(), buf.value.decode())
        ids = [None] * world
        if world > 1:
            dist.all_gather_object(ids, mine)
        else:
            ids = [mine]
        use_rccl = all(i is not None for i in ids) and len(set(ids)) == world
        if not use_rccl and want in ("rccl", "nccl"):
            raise RuntimeError(f"RCCL transport needs one GPU per rank; the ranks sit on {ids} (rank {rank}: LOCAL_RANK="
                               f"{os.environ.get('LOCAL_RANK')}, HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}, "
                               f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')})")
        if hip_available() and mine is not None:
            # LOCAL_RANK beyond the devices this process sees is bound by modulo (HipBackend): legitimate when the launcher
            # gives every rank its own one-device list, a doubled-up GPU otherwise -- the bus-id comparison above decides
            # which, this only says so early and by name
            lr, nvis = int(os.environ.get("LOCAL_RANK", "0")), n.value
            if lr >= nvis > 1:
                import warnings
                warnings.warn(f"numpywren_amd.dist: rank {rank} has LOCAL_RANK={lr} but sees {nvis} devices; it binds device {lr % nvis}")
        if not use_rccl and mine is not None and rank == 0:
            import warnings
            warnings.warn("numpywren_amd.dist: ranks share a GPU (PCI bus ids {0}): tiles are staged through host memory "
                          "over the gloo control group, not sent over RCCL / xGMI".format(ids))
    make = lambda: (RcclTransport if use_rccl else HostTransport)(rank, world, dist)
    comm = Comm(rank, world, make() if open_transport else None, dist)
    comm._make_transport = make
    comm.backend = "rccl" if use_rccl else "host"
    return comm


# ------------------------------------------------------------------------------------------------
# static tile metadata: what every rank can derive about every tile of the plan without asking anybody
# ------------------------------------------------------------------------------------------------
def _f32_iff_all(*metas):
    return np.dtype(np.float32) if all(m.dtype == np.float32 for m in metas) else np.dtype(np.float64)


def _meta_gemm(a, b, transpose_A=False, transpose_B=False, **kw):
    m = a.shape[1] if transpose_A else a.shape[0]
    n = b.shape[0] if transpose_B else b.shape[1]
    return [TileMeta((m, n), a.dtype if a.dtype == b.dtype else np.float64)]


_META_RULES = {
    "gemm": _meta_gemm,
    "syrk": lambda s, x, y, **kw: [TileMeta(s.shape, _f32_iff_all(s, x, y))],
    "trsm": lambda x, y, **kw: [TileMeta(y.shape, np.float64)],
    "chol": lambda x, **kw: [TileMeta(x.shape, x.dtype)],
    "add_matrices": lambda *a, **kw: [TileMeta(a[0].shape, np.float64)],
    "identity": lambda x, **kw: [x],
}


class TileMetaPlan(object):
    """Shape / dtype of every tile the program touches, propagated through the static task sequence with one rule per
    kernel (the arithmetic of the Cholesky and GEMM programs).  Every rank computes the same table, so a receiver can
    allocate and post its receive without a header from the producer.  Tiles of kernels without a rule (the QR
    family, user callables) or of inputs whose stored shape is not the nominal block shape stay unknown -> header."""

    def __init__(self, compiled):
        self.compiled = compiled
        self.table = {}

    def _input_meta(self, name, idx):
        m = self.compiled.matrices[name]
        try:
            blocks = [m._blocks(ax)[i] for ax, i in enumerate(idx)]
        except Exception:
            return None
        shape = tuple(int(e - s) for s, e in blocks)
        if getattr(m, "autosqueeze", True):
            shape = tuple(x for x in shape if x != 1) or (1,)
        return TileMeta(shape, m.dtype)

    def read(self, name, idx):
        key = (name, tuple(idx))
        if key in self.table:
            return self.table[key]
        if self.compiled.writer_of(name, idx) is not None:
            return None          # produced by a task we have no rule for
        m = self.compiled.matrices[name]
        if name not in self.compiled.inputs and getattr(m, "parent_fn", None) is not None:
            zs = getattr(m.parent_fn, "_npw_zero_shape", None)   # never written: a parent_fn zero tile
            return TileMeta(zs(m, tuple(idx)), np.float64) if zs is not None else None
        return self._input_meta(name, idx)

    def visit(self, task, kernel_name):
        """Record the metas of `task`'s outputs (call in plan order)."""
        rule = _META_RULES.get(kernel_name)
        outs = None
        if rule is not None:
            ins = [self.read(*r) for r in task.reads]
            if all(i is not None for i in ins):
                try:
                    args = [ins[j] for kind, j in task.arg_kinds if kind == "tile"]
                    outs = rule(*args, **task.kwargs)
                except Exception:
                    outs = None
        if outs is not None and len(outs) == len(task.writes):
            for w, o in zip(task.writes, outs):
                self.table[(w[0], tuple(w[1]))] = o
        return outs


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


def gemm_ownership(world):
    """Ownership for the GEMM program (reference algs.py:251-266): every partial product Temp[i, j, k, l] and the
    output Out[i, j] live with the C tile (i, j) on the Pr x Pc grid, so the reduction tree of a C tile is local and
    only input tiles move: A[i, k] to the Pc GPUs of grid row i, B[k, j] to the Pr GPUs of grid column j (SUMMA's
    traffic, pushed point-to-point in the prologue and consumed as the tiles arrive).  The generic map would own
    Temp by its (k, l) indices.  Install with `comm.ownership = gemm_ownership(world)`."""
    pr, pc = process_grid(world)

    def own(name, idx):
        if name == "Temp" and len(idx) == 4:
            return (idx[0] % pr) * pc + (idx[1] % pc)
        return None
    return own


def _stored_tile(bigm, idx):
    """The tile as stored (no `lambdav` shift: the receiver's own reads apply it), or the parent_fn / host form."""
    obj = bigm._raw(tuple(idx))[0] if hasattr(bigm, "_raw") else None
    if isinstance(obj, DeviceTile):
        return obj
    return bigm.get_tile(*idx)


TIMEOUT_CHECK_EVERY = 32     # positions of the common task sequence between two collective looks at the clock


class StallWatch(object):
    """Self-diagnosis of a distributed run that stops making progress (the first contact with a real 8-GPU node is the
    driver's, nobody is there to attach a debugger): the walk notes where it is -- position of the common task sequence,
    task, what it is about to wait for -- and a daemon thread prints ONE report per stalled position to stderr once nothing
    has moved for `report_s` seconds ($NUMPYWREN_AMD_DIST_STALL_REPORT_S, default 30; a position is milliseconds of host work):

        [numpywren_amd.dist stall] rank 3/8 transport rccl: no progress for 30.0 s at position 412 of the common sequence,
          task (5, {'i': 3, 'j': 9, 'k': 4}) = syrk, phase 'wait for own device (run-ahead bound)'; receives not complete: from
          rank 1 S[4,9,4] (134217728 B), from rank 5 O[9,3]; last sends: O[9,3] -> [1, 2]; bytes sent / received ...

    and, after `abort_s` seconds ($NUMPYWREN_AMD_DIST_STALL_ABORT_S, default 0 = never: the run's own `timeout` and the control
    group's time limit end it), aborts the communicator (npw_comm_abort) so that device-side waits inside RCCL return and
    the run fails loudly on every rank instead of hanging.  The counterpart of the progress poller of the reference's
    experiments (cholesky_experiment.py:177-287 prints up / busy workers, flops and read / write GB/s every few seconds)."""

    def __init__(self, comm, total, report_s=None, abort_s=None, out=None):
        import threading
        self.comm, self.total = comm, total
        self.report_s = float(os.environ.get("NUMPYWREN_AMD_DIST_STALL_REPORT_S", "30") if report_s is None else report_s)
        self.abort_s = float(os.environ.get("NUMPYWREN_AMD_DIST_STALL_ABORT_S", "0") if abort_s is None else abort_s)
        self.out = out if out is not None else sys.stderr
        self.position, self.task, self.kernel, self.phase = 0, None, "", "start"
        self.last = time.time()
        self.recv_log = collections.deque(maxlen=64)     # (src, key, nbytes, tile)
        self.send_log = collections.deque(maxlen=8)      # (key, dsts)
        self.reports = []
        self.aborted = False
        self._reported_at = None
        self._stop = threading.Event()
        self._thread = None
        if self.report_s > 0:
            self._thread = threading.Thread(target=self._loop, name="npw-dist-stall-watch", daemon=True)
            self._thread.start()

    def note(self, position=None, task=None, kernel=None, phase=None):
        if position is not None:
            self.position = position
        if task is not None:
            self.task = task
        if kernel is not None:
            self.kernel = kernel
        if phase is not None:
            self.phase = phase
        self.last = time.time()

    def received(self, src, key, nbytes, tile):
        self.recv_log.append((src, key, nbytes, tile))

    def sent(self, key, dsts):
        self.send_log.append((key, list(dsts)))

    @staticmethod
    def _name(key):
        return "?" if key is None else "{0}[{1}]".format(key[0], ",".join(str(int(x)) for x in key[1]))

    def report(self, why=None):
        """The report as a string (also kept in self.reports)."""
        comm = self.comm
        be = None
        try:
            be = get_backend()
        except Exception:
            pass
        waiting = []
        for src, key, nbytes, tile in list(self.recv_log):
            ev = getattr(tile, "ready", None)
            done = True
            if ev is not None and be is not None and hasattr(be, "event_done"):
                try:
                    done = bool(be.event_done(ev))
                except Exception:
                    done = True
            if not done:
                waiting.append("from rank {0} {1} ({2} B)".format(src, self._name(key), nbytes))
        idle = time.time() - self.last
        text = ("[numpywren_amd.dist stall] rank {r}/{w} transport {t}: {why} at position {p} of {n} of the common sequence, task {task} = "
                "{k}, phase '{ph}'; receives not complete: {wait}; last sends: {sends}; bytes sent / received {bs} / {br}".format(
                    r=comm.rank, w=comm.world, t=comm.backend, why=why or "no progress for {0:.1f} s".format(idle), p=self.position,
                    n=self.total, task=self.task, k=self.kernel, ph=self.phase,
                    wait=("; ".join(waiting) if waiting else "none on record (peers waited on, if any: the ranks named in the phase)"),
                    sends=("; ".join("{0} -> {1}".format(self._name(k), d) for k, d in self.send_log) or "none"),
                    bs=comm.bytes_sent, br=comm.bytes_received))
        self.reports.append(text)
        try:
            print(text, file=self.out, flush=True)
        except Exception:
            pass
        return text

    def _loop(self):
        while not self._stop.wait(min(1.0, max(0.05, self.report_s / 4))):
            idle = time.time() - self.last
            if idle >= self.report_s and self._reported_at != (self.position, self.phase):
                self._reported_at = (self.position, self.phase)
                self.report()
            if self.abort_s > 0 and idle >= self.abort_s and not self.aborted:
                self.aborted = True
                self.report("no progress for {0:.1f} s: aborting the communicator".format(idle))
                try:
                    self.comm.transport.abort_group()
                except Exception:
                    pass

    def close(self):
        self._stop.set()


def link_calibration(comm, nbytes=128 << 20, repeats=3):
    """What a tile transfer costs on THIS node, measured before the first timed step (bench.py --gpus N): the figure
    profiles/predicted_scaling.json assumes (64 GB/s per direction) has never been measured by the builder.

      * shift exchanges: for every distance d = 1 .. world - 1 all ranks at once send one `nbytes` tile to rank + d and
        receive one from rank - d (ONE grouped launch: every rank drives one outgoing and one incoming link) -- GB/s per
        direction per (rank, d), best of `repeats`;
      * fan-out: rank 0 pushes the same tile to every other rank in one group (the panel tile of a Cholesky step going to
        the owners of its consumers); the time until the last receiver has it, and the aggregate GB/s out of rank 0;
      * all-to-all: every rank to every other rank in one group (the GEMM program's prologue).
    A world of one exchanges with itself (a device-to-device copy through the transport: the fields exist, the numbers are
    HBM's, and the line says so).  Timed with HIP events on the transport stream (RCCL) or the host clock (host transport).
    Returns a dict on every rank (identical: the samples are gathered over the control group)."""
    be = get_backend()
    tr = comm.transport
    rank, world = comm.rank, comm.world
    n = max(8, int(nbytes) // 8)
    src = be.zeros((n,), np.float64)
    meta = TileMeta((n,), np.float64)
    rccl = tr.name == "rccl"
    # the calibration is the first payload that ever crosses this node's links: if it hangs, say where, and (RCCL) abort after
    # $NUMPYWREN_AMD_DIST_CALIB_ABORT_S (default 180 s) so that the job ends with an error instead of sitting in a stream wait
    watch = StallWatch(comm, 0, abort_s=float(os.environ.get("NUMPYWREN_AMD_DIST_CALIB_ABORT_S", "180")) if rccl else 0.0)
    what = ["start"]

    def timed(post):
        """seconds for one exchange described by post() (called between begin_group / end_group)"""
        watch.note(phase="link calibration: " + what[0] + " (control-group barrier)")
        comm.barrier()
        watch.note(phase="link calibration: " + what[0] + " (payload in flight)")
        if rccl:
            be.stream_sync(tr.stream)
            e0 = be.new_event(timing=True)
            be.record(e0, tr.stream)
        t0 = time.time()
        tr.begin_group()
        got = post()
        tr.end_group()
        if rccl:
            e1 = be.new_event(timing=True)
            be.record(e1, tr.stream)
            be.stream_sync(tr.stream)
            dt = be.elapsed_ms(e0, e1) * 1e-3
            be.recycle_event(e0)
            be.recycle_event(e1)
        else:
            tr.flush()
            dt = time.time() - t0
        del got
        return dt

    def shift(d):
        def post():
            to, frm = (rank + d) % world, (rank - d) % world
            # (blocking host transport: the lower rank of a pair sends first, so the pairs cannot lock up)
            if rccl or rank < to:
                tr.send(src, [to])
                return tr.recv(frm, meta)
            r = tr.recv(frm, meta)
            tr.send(src, [to])
            return r
        return post

    def self_exchange():
        if rccl:
            tr.send(src, [rank])
            return tr.recv(rank, meta)
        return be.copy(src)

    try:
        return _link_calibration_body(comm, be, tr, rank, world, n, src, meta, rccl, watch, what, timed, shift, self_exchange, repeats)
    finally:
        watch.close()


def _link_calibration_body(comm, be, tr, rank, world, n, src, meta, rccl, watch, what, timed, shift, self_exchange, repeats):
    samples = []
    if world == 1:
        what[0] = "self-exchange"
        best = min(timed(self_exchange) for _ in range(repeats))
        samples.append({"rank": 0, "d": 0, "GBps": round(n * 8 / best / 1e9, 2)})
    else:
        for d in range(1, world):
            if not rccl and world > 2 and d != 1 and d != world - 1:
                continue                      # (host transport: a ring of blocking pairs is enough for a test box)
            what[0] = "shift by %d: send to rank %d, receive from rank %d" % (d, (rank + d) % world, (rank - d) % world)
            best = min(timed(shift(d)) for _ in range(repeats))
            samples.append({"rank": rank, "d": d, "GBps": round(n * 8 / best / 1e9, 2)})
    out = {"bytes": n * 8, "transport": tr.name, "what": ("self-exchange on one GPU: a device copy, not a link" if world == 1 else
                                                          "GB/s per direction, one tile per (rank, distance), all ranks at once")}
    fan = a2a = None
    if world > 1 and rccl:
        def fanout():
            if rank == 0:
                for dst in range(1, world):
                    tr.send(src, [dst])
                return None
            return tr.recv(0, meta)
        what[0] = "fan-out from rank 0 to every rank"
        fan = min(timed(fanout) for _ in range(repeats))

        def alltoall():
            got = []
            for dst in range(world):
                if dst != rank:
                    tr.send(src, [dst])
            for s_ in range(world):
                if s_ != rank:
                    got.append(tr.recv(s_, meta))
            return got
        what[0] = "all-to-all"
        a2a = min(timed(alltoall) for _ in range(repeats))
    watch.note(phase="link calibration: gathering the samples over the control group")
    watch.close()
    if watch.aborted:
        raise RuntimeError("link calibration: no progress for %.0f s, communicator aborted (%s)" % (watch.abort_s, watch.reports[-1] if watch.reports else ""))
    gathered = [None] * world
    if world > 1:
        comm.dist.all_gather_object(gathered, {"samples": samples, "fan": fan, "a2a": a2a})
    else:
        gathered = [{"samples": samples, "fan": fan, "a2a": a2a}]
    allv = [s_["GBps"] for g in gathered for s_ in g["samples"]]
    out["xgmi_GBps"] = {"min": min(allv), "median": float(np.median(allv)), "max": max(allv)}
    out["samples"] = [s_ for g in gathered for s_ in g["samples"]]
    if world > 1 and rccl:
        fmax = max(g["fan"] for g in gathered)
        amax = max(g["a2a"] for g in gathered)
        out["fanout_1_to_all_ms"] = round(fmax * 1e3, 3)
        out["fanout_GBps_out_of_rank0"] = round((world - 1) * n * 8 / fmax / 1e9, 2)
        out["all_to_all_ms"] = round(amax * 1e3, 3)
        out["all_to_all_GBps_per_rank_out"] = round((world - 1) * n * 8 / amax / 1e9, 2)
    return out


def lambdapack_run_distributed(program, comm, pipeline_width=1, timeout=3600, max_inflight=16):
    """Distributed counterpart of job_runner.lambdapack_run: every rank calls it with the same program.

    Transfers are posted on the transport stream at the point of the common task sequence where the tile is produced:
    the producer's sends wait (on the device) for the producing kernel only -- not for the trailing updates queued
    behind it on the compute streams -- so a panel tile leaves as soon as its trsm has finished and the consumers'
    receives were posted long before; the panel exchange overlaps the trailing updates of the previous step.

    max_inflight: positions of the common sequence the host may run ahead of its own device (16: each is at least one
    kernel of 1 - 2 ms, usually a batch).  The bound is there so that the host waits HERE, where the wait is counted as
    `host_blocked_ms`, instead of inside a HIP launch call that returns late under back-pressure: with 64 the same
    65536^2 run booked 480 - 900 ms of such waiting as `host_walk_ms` (profiles/r05_dist_host_split.md); the run's wall
    time is the device's either way (1375 ms with 4, 1381 with 64)."""
    if getattr(program, "block_sparse", False):
        # an owner that skips storing a zero tile would never post the send its consumers wait for
        raise NotImplementedError("lambdapack_run_distributed: block_sparse programs are not supported (every planned "
                                  "transfer must have a payload); run them on one GPU")
    program.incr_up(1)
    t_start = time.time()
    cpu_start = time.thread_time()
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
    metas = TileMetaPlan(compiled)
    executed = []
    inflight = collections.deque()
    # diagnostics of this run (returned as "diag"; bench.py prints them per rank on every N > 1 line): host time blocked
    # on the device or on the control group, per-kernel device time (executor.task_timers), transport-stream time
    diag_on = bool(((program.config or {}).get("executor", {}) if isinstance(program.config, dict) else {}).get("task_timers", False))
    comm.transport.diag = diag_on
    blocked_s = 0.0
    # where the walk's host time goes (seconds; always on: a dozen clock reads per position): taking the next group off
    # the ready heap, looking up tasks / owners, running this rank's own tasks (by kernel name: ctypes marshalling, event
    # records, allocator, flag plumbing -- and any wait hidden inside a HIP call), the exchange plan (static metadata,
    # consumer ranks, posting sends / receives), dependency accounting
    split = collections.defaultdict(float)
    clock = time.perf_counter
    sent0, recv0 = comm.bytes_sent, comm.bytes_received
    program._defer_success = True
    program._distributed_world = int(getattr(comm, "world", 1) or 1)    # (checkpoint.save refuses a run over several ranks)
    watch = StallWatch(comm, len(compiled.tasks))
    comm.stall_watch = watch
    try:
        # prologue: input tiles read by tasks that live on another rank than the tile itself (one grouped push per tile)
        moves = collections.OrderedDict()
        for t in compiled.tasks:
            if not t.writes:
                continue
            consumer = comm.owner(*t.writes[0])
            for r in dict.fromkeys(t.reads):
                if r[0] in inputs and compiled.writer_of(*r) is None:
                    home = comm.owner(*r)
                    if home != consumer:
                        moves.setdefault(r, (home, []))
                        if consumer not in moves[r][1]:
                            moves[r][1].append(consumer)
        # the prologue is ONE grouped exchange: every link carries its input tiles at once (the GEMM program's A / B
        # panels: SUMMA's traffic as a single all-to-all-v launch instead of a queue of single transfers)
        comm.transport.begin_group()
        try:
            for r, (home, consumers) in moves.items():
                meta = metas.read(*r)
                if rank == home:
                    watch.note(phase="prologue: send {0} to {1}".format(StallWatch._name(r), list(consumers)))
                    comm.send_tile(_stored_tile(mats[r[0]], r[1]), consumers, known=meta is not None, meta=meta)
                    watch.sent(r, consumers)
                elif rank in consumers:
                    watch.note(phase="prologue: receive {0} from rank {1}".format(StallWatch._name(r), home))
                    got = comm.recv_tile(home, meta, key=r)
                    watch.received(home, r, getattr(got, "nbytes", 0), got)
                    mats[r[0]].put_tile(got, *r[1])
        except BaseException:
            comm.transport.abort_group()        # a partly posted group must not be launched (see RcclTransport.abort_group)
            raise
        else:
            comm.transport.end_group()
        step, timed_out = 0, False
        split["prologue"] = time.time() - t_start
        while program.program_status() == lp.PS.RUNNING and not program.all_terminators_done():
            c0 = clock()
            node = program.dequeue()
            if node is None:
                break
            # The time limit is a COLLECTIVE decision, taken at fixed positions of the common task sequence on the maximum
            # over the ranks: a rank that left the walk on its own clock while its peers post the next transfer would
            # leave them waiting inside RCCL for ever.  Every other decision in this loop is a function of the plan.
            if timeout is not None and step % TIMEOUT_CHECK_EVERY == TIMEOUT_CHECK_EVERY - 1:
                t_b = time.time()
                watch.note(position=step, phase="time-limit check: all_reduce over the control group (waits for EVERY rank to reach this position)")
                late = comm.max_over_ranks(t_b - t_start) > timeout
                blocked_s += time.time() - t_b
                if late:
                    # every rank says where it stands before the walk is left (the slowest rank's report names what it waited for)
                    watch.report("time limit of {0} s exceeded (collective decision)".format(timeout))
                    program._enqueue(node)
                    timed_out = True
                    break
            step += 1
            e, v = node
            watch.note(position=step, task=(int(e), dict(v)), kernel=getattr(compiled.kernel(e), "__name__", "task"), phase="walk")
            if watch.aborted:
                raise RuntimeError("lambdapack_run_distributed: the communicator was aborted after a stall (see the stall report on stderr)")
            # every rank forms the same group of ready tasks of one batchable kind and runs its own members of it as
            # one batched launch sequence; the exchange plan below is then walked in the common order
            group = [(e, v)]
            if ex.batch_fn(e) is not None:
                group += program.dequeue_matching(lambda e2, v2: e2 == e, ex.batch_tasks * comm.world - 1)
            c1 = clock()
            split["dequeue"] += c1 - c0
            tasks = [compiled.task(ge, gv) for ge, gv in group]
            owners = [comm.owner(*t.writes[0]) if t.writes else 0 for t in tasks]
            mine = [g for g, o in zip(group, owners) if o == rank]
            for ge, gv in group:
                program.set_node_status(ge, gv, lp.NS.RUNNING)
            c2 = clock()
            split["lookup"] += c2 - c1
            if mine:
                try:
                    last = ex.run_batch(mine) if len(mine) > 1 else ex.run_task(*mine[0])
                except Exception as exc:
                    program.handle_exception(exc, tb=traceback.format_exc(), expr_idx=e, var_values=v)
                    raise
                executed.extend([ge, gv] for ge, gv in mine)
                c3 = clock()
                split["run:" + getattr(compiled.kernel(e), "__name__", "task")] += c3 - c2
                if last is not None and last.ready is not None:
                    inflight.append(last)
                    if len(inflight) > max_inflight:
                        t_b = time.time()
                        watch.note(phase="wait for own device (run-ahead bound): a kernel or a receive it depends on has not finished")
                        be.wait_tile(inflight.popleft())
                        blocked_s += time.time() - t_b
                        watch.note(phase="walk")
            c4 = clock()
            # push the outputs to the remote consumers: both sides evaluate the same static plan here; the transfers of
            # one group of tasks (the right-hand sides of a batched solve, the nodes of a tree level) are one launch
            comm.transport.begin_group()
            try:
                for (ge, gv), task, owner in zip(group, tasks, owners):
                    kname = getattr(compiled.kernel(ge), "__name__", "")
                    out_metas = metas.visit(task, kname)
                    for pos, ranks in _consumer_ranks(comm, task).items():
                        name, idx = task.writes[pos]
                        meta = out_metas[pos] if out_metas is not None else None
                        if rank == owner:
                            watch.note(phase="send {0} to {1}".format(StallWatch._name((name, idx)), list(ranks)))
                            comm.send_tile(_stored_tile(mats[name], idx), ranks, known=meta is not None, meta=meta)
                            watch.sent((name, idx), ranks)
                            ex.sent(name, idx)
                        elif rank in ranks:
                            watch.note(phase="receive {0} from rank {1}".format(StallWatch._name((name, idx)), owner))
                            got = comm.recv_tile(owner, meta, key=(name, idx))
                            watch.received(owner, (name, idx), getattr(got, "nbytes", 0), got)
                            mats[name].put_tile(got, *idx)
                    c5 = clock()
                    program.post_op(ge, gv, lp.PS.SUCCESS, None)
                    program.set_node_status(ge, gv, lp.NS.FINISHED)
                    split["post_op"] += clock() - c5
            except BaseException:
                comm.transport.abort_group()
                raise
            else:
                comm.transport.end_group()
            split["exchange+post_op"] += clock() - c4
        t_walk_end = time.time()
        cpu_walk = time.thread_time() - cpu_start
        watch.note(phase="drain: transport flush + device synchronise (every posted receive must arrive)")
        comm.flush()
        be.synchronize()
        watch.note(phase="settle: control-group max over ranks")
        t_drained = time.time()
        ok = job_runner.settle_checks(program, be)
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
        ex.release_spill_plan()
        watch.close()
    diag = {"rank": rank, "transport": comm.backend, "positions": step, "tasks_run_here": len(executed),
            # the common walk on the host, without the time it spent blocked in dist.py's OWN waits (the run-ahead bound, the
            # control group).  It still contains what the walk spent blocked INSIDE HIP calls: a launch returns late when the
            # device is far behind (measured, profiles/r05_dist_host_split.md: 65536^2 on one GPU 61 - 115 ms with the host
            # kept 4 positions ahead, 480 - 900 ms of the same 1380 ms run with 64 -- back-pressure, not work) ...
            "host_walk_ms": round(1e3 * (t_walk_end - t_start - blocked_s), 3),
            # ... so the thread's CPU time over the walk says what the host actually computed
            "host_cpu_ms": round(1e3 * cpu_walk, 3),
            "host_blocked_ms": round(1e3 * blocked_s, 3),
            # host time between the end of the walk and the device being drained: how far the device ran behind the host
            "drain_ms": round(1e3 * (t_drained - t_walk_end), 3),
            "bytes_sent": comm.bytes_sent - sent0, "bytes_received": comm.bytes_received - recv0,
            "stall_reports": len(watch.reports)}
    split["exchange"] = split.pop("exchange+post_op", 0.0) - split["post_op"]
    diag["host_split_ms"] = {k: round(1e3 * v, 3) for k, v in sorted(split.items())}
    if diag_on:
        times = job_runner.collect_task_times(program)
        diag["kernel_busy_ms"] = round(sum(v["ms"] for v in times.values()), 3)
        diag["kernel_ms_by_name"] = {k: round(v["ms"], 3) for k, v in sorted(times.items())}
        diag["transfer_wait_ms"] = round(comm.transport.exchange_ms(), 3)
    return {"up_time": [t_start, time.time()], "exec_time": [], "executed_messages": executed,
            "operator_refs": [tuple(x) for x in executed], "log": pickle.dumps({}),
            "bytes_sent": comm.bytes_sent, "bytes_received": comm.bytes_received, "transfers": comm.transfers,
            "headers": comm.headers, "timed_out": timed_out, "steps": step, "diag": diag}


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
