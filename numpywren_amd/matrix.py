"""BigMatrix: an N-d array sharded into tiles that live in HBM (MI355X, 288 GB per GPU).

Same constructor, attributes and block-indexing semantics as the reference's S3-backed
BigMatrix / BigMatrixView (reference numpywren/matrix.py:37-782); what changed is the
substrate: a tile is a `DeviceTile` in device memory (tier "hbm") or a host ndarray (tier
"host": spill / CPU-side storage tests) held in a process-wide object table keyed exactly like
the reference's S3 objects -- `prefix + key + "/" + "{start}_{end}_{shard}_"...` (reference
matrix.py:457-464, 481-495).  As in the reference a BigMatrix object carries no tile state of
its own: two objects with the same (bucket, key) see the same tiles and header.

Two access paths:
  get_block / put_block   ndarray in/out -- the reference API (D2H / H2D under the hood)
  get_tile  / put_tile    DeviceTile in/out, asynchronous -- what the executor uses
Both apply the reference's read semantics (parent_fn for missing tiles, autosqueeze, the
`lambdav` diagonal shift on every read of a diagonal tile of a square matrix; reference
matrix.py:293-310) and write semantics (squeeze-reshape, `safe` shape check, no dtype cast;
reference matrix.py:343-361).
"""
import asyncio
import inspect
import itertools
import os
import threading

import numpy as np

from . import config as _config
from . import utils
from .device import DeviceTile, SpilledTile, get_backend
from .residency import Residency

DEFAULT_BUCKET = "hbm"
DEFAULT_REGION = "local"


class _ObjectTable(object):
    """Process-wide (bucket, object key) -> tile / header table: the S3 stand-in."""

    def __init__(self):
        self.lock = threading.RLock()
        self.objects = {}   # (bucket, key_base) -> {tile_key: tile}
        self.headers = {}   # (bucket, key_base) -> header dict

    def tiles(self, bucket, key_base, create=True):
        with self.lock:
            d = self.objects.get((bucket, key_base))
            if d is None and create:
                d = self.objects[(bucket, key_base)] = {}
            return d

    def clear(self):
        """Forget every tile and header (tests)."""
        with self.lock:
            self.objects.clear()
            self.headers.clear()
            RESIDENCY.reset()


OBJECTS = _ObjectTable()
RESIDENCY = Residency(OBJECTS)   # which stored tiles are in HBM, which in pinned host DRAM (residency.py)


_FILE_TIER = {}


def _store_tier():
    """config.default()["store"]["tier"] without building the whole configuration for every stored tile: the environment
    override is looked up each time (tests switch it), the file's / default value once per configuration file."""
    tier = os.environ.get("NUMPYWREN_AMD_STORE")
    if tier:
        return tier
    path = os.environ.get("NUMPYWREN_AMD_CONFIG_FILE")
    if path not in _FILE_TIER:
        _FILE_TIER[path] = _config.default()["store"]["tier"]
    return _FILE_TIER[path]


class BigMatrix(object):
    """
    A multidimensional array stored as tiles in HBM, sharded in blocks of a given size.

    Parameters mirror the reference (numpywren/matrix.py:77-89): key, shape, shard_sizes, bucket,
    prefix, dtype, parent_fn, write_header, autosqueeze, lambdav, region, safe.  `bucket` only
    namespaces the in-process object table; `region` is kept for signature compatibility.
    """

    def __init__(self, key, shape=None, shard_sizes=None, bucket=DEFAULT_BUCKET, prefix='numpywren.objects/',
                 dtype=np.float64, parent_fn=None, write_header=False, autosqueeze=True, lambdav=0.0,
                 region=DEFAULT_REGION, safe=True):
        if bucket is None:
            bucket = os.environ.get('PYWREN_LINALG_BUCKET')
            if bucket is None:
                raise Exception("Bucket not provided and environment variable " +
                                "PYWREN_LINALG_BUCKET not provided.")
        self.bucket = bucket
        self.safe = safe
        self.prefix = prefix
        self.key = key
        self.key_base = os.path.join(prefix, self.key)
        self.dtype = dtype
        self.parent_fn = parent_fn
        self.transposed = False
        self.autosqueeze = autosqueeze
        self.lambdav = lambdav
        self.region = region
        if shape is None or shard_sizes is None:
            header = self.__read_header__()
        else:
            header = None
        if header is None and shape is None:
            raise Exception("Header doesn't exist and no shape provided.")
        elif shape is None:
            self.shard_sizes = header['shard_sizes']
            self.shape = header['shape']
            self.dtype = header['dtype']
        else:
            self.shape = shape
            self.shard_sizes = shard_sizes
            self.dtype = dtype
        if (self.shard_sizes is None) or (len(self.shape) != len(self.shard_sizes)):
            raise Exception("shard_sizes should be same length as shape.")
        self.symmetric = False
        if write_header:
            self.__write_header__()
        if (self.lambdav != 0 and (len(self.shape) < 2 or len(set(self.shape)) != 1)):
            raise Exception("Lambda can only be prescribed for square matrices/tensors")

    # ------------------------------------------------------------------ views
    def submatrix(self, *block_slices):
        """View restricted per axis by None | int | (stop) | (start, stop) | (start, stop, step), in
        block units (reference matrix.py:133-154)."""
        return BigMatrixView(self, [utils.convert_to_slice(s) for s in block_slices])

    @property
    def T(self):
        """Transposed view on the same tiles (each tile is transposed on get / before put)."""
        return BigMatrixView(self, [slice(None, None, None)] * len(self.shape), transposed=True)

    def num_blocks(self, axis=None):
        return len(self._block_idxs(axis=axis))

    # ------------------------------------------------------------------ block bookkeeping
    def _tiles(self, create=True):
        return OBJECTS.tiles(self.bucket, self.key_base, create)

    @property
    def blocks_exist(self):
        d = self._tiles(False) or {}
        out = []
        for k in list(d.keys()):
            b = block_key_to_block(k)
            if b is not None:
                out.append(b)
        return out

    @property
    def blocks_not_exist(self):
        return list(filter(lambda x: x, list(set(self.blocks_exist).symmetric_difference(set(self.blocks)))))

    @property
    def blocks(self):
        """Absolute (start, end) ranges of every block, C order."""
        return self._blocks()

    @property
    def block_idxs_exist(self):
        exist = set(self.blocks_exist)
        return [bidx for bidx, blk in zip(self.block_idxs, self.blocks) if blk in exist]

    @property
    def block_idxs_not_exist(self):
        return list(filter(lambda x: x, list(set(self.block_idxs_exist).symmetric_difference(set(self.block_idxs)))))

    @property
    def block_idxs(self):
        return self._block_idxs()

    def true_block_idx(self, *block_idx):
        return block_idx

    def _axis_ranges(self, ax):
        """[(start, end)] of the tiles along axis `ax`: tile t covers [t s, min((t + 1) s, n)) for
        t = 0 .. ceil(n / s) - 1 -- full shards, the last one cut at the matrix edge.  (A zero-length axis has no tile; the
        reference indexes [-1] of an empty list there: the same IndexError.)"""
        n, s = int(self.shape[ax]), int(self.shard_sizes[ax])
        count = -(-n // s)
        if count == 0:
            raise IndexError("list index out of range")
        return [(t * s, min((t + 1) * s, n)) for t in range(count)]

    def _blocks(self, axis=None):
        """Tile extents: every axis' ranges (axis=None: their C-order product) -- reference matrix.py:426-443."""
        if axis is not None and type(axis) is not int:
            raise Exception("Axis must be an integer.")
        per_axis = [self._axis_ranges(ax) for ax in range(len(self.shape))]
        return list(itertools.product(*per_axis)) if axis is None else per_axis[axis]

    def _block_idxs(self, axis=None):
        """Tile indices, same convention (reference matrix.py:448-455; its message differs from _blocks' by design of neither)."""
        if axis is not None and type(axis) != int:
            raise Exception("Axis must be integer")
        counts = [len(self._axis_ranges(ax)) for ax in range(len(self.shape))]
        if axis is None:
            return list(itertools.product(*[range(c) for c in counts]))
        return list(range(counts[axis]))

    def _register_parent(self, parent_fn):
        self.parent_fn = parent_fn

    def __block_idx_to_real_idx__(self, block_idx):
        """((start, end), ...) of tile `block_idx`; indices beyond the nominal grid are allowed (safe=False matrices) and get
        an empty or inverted range exactly as in the reference (matrix.py:480-488)."""
        return tuple((int(t) * s, min((int(t) + 1) * s, n)) for t, s, n in zip(block_idx, self.shard_sizes, self.shape))

    def __get_matrix_shard_key__(self, real_idxs):
        """Object name of a tile: "<start>_<end>_<shard>_" per axis under the matrix' key base (matrix.py:457-464)."""
        name = "".join("%d_%d_%d_" % (lo, hi, sz) for (lo, hi), sz in zip(real_idxs, self.shard_sizes))
        return os.path.join(self.key_base, name)

    def __shard_idx_to_key__(self, block_idx):
        # memoised per (shape, shard_sizes, key_base): the executor asks for the same few hundred keys on every read and
        # write of a run, and the string formatting was the largest single item of its host time per task
        memo = self.__dict__.get("_key_memo")
        sig = (self.key_base, tuple(self.shape), tuple(self.shard_sizes))
        if memo is None or memo[0] != sig:
            memo = self.__dict__["_key_memo"] = (sig, {})
        idx = tuple(block_idx)
        key = memo[1].get(idx)
        if key is None:
            key = memo[1][idx] = self.__get_matrix_shard_key__(self.__block_idx_to_real_idx__(idx))
        return key

    # ------------------------------------------------------------------ header
    def __read_header__(self):
        with OBJECTS.lock:
            return OBJECTS.headers.get((self.bucket, self.key_base))

    def __write_header__(self):
        with OBJECTS.lock:
            OBJECTS.headers[(self.bucket, self.key_base)] = {
                'shape': self.shape, 'shard_sizes': self.shard_sizes, 'dtype': self.dtype}

    def __delete_header__(self):
        with OBJECTS.lock:
            OBJECTS.headers.pop((self.bucket, self.key_base), None)

    # ------------------------------------------------------------------ tile read path
    def _check_arity(self, block_idx):
        if len(block_idx) != len(self.shape):
            raise Exception("Get block query does not match shape {0} vs {1}".format(block_idx, self.shape))

    def _on_diagonal(self, block_idx):
        return len(set(block_idx)) == 1 and len(set(self.shape)) == 1 and len(self.shape) != 1

    def _call_parent(self, block_idx):
        fn = self.parent_fn
        res = fn(self, None, *block_idx)
        if inspect.isawaitable(res):
            def drive(coro):
                loop = asyncio.new_event_loop()
                try:
                    return loop.run_until_complete(coro)
                finally:
                    loop.close()
            try:
                asyncio.get_running_loop()
            except RuntimeError:
                return drive(res)
            # called from inside a running event loop (get_block_async, RemoteRead.__call__, LambdaPackExecutor.run):
            # a second loop cannot run on this thread, so the parent coroutine gets a thread of its own
            box = {}

            def worker():
                try:
                    box["v"] = drive(res)
                except BaseException as e:   # re-raised on the caller's thread
                    box["e"] = e
            t = threading.Thread(target=worker, name="npw-parent-fn")
            t.start()
            t.join()
            if "e" in box:
                raise box["e"]
            return box["v"]
        return res

    def _raw(self, block_idx):
        """(stored object or None, key) without any read semantics applied."""
        key = self.__shard_idx_to_key__(block_idx)
        d = self._tiles(False)
        return (d.get(key) if d is not None else None), key

    def tile_exists(self, *block_idx):
        return self._raw(block_idx)[0] is not None

    def get_block(self, *block_idx):
        """Contents of one block as a (fresh) numpy array."""
        self._check_arity(block_idx)
        obj, key = self._raw(block_idx)
        if obj is None:
            if self.parent_fn is None:
                raise Exception("Key does {0} not exist, and no parent function prescripted".format(key))
            X_block = np.array(self._call_parent(block_idx))
        elif isinstance(obj, DeviceTile):
            X_block = get_backend().to_host(obj)
        elif isinstance(obj, SpilledTile):
            X_block = get_backend().spilled_to_numpy(obj)   # straight from pinned memory, no trip through HBM
        else:
            X_block = np.array(obj)
        if self.autosqueeze:
            X_block = np.squeeze(X_block)
        if self._on_diagonal(block_idx):
            idxs = np.diag_indices(X_block.shape[0])
            X_block[idxs] += self.lambdav
        return X_block

    async def get_block_async(self, loop, *block_idx):
        return self.get_block(*block_idx)

    def get_tile(self, *block_idx, stream=None):
        """Device-resident read: a DeviceTile with the same semantics as get_block.  The returned tile
        shares the stored buffer unless `lambdav` has to be applied (then it is a shifted copy): callers
        must treat it as read-only."""
        self._check_arity(block_idx)
        be = get_backend()
        obj, key = self._raw(block_idx)
        if obj is None:
            if self.parent_fn is None:
                raise Exception("Key does {0} not exist, and no parent function prescripted".format(key))
            zshape = getattr(self.parent_fn, "_npw_zero_shape", None)
            if zshape is not None:
                tile = be.shared_zeros(zshape(self, block_idx), np.float64)
            else:
                tile = be.to_device(np.asarray(self._call_parent(block_idx)), stream)
        elif isinstance(obj, DeviceTile):
            tile = obj
            with OBJECTS.lock:
                RESIDENCY.touch((self.bucket, self.key_base, key))
        elif isinstance(obj, SpilledTile):
            with OBJECTS.lock:
                cur, _ = self._raw(block_idx)   # somebody else may have restored it meanwhile
                if isinstance(cur, SpilledTile):
                    tile = RESIDENCY.restore((self.bucket, self.key_base, key), cur, be)
                else:
                    tile = cur if isinstance(cur, DeviceTile) else be.to_device(cur, stream)
        else:
            tile = be.to_device(obj, stream)
        if self.autosqueeze:
            sq = tuple(s for s in tile.shape if s != 1)
            if sq != tile.shape:
                tile = tile.reshaped(sq)
        if self._on_diagonal(block_idx) and self.lambdav != 0:
            tile = be.add_diag(tile, self.lambdav, stream)
        return tile

    # ------------------------------------------------------------------ tile write path
    def _target_shape(self, block, block_idx):
        real_idxs = self.__block_idx_to_real_idx__(block_idx)
        current_shape = tuple([e - s for s, e in real_idxs])
        shape = tuple(block.shape)
        if self.autosqueeze:
            if list(shape) == [x for x in current_shape if x != 1]:
                shape = current_shape
        if self.safe and shape != current_shape:
            raise Exception("{2} Incompatible block size: {0} vs {1}".format(shape, current_shape, self))
        return shape

    def put_block(self, block, *block_idx):
        """Store one block (ndarray or DeviceTile).  The dtype is kept as given (not cast)."""
        if isinstance(block, DeviceTile):
            return self.put_tile(block, *block_idx)
        block = np.asarray(block)
        shape = self._target_shape(block, block_idx)
        key = self.__shard_idx_to_key__(block_idx)
        if _store_tier() == "host":
            obj = np.array(block).reshape(shape)
        else:
            obj = get_backend().to_device(block.reshape(shape))
        self._store(key, obj)
        return None

    def _store(self, key, obj):
        with OBJECTS.lock:
            self._tiles()[key] = obj
            tkey = (self.bucket, self.key_base, key)
            RESIDENCY.note_put(tkey, obj)
            if isinstance(obj, DeviceTile):
                be = get_backend()
                RESIDENCY._hook(be)
                if RESIDENCY.plan is not None and RESIDENCY._budget is not None:
                    RESIDENCY.write_through(tkey, obj, be)
                RESIDENCY.enforce(protect=(tkey,))

    async def put_block_async(self, block, loop=None, *block_idx, no_overwrite=False):
        if no_overwrite and self.tile_exists(*block_idx):
            assert np.allclose(self.get_block(*block_idx), block)
        return self.put_block(block, *block_idx)

    def put_tile(self, tile, *block_idx):
        """Store a DeviceTile without copying (the table takes a reference to its buffer)."""
        shape = self._target_shape(tile, block_idx)
        if shape != tile.shape:
            tile = tile.reshaped(shape)
        key = self.__shard_idx_to_key__(block_idx)
        if _store_tier() == "host":
            obj = get_backend().to_host(tile)
        else:
            obj = tile
        self._store(key, obj)
        return None

    def delete_block(self, *block_idx):
        key = self.__shard_idx_to_key__(block_idx)
        with OBJECTS.lock:
            d = self._tiles(False)
            if d is not None:
                d.pop(key, None)
            RESIDENCY.note_delete((self.bucket, self.key_base, key))
        return None

    async def delete_block_async(self, loop=None, *block_idx):
        return self.delete_block(*block_idx)

    def free(self):
        """Delete all allocated blocks while leaving the matrix metadata intact."""
        with OBJECTS.lock:
            d = self._tiles(False)
            if d is not None:
                for key in d:
                    RESIDENCY.note_delete((self.bucket, self.key_base, key))
                d.clear()
        return 0

    def delete(self):
        """Completely remove the matrix (tiles and header)."""
        self.free()
        self.__delete_header__()
        with OBJECTS.lock:
            OBJECTS.objects.pop((self.bucket, self.key_base), None)
        return 0

    def numpy(self, workers=1):
        """Gather the whole matrix into one host ndarray (reference matrix.py:410-424)."""
        from . import matrix_utils
        return matrix_utils.get_local_matrix(self, workers)

    def __str__(self):
        return "{0}({1})".format(self.__class__.__name__, self.key)


class BigMatrixView(BigMatrix):
    """A block-sliced and/or transposed window onto a parent BigMatrix (reference matrix.py:562-782)."""

    def __init__(self, parent, parent_slices, transposed=False):
        self.parent = parent
        self.transposed = transposed
        self.bucket = parent.bucket
        self.prefix = parent.prefix
        self.key = parent.key
        self.key_base = parent.key_base
        self.dtype = parent.dtype
        self.parent_fn = parent.parent_fn
        self.autosqueeze = parent.autosqueeze
        self.lambdav = parent.lambdav
        self.safe = parent.safe
        self.region = getattr(parent, "region", DEFAULT_REGION)
        self.symmetric = False
        self.shard_sizes = parent.shard_sizes
        if isinstance(parent_slices, (int, slice)):
            parent_slices = [parent_slices]
        self.axis_lens = [int(np.ceil(parent.shape[i] / self.shard_sizes[i])) for i in range(len(parent.shape))]
        self.parent_slices = []
        shape = []
        for i, sl in enumerate(parent_slices):
            start = 0 if sl.start is None else sl.start
            stop = self.axis_lens[i] if sl.stop is None else sl.stop
            step = 1 if sl.step is None else sl.step
            extent = self.shard_sizes[i] * int(np.ceil((stop - start) / step))
            # the last view block may be the parent's ragged final block
            if (stop == self.axis_lens[i] and (stop - 1 - start) % step == 0 and
                    parent.shape[i] % self.shard_sizes[i] != 0):
                extent += parent.shape[i] % self.shard_sizes[i] - self.shard_sizes[i]
            shape.append(extent)
            self.parent_slices.append(slice(start, stop, step))
        for i in range(len(self.parent_slices), len(parent.shape)):
            self.parent_slices.append(slice(0, self.axis_lens[i], 1))
            shape.append(parent.shape[i])
        self.shape = shape
        if self.transposed:
            self.shape = tuple(reversed(self.shape))
            self.shard_sizes = tuple(reversed(self.shard_sizes))
        assert len(self.shard_sizes) == len(self.shape)

    # views of views and further transposes compose through the generic constructors
    @property
    def blocks_exist(self):
        raise NotImplementedError

    @property
    def blocks_not_exist(self):
        raise NotImplementedError

    @property
    def blocks(self):
        raise NotImplementedError

    def _blocks(self, axis=None):
        raise NotImplementedError

    @property
    def block_idxs(self):
        return self._block_idxs()

    @property
    def block_idxs_exist(self):
        return [self.__parent_to_view_block_idx__(p) for p in self.parent.block_idxs_exist
                if self.__is_valid_parent_block_idx__(p)]

    @property
    def block_idxs_not_exist(self):
        return [self.__parent_to_view_block_idx__(p) for p in self.parent.block_idxs_not_exist
                if self.__is_valid_parent_block_idx__(p)]

    def true_block_idx(self, *block_idx):
        return self.parent.true_block_idx(*self.__view_to_parent_block_idx__(block_idx))

    def _block_idxs(self, axis=None):
        if axis is None:
            per_axis = [self._block_idxs(axis=a) for a in range(len(self.shape))]
            return list(itertools.product(*per_axis))
        parent_axis = self.__view_to_parent_axis__(axis)
        out = []
        for p in self.parent._block_idxs(axis=parent_axis):
            if self.__is_valid_parent_block_idx__(p, axis=parent_axis):
                out.append(self.__parent_to_view_block_idx__(p, axis=parent_axis))
        return out

    def __view_to_parent_axis__(self, view_axis):
        if self.transposed:
            view_axis = len(self.shape) - view_axis - 1
        return view_axis

    def __view_to_parent_block_idx__(self, view_idx):
        idx = list(view_idx)
        if len(view_idx) < len(self.shape):
            # single-block axes may be omitted from the index
            for i in range(len(self.shape)):
                if self.shape[i] <= self.shard_sizes[i]:
                    idx.insert(i, 0)
        if len(idx) != len(self.shape):
            raise ValueError("Invalid index length.")
        if self.transposed:
            idx = list(reversed(idx))
        parent_idx = []
        for sl, elt in zip(self.parent_slices, idx):
            p = elt * sl.step + sl.start
            if p < 0:
                raise NotImplementedError
            if p >= sl.stop:
                raise IndexError("Array index out of bounds.")
            parent_idx.append(p)
        return tuple(parent_idx)

    def __parent_to_view_block_idx__(self, parent_idx, axis=None):
        if axis is not None:
            sl = self.parent_slices[axis]
            return (parent_idx - sl.start) // sl.step
        view = tuple((p - sl.start) // sl.step for p, sl in zip(parent_idx, self.parent_slices))
        return tuple(reversed(view)) if self.transposed else view

    def __is_valid_parent_block_idx__(self, parent_idx, axis=None):
        if axis is not None:
            pairs = [(parent_idx, self.parent_slices[axis])]
        else:
            pairs = list(zip(parent_idx, self.parent_slices))
        for p, sl in pairs:
            if p < 0:
                raise NotImplementedError("Negative indexing not yet supported.")
            if p < sl.start or p >= sl.stop or (p - sl.start) % sl.step != 0:
                return False
        return True

    # ------------------------------------------------------------------ data access through the parent
    def get_block(self, *block_idx):
        block = self.parent.get_block(*self.__view_to_parent_block_idx__(block_idx))
        return block.T if self.transposed else block

    async def get_block_async(self, loop, *block_idx):
        return self.get_block(*block_idx)

    def get_tile(self, *block_idx, stream=None):
        tile = self.parent.get_tile(*self.__view_to_parent_block_idx__(block_idx), stream=stream)
        return get_backend().transpose(tile, stream) if self.transposed else tile

    def put_block(self, block, *block_idx):
        if isinstance(block, DeviceTile):
            return self.put_tile(block, *block_idx)
        if self.transposed:
            block = np.asarray(block).T
        return self.parent.put_block(block, *self.__view_to_parent_block_idx__(block_idx))

    async def put_block_async(self, block, loop=None, *block_idx):
        return self.put_block(block, *block_idx)

    def put_tile(self, tile, *block_idx):
        if self.transposed:
            tile = get_backend().transpose(tile)
        return self.parent.put_tile(tile, *self.__view_to_parent_block_idx__(block_idx))

    def tile_exists(self, *block_idx):
        return self.parent.tile_exists(*self.__view_to_parent_block_idx__(block_idx))

    def delete_block(self, *block_idx):
        return self.parent.delete_block(*self.__view_to_parent_block_idx__(block_idx))

    async def delete_block_async(self, loop, *block_idx):
        return self.delete_block(*block_idx)

    def free(self):
        for idx in self._block_idxs():
            self.delete_block(*idx)
        return 0

    def __str__(self):
        reps = []
        last = 0
        for i, (sl, n) in enumerate(zip(self.parent_slices, self.axis_lens)):
            if sl != slice(0, n, 1):
                last = i
            if sl.start == sl.stop - 1:
                reps.append(str(sl.start))
            else:
                step = "" if sl.step == 1 else ":" + str(sl.step)
                start = "" if sl.start == 0 else str(sl.start)
                stop = "" if sl.stop == n else str(sl.stop)
                reps.append(start + ":" + stop + step)
        rep = self.parent.__str__()
        if last != 0:
            rep += "[" + ",".join(reps[:last + 1]) + "]"
        if self.transposed:
            rep += ".T"
        rep += str(tuple(self.shape))
        return rep


def block_key_to_block(key):
    """Parse "…/{s}_{e}_{shard}_{s}_{e}_{shard}_" back into ((s, e), …); None for the header object
    (same contract as reference numpywren/matrix_utils.py:123-139)."""
    block_key = key.strip().split("/")[-1]
    if block_key == "header":
        return None
    parts = block_key.strip('_').split("_")
    assert len(parts) % 3 == 0
    return tuple((int(parts[i]), int(parts[i + 1])) for i in range(0, len(parts), 3))
