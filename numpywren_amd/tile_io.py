"""Host / disk form of a BigMatrix: the reference's object format on a file system.

The reference keeps every tile as one S3 object holding `np.save` bytes under the key
`prefix + key + "/" + "{start}_{end}_{shard}_"...` (numpywren/matrix.py:457-464, 519-533) and one `header` object
with JSON `{"shape", "shard_sizes", "dtype"}` where dtype is base64(pickle(dtype)) (matrix.py:535-556).  This
module reads and writes exactly those objects as files `<root>/<bucket>/<object key>`, so

  * a bucket copied to disk (`aws s3 sync s3://bucket root/bucket`) loads into HBM with `import_matrix`,
  * `export_matrix` output is readable by the reference (or by NumPy alone),
  * `spill` / `restore` give the HBM store a disk tier: tiles leave HBM and come back bit for bit.

Tiles are written in the full shard shape (as the reference stores them, before autosqueeze) and in the
matrix's tile dtype; nothing is cast.
"""
import base64
import io
import json
import os
import pickle

import numpy as np

from .matrix import DEFAULT_BUCKET, OBJECTS, BigMatrix, block_key_to_block

DEFAULT_PREFIX = "numpywren.objects/"

_DTYPE_NAMES = {"float64", "float32", "float16", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16",
                "uint8", "bool_", "complex64", "complex128", "dtype"}


class _DtypeUnpickler(pickle.Unpickler):
    """The header's dtype field is a pickle; only NumPy's scalar types / dtype may come out of it."""

    def find_class(self, module, name):
        if module in ("numpy", "numpy.core.multiarray", "numpy._core.multiarray") and name in _DTYPE_NAMES:
            return getattr(np, name)
        raise pickle.UnpicklingError(f"header dtype refers to {module}.{name}: refused")


def encode_dtype(dtype):
    """reference matrix.py:548-551."""
    return base64.b64encode(pickle.dumps(dtype)).decode("utf-8")


def decode_dtype(text):
    """reference matrix.py:553-556, restricted to NumPy dtypes."""
    return _DtypeUnpickler(io.BytesIO(base64.b64decode(text))).load()


def _object_path(root, bucket, key):
    return os.path.join(root, bucket, *key.split("/"))


def _host_array(tile):
    if isinstance(tile, np.ndarray):
        return tile
    from .device import SpilledTile, get_backend
    if isinstance(tile, SpilledTile):
        return get_backend().spilled_to_numpy(tile)
    return get_backend().to_host(tile)


def tile_bytes(array):
    """The object body of one tile: np.save bytes (reference matrix.py:526-527)."""
    bio = io.BytesIO()
    np.save(bio, np.ascontiguousarray(array))
    return bio.getvalue()


def export_matrix(bigm, root, atomic=False, return_keys=False):
    """Write the header object and every stored tile of `bigm` under `root`; returns the number of tile objects (or, with
    return_keys, their object names relative to the matrix).  atomic: every object goes to a temporary name first and is
    renamed into place (checkpoint.save)."""
    def put(path, data, mode):
        if not atomic:
            with open(path, mode) as f:
                f.write(data)
            return
        tmp = path + ".tmp"
        with open(tmp, mode) as f:
            f.write(data)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)

    hdr = {"shape": [int(s) for s in bigm.shape], "shard_sizes": [int(s) for s in bigm.shard_sizes],
           "dtype": encode_dtype(bigm.dtype)}
    hpath = _object_path(root, bigm.bucket, os.path.join(bigm.key_base, "header"))
    os.makedirs(os.path.dirname(hpath), exist_ok=True)
    put(hpath, json.dumps(hdr), "w")
    with OBJECTS.lock:
        tiles = dict(OBJECTS.tiles(bigm.bucket, bigm.key_base, create=False) or {})
    for key, tile in tiles.items():
        path = _object_path(root, bigm.bucket, key)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        put(path, tile_bytes(_host_array(tile)), "wb")
    if return_keys:
        return sorted(os.path.basename(k) for k in tiles)
    return len(tiles)


def read_header(root, key, bucket=DEFAULT_BUCKET, prefix=DEFAULT_PREFIX):
    with open(_object_path(root, bucket, os.path.join(prefix + key, "header"))) as f:
        hdr = json.load(f)
    return {"shape": tuple(hdr["shape"]), "shard_sizes": tuple(hdr["shard_sizes"]), "dtype": decode_dtype(hdr["dtype"])}


def _load_tiles(bigm, root, only=None):
    base = _object_path(root, bigm.bucket, bigm.key_base)
    n = 0
    if not os.path.isdir(base):
        return 0
    for name in sorted(os.listdir(base)):
        if name.endswith(".tmp") or (only is not None and name not in only):
            continue            # (an interrupted atomic write; an object that is not part of the checkpoint being loaded)
        ranges = block_key_to_block(name)
        if ranges is None:      # the header object
            continue
        with open(os.path.join(base, name), "rb") as f:
            arr = np.load(io.BytesIO(f.read()), allow_pickle=False)
        idx = tuple(s // sh for (s, _), sh in zip(ranges, bigm.shard_sizes))
        bigm.put_block(arr, *idx)
        n += 1
    return n


def import_matrix(root, key, bucket=DEFAULT_BUCKET, prefix=DEFAULT_PREFIX, **kwargs):
    """Build the BigMatrix described by `<root>/<bucket>/<prefix><key>/header` and load its tile objects into the
    current store tier (HBM by default).  Extra kwargs go to BigMatrix (parent_fn, lambdav, safe, ...)."""
    hdr = read_header(root, key, bucket, prefix)
    m = BigMatrix(key, shape=hdr["shape"], shard_sizes=hdr["shard_sizes"], bucket=bucket, prefix=prefix,
                  dtype=hdr["dtype"], write_header=True, **kwargs)
    _load_tiles(m, root)
    return m


def spill(bigm, root):
    """Disk tier: write the matrix out and release its tiles (HBM or host); the metadata stays."""
    n = export_matrix(bigm, root)
    bigm.free()
    return n


def restore(bigm, root, only=None):
    """Inverse of spill: bring the tile objects under `root` back into the store (`only`: just these object names)."""
    return _load_tiles(bigm, root, only)
