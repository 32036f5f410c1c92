"""Helpers around BigMatrix with the reference's names (numpywren/matrix_utils.py): the
`constant_zeros*` parent functions used by alg_wrappers, key-name / hash helpers, and the bulk
gather behind BigMatrix.numpy().  The reference's S3 listing / mmap download plumbing has no
counterpart here (tiles are already local to the node).
"""
import hashlib
import inspect
import itertools
import pickle

import numpy as np

from .matrix import block_key_to_block  # noqa: F401  (re-export, same name as the reference)
from .utils import chunk  # noqa: F401


def hash_string(s):
    return hashlib.sha1(s.encode('utf-8')).hexdigest()


def hash_array(s):
    s = np.ascontiguousarray(s)
    return hashlib.sha1(s.view(np.uint8)).hexdigest()


def hash_function(f):
    return hashlib.sha1(inspect.getsource(f).encode()).hexdigest()


def hash_bytes(byte_string):
    return hashlib.sha1(byte_string.encode('utf-8')).hexdigest()


def hash_args(args):
    return hashlib.sha1(pickle.dumps(args)).hexdigest()


def generate_key_name_binop(X, Y, op):
    assert op == "gemm" or op == "trisolve"
    return "{0}({1}, {2})".format(op, str(X), str(Y))


def generate_key_name_uop(X, op):
    assert op == "chol"
    return "chol({0})".format(str(X))


def generate_key_name_local_matrix(X_local):
    return hash_array(X_local)


def _real_shape(bigm, block_idx):
    return tuple(e - s for s, e in bigm.__block_idx_to_real_idx__(block_idx))


def make_constant_parent(cnst):
    def constant_parent(bigm, *block_idx):
        return np.full(_real_shape(bigm, block_idx), cnst)
    return constant_parent


async def constant_zeros(bigm, loop, *block_idx):
    """parent_fn: a missing tile reads as zeros of the tile's real (edge-truncated) shape
    (reference matrix_utils.py:314-317)."""
    return np.zeros(_real_shape(bigm, block_idx))


constant_zeros._npw_zero_shape = _real_shape


def _ext_shape(bigm, block_idx):
    return (bigm.shard_sizes[-1], bigm.shard_sizes[-1])


async def constant_zeros_ext(bigm, loop, *block_idx):
    """parent_fn: zeros of shape (shard[-1], shard[-1]) whatever the index (reference
    matrix_utils.py:319-325; used by BDFAC's L_LQ / S_LQ)."""
    return np.zeros(_ext_shape(bigm, block_idx))


constant_zeros_ext._npw_zero_shape = _ext_shape


def get_local_matrix(bigm, workers=1, mmap_loc=None, big_axis=0):
    """Gather every block of `bigm` (matrix or view) into one host array: D2H copies of the tiles
    placed at their element ranges (reference matrix_utils.py:156-167, 258-304 without the process
    pool / /dev/shm memmap)."""
    out = np.zeros(tuple(bigm.shape), dtype=bigm.dtype)
    per_axis = [bigm._block_idxs(i) for i in range(len(bigm.shape))]
    for bidx in itertools.product(*per_axis):
        real = BigMatrixRealIdx(bigm, bidx)
        sl = tuple(slice(s, e) for s, e in real)
        out[sl] = np.asarray(bigm.get_block(*bidx)).reshape(out[sl].shape)  # autosqueeze drops singleton shard axes
    return out


def BigMatrixRealIdx(bigm, bidx):
    # element ranges in the coordinates of `bigm` itself (a view uses its own shape / shard sizes)
    out = []
    for i in range(len(bigm.shape)):
        start = bidx[i] * bigm.shard_sizes[i]
        out.append((start, min(start + bigm.shard_sizes[i], bigm.shape[i])))
    return tuple(out)


def get_row(bigm, row, workers=1, mmap_loc=None):
    assert len(bigm.shape) == 2
    return np.hstack([np.atleast_2d(bigm.get_block(row, j)) for j in bigm._block_idxs(1)])


def get_col(bigm, col, workers=1, mmap_loc=None):
    assert len(bigm.shape) == 2
    return np.vstack([np.atleast_2d(bigm.get_block(i, col)) for i in bigm._block_idxs(0)])


def get_rows(bigm, rows, workers=1, mmap_loc=None):
    """Several block rows stacked (reference matrix_utils.py:221-232; there through /dev/shm memory maps)."""
    assert len(bigm.shape) == 2
    return np.vstack([get_row(bigm, r) for r in rows])


def chunk(l, n):
    """Yield successive n-sized chunks from l (reference matrix_utils.py:56-60)."""
    if n == 0:
        return []
    for i in range(0, len(l), n):
        yield l[i:i + n]


def block_key_to_block(key):
    """Object key -> ((start, end), ...) block ranges, None for the header (reference matrix_utils.py:123-139)."""
    from .matrix import block_key_to_block as _impl
    return _impl(key)


def put_row(bigm, data, row, workers=1, mmap_loc=None, big_axis=0):
    assert len(bigm.shape) == 2
    for bidx, block in zip(bigm.block_idxs, bigm.blocks):
        if bidx[0] == row:
            bigm.put_block(np.ascontiguousarray(data[:, block[1][0]:block[1][1]]), *bidx)


def put_col(bigm, data, col, workers=1, mmap_loc=None, big_axis=0):
    assert len(bigm.shape) == 2
    for bidx, block in zip(bigm.block_idxs, bigm.blocks):
        if bidx[1] == col:
            bigm.put_block(np.ascontiguousarray(data[block[0][0]:block[0][1], :]), *bidx)
