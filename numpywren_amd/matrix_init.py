"""Host ndarray <-> BigMatrix tiles (reference numpywren/matrix_init.py): `shard_matrix` is the
host -> HBM scatter, `local_numpy_init` names the matrix by the hash of its contents."""
import itertools

import numpy as np

from . import matrix, matrix_utils
from .matrix import BigMatrix
from .matrix_utils import generate_key_name_local_matrix


def shard_matrix(bigm, X_local, n_jobs=1, executor=None, overwrite=True):
    """Scatter the host array into the tiles of `bigm` (reference matrix_init.py:73-96: one
    put_block per block; the reference's /tmp memmap + thread pool is an S3 bandwidth trick with no
    equivalent need here)."""
    if overwrite:
        all_bidxs, all_blocks = bigm.block_idxs, bigm.blocks
    else:
        all_bidxs, all_blocks = bigm.block_idxs_not_exist, bigm.blocks_not_exist
    X_local = np.asarray(X_local)
    for bidxs, blocks in zip(all_bidxs, all_blocks):
        slices = tuple(slice(s, e) for s, e in blocks)
        bigm.put_block(np.ascontiguousarray(X_local[slices]), *bidxs)
    return bigm


def local_numpy_init(X_local, shard_sizes, n_jobs=1, symmetric=False, exists=False, executor=None,
                     write_header=False, bucket=matrix.DEFAULT_BUCKET, overwrite=True):
    key = generate_key_name_local_matrix(X_local)
    bigm = BigMatrix(key, shape=X_local.shape, shard_sizes=shard_sizes, dtype=X_local.dtype,
                     write_header=write_header, bucket=bucket)
    if not exists:
        return shard_matrix(bigm, X_local, n_jobs=n_jobs, executor=executor, overwrite=overwrite)
    return bigm


def empty_result_matrix(X_sharded, function, args, shape=None, shard_sizes=None, symmetric=False, dtype=None,
                        write_header=False):
    """An (empty) BigMatrix named by the hash of (function, source key, args) -- reference matrix_init.py:33-49.  The
    reference's `symmetric=True` branch names a class that does not exist there either."""
    if symmetric:
        raise NotImplementedError("BigSymmetricMatrix does not exist (reference matrix_init.py:47-48 would raise NameError)")
    dtype = X_sharded.dtype if dtype is None else dtype
    shape = X_sharded.shape if shape is None else shape
    shard_sizes = X_sharded.shard_sizes if shard_sizes is None else shard_sizes
    key = matrix_utils.hash_string(matrix_utils.hash_function(function) + X_sharded.key + matrix_utils.hash_args(args))
    return BigMatrix(key, shape=shape, shard_sizes=shard_sizes, dtype=dtype, write_header=write_header, bucket=X_sharded.bucket)


def reshard_down(bigm, breakdowns, pwex=None):
    """A new BigMatrix whose shard sizes are bigm.shard_sizes / breakdowns: every block of `bigm` becomes
    prod(breakdowns) sub-blocks (reference matrix_init.py:100-148; `pwex` fanned that out over Lambdas).  2-D tiles are
    cut on the GPU (strided device copies); other ranks go through the host."""
    for x, y in zip(bigm.shard_sizes, breakdowns):
        assert x % y == 0
    new_shard_sizes = [int(x / y) for x, y in zip(bigm.shard_sizes, breakdowns)]
    X_new = BigMatrix("reshard({0},{1})".format(bigm.key, breakdowns), bucket=bigm.bucket, shape=bigm.shape,
                      shard_sizes=new_shard_sizes, dtype=bigm.dtype)
    on_device = len(bigm.shape) == 2 and matrix._store_tier() == "hbm"
    be = matrix.get_backend() if on_device else None
    for idx_old in bigm._block_idxs():
        old_ranges = bigm.__block_idx_to_real_idx__(idx_old)
        offsets = [r[0] for r in old_ranges]
        per_axis = []
        for ax, (s, e) in enumerate(old_ranges):
            first = s // new_shard_sizes[ax]
            count = -(-(e - s) // new_shard_sizes[ax])
            per_axis.append(range(first, first + count))
        data = bigm.get_tile(*idx_old) if on_device else bigm.get_block(*idx_old)
        if not on_device:
            data = np.reshape(data, [e - s for s, e in old_ranges])
        for idx_new in itertools.product(*per_axis):
            rng = X_new.__block_idx_to_real_idx__(idx_new)
            local = [(s - o, e - o) for (s, e), o in zip(rng, offsets)]
            if on_device:
                (r0, r1), (c0, c1) = local
                X_new.put_tile(be.block(data, r0, r1, c0, c1), *idx_new)
            else:
                X_new.put_block(np.ascontiguousarray(data[tuple(slice(s, e) for s, e in local)]), *idx_new)
    return X_new
