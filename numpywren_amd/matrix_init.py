"""Host ndarray <-> BigMatrix tiles (reference numpywren/matrix_init.py): `shard_matrix` is the
host -> HBM scatter, `local_numpy_init` names the matrix by the hash of its contents."""
import numpy as np

from . import matrix
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
