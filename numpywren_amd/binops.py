"""Binary-operator surface of the reference (numpywren/binops.py).

`gemm` there is a non-LambdaPACK blocked matmul fanned out with pywren.map -- every experiment uses it to form
X.X^T before the factorisation (binops.py:107-174 with the per-block worker _gemm_remote_0, 20-34).  Here the
same output matrix (key `gemm(BigMatrix(x), BigMatrix(y))`, shape, shard sizes, dtype, header) is produced by the
GPU the caller runs on: one accumulating MFMA GEMM per (output block, reduction index), tiles resident in HBM.
`pwex`, `tasks_per_job`, `local`, `gemm_impl`, `gemm_chunk_size` are accepted and ignored (they steer the pywren
fan-out).  The remaining names are stubs in the reference too."""
import numpy as np

from . import matrix_utils
from .device import get_backend
from .matrix import BigMatrix


def gemm(pwex, X, Y, out_bucket=None, tasks_per_job=1, local=False, dtype=np.float64, overwrite=True, gemm_impl=0,
         gemm_chunk_size=16):
    """XY = X . Y over BigMatrices (or views such as X.T).  Reference binops.py:107-174."""
    reduce_idxs = Y._block_idxs(axis=0)
    if out_bucket is None:
        out_bucket = X.bucket
    root_key = matrix_utils.generate_key_name_binop(X, Y, "gemm")
    if Y.shard_sizes[0] != X.shard_sizes[1]:
        raise Exception("X dim 1 shard size must match Y dim 0 shard size")
    XY = BigMatrix(root_key, shape=(X.shape[0], Y.shape[1]), bucket=out_bucket,
                   shard_sizes=[X.shard_sizes[0], Y.shard_sizes[1]], dtype=dtype, write_header=True)
    todo = list(XY.block_idxs) if overwrite else list(XY.block_idxs_not_exist)
    be = get_backend()
    f32 = np.dtype(dtype) == np.float32
    for i, j in todo:
        acc = None
        for r in reduce_idxs:
            a, b = X.get_tile(i, r), Y.get_tile(r, j)
            if f32:
                a, b = be.convert(a, np.float32), be.convert(b, np.float32)
            else:
                a, b = be.as_f64(a), be.as_f64(b)
            acc = be.gemm(a, b, False, False, alpha=1.0, beta=0.0 if acc is None else 1.0, C=acc, out=acc)
        XY.put_tile(acc, i, j)
    return XY


def gemm_with_prefetch(X, Y, bidx0, bidx1, block_chunk_size=16):
    """One output block (bidx0, bidx1) of X . Y as an ndarray (reference binops.py:60-105: a double-buffered S3
    prefetch around `result += b1.dot(b2)`).  Here the operand tiles are already in HBM and the products accumulate
    in one GPU buffer; `block_chunk_size` (the prefetch depth) has no counterpart."""
    assert X._block_idxs(1) == Y._block_idxs(0)
    be = get_backend()
    acc = None
    for r in X._block_idxs(1):
        a, b = be.as_f64(X.get_tile(bidx0, r)), be.as_f64(Y.get_tile(r, bidx1))
        acc = be.gemm(a, b, False, False, alpha=1.0, beta=0.0 if acc is None else 1.0, C=acc, out=acc)
    return be.to_host(acc)


def _stub(name):
    def f(*args, **kwargs):
        raise NotImplementedError(f"binops.{name} is not implemented (a stub in the reference as well)")

    f.__name__ = name
    return f


for _n in ("gemv", "syrk", "posv", "add", "sub", "mul", "div", "logical_and", "logical_or", "xor", "elemwise_binop_func",
           "trisolve"):
    globals()[_n] = _stub(_n)
