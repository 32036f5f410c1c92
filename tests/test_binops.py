"""binops.gemm (reference numpywren/binops.py:107-174): the non-LambdaPACK blocked matmul every experiment uses to form
X.X^T.  Output matrix identity (key, shape, shard sizes, dtype) as in the reference; values against NumPy."""
import numpy as np
import pytest

from numpywren_amd import binops
from numpywren_amd.matrix import BigMatrix
from numpywren_amd.matrix_init import shard_matrix


def _check(dtype, tol):
    rng = np.random.default_rng(12)
    Xh = rng.standard_normal((40, 24)).astype(dtype)
    X = BigMatrix("binop_X", shape=Xh.shape, shard_sizes=(16, 8), dtype=dtype)      # ragged last row block
    shard_matrix(X, Xh)
    XXT = binops.gemm(None, X, X.T, dtype=dtype)
    assert XXT.key == "gemm(BigMatrix(binop_X), BigMatrix(binop_X).T)" or XXT.key.startswith("gemm(BigMatrix(binop_X)")
    assert XXT.shape == (40, 40) and tuple(XXT.shard_sizes) == (16, 16)
    got = XXT.numpy()
    assert got.dtype == np.dtype(dtype)
    np.testing.assert_allclose(got, Xh @ Xh.T, rtol=tol, atol=tol)
    Yh = rng.standard_normal((24, 20)).astype(dtype)
    Y = BigMatrix("binop_Y", shape=Yh.shape, shard_sizes=(8, 10), dtype=dtype)
    shard_matrix(Y, Yh)
    XY = binops.gemm(None, X, Y, dtype=dtype)
    assert XY.shape == (40, 20) and tuple(XY.shard_sizes) == (16, 10)
    np.testing.assert_allclose(XY.numpy(), Xh @ Yh, rtol=tol, atol=tol)
    # one output block the way the reference's prefetching worker forms it (binops.py:60-105)
    blk = binops.gemm_with_prefetch(X, Y, 1, 1)
    np.testing.assert_allclose(blk, (Xh @ Yh)[16:32, 10:20], rtol=tol * 50, atol=tol * 50)
    Z = BigMatrix("binop_Z", shape=(24, 20), shard_sizes=(6, 10))
    with pytest.raises(Exception, match="shard size"):
        binops.gemm(None, X, Z)


def test_gemm_host_logic(oracle_backend):
    _check(np.float64, 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
def test_gemm_gpu(dtype, tol, hbm_store):
    _check(dtype, tol)


def test_stub_surface_matches_the_reference():
    """Everything else in binops / uops raises NotImplementedError in the reference as well: same names here."""
    from numpywren_amd import uops
    for name in ("gemv", "syrk", "posv", "add", "sub", "mul", "div", "logical_and", "logical_or", "xor",
                 "elemwise_binop_func", "trisolve"):
        with pytest.raises(NotImplementedError):
            getattr(binops, name)(None, None, None)
    for name in ("reshard", "sum", "prod", "min", "max", "norm", "abs", "neg", "square", "sqrt", "sin", "cos", "tan", "exp",
                 "sign", "elemwise_uop_func", "power"):
        with pytest.raises(NotImplementedError):
            getattr(uops, name)(None, None)
