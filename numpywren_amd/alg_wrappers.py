"""User API: build the output / intermediate BigMatrices and the LambdaPACK program for a
blocked algorithm.  Same functions, matrix names, shapes, shard sizes, `safe` / `parent_fn`
choices and return value `(LambdaPackProgram, {"outputs", "intermediates", "compile_time"})` as
the reference (numpywren/alg_wrappers.py:16-114), so callers switch by changing the import.
"""
import time

import numpy as np

from . import config as npw_config
from . import lambdapack as lp
from .algs import BDFAC, CHOLESKY, GEMM, QR, TSQR
from .compiler import lpcompile_for_execution
from .matrix import BigMatrix
from .matrix_utils import constant_zeros, constant_zeros_ext


def _levels(num_blocks, fan_in):
    return max(int(np.ceil(np.log2(num_blocks) / np.log2(fan_in))), 1)


def cholesky(X, truncate=0):
    """Lower Cholesky factor of the SPD BigMatrix X (reference alg_wrappers.py:16-27)."""
    n, b = X.shape[0], X.shard_sizes[0]
    S = BigMatrix("Cholesky.Intermediate({0})".format(X.key), shape=(X.num_blocks(1) + 1, n, n),
                  shard_sizes=(1, b, b), bucket=X.bucket, write_header=True, parent_fn=constant_zeros)
    O = BigMatrix("Cholesky({0})".format(X.key), shape=(n, n), shard_sizes=(b, b), write_header=True,
                  parent_fn=constant_zeros)
    t = time.time()
    p1 = lpcompile_for_execution(CHOLESKY, inputs=["I"], outputs=["O"])(O, X, S, int(np.ceil(n / b)), truncate)
    c_time = time.time() - t
    program = lp.LambdaPackProgram(p1, config=npw_config.default())
    return program, {"outputs": [O], "intermediates": [S], "compile_time": c_time}


def tsqr(X, truncate=0):
    """R factor of the tall-skinny BigMatrix X by a binary reduction tree (reference alg_wrappers.py:30-47)."""
    b_fac = 2
    assert X.shard_sizes[1] == X.shape[1]
    b = X.shard_sizes[0]
    levels = _levels(X.num_blocks(0), b_fac)
    R_sharded = BigMatrix("tsqr_R({0})".format(X.key), shape=(levels * b, X.shape[0]), shard_sizes=X.shard_sizes,
                          write_header=True, safe=False)
    T_sharded = BigMatrix("tsqr_T({0})".format(X.key), shape=(levels * b * b_fac, X.shape[0]),
                          shard_sizes=(b * b_fac, b), write_header=True, safe=False)
    V_sharded = BigMatrix("tsqr_V({0})".format(X.key), shape=(levels * b * b_fac, X.shape[0]),
                          shard_sizes=(b * b_fac, b), write_header=True, safe=False)
    t = time.time()
    p1 = lpcompile_for_execution(TSQR, inputs=["A"], outputs=["Rs"])(X, V_sharded, T_sharded, R_sharded,
                                                                     X.num_blocks(0))
    c_time = time.time() - t
    program = lp.LambdaPackProgram(p1, config=npw_config.default())
    return program, {"outputs": [R_sharded, V_sharded, T_sharded], "intermediates": [], "compile_time": c_time}


def gemm(A, B):
    """C = A.B with a fan-in-4 reduction tree over the contraction blocks (reference alg_wrappers.py:49-65).
    As in the reference the program receives (M, N, K) = (A.num_blocks(0), A.num_blocks(1), B.num_blocks(1)),
    which is only consistent for square block grids."""
    b_fac = 4
    assert A.shape[1] == B.shape[0]
    assert A.shard_sizes[1] == B.shard_sizes[0]
    levels = _levels(A.num_blocks(1), b_fac)
    Temp = BigMatrix(f"matmul_test_Temp({A.key},{B.key})", shape=(A.shape[0], B.shape[1], B.shape[0], levels),
                     shard_sizes=[A.shard_sizes[0], B.shard_sizes[1], 1, 1], write_header=True, safe=False,
                     parent_fn=constant_zeros)
    C_sharded = BigMatrix("matmul_test_C", shape=(A.shape[0], B.shape[1]),
                          shard_sizes=(A.shard_sizes[0], B.shard_sizes[1]), write_header=True)
    t = time.time()
    p1 = lpcompile_for_execution(GEMM, inputs=["A", "B"], outputs=["Out"])(
        A, B, A.num_blocks(0), A.num_blocks(1), B.num_blocks(1), Temp, C_sharded)
    c_time = time.time() - t
    program = lp.LambdaPackProgram(p1, config=npw_config.default())
    return program, {"outputs": [C_sharded], "intermediates": [Temp], "compile_time": c_time}


def qr(A):
    """Blocked Householder QR (reference alg_wrappers.py:67-89): same matrices, names, shapes and program.
    Runs as the reference does, including what kernels.qr_factor_triangular and kernels.qr_leaf do as written
    (only the first block row of Rs is a true QR factor of A; tests/golden/make_golden_qr.py pins the rest)."""
    b_fac = 2
    N = A.shape[0]
    b = A.shard_sizes[0]
    levels = _levels(A.num_blocks(0), b_fac) + 1
    mk = lambda name, shape, shards: BigMatrix(name, shape=shape, shard_sizes=shards, write_header=True,
                                               parent_fn=constant_zeros, safe=False)
    Vs = mk("Vs", (2 * N, 2 * N, levels), (b, b, 1))
    Ts = mk("Ts", (2 * N, 2 * N, levels), (b, b, 1))
    Rs = mk("Rs", (2 * N, 2 * N, levels), (b, b, 1))
    Ss = mk("Ss", (2 * N, 2 * N, 2 * N, levels * b), (b, b, 1, 1))
    t = time.time()
    p1 = lpcompile_for_execution(QR, inputs=["I"], outputs=["Rs"])(A, Vs, Ts, Rs, Ss, A.num_blocks(0), 0)
    c_time = time.time() - t
    program = lp.LambdaPackProgram(p1, config=npw_config.default())
    return program, {"outputs": [Rs, Vs, Ts], "intermediates": [Ss], "compile_time": c_time}


def bdfac(A, truncate=0):
    """Reduction to block-bidiagonal form by alternating TSQR / TSLQ sweeps (reference alg_wrappers.py:92-114)."""
    b_fac = 2
    N = A.shape[0]
    b = A.shard_sizes[0]
    levels = _levels(A.num_blocks(0), b_fac) + 1
    mk = lambda name, shape, shards, pf=None: BigMatrix(name, shape=shape, shard_sizes=shards, write_header=True,
                                                        safe=False, parent_fn=pf)
    V_QR = mk("V_QR", (2 * N, levels, 2 * N), (1, 1, b))
    T_QR = mk("T_QR", (2 * N, levels, 2 * N), (1, 1, b))
    R_QR = mk("R_QR", (2 * N, levels, 2 * N), (b, 1, b), constant_zeros)
    S_QR = mk("S_QR", (2 * N, levels, 2 * N, 2 * N), (1, 1, b, b), constant_zeros)
    V_LQ = mk("V_LQ", (2 * N, levels, 2 * N), (1, 1, b))
    T_LQ = mk("T_LQ", (2 * N, levels, 2 * N), (1, 1, b))
    L_LQ = mk("L_LQ", (2 * N, levels, 2 * N), (1, 1, b), constant_zeros_ext)
    S_LQ = mk("S_LQ", (2 * N, levels, 2 * N, 2 * N), (1, 1, b, b), constant_zeros_ext)
    t = time.time()
    p1 = lpcompile_for_execution(BDFAC, inputs=["I"], outputs=["R_QR", "L_LQ"])(
        A, V_QR, T_QR, S_QR, R_QR, V_LQ, T_LQ, S_LQ, L_LQ, A.num_blocks(0), truncate)
    c_time = time.time() - t
    program = lp.LambdaPackProgram(p1, config=npw_config.default())
    return program, {"outputs": [L_LQ, R_QR], "intermediates": [S_LQ, S_QR, T_QR, V_QR, V_LQ, T_LQ],
                     "compile_time": c_time}
