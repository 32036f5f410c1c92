"""LambdaPACK sources of the blocked algorithms (the DSL accepted by frontend.py).

These are the programs `alg_wrappers` compiles; statement order fixes the `expr_idx` of every
task and the index expressions fix which tiles each task reads and writes, so both are kept
identical to the reference's algorithms (reference numpywren/algs.py) -- the DAG fixtures under
tests/golden/dag.json, generated from the reference's own compiler, pin that equivalence task
by task.  Matrix roles:

  CHOLESKY  O = output factor, I = input, S[v, j, k] = version v of trailing tile (j, k)
  GEMM      Temp[i, j, k, l] = partial sums of C[i, j] at reduction-tree level l (fan-in 4)
  TSQR      Vs/Ts/Rs[level, j] = Householder factors of the binary reduction tree
  BDFAC     alternating column-TSQR (V_QR, T_QR, R_QR, S_QR) / row-TSLQ (V_LQ, T_LQ, L_LQ, S_LQ)
            sweeps; first index = sweep, second = tree level
  QR        TSQR per block column with flat + tree trailing updates
"""
from numpywren_amd.matrix import BigMatrix


def SimpleTestLinear(A: BigMatrix, B: BigMatrix, N: int):
    for i in range(N):
        for j in range(i + 1, N):
            A[j, i] = identity(A[i, j])
    for z in range(N):
        for k in range(N):
            B[z, k] = identity(A[z, k])


def SimpleTestLinear2(A: BigMatrix, B: BigMatrix, N: int):
    for i in range(N):
        for j in range(i + 1, N):
            A[j + 1, i + j] = identity(A[i, j])
    for z in range(N):
        for k in range(N):
            B[z, k] = identity(A[z, k])


def SimpleTestNonLinear(A: BigMatrix, B: BigMatrix, N: int):
    for i in range(N):
        N_tree = ceiling(log(N - i) / log(2))
        for level in range(0, ceiling(log(N - i) / log(2))):
            for k in range(0, N, 2 ** (level + 1)):
                A[N_tree - level - 1, i, k] = add(A[N_tree - level, i, k], A[N_tree - level, i, k + 2 ** level])
        B[i] = identity(A[1, i, 0])


def CHOLESKY(O: BigMatrix, I: BigMatrix, S: BigMatrix, N: int, truncate: int):
    # step 0 reads the input matrix, later steps read the SSA trailing matrix S
    O[0, 0] = chol(I[0, 0])
    for j in range(1, N - truncate):
        O[j, 0] = trsm(O[0, 0], I[j, 0])
        for k in range(1, j + 1):
            S[1, j, k] = syrk(I[j, k], O[j, 0], O[k, 0])
    for i in range(1, N - truncate):
        O[i, i] = chol(S[i, i, i])
        for j in range(i + 1, N - truncate):
            O[j, i] = trsm(O[i, i], S[i, j, i])
            for k in range(i + 1, j + 1):
                S[i + 1, j, k] = syrk(S[i, j, k], O[j, i], O[k, i])


def GEMM(A: BigMatrix, B: BigMatrix, M: int, N: int, K: int, Temp: BigMatrix, Out: BigMatrix):
    tree_depth = ceiling(log(K) / log(4))
    for i in range(0, M):
        for j in range(0, N):
            for k in range(0, K):
                Temp[i, j, k, 0] = gemm(A[i, k], B[k, j])
    for i in range(0, M):
        for j in range(0, N):
            for level in range(0, tree_depth):
                for k in range(0, K, 4 ** (level + 1)):
                    Temp[i, j, k, level + 1] = add_matrices(Temp[i, j, k, level], Temp[i, j, k + 4 ** level, level], Temp[i, j, k + 2 * 4 ** level, level], Temp[i, j, k + 3 * 4 ** level, level])
    for i in range(0, M):
        for j in range(0, N):
            Out[i, j] = identity(Temp[i, j, 0, tree_depth])


def TSQR(A: BigMatrix, Vs: BigMatrix, Ts: BigMatrix, Rs: BigMatrix, N: int):
    for j in range(0, N):
        Vs[0, j], Ts[0, j], Rs[0, j] = qr_factor(A[j, 0])
    for level in range(0, ceiling(log(N) / log(2))):
        for j in range(0, N, 2 ** (level + 1)):
            Vs[level + 1, j], Ts[level + 1, j], Rs[level + 1, j] = qr_factor(Rs[level, j], Rs[level, j + 2 ** level])


def BDFAC(I: BigMatrix, V_QR: BigMatrix, T_QR: BigMatrix, S_QR: BigMatrix, R_QR: BigMatrix, V_LQ: BigMatrix, T_LQ: BigMatrix, S_LQ: BigMatrix, L_LQ: BigMatrix, N: int, truncate: int):
    # ---- sweep 0, column part: TSQR of block column 0 with the reflectors applied to the rest ----
    QR_depth0 = ceiling(log(N) / log(2))
    for j in range(0, N):
        V_QR[0, 0, j], T_QR[0, 0, j], R_QR[0, 0, j] = qr_factor(I[j, 0])
        for k in range(1, N):
            S_QR[0, 0, j, k] = qr_leaf(V_QR[0, 0, j], T_QR[0, 0, j], I[j, k])
    for level in range(1, QR_depth0 + 1):
        for j in range(0, N, 2 ** level):
            V_QR[0, level, j], T_QR[0, level, j], R_QR[0, level, j] = qr_factor(R_QR[0, level - 1, j], R_QR[0, level - 1, j + 2 ** (level - 1)])
            for k in range(1, N):
                S_QR[0, level, j, k], S_QR[0, QR_depth0, j + 2 ** (level - 1), k] = qr_trailing_update(V_QR[0, level, j], T_QR[0, level, j], S_QR[0, level - 1, j, k], S_QR[0, level - 1, j + 2 ** (level - 1), k])
    # ---- sweep 0, row part: TSLQ of block row 0 (columns 1..N-1) ----
    LQ_depth0 = ceiling(log(N - 1) / log(2))
    for k in range(1, N):
        V_LQ[0, 0, k], T_LQ[0, 0, k], L_LQ[0, 0, k] = lq_factor(S_QR[0, QR_depth0, 0, k])
        for j in range(1, N):
            S_LQ[0, 0, j, k] = lq_leaf(V_LQ[0, 0, k], T_LQ[0, 0, k], S_QR[0, QR_depth0, j, k])
    for level in range(1, LQ_depth0 + 1):
        for k in range(1, N, 2 ** level):
            V_LQ[0, level, k], T_LQ[0, level, k], L_LQ[0, level, k] = lq_factor(L_LQ[0, level - 1, k], L_LQ[0, level - 1, k + 2 ** (level - 1)])
            for j in range(1, N):
                S_LQ[0, level, j, k], S_LQ[0, LQ_depth0, j, k + 2 ** (level - 1)] = lq_trailing_update(V_LQ[0, level, k], T_LQ[0, level, k], S_LQ[0, level - 1, j, k], S_LQ[0, level - 1, j, k + 2 ** (level - 1)])
    # ---- sweeps 1 .. N-2 ----
    for i in range(1, N - 1 - truncate):
        QR_depth = ceiling(log(N - i) / log(2))
        LQ_prev_depth = ceiling(log(N - i) / log(2))
        for j in range(i, N):
            V_QR[i, 0, j], T_QR[i, 0, j], R_QR[i, 0, j] = qr_factor(S_LQ[i - 1, LQ_prev_depth, j, i])
            for k in range(i + 1, N):
                S_QR[i, 0, j, k] = qr_leaf(V_QR[i, 0, j], T_QR[i, 0, j], S_LQ[i - 1, LQ_prev_depth, j, k])
        for level in range(1, QR_depth + 1):
            for j in range(i, N, 2 ** level):
                V_QR[i, level, j], T_QR[i, level, j], R_QR[i, level, j] = qr_factor(R_QR[i, level - 1, j], R_QR[i, level - 1, j + 2 ** (level - 1)])
                for k in range(i + 1, N):
                    S_QR[i, level, j, k], S_QR[i, QR_depth, j + 2 ** (level - 1), k] = qr_trailing_update(V_QR[i, level, j], T_QR[i, level, j], S_QR[i, level - 1, j, k], S_QR[i, level - 1, j + 2 ** (level - 1), k])
        LQ_depth = ceiling(log(N - i - 1) / log(2))
        for k in range(i + 1, N):
            V_LQ[i, 0, k], T_LQ[i, 0, k], L_LQ[i, 0, k] = lq_factor(S_QR[i, QR_depth, i, k])
            for j in range(i + 1, N):
                S_LQ[i, 0, j, k] = lq_leaf(V_LQ[i, 0, k], T_LQ[i, 0, k], S_QR[i, QR_depth, j, k])
        for level in range(1, LQ_depth + 1):
            for k in range(i + 1, N, 2 ** level):
                V_LQ[i, level, k], T_LQ[i, level, k], L_LQ[i, level, k] = lq_factor(L_LQ[i, level - 1, k], L_LQ[i, level - 1, k + 2 ** (level - 1)])
                for j in range(i + 1, N):
                    S_LQ[i, level, j, k], S_LQ[i, LQ_depth, j, k + 2 ** (level - 1)] = lq_trailing_update(V_LQ[i, level, k], T_LQ[i, level, k], S_LQ[i, level - 1, j, k], S_LQ[i, level - 1, j, k + 2 ** (level - 1)])
    # ---- last sweep: a lone QR of the bottom-right tile ----
    V_QR[N - 1, 0, N - 1], T_QR[N - 1, 0, N - 1], R_QR[N - 1, 0, N - 1] = qr_factor(S_LQ[N - 2, 0, N - 1, N - 1])


def QR(I: BigMatrix, Vs: BigMatrix, Ts: BigMatrix, Rs: BigMatrix, S: BigMatrix, N: int, truncate: int):
    depth0 = ceiling(log(N) / log(2))
    for j in range(0, N):
        Vs[j, 0, depth0], Ts[j, 0, depth0], Rs[j, 0, depth0] = qr_factor(I[j, 0])
    for level in range(0, depth0):
        for j in range(0, N, 2 ** (level + 1)):
            Vs[j, 0, depth0 - level - 1], Ts[j, 0, depth0 - level - 1], Rs[j, 0, depth0 - level - 1] = qr_factor_triangular(Rs[j, 0, depth0 - level], Rs[j + 2 ** level, 0, depth0 - level])
    for j in range(0, N):
        for k in range(1, N):
            S[j, k, 1, depth0] = qr_leaf(Vs[j, 0, depth0], Ts[j, 0, depth0], I[j, k])
    for k in range(1, N):
        for level in range(0, depth0):
            for j in range(0, N, 2 ** (level + 1)):
                S[j, k, 1, depth0 - 1 - level], S[j + 2 ** level, k, 1, 0] = qr_trailing_update(Vs[j, 0, depth0 - 1 - level], Ts[j, 0, depth0 - 1 - level], S[j, k, 1, depth0 - level], S[j + 2 ** level, k, 1, depth0 - level])
    for k in range(1, N):
        Rs[0, k, 0] = identity(S[0, k, 1, 0])
    for i in range(1, N):
        depth = ceiling(log(N - i) / log(2))
        for j in range(i, N):
            Vs[j, i, depth], Ts[j, i, depth], Rs[j, i, depth] = qr_factor(S[j, i, i, 0])
        for level in range(0, depth):
            for j in range(i, N, 2 ** (level + 1)):
                Vs[j, i, depth - level - 1], Ts[j, i, depth - level - 1], Rs[j, i, depth - level - 1] = qr_factor_triangular(Rs[j, i, depth - level], Rs[j + 2 ** level, i, depth - level])
        for j in range(i, N):
            for k in range(i + 1, N):
                S[j, k, i + 1, depth] = qr_leaf(Vs[j, i, depth], Ts[j, i, depth], S[j, k, i, 0])
        for k in range(i + 1, N):
            for level in range(0, depth):
                for j in range(i, N, 2 ** (level + 1)):
                    S[j, k, i + 1, depth - 1 - level], S[j + 2 ** level, k, i + 1, 0] = qr_trailing_update(Vs[j, i, depth - 1 - level], Ts[j, i, depth - 1 - level], S[j, k, i + 1, depth - level], S[j + 2 ** level, k, i + 1, depth - level])
        for k in range(i + 1, N):
            Rs[i, k, 0] = identity(S[i, k, i + 1, 0])
