"""Tile kernels with the reference's names, arity, kwargs and `.flops` models
(reference numpywren/kernels.py), executed by hand-written HIP kernels on the MI355X.

Calling convention (the "kernel seam"):
  * ndarray arguments  -> H2D, HIP kernels on the default stream, D2H: ndarray results.  This is
    the drop-in mode the reference's RemoteCall uses (`compute(*ndarrays, **kwargs)`,
    reference numpywren/lambdapack.py:360-381).
  * DeviceTile arguments -> DeviceTile results, asynchronous on the calling thread's current
    stream (`stream_scope`): the executor's fast path, no host round trips.
Inputs are never modified.  There is no CPU arithmetic in this module: without the HIP
extension / a GPU every kernel raises HipExtensionError.

Reference quirks reproduced on purpose (SURVEY.md section 8a notes): the allclose(x, 0)
short-circuits of syrk / trsm (device-side flags), trsm's odd zero-result shape, add_matrices'
float64 promotion, qr_leaf's formula as written (S0 - V^T S0).
"""
import contextlib
import threading

import numpy as np

from .device import DeviceTile, get_backend

_tls = threading.local()


@contextlib.contextmanager
def stream_scope(stream, info_sink=None, exact_zero=True, unwanted=None):
    """Run kernels of this thread on `stream`; Cholesky info flags are appended to `info_sink`.
    `unwanted`: per task of the call (one entry for a single task, one per task of a batched call) the set of output
    positions whose tiles will be dropped as soon as they are stored (the executor's `drop_unread_outputs`); a kernel
    may return None in such a position instead of computing the tile."""
    prev = getattr(_tls, "ctx", None)
    prev_unwanted = getattr(_tls, "unwanted", None)
    _tls.ctx = (stream, info_sink, exact_zero)
    _tls.unwanted = unwanted
    try:
        yield
    finally:
        _tls.ctx = prev
        _tls.unwanted = prev_unwanted


def _all_unwanted(position, ntasks):
    """True when every task of the current call has output `position` marked as dropped on store."""
    u = getattr(_tls, "unwanted", None)
    return bool(u) and len(u) == ntasks and all(position in x for x in u)


def _ctx():
    c = getattr(_tls, "ctx", None)
    return c if c is not None else (None, None, True)


def _is_host(a):
    return isinstance(a, np.ndarray)


def _kernel(impl):
    """Wrap a DeviceTile implementation `impl(be, stream, *tiles, **kw)` into the dual-mode callable."""

    def wrapper(*args, **kwargs):
        be = get_backend()
        stream, _, _ = _ctx()
        host_mode = any(_is_host(a) for a in args)
        if host_mode:
            targs = [be.to_device(np.asarray(a), stream) if _is_host(a) else a for a in args]
        else:
            targs = list(args)
        res = impl(be, stream, *targs, **kwargs)
        if not host_mode:
            return res

        def back(r):
            return be.to_host(r, stream) if isinstance(r, DeviceTile) else r

        if isinstance(res, tuple):
            return tuple(back(r) for r in res)
        return back(res)

    wrapper.__name__ = impl.__name__.lstrip("_")
    wrapper.__doc__ = impl.__doc__
    wrapper._npw_device_kernel = True
    return wrapper


def _all_f32(*tiles):
    return all(t.dtype == np.float32 for t in tiles)


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
@_kernel
def _gemm(be, stream, A, B, *args, **kwargs):
    """op(A) . op(B) with transpose_A / transpose_B kwargs (reference kernels.py:239-244)."""
    return be.gemm(A, B, bool(kwargs.get("transpose_A", False)), bool(kwargs.get("transpose_B", False)), stream)


gemm = _gemm


def _gemm_flops(A, B):
    m, n = A.shape
    k = B.shape[1]
    return 2 * m * n * k


gemm.flops = _gemm_flops


def _gemm_batch(be, stream, arg_lists, kwargs_list):
    """Ready gemm tasks of one statement (the M N K starters of the GEMM program, Temp[i, j, k, 0] = gemm(A[i, k], B[k, j]),
    reference algs.py:253-256) as batched launches of up to 16 products (HipBackend.gemm_batched): tasks with two device tiles
    and the same transpose kwargs together, anything else one by one.  Same outputs as `gemm` for each task."""
    out = [None] * len(arg_lists)
    groups = {}
    for pos, (args, kw) in enumerate(zip(arg_lists, kwargs_list)):
        plain = (len(args) == 2 and all(isinstance(a, DeviceTile) and a.ndim == 2 for a in args) and
                 set(kw) <= {"transpose_A", "transpose_B"})
        if plain:
            groups.setdefault((bool(kw.get("transpose_A", False)), bool(kw.get("transpose_B", False))), []).append(pos)
        else:
            out[pos] = _gemm(*args, **kw)
    for (ta, tb), members in groups.items():
        res = be.gemm_batched([tuple(arg_lists[p]) for p in members], ta, tb, stream)
        for p, r in zip(members, res):
            out[p] = r
    return out


gemm._npw_batch = _gemm_batch


@_kernel
def _syrk(be, stream, s, x, y, *args, **kwargs):
    """s - x . y^T; returns s itself when x or y is allclose to 0 (reference kernels.py:212-215)."""
    exact = _ctx()[2]
    out = be.syrk(s, x, y, stream, inplace=False, exact_zero=exact)
    if _all_f32(s, x, y):
        out = be.convert(out, np.float32, stream)
    return out


syrk = _syrk


def _syrk_flops(s, x, y):
    m = x.shape[0]
    n = x.shape[1]
    z = y.shape[1]
    return 2 * m * n * z + m * z


syrk.flops = _syrk_flops


def _syrk_batch(be, stream, arg_lists, kwargs_list):
    """Ready syrk tasks of one statement (the trailing updates of a block column of the Cholesky DAG,
    S[i+1, j, k] = syrk(S[i, j, k], O[j, i], O[k, i]), reference algs.py:248) as batched launches of up to 16 tiles
    (HipBackend.syrk_batched); the diagonal ones (x is y) and anything that is not three fp64 device tiles one by one.
    Same outputs as `syrk` for each task."""
    exact = _ctx()[2]
    out = [None] * len(arg_lists)
    plain = []
    for pos, (args, kw) in enumerate(zip(arg_lists, kwargs_list)):
        if len(args) == 3 and not kw and all(isinstance(a, DeviceTile) and a.ndim == 2 and a.dtype == np.float64 for a in args):
            plain.append(pos)
        else:
            out[pos] = _syrk(*args, **kw)
    if plain:
        res = be.syrk_batched([tuple(arg_lists[p]) for p in plain], stream, exact_zero=exact)
        for p, r in zip(plain, res):
            out[p] = r
    return out


syrk._npw_batch = _syrk_batch


def _solve_right_upper(be, stream, L, Y, exact):
    """X with X L^T = Y, L lower triangular: the one triangular solve the library has (npw_dtrsm_rltn_inv)."""
    return be.trsm(L, Y, stream, exact_zero=exact)


def _solve_right_lower(be, stream, M, Y, exact):
    """X with X M = Y, M lower triangular.  With J the reversal permutation, (X J)(J M J) = Y J and J M J is upper
    triangular: the library's solve with L' = (J M J)^T = J M^T J and the columns of Y reversed; X is the result with
    its columns reversed again (npw_dflip)."""
    Lp = be.flip(be.transpose(M, stream), True, True, stream)
    return be.flip(be.trsm(Lp, be.flip(Y, False, True, stream), stream, exact_zero=exact), False, True, stream)


@_kernel
def _trsm(be, stream, x, y, lower=False, right=True, *args, **kwargs):
    """scipy.linalg.blas.dtrsm(1.0, x.T, y, lower=lower, side=int(right)) (reference kernels.py:254-257); zeros((x.shape[1],
    y.shape[0])) when y is allclose to 0.  With a = x.T, DTRSM reads only a's upper (lower=False) or lower triangle:
      lower=False, right=True  (the default, all the LambdaPACK programs use):  X triu(a) = y  <=>  X tril(x)^T = y
      lower=True,  right=True :  X tril(a) = y          -- a lower triangular matrix on the right (see _solve_right_lower)
      lower=True,  right=False:  tril(a) X = y  <=>  X^T tril(a)^T = y^T       -- the library's solve on transposes
      lower=False, right=False:  triu(a) X = y  <=>  X^T tril(x) = y^T         -- lower triangular on the right, transposed
    """
    exact = _ctx()[2]
    lower, right = bool(lower), bool(right)
    if not lower and right:
        out = _solve_right_upper(be, stream, x, y, exact)
    elif lower and right:
        out = _solve_right_lower(be, stream, be.tri(be.transpose(x, stream), "L", False, stream), y, exact)
    elif lower and not right:
        out = be.transpose(_solve_right_upper(be, stream, be.tri(be.transpose(x, stream), "L", False, stream),
                                              be.transpose(y, stream), exact), stream)
    else:
        out = be.transpose(_solve_right_lower(be, stream, be.tri(x, "L", False, stream), be.transpose(y, stream), exact), stream)
    zshape = (x.shape[1], y.shape[0])
    if exact and tuple(out.shape) != zshape:
        # the reference's zero short-circuit changes the result SHAPE, which needs the flag on the host
        if be.read_flag(be.zero_flag(y, stream), stream):
            return be.zeros(zshape, np.float64, stream)
    return out


trsm = _trsm


def _trsm_batch(be, stream, arg_lists, kwargs_list):
    """Ready trsm tasks of one statement as batched solves: tasks that share their factor tile (one block column of
    the Cholesky DAG: O[j, i] = trsm(O[i, i], S[i, j, i]) for every j > i, reference algs.py:242, 246) go to the device as
    ONE solve with stacked right-hand sides (HipBackend.trsm_batched); anything else one by one.  Same outputs as
    `trsm` for each task."""
    exact = _ctx()[2]
    out = [None] * len(arg_lists)
    groups = {}
    for pos, (args, kw) in enumerate(zip(arg_lists, kwargs_list)):
        x, y = args[0], args[1]
        plain = (len(args) == 2 and not kw and isinstance(x, DeviceTile) and isinstance(y, DeviceTile) and x.ndim == 2 and
                 y.ndim == 2 and x.shape[0] == x.shape[1] == y.shape[1] and y.shape[0] == x.shape[1])
        if plain:
            groups.setdefault((id(x.buf), x.offset, x.shape, y.shape), []).append(pos)
        else:
            out[pos] = _trsm(*args, **kw)
    for key, members in groups.items():
        L = arg_lists[members[0]][0]
        for i in range(0, len(members), 16):
            part = members[i:i + 16]
            if len(part) < _TRSM_BATCH_MIN:
                # a handful of right-hand sides: one solve each fills the chip as well (4096^2 tiles: 1.10 ms each
                # against 1.14 in a batch of 3; 1.09 in a batch of 15)
                for p in part:
                    out[p] = be.trsm(L, arg_lists[p][1], stream, exact_zero=exact)
                continue
            res = be.trsm_batched(L, [arg_lists[p][1] for p in part], stream, exact_zero=exact)
            for p, r in zip(part, res):
                out[p] = r
    return out


_TRSM_BATCH_MIN = 8


trsm._npw_batch = _trsm_batch
trsm._npw_batch_gather = True   # worth reordering for: the executor first runs the ready tasks that enable further siblings


def _trsm_flops(x, y):
    # defined by the reference but never attached (kernels.py:259-263): trsm counts 0 flops there
    if len(y.shape) == 0:
        return x.shape[0] * x.shape[1]
    return x.shape[0] * x.shape[1] * y.shape[1]


@_kernel
def _chol(be, stream, x, *args, **kwargs):
    """Lower Cholesky factor, upper part zero (reference kernels.py:225-226 np.linalg.cholesky);
    raises numpy.linalg.LinAlgError for a non positive definite tile (deferred to the executor's
    completion check on the asynchronous path)."""
    L, info = be.chol(x, stream)
    sink = _ctx()[1]
    if sink is not None:
        sink.append(info)
    else:
        code = be.read_flag(info, stream)
        if code != 0:
            raise np.linalg.LinAlgError("Matrix is not positive definite")
    if x.dtype == np.float32:
        L = be.convert(L, np.float32, stream)
    return L


chol = _chol


def _chol_flops(x):
    return (x.shape[0] ** 3) / 3


chol.flops = _chol_flops


@_kernel
def _add_matrices(be, stream, *args, **kwargs):
    """n-ary sum, always float64 (reference kernels.py:16-20: np.zeros(args[0].shape) += a)."""
    return be.add_n(list(args), stream)


add_matrices = _add_matrices


def identity(x, *args, **kwargs):
    """Returns its input (reference kernels.py:236-237)."""
    return x


identity._npw_device_kernel = True


@_kernel
def _mul(be, stream, x, y, *args, **kwargs):
    """x * y (reference kernels.py:233-234): a scalar and a tile, or two tiles of one shape elementwise (npw_dmul)."""
    if isinstance(x, DeviceTile) and isinstance(y, DeviceTile):
        return be.mul(x, y, stream)
    if isinstance(y, DeviceTile):
        x, y = y, x
    return be.axpby(float(y), x, 0.0, x, stream)


mul = _mul


# ------------------------------------------------------------------------------------------------
# Householder QR family
# ------------------------------------------------------------------------------------------------
@_kernel
def _qr_factor(be, stream, *blocks, **kwargs):
    """QR of vstack(blocks): (V unit-lower-trapezoid m x n, T n x n upper with Q = I - V T V^T,
    R n x n upper) -- reference kernels.py:127-130 -> fast_qr 86-105 (LAPACK dgeqrt3).  A stack with more columns
    than rows (fast_qr hands it to slow_qr, 94-95 -> 67-84) gives V m x m, T m x m and R m x n."""
    want_t = not _all_unwanted(1, 1)   # T (output 1) is only left out when the executor will drop it unread
    if _stacked_triangles(blocks):
        return be.tpqrt_batched([tuple(blocks)], stream, want_t=want_t)[0]
    ins = be.vstack(list(blocks), stream)
    return be.geqrt(ins, stream, want_t=want_t)


def _stacked_triangles(blocks):
    """Two square tiles of one size, both known to be upper triangular with exact zeros below the diagonal (R factors
    this backend produced): the node of a TSQR tree.  Then the structured factorisation applies."""
    return (len(blocks) == 2 and all(getattr(b, "upper", False) for b in blocks) and blocks[0].shape == blocks[1].shape
            and len(blocks[0].shape) == 2 and blocks[0].shape[0] == blocks[0].shape[1])


qr_factor = _qr_factor
fast_qr = _qr_factor


def _qr_factor_batch(be, stream, arg_lists, kwargs_list):
    """Several independent qr_factor tasks as one batched launch sequence (npw_dgeqrt_batched): the TSQR leaves and
    the nodes of one tree level (reference algs.py:30-36) are independent and latency-bound one by one.
    arg_lists[i] are the block tiles of task i; returns [(V, T, R), ...] in the same order."""
    out = [None] * len(arg_lists)
    want_t = not _all_unwanted(1, len(arg_lists))
    want_v = not _all_unwanted(0, len(arg_lists))
    tri = [i for i, blocks in enumerate(arg_lists) if _stacked_triangles(blocks)]
    if tri and len({arg_lists[i][0].shape for i in tri}) == 1:
        for i, res in zip(tri, be.tpqrt_batched([tuple(arg_lists[i]) for i in tri], stream, want_t=want_t, want_v=want_v)):
            out[i] = res
    rest = [i for i in range(len(arg_lists)) if out[i] is None]
    if rest:
        ins = [be.vstack(list(arg_lists[i]), stream) for i in rest]
        for i, res in zip(rest, be.geqrt_batched(ins, stream, want_t=want_t, want_v=want_v)):
            out[i] = res
    return out


qr_factor._npw_batch = _qr_factor_batch
# the panel kernel's workgroups exchange partial sums: every workgroup of a launch must be resident (job_runner._fence_in)
qr_factor._npw_needs_whole_cus = True


def _qr_flops(*blocks):
    m = sum(b.shape[0] for b in blocks)
    n = blocks[0].shape[1]
    return 2 * m * n * n - (2 * n ** 3) / 3


qr_factor.flops = _qr_flops


@_kernel
def _lq_factor(be, stream, *blocks, **kwargs):
    """LQ of hstack(blocks) by transposition: fast_qr(ins.T) -> (v.T, t.T, r.T)
    (reference kernels.py:145-150)."""
    if len(blocks) == 2:
        assert blocks[0].shape[0] == blocks[1].shape[0]
    ins_t = be.vstack([be.transpose(b, stream) for b in blocks], stream)  # == hstack(blocks).T
    V, T, R = be.geqrt(ins_t, stream)
    return be.transpose(V, stream), be.transpose(T, stream), be.transpose(R, stream)


lq_factor = _lq_factor
lq_factor.flops = _qr_flops


def _lq_factor_batch(be, stream, arg_lists, kwargs_list):
    """Independent lq_factor tasks (the leaves / one tree level of a row sweep of BDFAC) as one batched QR of the
    transposed blocks; same outputs as _lq_factor for each."""
    ins = [be.vstack([be.transpose(b, stream) for b in blocks], stream) for blocks in arg_lists]
    return [(be.transpose(V, stream), be.transpose(T, stream), be.transpose(R, stream))
            for V, T, R in be.geqrt_batched(ins, stream)]


lq_factor._npw_batch = _lq_factor_batch
lq_factor._npw_needs_whole_cus = True


@_kernel
def _qr_leaf(be, stream, V, T, S0, *args, **kwargs):
    """S0 - V^T S0, exactly as the reference writes it (kernels.py:160-164; the WY form is commented
    out there)."""
    return be.gemm(V, S0, True, False, stream, alpha=-1.0, beta=1.0, C=S0)


qr_leaf = _qr_leaf


@_kernel
def _qr_leaf_wy(be, stream, V, T, S0, *args, **kwargs):
    """S0 - V T^T (V^T S0): the compact-WY application of Q^T that the reference has commented out
    in qr_leaf (kernels.py:162).  Not used by default (parity = the reference as written); pass
    `kernels={"qr_leaf": kernels.qr_leaf_wy}` to lpcompile_for_execution to run BDFAC / QR with a
    mathematically valid leaf update."""
    w = be.gemm(V, S0, True, False, stream)
    w = be.gemm(T, w, True, False, stream)
    return be.gemm(V, w, False, False, stream, alpha=-1.0, beta=1.0, C=S0)


qr_leaf_wy = _qr_leaf_wy


@_kernel
def _lq_leaf(be, stream, V, T, S0, *args, **kwargs):
    """S0 - S0 V^T T^T V (reference kernels.py:154-157)."""
    a1 = be.gemm(S0, V, False, True, stream)
    a2 = be.gemm(a1, T, False, True, stream)
    return be.gemm(a2, V, False, False, stream, alpha=-1.0, beta=1.0, C=S0)


lq_leaf = _lq_leaf


def _qr_leaf_flops(V, T, S0):
    c0 = V.shape[0] * S0.shape[0] * S0.shape[1]
    c1 = T.shape[0] * V.shape[0] * S0.shape[1]
    c2 = V.shape[0] * T.shape[0] * T.shape[1]
    return c0 + c1 + c2 + S0.shape[0] * S0.shape[1]


qr_leaf.flops = _qr_leaf_flops
qr_leaf_wy.flops = _qr_leaf_flops
lq_leaf.flops = _qr_leaf_flops


@_kernel
def _qr_trailing_update(be, stream, V, T, S0, S1=None, *args, **kwargs):
    """V = V[-S0.rows:];  W = T^T (S0 + V^T S1);  returns (S0 - W, S1 - V W)
    (reference kernels.py:181-188)."""
    if S1 is None:
        return be.gemm(V, S0, True, False, stream, alpha=-1.0, beta=1.0, C=S0), be.zeros(S0.shape, np.float64, stream)
    rows = S0.shape[0]
    Vb = be.rows(be.as_f64(V, stream), V.shape[0] - rows, V.shape[0], stream) if V.shape[0] != rows else V
    X = be.gemm(Vb, S1, True, False, stream, alpha=1.0, beta=1.0, C=S0)
    W = be.gemm(T, X, True, False, stream)
    S01 = be.axpby(1.0, S0, -1.0, W, stream)
    S11 = be.gemm(Vb, W, False, False, stream, alpha=-1.0, beta=1.0, C=S1)
    return S01, S11


qr_trailing_update = _qr_trailing_update


def _qr_trailing_flops(V, T, S0, S1):
    M, N = V.shape
    c0 = M * S1.shape[0] * S1.shape[1]
    c1 = T.shape[0] * T.shape[1] * S0.shape[1]
    return 2 * c1 + c0 + T.shape[0] * T.shape[1]


qr_trailing_update.flops = _qr_trailing_flops


@_kernel
def _lq_trailing_update(be, stream, V, T, S0, S1=None, *args, **kwargs):
    """V = V[:, -S0.rows:];  W = (S0 + S1 V^T) T^T;  returns (S0 - W, S1 - W V)
    (reference kernels.py:199-208)."""
    if S1 is None:
        a1 = be.gemm(S0, V, False, True, stream)
        a2 = be.gemm(a1, T, False, True, stream)
        return be.gemm(a2, V, False, False, stream, alpha=-1.0, beta=1.0, C=S0), be.zeros(S0.shape, np.float64, stream)
    cols = S0.shape[0]
    if V.shape[1] != cols:
        # last `cols` columns of V == (last `cols` rows of V^T)^T
        Vt = be.transpose(be.as_f64(V, stream), stream)
        Vr = be.transpose(be.rows(Vt, Vt.shape[0] - cols, Vt.shape[0], stream), stream)
    else:
        Vr = V
    X = be.gemm(S1, Vr, False, True, stream, alpha=1.0, beta=1.0, C=S0)
    W = be.gemm(X, T, False, True, stream)
    S01 = be.axpby(1.0, S0, -1.0, W, stream)
    S11 = be.gemm(W, Vr, False, False, stream, alpha=-1.0, beta=1.0, C=S1)
    assert S0.shape == S01.shape
    assert S1.shape == S11.shape
    return S01, S11


lq_trailing_update = _lq_trailing_update
lq_trailing_update.flops = _qr_trailing_flops


# panel kernels sit on the critical path of the factorizations and are latency-bound: the executor
# issues them on its high-priority stream so they overtake queued trailing updates
for _k in (qr_factor, lq_factor):
    _k._npw_handoff = True   # runs the panel kernel: the executor checks its expired-wait counter when the run settles
for _k in (chol, trsm, qr_factor, lq_factor):
    _k._npw_latency_bound = True
chol._npw_needs_whole_cus = True   # see job_runner.LambdaPackExecutor.run_task


# The chain partition of the executor (job_runner.LambdaPackExecutor.run_chain): `chol` on a stream masked to a quarter
# of the CUs, throughput kernels of the same tile size beside it on the rest.  Weights in units of that window,
# measured at the 4096^2 tile (tools/overlap_probe.py): chol on 64 CUs 2.86 ms; on the other 192 a general trailing
# update 2.92 ms, one whose x and y are the same tile (lower triangle + mirror, half the products) 2.0 ms.
chol._npw_chain_resident_cus = lambda be, rows: be.chol_resident_cus(rows)
syrk._npw_chain_weight = lambda task: 0.69 if len(task.reads) >= 3 and task.reads[1] == task.reads[2] else 1.0
gemm._npw_chain_weight = lambda task: 1.0


# ------------------------------------------------------------------------------------------------
# surface kept for import compatibility; not on the gemm / cholesky / tsqr / bdfac paths
# ------------------------------------------------------------------------------------------------
slow_qr = _qr_factor   # reference kernels.py:67-84 (DGEQRF + DLARFT): same (V, T, R); npw_dgeqrt takes any m, n


def _tpqrt_shapes(x0, x1):
    """The shapes the reference's DTPQRT call (kernels.py:107-124) accepts: it passes m = l = x0.shape[0], n = x0.shape[1],
    a = x0, b = x1.  LAPACK wants A n x n (LDA >= n), B with at least m rows and l <= min(m, n), so x0 has to be square
    (m > n: l > min(m, n), argument 3; m < n: LDA < n, argument 6) and x1 needs n columns and at least n rows, of which the
    routine only touches the first n.  Anything else is LAPACK's "illegal value" error there, a ValueError here."""
    if x0.ndim != 2 or x1.ndim != 2:
        raise ValueError(f"qr_factor_triangular: 2-D tiles expected, got {tuple(x0.shape)} over {tuple(x1.shape)}")
    m, n = x0.shape
    if m > n:
        raise ValueError(f"qr_factor_triangular: x0 is {m} x {n}; DTPQRT(m={m}, n={n}, l={m}) has l > min(m, n) (illegal argument 3)")
    if m < n:
        raise ValueError(f"qr_factor_triangular: x0 is {m} x {n}; DTPQRT's A must be n x n (illegal argument 6, lda < n)")
    if x1.shape[1] != n or x1.shape[0] < n:
        raise ValueError(f"qr_factor_triangular: x1 is {tuple(x1.shape)}; DTPQRT(m={n}, n={n}) needs at least {n} rows of {n} columns")
    return n


@_kernel
def _qr_factor_triangular(be, stream, x0, x1, **kwargs):
    """QR of two stacked upper-triangular tiles (reference kernels.py:107-124: LAPACK DTPQRT with l = m,
    nb = min(n, 32) on x0 over x1).  Returns, exactly as the reference does:
      r  the n x n upper-triangular factor of [triu(x0); triu(x1)];
      t  an n x n tile whose first nb rows hold DTPQRT's blocked T (the nb x nb diagonal blocks of the compact-WY
         factor side by side), zero elsewhere -- the caller (qr_trailing_update) then uses it as if it were the full T;
      v  `tril` of what DTPQRT leaves in x1 with a unit diagonal.  DTPQRT stores its reflectors in the *upper* triangle
         of x1 and does not touch the part below, so this is tril(x1, -1) + I: the identity for the triangular inputs of
         the QR tree.  (The reference's alg_wrappers.qr therefore only gets the first block row of R right; we reproduce
         it as written -- see tests/golden/make_golden_qr.py and DESIGN.md section 7.)
    On the GPU: npw_dtpqrt_batched (the structured factorisation of two stacked triangles) gives R and the full n x n T,
    whose nb x nb diagonal blocks are DTPQRT's blocked T (npw_dblockdiag_rows lays them side by side)."""
    n = _tpqrt_shapes(x0, x1)
    x1top = x1 if x1.shape[0] == n else be.rows(x1, 0, n, stream)
    _, T, R = be.tpqrt_batched([(be.tri(x0, "U", False, stream), be.tri(x1top, "U", False, stream))], stream)[0]
    v = be.tri(x1, "L", True, stream)
    t = be.blockdiag_rows(T, min(n, 32), stream)
    return v, t, R


qr_factor_triangular = _qr_factor_triangular
fast_qr_triangular = _qr_factor_triangular
qr_factor_triangular._npw_latency_bound = True


def _qr_factor_triangular_batch(be, stream, arg_lists, kwargs_list):
    """Independent qr_factor_triangular tasks (the nodes of one level of the QR tree, reference algs.py:182-234) as one
    batched factorisation; same outputs as _qr_factor_triangular for each."""
    for x0, x1 in arg_lists:
        _tpqrt_shapes(x0, x1)
    if len({(tuple(x0.shape), tuple(x1.shape)) for x0, x1 in arg_lists}) != 1 or arg_lists[0][1].shape[0] != arg_lists[0][0].shape[0]:
        return [_qr_factor_triangular(x0, x1) for x0, x1 in arg_lists]
    stacked = [(be.tri(x0, "U", False, stream), be.tri(x1, "U", False, stream)) for x0, x1 in arg_lists]
    out = []
    for (x0, x1), (_, T, R) in zip(arg_lists, be.tpqrt_batched(stacked, stream)):
        out.append((be.tri(x1, "L", True, stream), be.blockdiag_rows(T, min(x0.shape[-1], 32), stream), R))
    return out


qr_factor_triangular._npw_batch = _qr_factor_triangular_batch
qr_factor_triangular._npw_handoff = True
qr_factor_triangular._npw_needs_whole_cus = True


def banded_to_bidiagonal(x):
    """(diag_out, offdiag_out) of the bidiagonal form of the band matrix packed from the blocks `x` (reference
    kernels.py:43-65).  As written there, block i is placed at rows and columns [i s, (i + 1) s) of a band matrix with
    kl = ku = s - 1 (s = x[0].shape[0]; every block has to be s x s, otherwise the reference's packing loop raises) --
    a block-diagonal matrix -- and LAPACK DGBBRD(vect='N') returns d (length s * len(x)) and e (one shorter).  Every block
    is reduced on its own on the GPU (npw_dgebd2); the entries of e between two blocks are exactly zero.  The
    bidiagonal form is unique up to the signs of its entries: magnitudes equal DGBBRD's, signs follow DLARFG.
    ndarray blocks -> ndarrays; DeviceTile blocks -> device vectors."""
    be = get_backend()
    stream = _ctx()[0]
    blocks = list(x)
    if not blocks:
        raise IndexError("list index out of range")          # x[0] in the reference
    s = blocks[0].shape[0]
    host = any(_is_host(b) for b in blocks)
    ds, es = [], []
    for b in blocks:
        if tuple(b.shape) != (s, s):
            # the reference's packing `packed_x[a:a + s, col] = block[:, j]` cannot broadcast anything else
            raise ValueError(f"could not broadcast input array from shape {tuple(b.shape)} into shape ({s},)")
        t = be.to_device(np.asarray(b, dtype=np.float64), stream) if _is_host(b) else b
        d, e = be.gebd2(t, stream)
        ds.append(d)
        es.append(e)
    if host:
        n = s * len(blocks)
        d_out, e_out = np.zeros(n), np.zeros(max(n - 1, 0))
        for i, (d, e) in enumerate(zip(ds, es)):
            d_out[i * s:(i + 1) * s] = be.to_host(d, stream)
            if s > 1:
                e_out[i * s:(i + 1) * s - 1] = be.to_host(e, stream)
        return d_out, e_out
    zero = be.zeros((1,), np.float64, stream)
    parts_e = []
    for i, e in enumerate(es):
        if s > 1:
            parts_e.append(e.reshaped((s - 1, 1)))
        if i + 1 < len(es):
            parts_e.append(zero.reshaped((1, 1)))
    d_all = be.vstack([d.reshaped((s, 1)) for d in ds], stream)
    e_all = be.vstack(parts_e, stream) if parts_e else be.zeros((0, 1), np.float64, stream)
    return d_all.reshaped((d_all.shape[0],)), e_all.reshaped((e_all.shape[0],))


@_kernel
def _trsm_sub(be, stream, L, S, x, *args, **kwargs):
    """scipy.linalg.solve_triangular(L, x - S) (reference kernels.py:178-179): solve_triangular's default is lower=False, so
    this is X with triu(L) X = x - S.  Transposed, X^T triu(L)^T = (x - S)^T: a lower triangular matrix on the right
    (_solve_right_lower).  Unused by the LambdaPACK programs; kept because it is part of the reference's kernel surface."""
    rhs = be.axpby(1.0, x, -1.0, S, stream)
    if rhs.ndim != 2 or L.ndim != 2 or L.shape[0] != L.shape[1] or rhs.shape[0] != L.shape[0]:
        raise ValueError(f"trsm_sub: incompatible shapes L{tuple(L.shape)} x{tuple(rhs.shape)}")
    M = be.transpose(be.tri(L, "U", False, stream), stream)
    return be.transpose(_solve_right_lower(be, stream, M, be.transpose(rhs, stream), False), stream)


trsm_sub = _trsm_sub
