#!/usr/bin/env python
"""Generate the committed golden fixtures by RUNNING THE REFERENCE (authoring container only).

    python tests/golden/make_golden.py        # writes tests/golden/*.npz / *.json

What is recorded:
  kernels_kat.npz  -- known-answer vectors of every hot-path function of the reference's
                      numpywren/kernels.py (inputs + outputs, tiles of 8..32)
  dag.json         -- for CHOLESKY / TSQR / GEMM / BDFAC / QR / SimpleTest* at small sizes: starters,
                      num_terminators, every task's (reads, kernel, kwargs, writes), children, parents
                      as computed by the reference's frontend.py + compiler.py
  indexing.json    -- BigMatrix block-indexing vectors (blocks, block idxs, key strings, views)
  algos.npz        -- whole-algorithm inputs/outputs: the reference's compiled programs executed
                      sequentially with the reference's own RemoteRead/RemoteCall/RemoteWrite objects and
                      BigMatrix get/put logic over an in-memory object store (the S3 calls are the only
                      thing replaced)

The reference is imported from /root/reference via _ref_import.py (stub modules for the AWS
dependencies, scipy-LAPACK shims for the f2py modules it would download from S3).
"""
import asyncio
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

npw = _ref_import.import_reference()
from numpywren import algs, compiler, kernels, matrix  # noqa: E402
from numpywren import lambdapack as lp  # noqa: E402
from numpywren.matrix import BigMatrix  # noqa: E402
from numpywren.matrix_utils import constant_zeros, constant_zeros_ext  # noqa: E402

# ---------------------------------------------------------------------------------------------
# in-memory object store under the reference's S3 calls
# ---------------------------------------------------------------------------------------------
STORE = {}


async def _key_exists_async(bucket, key, loop=None):
    return (bucket, key) in STORE


async def _s3_key_to_byte_io(self, key, loop=None):
    return io.BytesIO(STORE[(self.bucket, key)])


async def _save_matrix_to_s3(self, X, out_key, loop, client=None):
    bio = io.BytesIO()
    np.save(bio, X)
    STORE[(self.bucket, out_key)] = bio.getvalue()
    return None


matrix.key_exists_async = _key_exists_async
BigMatrix.__s3_key_to_byte_io__ = _s3_key_to_byte_io
BigMatrix.__save_matrix_to_s3__ = _save_matrix_to_s3


def shard(bigm, X):
    """Equivalent of matrix_init.shard_matrix (reference matrix_init.py:73-96) minus the /tmp memmap."""
    for bidx, blk in zip(bigm.block_idxs, bigm.blocks):
        sl = tuple(slice(s, e) for s, e in blk)
        bigm.put_block(np.ascontiguousarray(X[sl]), *bidx)
    return bigm


def gather(bigm):
    """Equivalent of BigMatrix.numpy() (reference matrix_utils.py:156-167) without the process pool."""
    out = np.zeros(bigm.shape, dtype=bigm.dtype)
    for bidx in bigm._block_idxs():
        real = bigm.__block_idx_to_real_idx__(bidx) if not isinstance(bigm, matrix.BigMatrixView) else None
        blk = bigm.get_block(*bidx)
        if real is None:
            # views: place by view block index * shard size
            real = tuple((i * s, i * s + d) for i, s, d in zip(bidx, bigm.shard_sizes, np.atleast_1d(blk.shape)))
        sl = tuple(slice(s, e) for s, e in real)
        out[sl] = blk.reshape(out[sl].shape)
    return out


def run_program(program, max_tasks=100000):
    """Sequential driver: BFS over the reference's implicit DAG using its own find_children /
    find_parents and instruction objects (reference job_runner.py:78-159 minus Redis/SQS)."""
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    done = set()
    edge_count = {}
    ready = [(e, dict(v)) for e, v in program.starters]
    order = []

    def key(node):
        return (node[0], tuple(sorted(node[1].items())))

    n = 0
    while ready:
        node = ready.pop(0)
        if key(node) in done:
            continue
        ib = program.eval_expr(node[0], node[1])
        for instr in ib.instrs:
            if isinstance(instr, lp.RemoteWrite):
                loop.run_until_complete(instr(False))
            else:
                loop.run_until_complete(instr())
        done.add(key(node))
        order.append(node)
        n += 1
        assert n < max_tasks
        for child in program.find_children(node[0], node[1]):
            ck = key(child)
            edge_count.setdefault(ck, set()).add(key(node))
            parents = program.find_parents(child[0], child[1])
            if len(edge_count[ck]) == len(parents) and ck not in done:
                ready.append((child[0], {str(k): int(v) for k, v in child[1].items()}))
    loop.close()
    return order


# ---------------------------------------------------------------------------------------------
# (i) kernel known-answer vectors
# ---------------------------------------------------------------------------------------------
def make_kernel_kats():
    rng = np.random.default_rng(20260928)
    out = {}

    def rec(name, ins, outs, **meta):
        for i, a in enumerate(ins):
            out[f"{name}/in{i}"] = np.asarray(a)
        if not isinstance(outs, tuple):
            outs = (outs,)
        for i, a in enumerate(outs):
            out[f"{name}/out{i}"] = np.asarray(a)
        out[f"{name}/nout"] = np.asarray(len(outs))

    for b in (8, 16):
        A = rng.standard_normal((b, b))
        B = rng.standard_normal((b, b))
        C = rng.standard_normal((b, b))
        rec(f"gemm_nn_{b}", [A, B], kernels.gemm(A, B))
        rec(f"gemm_tn_{b}", [A, B], kernels.gemm(A, B, transpose_A=True))
        rec(f"gemm_nt_{b}", [A, B], kernels.gemm(A, B, transpose_B=True))
        rec(f"gemm_tt_{b}", [A, B], kernels.gemm(A, B, transpose_A=True, transpose_B=True))
        A32, B32 = A.astype(np.float32), B.astype(np.float32)
        rec(f"gemm_f32_{b}", [A32, B32], kernels.gemm(A32, B32))
        rec(f"syrk_{b}", [C, A, B], kernels.syrk(C, A, B))
        rec(f"syrk_same_{b}", [C, A], kernels.syrk(C, A, A))
        tiny = 1e-9 * rng.standard_normal((b, b))
        rec(f"syrk_xzero_{b}", [C, tiny, B], kernels.syrk(C, tiny, B))
        rec(f"syrk_yzero_{b}", [C, A, np.zeros((b, b))], kernels.syrk(C, A, np.zeros((b, b))))
        spd = A @ A.T + b * np.eye(b)
        L = kernels.chol(spd)
        rec(f"chol_{b}", [spd], L)
        rec(f"trsm_{b}", [L, B], kernels.trsm(L, B))
        rec(f"trsm_yzero_{b}", [L, tiny], kernels.trsm(L, tiny))
        rec(f"add4_{b}", [A, B, C, A.T.copy()], kernels.add_matrices(A, B, C, A.T.copy()))
        rec(f"add_f32_{b}", [A32, B32], kernels.add_matrices(A32, B32))
        rec(f"identity_{b}", [A], kernels.identity(A))
        # QR family
        rec(f"qr_factor_{b}", [A], kernels.qr_factor(A))
        rec(f"qr_factor_stack_{b}", [A, B], kernels.qr_factor(A, B))
        R0 = np.triu(A)
        R1 = np.triu(B)
        rec(f"qr_factor_rr_{b}", [R0, R1], kernels.qr_factor(R0, R1))
        rec(f"lq_factor_{b}", [A], kernels.lq_factor(A))
        rec(f"lq_factor_pair_{b}", [A, B], kernels.lq_factor(A, B))
        V, T, R = kernels.qr_factor(A)
        rec(f"qr_leaf_{b}", [V, T, C], kernels.qr_leaf(V, T, C))
        Vl, Tl, Ll = kernels.lq_factor(A)
        rec(f"lq_leaf_{b}", [Vl, Tl, C], kernels.lq_leaf(Vl, Tl, C))
        V2, T2, R2 = kernels.qr_factor(R0, R1)
        rec(f"qr_trailing_{b}", [V2, T2, A, C], kernels.qr_trailing_update(V2, T2, A, C))
        Vl2, Tl2, Ll2 = kernels.lq_factor(np.tril(A), np.tril(B))
        rec(f"lq_trailing_{b}", [Vl2, Tl2, A, C], kernels.lq_trailing_update(Vl2, Tl2, A, C))
    # ragged / non-square
    A = rng.standard_normal((5, 7))
    B = rng.standard_normal((7, 3))
    rec("gemm_ragged", [A, B], kernels.gemm(A, B))
    L = np.linalg.cholesky(np.eye(7) * 7 + 0.1 * np.ones((7, 7)))
    Y = rng.standard_normal((5, 7))
    rec("trsm_ragged", [L, Y], kernels.trsm(L, Y))
    rec("trsm_ragged_yzero", [L, np.zeros((5, 7))], kernels.trsm(L, np.zeros((5, 7))))
    S = rng.standard_normal((5, 3))
    X = rng.standard_normal((5, 7))
    Yy = rng.standard_normal((3, 7))
    rec("syrk_ragged", [S, X, Yy], kernels.syrk(S, X, Yy))
    A = rng.standard_normal((24, 8))
    rec("qr_factor_tall", [A], kernels.qr_factor(A))
    # flop models
    a8 = np.zeros((8, 8))
    a16x8 = np.zeros((16, 8))
    flops = {
        "gemm": kernels.gemm.flops(a8, a8),
        "syrk": kernels.syrk.flops(a8, a8, a8),
        "chol": kernels.chol.flops(a8),
        "qr_factor": kernels.qr_factor.flops(a8),
        "qr_factor_stack": kernels.qr_factor.flops(a8, a8),
        "lq_factor": kernels.lq_factor.flops(a8),
        "qr_leaf": kernels.qr_leaf.flops(a8, a8, a8),
        "lq_leaf": kernels.lq_leaf.flops(a8, a8, a8),
        "qr_trailing_update": kernels.qr_trailing_update.flops(a16x8, a8, a8, a8),
        "lq_trailing_update": kernels.lq_trailing_update.flops(a16x8, a8, a8, a8),
        "trsm_has_flops": float(hasattr(kernels.trsm, "flops")),
    }
    out["flops_json"] = np.frombuffer(json.dumps({k: float(v) for k, v in flops.items()}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "kernels_kat.npz"), **out)
    print("kernels_kat.npz:", len(out), "arrays")


# ---------------------------------------------------------------------------------------------
# (ii) DAG fixtures
# ---------------------------------------------------------------------------------------------
def dummy(name, ndims):
    shape = tuple(1 for _ in range(ndims))
    return BigMatrix(name, shape=shape, shard_sizes=shape, write_header=False, safe=False)


def node_key(node):
    return [int(node[0]), {str(k): int(v) for k, v in sorted(node[1].items())}]


def describe_block(ib):
    reads, writes, call = [], [], None
    for ins in ib.instrs:
        if isinstance(ins, lp.RemoteRead):
            reads.append([ins.matrix.key, [int(x) for x in ins.bidxs]])
        elif isinstance(ins, lp.RemoteWrite):
            writes.append([ins.matrix.key, [int(x) for x in ins.bidxs], int(ins.data_idx)])
        elif isinstance(ins, lp.RemoteCall):
            fargs = [float(a) if isinstance(a, float) else None for a in ins.argv_instr]
            call = {"kernel": ins.compute.__name__, "kwargs": {k: (v if isinstance(v, (int, float, bool, str)) else str(v))
                                                                for k, v in ins.kwargs.items()},
                    "num_outputs": len(ins.results), "float_args": fargs}
    return {"reads": reads, "call": call, "writes": writes}


def dag_fixture(name, fn, args, inputs, outputs, with_edges=True):
    prog = compiler.lpcompile_for_execution(fn, inputs=inputs, outputs=outputs)(*args)
    states = compiler.walk_program(prog.remote_calls)
    tasks = []
    for e, v in states:
        v = {str(k): int(x) for k, v_ in [(0, 0)] for k, x in v.items()} if False else {str(k): int(x) for k, x in v.items()}
        d = {"node": node_key((e, v)), "block": describe_block(prog.eval_expr(e, v)),
             "is_terminator": bool(prog.is_terminator(e))}
        if with_edges:
            d["children"] = sorted((node_key(c) for c in prog.find_children(e, v)), key=json.dumps)
            d["parents"] = sorted((node_key(p) for p in prog.find_parents(e, v)), key=json.dumps)
        tasks.append(d)
    fx = {"name": name, "inputs": inputs, "outputs": outputs,
          "args": [a.key if isinstance(a, BigMatrix) else a for a in args],
          "starters": sorted((node_key(s) for s in prog.starters), key=json.dumps),
          "num_terminators": int(prog.num_terminators), "tasks": tasks}
    print(f"  dag {name}: {len(tasks)} tasks, {len(fx['starters'])} starters, {fx['num_terminators']} terminators")
    return fx


def make_dags():
    fxs = []
    for n in (1, 2, 3, 4, 5, 8):
        fxs.append(dag_fixture(f"cholesky_{n}", algs.CHOLESKY, (dummy("O", 2), dummy("I", 2), dummy("S", 3), n, 0),
                               ["I"], ["O"]))
    fxs.append(dag_fixture("cholesky_5_trunc2", algs.CHOLESKY, (dummy("O", 2), dummy("I", 2), dummy("S", 3), 5, 2),
                           ["I"], ["O"]))
    for n in (1, 2, 4, 8, 16):
        fxs.append(dag_fixture(f"tsqr_{n}", algs.TSQR, (dummy("A", 2), dummy("Vs", 2), dummy("Ts", 2), dummy("Rs", 2), n),
                               ["A"], ["Rs"]))
    for (m, n, k) in ((1, 1, 1), (2, 2, 2), (4, 4, 4), (5, 5, 5), (2, 3, 4)):
        fxs.append(dag_fixture(f"gemm_{m}_{n}_{k}", algs.GEMM,
                               (dummy("A", 2), dummy("B", 2), m, n, k, dummy("Temp", 4), dummy("Out", 2)),
                               ["A", "B"], ["Out"]))
    for n in (2, 3, 4):
        mats = [dummy(nm, d) for nm, d in (("I", 2), ("V_QR", 3), ("T_QR", 3), ("S_QR", 4), ("R_QR", 3),
                                             ("V_LQ", 3), ("T_LQ", 3), ("S_LQ", 4), ("L_LQ", 3))]
        fxs.append(dag_fixture(f"bdfac_{n}", algs.BDFAC, tuple(mats) + (n, 0), ["I"], ["R_QR", "L_LQ"]))
    for n in (2, 3, 4):
        mats = [dummy(nm, d) for nm, d in (("I", 2), ("Vs", 3), ("Ts", 3), ("Rs", 3), ("S", 4))]
        fxs.append(dag_fixture(f"qr_{n}", algs.QR, tuple(mats) + (n, 0), ["I"], ["Rs"]))
    fxs.append(dag_fixture("simple_linear_4", algs.SimpleTestLinear, (dummy("A", 2), dummy("B", 2), 4), ["A"], ["B"]))
    fxs.append(dag_fixture("simple_linear2_4", algs.SimpleTestLinear2, (dummy("A", 2), dummy("B", 2), 4), ["A"], ["B"]))
    fxs.append(dag_fixture("simple_nonlinear_4", algs.SimpleTestNonLinear, (dummy("A", 3), dummy("B", 1), 4), ["A"], ["B"]))
    # the counts pinned by the reference's tests/test_starters_terminators.py (no edges: too slow / large)
    counts = {}
    prog = compiler.lpcompile(algs.CHOLESKY)(dummy("O", 2), dummy("I", 2), dummy("S", 3), 313, 0)
    counts["cholesky_313"] = {"starters": [node_key(s) for s in compiler.find_starters(prog, ["I"])],
                              "num_terminators": len(compiler.find_terminators(prog, ["O"]))}
    prog = compiler.lpcompile(algs.GEMM)(dummy("A", 2), dummy("B", 2), 4, 4, 4, dummy("Temp", 4), dummy("Out", 3))
    counts["gemm_4"] = {"num_starters": len(compiler.find_starters(prog, ["A", "B"])),
                        "num_terminators": len(compiler.find_terminators(prog, ["Out"]))}
    prog = compiler.lpcompile(algs.QR)(dummy("I", 2), dummy("Vs", 2), dummy("Ts", 2), dummy("Rs", 2), dummy("S", 4), 64, 0)
    counts["qr_64"] = {"num_starters": len(compiler.find_starters(prog, ["I"])),
                       "num_terminators": len(compiler.find_terminators(prog, ["Rs"]))}
    with open(os.path.join(HERE, "dag.json"), "w") as f:
        json.dump({"programs": fxs, "counts": counts}, f, separators=(",", ":"), sort_keys=True)


# ---------------------------------------------------------------------------------------------
# (iii) block indexing
# ---------------------------------------------------------------------------------------------
def make_indexing():
    cases = []
    specs = [((128, 128), (128, 128)), ((128, 128), (64, 64)), ((200, 200), (101, 101)), ((21, 67, 53), (21, 16, 11)),
             ((8, 8, 8, 8), (4, 4, 4, 4)), ((5, 32, 32), (1, 8, 8)), ((100,), (30,)), ((64, 8), (8, 8)),
             ((3 * 8, 64), (8, 8))]
    for ci, (shape, shards) in enumerate(specs):
        bm = BigMatrix(f"idx_{ci}", shape=shape, shard_sizes=shards, write_header=False)
        c = {"shape": list(shape), "shard_sizes": list(shards),
             "blocks_axis": [[list(b) for b in bm._blocks(axis=a)] for a in range(len(shape))],
             "block_idxs_axis": [bm._block_idxs(axis=a) for a in range(len(shape))],
             "num_blocks_axis": [bm.num_blocks(a) for a in range(len(shape))],
             "num_blocks": bm.num_blocks(),
             "block_idxs": [list(b) for b in bm.block_idxs][:64],
             "blocks": [[list(x) for x in b] for b in bm.blocks][:64],
             "str": str(bm), "keys": []}
        for bidx in bm.block_idxs[:64]:
            real = bm.__block_idx_to_real_idx__(bidx)
            c["keys"].append({"bidx": list(bidx), "real": [list(x) for x in real],
                              "key": bm.__shard_idx_to_key__(bidx)})
        # one index beyond the nominal shape (tolerated when safe=False; TSQR/BDFAC rely on it)
        beyond = tuple(n + 1 for n in [bm.num_blocks(a) for a in range(len(shape))])
        c["beyond"] = {"bidx": list(beyond), "real": [list(x) for x in bm.__block_idx_to_real_idx__(beyond)],
                       "key": bm.__shard_idx_to_key__(beyond)}
        cases.append(c)
    views = []
    vspecs = [((128, 128), (32, 32), [[2]]), ((128, 128), (32, 32), [[2, None]]), ((128, 128), (32, 32), [None, [0, 3]]),
              ((128, 128), (32, 32), [None, [3, None]]), ((128, 128), (16, 16), [[None, None, 2]]),
              ((128, 128), (16, 16), [[1, None, 2]]), ((128, 128), (16, 16), [None, [0, 6, 4]]),
              ((128, 128), (16, 16), [None, [6, 8, 4]]), ((128, 128), (64, 64), [0]), ((128, 128), (64, 64), [1, 1]),
              ((128, 128), (64, 64), [None, 0]), ((21, 67, 53), (21, 16, 11), [0, 4, 4]),
              ((200, 200), (101, 101), [[1, None]]), ((200, 200), (101, 101), [None, 1])]
    for vi, (shape, shards, sl) in enumerate(vspecs):
        bm = BigMatrix(f"view_{vi}", shape=shape, shard_sizes=shards, write_header=False)
        for transposed in (False, True):
            v = bm.submatrix(*sl)
            if transposed:
                if len(shape) != 2:
                    continue
                v = matrix.BigMatrixView(bm, [npw.utils.convert_to_slice(s) for s in sl], transposed=True)
            entry = {"shape": list(shape), "shard_sizes": list(shards), "slices": sl, "transposed": transposed,
                     "view_shape": [int(x) for x in v.shape], "view_shard_sizes": [int(x) for x in v.shard_sizes],
                     "str": str(v), "block_idxs_axis": [], "maps": []}
            for a in range(len(v.shape)):
                entry["block_idxs_axis"].append([int(x) for x in v._block_idxs(axis=a)])
            import itertools
            for vidx in itertools.product(*entry["block_idxs_axis"]):
                entry["maps"].append([list(vidx), [int(x) for x in v.true_block_idx(*vidx)]])
            views.append(entry)
    T = BigMatrix("tr", shape=(128, 64), shard_sizes=(32, 16), write_header=False).T
    views.append({"shape": [128, 64], "shard_sizes": [32, 16], "slices": [], "transposed": True,
                  "view_shape": [int(x) for x in T.shape], "view_shard_sizes": [int(x) for x in T.shard_sizes],
                  "str": str(T), "block_idxs_axis": [[int(x) for x in T._block_idxs(axis=a)] for a in range(2)],
                  "maps": [[[i, j], [int(x) for x in T.true_block_idx(i, j)]] for i in range(4) for j in range(4)]})
    cs = [[x, [[s.start, s.stop, s.step]]] for x in (None, 3, [5], [1, 4], [1, 9, 2])
          for s in [npw.utils.convert_to_slice(x)]]
    with open(os.path.join(HERE, "indexing.json"), "w") as f:
        json.dump({"matrices": cases, "views": views, "convert_to_slice": cs}, f, separators=(",", ":"), sort_keys=True)
    print("indexing.json:", len(cases), "matrices,", len(views), "views")


# ---------------------------------------------------------------------------------------------
# (iv) whole algorithms through the reference program objects
# ---------------------------------------------------------------------------------------------
def make_algos():
    out = {}
    rng = np.random.default_rng(7)

    # --- cholesky (reference alg_wrappers.py:16-27 minus config/LambdaPackProgram) -------------
    def run_cholesky(tag, n, b, lambdav=0.0, truncate=0):
        STORE.clear()
        X = rng.standard_normal((n, n))
        A = X @ X.T + np.eye(n)
        Ab = BigMatrix(f"chol_in_{tag}", shape=A.shape, shard_sizes=(b, b), write_header=False, lambdav=lambdav)
        shard(Ab, A)
        nb = Ab.num_blocks(1)
        S = BigMatrix("Cholesky.Intermediate({0})".format(Ab.key), shape=(nb + 1, n, n), shard_sizes=(1, b, b),
                      bucket=Ab.bucket, write_header=False, parent_fn=constant_zeros)
        O = BigMatrix("Cholesky({0})".format(Ab.key), shape=(n, n), shard_sizes=(b, b), write_header=False,
                      parent_fn=constant_zeros)
        prog = compiler.lpcompile_for_execution(algs.CHOLESKY, inputs=["I"], outputs=["O"])(
            O, Ab, S, int(np.ceil(n / b)), truncate)
        order = run_program(prog)
        out[f"cholesky_{tag}/A"] = A
        out[f"cholesky_{tag}/L"] = gather(O)
        out[f"cholesky_{tag}/meta"] = np.asarray([n, b, lambdav, truncate, len(order)], dtype=np.float64)

    run_cholesky("32_8", 32, 8)
    run_cholesky("20_8", 20, 8)           # ragged last tile
    run_cholesky("24_8_lam", 24, 8, lambdav=3.5)
    run_cholesky("40_8_t2", 40, 8, truncate=2)

    # --- tsqr (reference alg_wrappers.py:30-47) ------------------------------------------------------
    def run_tsqr(tag, m, b):
        STORE.clear()
        X = rng.standard_normal((m, b))
        Xb = BigMatrix(f"tsqr_in_{tag}", shape=X.shape, shard_sizes=(b, b), write_header=False)
        shard(Xb, X)
        levels = max(int(np.ceil(np.log2(Xb.num_blocks(0)))), 1)
        R = BigMatrix("tsqr_R({0})".format(Xb.key), shape=(levels * b, X.shape[0]), shard_sizes=(b, b),
                      write_header=False, safe=False)
        T = BigMatrix("tsqr_T({0})".format(Xb.key), shape=(levels * b * 2, X.shape[0]), shard_sizes=(b * 2, b),
                      write_header=False, safe=False)
        V = BigMatrix("tsqr_V({0})".format(Xb.key), shape=(levels * b * 2, X.shape[0]), shard_sizes=(b * 2, b),
                      write_header=False, safe=False)
        prog = compiler.lpcompile_for_execution(algs.TSQR, inputs=["A"], outputs=["Rs"])(Xb, V, T, R, Xb.num_blocks(0))
        run_program(prog)
        out[f"tsqr_{tag}/X"] = X
        out[f"tsqr_{tag}/R_final"] = R.get_block(levels, 0)
        out[f"tsqr_{tag}/R_leaf0"] = R.get_block(0, 0)
        out[f"tsqr_{tag}/V_leaf0"] = V.get_block(0, 0)
        out[f"tsqr_{tag}/T_leaf0"] = T.get_block(0, 0)
        out[f"tsqr_{tag}/V_top"] = V.get_block(levels, 0)
        out[f"tsqr_{tag}/T_top"] = T.get_block(levels, 0)

    run_tsqr("64_8", 64, 8)
    run_tsqr("32_16", 32, 16)

    # --- gemm (reference alg_wrappers.py:49-65) -----------------------------------------------------------
    def run_gemm(tag, n, b, dtype=np.float64):
        STORE.clear()
        A = rng.standard_normal((n, n)).astype(dtype)
        B = rng.standard_normal((n, n)).astype(dtype)
        Ab = BigMatrix(f"gemm_A_{tag}", shape=A.shape, shard_sizes=(b, b), write_header=False, dtype=dtype)
        Bb = BigMatrix(f"gemm_B_{tag}", shape=B.shape, shard_sizes=(b, b), write_header=False, dtype=dtype)
        shard(Ab, A)
        shard(Bb, B)
        levels = max(int(np.ceil(np.log2(Ab.num_blocks(1)) / np.log2(4))), 1)
        Temp = BigMatrix(f"matmul_test_Temp({Ab.key},{Bb.key})", shape=(n, n, n, levels), shard_sizes=[b, b, 1, 1],
                         write_header=False, safe=False, parent_fn=constant_zeros)
        C = BigMatrix("matmul_test_C", shape=(n, n), shard_sizes=(b, b), write_header=False)
        prog = compiler.lpcompile_for_execution(algs.GEMM, inputs=["A", "B"], outputs=["Out"])(
            Ab, Bb, Ab.num_blocks(0), Ab.num_blocks(1), Bb.num_blocks(1), Temp, C)
        run_program(prog)
        out[f"gemm_{tag}/A"] = A
        out[f"gemm_{tag}/B"] = B
        out[f"gemm_{tag}/C"] = gather(C)

    run_gemm("32_8", 32, 8)       # K = 4 tiles: one tree level
    run_gemm("40_8", 40, 8)       # K = 5 tiles: two levels, zero-padded operands
    run_gemm("16_8_f32", 16, 8, dtype=np.float32)

    # --- bdfac (reference alg_wrappers.py:92-114) ----------------------------------------------------------
    def run_bdfac(tag, n, b):
        STORE.clear()
        X = rng.standard_normal((n, n))
        Xb = BigMatrix(f"bdfac_in_{tag}", shape=X.shape, shard_sizes=(b, b), write_header=False)
        shard(Xb, X)
        nbk = Xb.num_blocks(0)
        levels = max(int(np.ceil(np.log2(nbk))), 1) + 1
        mk = lambda name, shape, shards, pf=None: BigMatrix(name + "_" + tag, shape=shape, shard_sizes=shards,
                                                             write_header=False, safe=False, parent_fn=pf)
        V_QR = mk("V_QR", (2 * n, levels, 2 * n), (1, 1, b))
        T_QR = mk("T_QR", (2 * n, levels, 2 * n), (1, 1, b))
        R_QR = mk("R_QR", (2 * n, levels, 2 * n), (b, 1, b), constant_zeros)
        S_QR = mk("S_QR", (2 * n, levels, 2 * n, 2 * n), (1, 1, b, b), constant_zeros)
        V_LQ = mk("V_LQ", (2 * n, levels, 2 * n), (1, 1, b))
        T_LQ = mk("T_LQ", (2 * n, levels, 2 * n), (1, 1, b))
        L_LQ = mk("L_LQ", (2 * n, levels, 2 * n), (1, 1, b), constant_zeros_ext)
        S_LQ = mk("S_LQ", (2 * n, levels, 2 * n, 2 * n), (1, 1, b, b), constant_zeros_ext)
        prog = compiler.lpcompile_for_execution(algs.BDFAC, inputs=["I"], outputs=["R_QR", "L_LQ"])(
            Xb, V_QR, T_QR, S_QR, R_QR, V_LQ, T_LQ, S_LQ, L_LQ, nbk, 0)
        order = run_program(prog)
        out[f"bdfac_{tag}/X"] = X
        out[f"bdfac_{tag}/ntasks"] = np.asarray(len(order))
        # the blocks the reference's test assembles (tests/test_alg_correctness.py:257-278)
        if nbk == 4:
            blocks = {"R_0_2_0": R_QR.get_block(0, 2, 0), "L_0_2_1": L_LQ.get_block(0, 2, 1),
                      "R_1_2_1": R_QR.get_block(1, 2, 1), "L_1_1_2": L_LQ.get_block(1, 1, 2),
                      "R_2_1_2": R_QR.get_block(2, 1, 2), "L_2_0_3": L_LQ.get_block(2, 0, 3),
                      "R_3_0_3": R_QR.get_block(3, 0, 3)}
            for k, v in blocks.items():
                out[f"bdfac_{tag}/{k}"] = v

    run_bdfac("16_4", 16, 4)
    np.savez_compressed(os.path.join(HERE, "algos.npz"), **out)
    print("algos.npz:", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["kernels", "indexing", "algos", "dags"]
    if "kernels" in which:
        make_kernel_kats()
    if "indexing" in which:
        make_indexing()
    if "algos" in which:
        make_algos()
    if "dags" in which:
        make_dags()
