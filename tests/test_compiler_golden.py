"""numpywren_amd's LambdaPACK front end + compiler against the DAG fixtures produced by the
reference's frontend.py / compiler.py (tests/golden/dag.json), plus the invariants the reference's
tests/test_dependency_analyze.py and tests/test_starters_terminators.py assert.  CPU only."""
import json
import os

import pytest

from conftest import GOLDEN
from numpywren_amd import algs, compiler
from numpywren_amd.matrix import BigMatrix

FX = json.load(open(os.path.join(GOLDEN, "dag.json")))
PROGRAMS = {p["name"]: p for p in FX["programs"]}
NDIMS = {'O': 2, 'I': 2, 'S': 3, 'A': 2, 'Vs': 2, 'Ts': 2, 'Rs': 2, 'B': 2, 'Temp': 4, 'Out': 2, 'V_QR': 3, 'T_QR': 3,
         'S_QR': 4, 'R_QR': 3, 'V_LQ': 3, 'T_LQ': 3, 'S_LQ': 4, 'L_LQ': 3}
FNS = {'cholesky': algs.CHOLESKY, 'tsqr': algs.TSQR, 'gemm': algs.GEMM, 'bdfac': algs.BDFAC, 'qr': algs.QR,
       'simple_linear_4': algs.SimpleTestLinear, 'simple_linear2_4': algs.SimpleTestLinear2,
       'simple_nonlinear_4': algs.SimpleTestNonLinear}


def dummy(name, ndims):
    shape = tuple(1 for _ in range(ndims))
    return BigMatrix(name, shape=shape, shard_sizes=shape, write_header=False, safe=False)


def nk(node):
    return (int(node[0]), tuple(sorted((str(k), int(v)) for k, v in node[1].items())))


def build(fx):
    fn = FNS.get(fx["name"]) or FNS[fx["name"].split("_")[0]]
    args = [dummy(a, NDIMS.get(a, 3)) if isinstance(a, str) else a for a in fx["args"]]
    return compiler.lpcompile_for_execution(fn, fx["inputs"], fx["outputs"])(*args)


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_dag_matches_reference(name, host_store):
    fx = PROGRAMS[name]
    p = build(fx)
    mine = {nk(t.node): t for t in p.tasks}
    ref = {nk(t["node"]): t for t in fx["tasks"]}
    assert set(mine) == set(ref)
    assert sorted(nk(s) for s in p.starters) == sorted(nk(s) for s in fx["starters"])
    assert p.num_terminators == fx["num_terminators"]
    for key, rt in ref.items():
        mt = mine[key]
        assert [[m, list(i)] for m, i in mt.reads] == rt["block"]["reads"]
        assert [[m, list(i), n] for n, (m, i) in enumerate(mt.writes)] == rt["block"]["writes"]
        assert p.kernel(mt.expr_idx).__name__ == rt["block"]["call"]["kernel"]
        assert len(mt.writes) == rt["block"]["call"]["num_outputs"]
        assert sorted(nk(c) for c in p.find_children(*mt.node)) == sorted(nk(c) for c in rt["children"])
        assert sorted(nk(c) for c in p.find_parents(*mt.node)) == sorted(nk(c) for c in rt["parents"])
        assert p.is_terminator(mt.expr_idx) == rt["is_terminator"]
        # the instruction block view agrees with the task
        ib = p.eval_expr(*mt.node)
        kinds = [type(i).__name__ for i in ib.instrs]
        assert kinds == ["RemoteRead"] * len(mt.reads) + ["RemoteCall"] + ["RemoteWrite"] * len(mt.writes)


@pytest.mark.parametrize("name", ["cholesky_8", "tsqr_16", "gemm_4_4_4", "simple_linear_4", "simple_linear2_4",
                                  "simple_nonlinear_4", "bdfac_4"])
def test_children_parents_are_inverse(name, host_store):
    """reference tests/test_dependency_analyze.py:18-31 verify_program"""
    p = build(PROGRAMS[name])
    for e, v in compiler.walk_program(p):
        for c in compiler.find_children(p, e, v):
            assert (e, v) in compiler.find_parents(p, *c)
        for q in compiler.find_parents(p, e, v):
            assert (e, v) in compiler.find_children(p, *q)


def test_reference_counts(host_store):
    """reference tests/test_starters_terminators.py:14-43 (N = 313 Cholesky, 4^3 GEMM) + QR at 64"""
    c = FX["counts"]
    p = compiler.lpcompile(algs.CHOLESKY)(dummy("O", 2), dummy("I", 2), dummy("S", 3), 313, 0)
    assert compiler.find_starters(p, ["I"]) == [(0, {})] == [tuple(x) for x in c["cholesky_313"]["starters"]]
    assert len(compiler.find_terminators(p, ["O"])) == 49141 == c["cholesky_313"]["num_terminators"]
    p = compiler.lpcompile(algs.GEMM)(dummy("A", 2), dummy("B", 2), 4, 4, 4, dummy("Temp", 4), dummy("Out", 3))
    assert len(compiler.find_starters(p, ["A", "B"])) == 64 == c["gemm_4"]["num_starters"]
    assert len(compiler.find_terminators(p, ["Out"])) == 16 == c["gemm_4"]["num_terminators"]
    p = compiler.lpcompile(algs.QR)(dummy("I", 2), dummy("Vs", 2), dummy("Ts", 2), dummy("Rs", 2), dummy("S", 4), 64, 0)
    assert len(compiler.find_starters(p, ["I"])) == c["qr_64"]["num_starters"]
    assert len(compiler.find_terminators(p, ["Rs"])) == c["qr_64"]["num_terminators"]


def test_headline_task_counts(host_store):
    """config 3 Cholesky = 816 tasks (16 chol + 120 trsm + 680 syrk), 136 terminators."""
    p = compiler.lpcompile_for_execution(algs.CHOLESKY, ["I"], ["O"])(dummy("O", 2), dummy("I", 2), dummy("S", 3), 16, 0)
    names = [p.kernel(t.expr_idx).__name__ for t in p.tasks]
    assert (names.count("chol"), names.count("trsm"), names.count("syrk")) == (16, 120, 680)
    assert p.num_terminators == 136 and p.starters == [(0, {})]
    p = compiler.lpcompile_for_execution(algs.TSQR, ["A"], ["Rs"])(dummy("A", 2), dummy("Vs", 2), dummy("Ts", 2), dummy("Rs", 2), 256)
    assert len(p.tasks) == 511
    p = compiler.lpcompile_for_execution(algs.GEMM, ["A", "B"], ["Out"])(dummy("A", 2), dummy("B", 2), 8, 8, 8, dummy("Temp", 4), dummy("Out", 2))
    assert len(p.tasks) == 512 + 192 + 64


def test_static_if_and_scalar_args(host_store):
    """static `if` on loop variables and scalar kernel arguments (reference tests/test_if.py)."""
    def prog(A: BigMatrix, B: BigMatrix, N: int):
        for i in range(N):
            if i % 2 == 0:
                B[i] = mul(2.0, A[i])
            else:
                B[i] = identity(A[i])

    p = compiler.lpcompile_for_execution(prog, ["A"], ["B"])(dummy("A", 1), dummy("B", 1), 5)
    ks = [(t.vars["i"], p.kernel(t.expr_idx).__name__, t.consts) for t in p.tasks]
    assert ks == [(0, "mul", [2.0]), (1, "identity", []), (2, "mul", [2.0]), (3, "identity", []), (4, "mul", [2.0])]
    import operator
    assert p.kernel(0) is operator.mul  # reference frontend.py:15 `from operator import *` shadows kernels.mul


def test_non_ssa_program_rejected(host_store):
    from numpywren_amd.exceptions import LambdaPackBackendGenerationException

    def bad(A: BigMatrix, B: BigMatrix, N: int):
        for i in range(N):
            B[0] = identity(A[i])

    p = compiler.lpcompile_for_execution(bad, ["A"], ["B"])(dummy("A", 1), dummy("B", 1), 3)
    with pytest.raises(LambdaPackBackendGenerationException):
        p.tasks
