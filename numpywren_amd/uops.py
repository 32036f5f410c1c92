"""Unary-operator surface of the reference (numpywren/uops.py).  There every function except
`chol` is a stub that raises NotImplementedError (uops.py:18-100) and `chol` calls a
LambdaPACK entry point that no longer exists (`lp._chol`, uops.py:102-119).  The same names are
exposed here; `chol` is given its evident meaning: build the Cholesky program, run it on the
local GPU, return the factor as a BigMatrix."""
from . import alg_wrappers, job_runner
from . import lambdapack as lp

_STUBS = ("reshard", "sum", "prod", "argmin", "argmax", "min", "max", "norm", "sqrt", "neg", "abs", "square", "sign", "ceil",
          "floor", "round", "exp", "log", "log10", "log2", "sin", "cos", "tan", "power", "elemwise_uop_func")


def _make_stub(name):
    def stub(*args, **kwargs):
        raise NotImplementedError(f"uops.{name} is not implemented (it is a stub in the reference as well)")

    stub.__name__ = name
    return stub


for _n in _STUBS:
    globals()[_n] = _make_stub(_n)


def chol(pwex, X, out_bucket=None, tasks_per_job=1):
    """Cholesky factor of the BigMatrix X; `pwex` (a pywren executor in the reference) is ignored."""
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    job_runner.lambdapack_run(program)
    program.wait()
    if program.program_status() != lp.PS.SUCCESS:
        raise Exception("cholesky failed: {0}".format(program.exceptions))
    program.free()
    return meta["outputs"][0]
