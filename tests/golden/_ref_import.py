"""Import the reference (Vaishaal/numpywren at /root/reference) in the authoring container.

Only used by tests/golden/make_golden.py to GENERATE the committed fixtures; nothing on the
GPU box imports this (the reference does not travel).  The reference needs AWS/pywren/redis
modules that are absent here, so empty stand-in modules are registered for the *imports
only* -- none of their functionality is on the kernel / DAG / block-indexing path we record.

The f2py LAPACK modules the reference downloads from S3 at run time (dgeqrt3, dtpqrt;
reference numpywren/kernels.py:22-40,86-124) are not in /root/reference either; a shim
module `dgeqrt3` backed by scipy's LAPACK DGEQRT with nb = n (a single DGEQRT3 call) is
registered so that the reference's own fast_qr post-processing (kernels.py:99-105) runs
unchanged.  netlib LAPACK 3.8.0 is the pinned third-party arithmetic (reference
scripts/lapack.py:12-13).
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    if "pywren" in sys.modules and getattr(sys.modules["pywren"], "_npw_stub", False):
        return

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    _mod("boto3", client=lambda *a, **k: _Any(), resource=lambda *a, **k: _Any())
    be = _mod("botocore.exceptions", ClientError=type("ClientError", (Exception,), {}))
    _mod("botocore", exceptions=be)
    _mod("aiobotocore", get_session=lambda *a, **k: _Any())
    ce = _mod("aiohttp.client_exceptions", ClientPayloadError=type("ClientPayloadError", (Exception,), {}))
    _mod("aiohttp", client_exceptions=ce)
    wc = _mod("pywren.wrenconfig",
              default=lambda *a, **k: {"s3": {"bucket": "stub-bucket"}, "account": {"aws_region": "stub-region"},
                                        "runtime": {}})
    ser = _mod("pywren.serialize", serialize=_Any())
    ex = _mod("pywren.executor", Executor=_Any)
    pw = _mod("pywren", wrenconfig=wc, serialize=ser, executor=ex, ec2standalone=_mod("pywren.ec2standalone"),
              future=_mod("pywren.future"), storage=_mod("pywren.storage"), wait=lambda *a, **k: None,
              default_executor=lambda *a, **k: _Any(), standalone_executor=lambda *a, **k: _Any())
    pw._npw_stub = True
    rex = _mod("redis.exceptions", TimeoutError=type("TimeoutError", (Exception,), {}),
               WatchError=type("WatchError", (Exception,), {}))
    _mod("redis", exceptions=rex, StrictRedis=_Any, WatchError=rex.WatchError)
    _mod("astor", dump=lambda *a, **k: "", dump_tree=lambda *a, **k: "", to_source=lambda *a, **k: "")
    for name in ("numba", "tblib", "watchtower", "glob2", "flaky"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)
    # numpy-2 spellings used by the 2018-era reference
    if not hasattr(np, "product"):
        np.product = np.prod
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float

    # LAPACK shims (see module docstring)
    import scipy.linalg.lapack as lapack

    def dgeqrt3(m, n, a, t, info=0):
        # in-place semantics of the f2py wrapper: a <- factored matrix, t <- T
        out_a, out_t, inf = lapack.dgeqrt(n, np.asfortranarray(a))
        a[...] = out_a
        t[...] = out_t
        return inf

    _mod("dgeqrt3", dgeqrt3=dgeqrt3)

    def dtpqrt(m, n, nb, l, a, b, t, work, info=0):
        # scipy.linalg.lapack.dtpqrt(l, nb, a, b) -> a, b, t, info
        out_a, out_b, out_t, inf = lapack.dtpqrt(l, nb, np.asfortranarray(a), np.asfortranarray(b))
        a[...] = out_a
        b[...] = out_b
        t[: out_t.shape[0], :] = out_t
        return inf

    _mod("dtpqrt", dtpqrt=dtpqrt)


def import_reference():
    """Returns the reference `numpywren` package (kernels, matrix, compiler, algs, lambdapack)."""
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import numpywren  # noqa: F401
    import numpywren.kernels as kernels

    kernels.get_shared_so = lambda name: None  # the .so fetch from S3 is replaced by the shims
    import numpywren.matrix  # noqa: F401
    import numpywren.matrix_utils  # noqa: F401
    import numpywren.lambdapack  # noqa: F401
    import numpywren.compiler  # noqa: F401
    import numpywren.algs  # noqa: F401

    return numpywren
