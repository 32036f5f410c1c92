"""chol on a CU-masked stream beside syrk launches on the complementary mask: does the pair overlap?

usage: python tools/overlap_probe.py [chain_cus ...]     (default 32 64)
"""
import sys, os, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from numpywren_amd.device import get_backend, Stream
from numpywren_amd import _ffi

be = get_backend()
n = 4096
ncu = be.compute_units
words = (ncu + 31) // 32


def masked(bits, name):
    mask = [0] * words
    for b in bits:
        mask[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * words)(*mask)
    h = ctypes.c_void_p(0)
    _ffi.check(be.lib.npw_stream_create_masked(ctypes.byref(h), arr, words), "masked")
    return Stream(h.value, False, name)


G = be.fill_random((n, 256), seed=5)
A = be.add_diag(be.gemm(G, G, False, True), float(n))
S = be.fill_random((n, n), 1); X = be.fill_random((n, n), 2); Y = be.fill_random((n, n), 3)
be.synchronize()


def timed(fn, reps=5):
    fn(); be.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    be.synchronize()
    return (time.time() - t0) / reps * 1e3


full = be.create_stream(name="full")
print("full chip: chol %.3f ms  syrk %.3f ms" % (timed(lambda: be.chol(A, stream=full)), timed(lambda: be.syrk(S, X, Y, stream=full, exact_zero=False))))
for kind in ("interleaved", "leading"):
    for c in [int(a) for a in sys.argv[1:]] or [32, 64]:
        if kind == "interleaved":   # every (ncu/c)-th CU
            step = ncu // c
            chain_bits = list(range(0, ncu, step))[:c]
        else:                       # the first c bits
            chain_bits = list(range(c))
        rest = [b for b in range(ncu) if b not in set(chain_bits)]
        sb = masked(chain_bits, "chain")
        sa = masked(rest, "rest")
        t_chol = timed(lambda: be.chol(A, stream=sb))
        t_syrk = timed(lambda: be.syrk(S, X, Y, stream=sa, exact_zero=False))

        def both(k):
            def f():
                be.chol(A, stream=sb)
                for _ in range(k):
                    be.syrk(S, X, Y, stream=sa, exact_zero=False)
            return f
        res = ["%d syrk: %.3f" % (k, timed(both(k))) for k in (1, 2, 3)]
        print("%s chain=%d CUs: chol alone %.3f  syrk alone on %d CUs %.3f   chol || k syrk: %s" % (kind, c, t_chol, len(rest), t_syrk, "  ".join(res)))
