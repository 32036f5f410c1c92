"""The parts of the reference's kernels.py surface that no LambdaPACK program uses (VERDICT r5 "missing" 3-5): the three
non-default forms of kernels.trsm (reference kernels.py:254-257 hands `lower` and `right` to DTRSM), kernels.trsm_sub
(kernels.py:178-179), kernels.mul on two tiles (kernels.py:233-234) and kernels.qr_factor_triangular on the shapes the
reference's DTPQRT call accepts (kernels.py:107-124).  Each body runs twice: on the CPU over the checker backend (the
transposition / reversal algebra of kernels.py itself, no GPU needed) and on the GPU through the C-ABI; both against the
oracle.  fp64 tolerances: 1e-11 relative to the solution's magnitude for the solves (the library's solve multiplies by
inverted diagonal blocks, so it is not DTRSM's substitution order), bit-exact for mul."""
import numpy as np
import pytest

import npw_oracle as oracle
from numpywren_amd import kernels


def _tri_system(rng, n):
    x = rng.standard_normal((n, n)) + n * np.eye(n)     # well conditioned in both triangles
    return x


def check_trsm_forms(n, m):
    rng = np.random.default_rng(100 * n + m)
    x = _tri_system(rng, n)
    for lower in (False, True):
        for right in (True, False):
            y = rng.standard_normal((m, n) if right else (n, m))
            got = kernels.trsm(x, y, lower=lower, right=right)
            ref = oracle.trsm(x, y, lower=lower, right=right)
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, atol=1e-11 * np.abs(ref).max(), err_msg=f"lower={lower} right={right}")
            # the defining equation, with the triangle DTRSM reads
            a = np.tril(x.T) if lower else np.triu(x.T)
            res = got @ a - y if right else a @ got - y
            assert np.abs(res).max() <= 1e-11 * n * np.abs(y).max()
            # the zero short-circuit keeps the reference's shape rule in every form
            z = kernels.trsm(x, np.zeros_like(y), lower=lower, right=right)
            assert z.shape == (x.shape[1], y.shape[0]) and not z.any()


def check_trsm_sub(n, m):
    rng = np.random.default_rng(7 * n + m)
    L = _tri_system(rng, n)
    S, x = rng.standard_normal((n, m)), rng.standard_normal((n, m))
    got = kernels.trsm_sub(L, S, x)
    ref = oracle.trsm_sub(L, S, x)
    np.testing.assert_allclose(got, ref, atol=1e-11 * np.abs(ref).max())
    assert np.abs(np.triu(L) @ got - (x - S)).max() <= 1e-11 * n


def check_mul():
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((33, 70)), rng.standard_normal((33, 70))
    assert np.array_equal(kernels.mul(a, b), oracle.mul(a, b))
    assert np.array_equal(kernels.mul(a, 2.5), oracle.mul(a, 2.5))
    assert np.array_equal(kernels.mul(2.5, a), oracle.mul(2.5, a))
    with pytest.raises(ValueError):
        kernels.mul(a, b[:, :5])


def check_qr_factor_triangular_shapes(n, extra):
    rng = np.random.default_rng(n + extra)
    x0 = np.triu(rng.standard_normal((n, n)))
    x1 = rng.standard_normal((n + extra, n))
    x1[:n] = np.triu(x1[:n])
    v, t, r = kernels.qr_factor_triangular(x0, x1)
    vo, to, ro = oracle.qr_factor_triangular(x0, x1)
    assert v.shape == vo.shape and t.shape == to.shape and r.shape == ro.shape
    sgn = np.sign(np.diag(r)) * np.sign(np.diag(ro))     # DTPQRT and the library agree up to rounding, not up to signs
    assert np.all(sgn == 1)
    np.testing.assert_allclose(r, ro, atol=1e-11 * n)
    np.testing.assert_allclose(t, to, atol=1e-11 * n)
    np.testing.assert_allclose(v, vo, atol=1e-11 * n)
    for bad0, bad1 in (((n + 1, n), (n + 1, n)), ((n, n + 1), (n, n + 1)), ((n, n), (n - 1, n)), ((n, n), (n, n - 1))):
        with pytest.raises(ValueError):
            kernels.qr_factor_triangular(np.zeros(bad0), np.zeros(bad1))
        with pytest.raises(ValueError):
            oracle.qr_factor_triangular(np.zeros(bad0), np.zeros(bad1))


SHAPES = [(8, 8), (40, 17), (96, 130)]


@pytest.mark.parametrize("n,m", SHAPES)
def test_trsm_forms_host_algebra(n, m, oracle_backend):
    check_trsm_forms(n, m)


@pytest.mark.parametrize("n,m", SHAPES)
def test_trsm_sub_host_algebra(n, m, oracle_backend):
    check_trsm_sub(n, m)


def test_mul_host(oracle_backend):
    check_mul()


@pytest.mark.parametrize("n,extra", [(8, 0), (24, 5), (64, 64)])
def test_qr_factor_triangular_shapes_host(n, extra, oracle_backend):
    check_qr_factor_triangular_shapes(n, extra)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", SHAPES + [(512, 384)])
def test_trsm_forms_gpu(n, m):
    check_trsm_forms(n, m)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", SHAPES + [(512, 384)])
def test_trsm_sub_gpu(n, m):
    check_trsm_sub(n, m)


@pytest.mark.gpu
def test_mul_gpu():
    check_mul()


@pytest.mark.gpu
@pytest.mark.parametrize("n,extra", [(8, 0), (24, 5), (64, 64), (256, 32)])
def test_qr_factor_triangular_shapes_gpu(n, extra):
    check_qr_factor_triangular_shapes(n, extra)
