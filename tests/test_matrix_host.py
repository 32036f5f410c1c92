"""BigMatrix block indexing / views / read-write semantics on the host tier (no GPU): bit-exact against
the vectors recorded from the reference's matrix.py (tests/golden/indexing.json) and the behaviours
the reference's tests/test_simple.py, test_indexing.py, test_transpose.py, test_multiaxis.py pin."""
import itertools
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from numpywren_amd import utils
from numpywren_amd.matrix import BigMatrix, BigMatrixView, block_key_to_block
from numpywren_amd.matrix_init import local_numpy_init, shard_matrix
from numpywren_amd.matrix_utils import constant_zeros, constant_zeros_ext, get_col, get_row, put_row

FX = json.load(open(os.path.join(GOLDEN, "indexing.json")))


@pytest.mark.parametrize("ci", range(len(FX["matrices"])))
def test_block_indexing_vectors(ci, host_store):
    c = FX["matrices"][ci]
    shape, shards = tuple(c["shape"]), tuple(c["shard_sizes"])
    bm = BigMatrix(f"idx_{ci}", shape=shape, shard_sizes=shards)
    assert str(bm) == c["str"]
    for a in range(len(shape)):
        assert [list(b) for b in bm._blocks(axis=a)] == c["blocks_axis"][a]
        assert bm._block_idxs(axis=a) == c["block_idxs_axis"][a]
        assert bm.num_blocks(a) == c["num_blocks_axis"][a]
    assert bm.num_blocks() == c["num_blocks"]
    assert [list(b) for b in bm.block_idxs][:64] == c["block_idxs"]
    assert [[list(x) for x in b] for b in bm.blocks][:64] == c["blocks"]
    for k in c["keys"] + [c["beyond"]]:
        assert [list(x) for x in bm.__block_idx_to_real_idx__(k["bidx"])] == k["real"]
        assert bm.__shard_idx_to_key__(k["bidx"]) == k["key"]
        if k is not c["beyond"]:
            assert [list(x) for x in block_key_to_block(k["key"])] == k["real"]


@pytest.mark.parametrize("vi", range(len(FX["views"])))
def test_view_vectors(vi, host_store):
    v = FX["views"][vi]
    key = v["str"][len("BigMatrix("):v["str"].index(")")]   # the fixture generator's matrix name
    bm = BigMatrix(key, shape=tuple(v["shape"]), shard_sizes=tuple(v["shard_sizes"]))
    if v["slices"]:
        view = BigMatrixView(bm, [utils.convert_to_slice(s) for s in v["slices"]], transposed=v["transposed"])
    else:
        view = bm.T
    assert [int(x) for x in view.shape] == v["view_shape"]
    assert [int(x) for x in view.shard_sizes] == v["view_shard_sizes"]
    assert str(view) == v["str"]
    for a in range(len(view.shape)):
        assert [int(x) for x in view._block_idxs(axis=a)] == v["block_idxs_axis"][a]
    for vidx, pidx in v["maps"]:
        assert list(view.true_block_idx(*vidx)) == pidx


def test_convert_to_slice():
    for arg, (expect,) in FX["convert_to_slice"]:
        s = utils.convert_to_slice(arg)
        assert [s.start, s.stop, s.step] == expect


def test_roundtrips(host_store):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((128, 128))
    a = local_numpy_init(X, X.shape)
    assert np.all(a.numpy() == X)
    b = local_numpy_init(X, shard_sizes=(64, 64), write_header=True)
    assert np.all(b.numpy() == X)
    assert np.all(BigMatrix(b.key).numpy() == X)          # header reload (reference test_matrix_header)
    Y = rng.standard_normal((200, 200))
    c = local_numpy_init(Y, shard_sizes=(101, 101))       # uneven tiles
    assert np.all(c.numpy() == Y)
    assert c.get_block(1, 1).shape == (99, 99)
    c.free()
    assert c.block_idxs_exist == []
    Z = rng.standard_normal((8, 8, 8, 8))
    d = BigMatrix("multiaxis", shape=Z.shape, shard_sizes=(4, 4, 4, 4))
    shard_matrix(d, Z)
    assert len(d.block_idxs_exist) == 16
    assert np.all(d.numpy() == Z)


def test_submatrix_and_transpose(host_store):
    rng = np.random.default_rng(1)
    X = rng.standard_normal((128, 128))
    m = BigMatrix("t2", shape=X.shape, shard_sizes=[64, 64])
    shard_matrix(m, X)
    assert np.all(X[0:64, 0:64] == m.submatrix(0).get_block(0))
    assert np.all(X[64:128, 64:128] == m.submatrix(1, 1).get_block())
    assert np.all(X[0:64, 64:128] == m.submatrix(0, 1).get_block())
    assert np.all(X[64:128, 0:64] == m.submatrix(None, 0).get_block(1))
    assert np.all(m.T.numpy() == X.T)
    m2 = BigMatrix("t3", shape=X.shape, shard_sizes=[32, 32])
    shard_matrix(m2, X)
    assert np.all(X[0:64] == m2.submatrix([2]).numpy())
    assert np.all(X[64:128] == m2.submatrix([2, None]).numpy())
    assert np.all(X[:, 0:96] == m2.submatrix(None, [0, 3]).numpy())
    m3 = BigMatrix("t4", shape=X.shape, shard_sizes=[16, 16])
    shard_matrix(m3, X)
    assert np.all(X[::32] == m3.submatrix([None, None, 2]).numpy()[::16])
    assert np.all(X[:, 96:128:64] == m3.submatrix(None, [6, 8, 4]).numpy()[:, ::16])
    # put through a view
    m4 = BigMatrix("t5", shape=X.shape, shard_sizes=X.shape)
    m4.submatrix(0, 0).put_block(X)
    assert np.all(m4.numpy() == X)
    m5 = BigMatrix("t6", shape=(64, 32), shard_sizes=(32, 16))
    blk = rng.standard_normal((16, 32))
    m5.T.put_block(blk, 1, 0)
    assert np.all(m5.get_block(0, 1) == blk.T)
    W = rng.standard_normal((21, 67, 53))
    m6 = BigMatrix("t7", shape=W.shape, shard_sizes=[21, 16, 11])
    shard_matrix(m6, W)
    assert np.all(W[:, 64:67, 44:53] == m6.submatrix(0, 4, 4).numpy())


def test_read_write_semantics(host_store):
    # arity check, missing tile, parent_fn, autosqueeze, lambdav on every diagonal read, safe shape check
    S = BigMatrix("S", shape=(3, 16, 16), shard_sizes=(1, 8, 8), parent_fn=constant_zeros)
    with pytest.raises(Exception, match="does not match shape"):
        S.get_block(0, 0)
    z = S.get_block(2, 1, 1)
    assert z.shape == (8, 8) and not z.any()                       # squeezed zeros from parent_fn
    S.put_block(np.ones((8, 8)), 1, 0, 1)                          # squeezed input is reshaped to (1, 8, 8)
    assert S.get_block(1, 0, 1).shape == (8, 8)
    with pytest.raises(Exception, match="Incompatible block size"):
        S.put_block(np.ones((4, 8)), 1, 0, 0)
    N = BigMatrix("nofn", shape=(16, 16), shard_sizes=(8, 8))
    with pytest.raises(Exception, match="no parent function"):
        N.get_block(0, 0)
    A = np.arange(256.0).reshape(16, 16)
    L = BigMatrix("lam", shape=(16, 16), shard_sizes=(8, 8), lambdav=2.5)
    shard_matrix(L, A)
    d = L.get_block(1, 1)
    assert np.allclose(d, A[8:, 8:] + 2.5 * np.eye(8))
    assert np.allclose(L.get_block(1, 1), d)                       # applied on read, never stored
    assert np.allclose(L.get_block(0, 1), A[:8, 8:])
    with pytest.raises(Exception, match="square"):
        BigMatrix("bad", shape=(16, 8), shard_sizes=(8, 8), lambdav=1.0)
    E = BigMatrix("ext", shape=(32, 3, 32), shard_sizes=(1, 1, 8), parent_fn=constant_zeros_ext, safe=False)
    assert E.get_block(5, 1, 2).shape == (8, 8)
    U = BigMatrix("unsafe", shape=(8, 16), shard_sizes=(8, 8), safe=False)
    U.put_block(np.ones((16, 8)), 3, 0)                            # index beyond the nominal shape tolerated
    assert U.get_block(3, 0).shape == (16, 8)
    f32 = BigMatrix("f32", shape=(8, 8), shard_sizes=(8, 8))       # dtype is not cast on put
    f32.put_block(np.ones((8, 8), dtype=np.float32), 0, 0)
    assert f32.get_block(0, 0).dtype == np.float32


def test_rows_cols(host_store):
    X = np.random.default_rng(3).standard_normal((64, 48))
    m = BigMatrix("rc", shape=X.shape, shard_sizes=(16, 16))
    shard_matrix(m, X)
    assert np.all(get_row(m, 2) == X[32:48])
    assert np.all(get_col(m, 1) == X[:, 16:32])
    put_row(m, np.zeros((16, 48)), 0)
    assert not m.numpy()[:16].any()


def test_matrix_utils_row_col_helpers(host_store):
    """get_row / get_col / get_rows / put_row / put_col / chunk / block_key_to_block (reference matrix_utils.py)."""
    from numpywren_amd import matrix_utils
    rng = np.random.default_rng(4)
    Xh = rng.standard_normal((20, 14))
    X = BigMatrix("mu_rows", shape=Xh.shape, shard_sizes=(8, 6), write_header=True)
    shard_matrix(X, Xh)
    assert np.array_equal(matrix_utils.get_row(X, 1), Xh[8:16])
    assert np.array_equal(matrix_utils.get_col(X, 2), Xh[:, 12:14])
    assert np.array_equal(matrix_utils.get_rows(X, [0, 2]), np.vstack([Xh[0:8], Xh[16:20]]))
    new = rng.standard_normal((8, 14))
    matrix_utils.put_row(X, new, 0)
    assert np.array_equal(X.numpy()[:8], new)
    newc = rng.standard_normal((20, 6))
    matrix_utils.put_col(X, newc, 1)
    assert np.array_equal(X.numpy()[:, 6:12], newc)
    assert [list(c) for c in matrix_utils.chunk(list(range(7)), 3)] == [[0, 1, 2], [3, 4, 5], [6]]
    assert list(matrix_utils.chunk([1, 2], 0)) == []
    key = X.__shard_idx_to_key__((2, 1))
    assert matrix_utils.block_key_to_block(key) == ((16, 20), (6, 12))
    assert matrix_utils.block_key_to_block(X.key_base + "/header") is None


@pytest.mark.parametrize("store", ["host", "oracle"])
def test_reshard_down_and_empty_result_matrix(store, request):
    """matrix_init.reshard_down / empty_result_matrix (reference matrix_init.py:33-49, 100-148)."""
    request.getfixturevalue("host_store" if store == "host" else "oracle_backend")
    from numpywren_amd import matrix_init, matrix_utils
    rng = np.random.default_rng(8)
    Xh = rng.standard_normal((20, 12))
    X = BigMatrix("rs_in", shape=Xh.shape, shard_sizes=(8, 6), write_header=True)     # ragged last row block
    shard_matrix(X, Xh)
    Y = matrix_init.reshard_down(X, [2, 3])
    assert Y.key == "reshard(rs_in,[2, 3])" and tuple(Y.shard_sizes) == (4, 2) and tuple(Y.shape) == (20, 12)
    assert len(Y.block_idxs_exist) == 5 * 6
    assert np.array_equal(Y.numpy(), Xh)
    assert np.array_equal(Y.get_block(4, 5), Xh[16:20, 10:12])
    with pytest.raises(AssertionError):
        matrix_init.reshard_down(X, [3, 1])
    E = matrix_init.empty_result_matrix(X, np.sum, (1, 2), shape=(20, 20), shard_sizes=(8, 8))
    assert E.shape == (20, 20) and E.block_idxs_exist == []
    assert E.key == matrix_utils.hash_string(matrix_utils.hash_function(np.sum) + "rs_in" + matrix_utils.hash_args((1, 2)))
