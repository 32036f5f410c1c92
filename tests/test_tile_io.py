"""Host/disk tile format: numpywren_amd.tile_io against objects WRITTEN BY THE REFERENCE
(tests/golden/objects.npz, made by tests/golden/make_golden_objects.py) and round trips through both store tiers."""
import base64
import io
import json
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN
from numpywren_amd import tile_io
from numpywren_amd.matrix import BigMatrix
from numpywren_amd.matrix_init import shard_matrix
from numpywren_amd.matrix_utils import constant_zeros

OBJ = np.load(os.path.join(GOLDEN, "objects.npz"))
CASES = ["objA", "objB"]


def _materialise(name, root):
    """write the reference's objects of one matrix as files; returns its meta dict"""
    meta = json.loads(bytes(OBJ[f"{name}|meta"]).decode())
    for k in OBJ.files:
        mat, _, key = k.partition("|")
        if mat != name or key in ("dense", "meta"):
            continue
        path = os.path.join(root, meta["bucket"], *key.split("/"))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(bytes(OBJ[k]))
    return meta


def _check_import(name, tmp_path):
    meta = _materialise(name, str(tmp_path))
    m = tile_io.import_matrix(str(tmp_path), meta["key"], bucket=meta["bucket"], prefix=meta["prefix"])
    dense = OBJ[f"{name}|dense"]
    assert m.shape == dense.shape and np.dtype(m.dtype) == dense.dtype
    got = m.numpy()
    assert got.dtype == dense.dtype and np.array_equal(got, dense)
    return m, meta, dense


@pytest.mark.parametrize("name", CASES)
def test_import_reference_written_objects(name, tmp_path, host_store):
    m, meta, dense = _check_import(name, tmp_path)
    assert m.key_base == meta["key_base"]
    # a second handle with only the key sees the header the import registered (reference matrix.py:96-104)
    again = BigMatrix(meta["key"], bucket=meta["bucket"], prefix=meta["prefix"])
    assert again.shape == dense.shape and tuple(again.shard_sizes) == tuple(m.shard_sizes)


@pytest.mark.parametrize("name", CASES)
def test_export_matches_reference_objects(name, tmp_path, host_store):
    m, meta, dense = _check_import(name, tmp_path / "in")
    out = tmp_path / "out"
    n = tile_io.export_matrix(m, str(out))
    ref_keys = sorted(k.partition("|")[2] for k in OBJ.files if k.startswith(name + "|") and k.split("|")[1] not in ("dense", "meta"))
    assert n == len(ref_keys) - 1
    for key in ref_keys:
        with open(os.path.join(str(out), meta["bucket"], *key.split("/")), "rb") as f:
            mine = f.read()
        ref = bytes(OBJ[f"{name}|{key}"])
        if key.endswith("header"):
            a, b = json.loads(mine.decode()), json.loads(ref.decode())
            assert a["shape"] == b["shape"] and a["shard_sizes"] == b["shard_sizes"]
            assert np.dtype(tile_io.decode_dtype(a["dtype"])) == np.dtype(tile_io.decode_dtype(b["dtype"]))
        else:
            x, y = np.load(io.BytesIO(mine)), np.load(io.BytesIO(ref))
            assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y)


def test_spill_restore_sparse_matrix(tmp_path, host_store):
    """safe=False matrix with tiles beyond its nominal shape (the TSQR / QR trees), only some tiles present"""
    R = BigMatrix("spill_R", shape=(16, 8), shard_sizes=(8, 8), parent_fn=constant_zeros, safe=False)
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((8, 8)), rng.standard_normal((8, 8))
    R.put_block(a, 0, 0)
    R.put_block(b, 3, 0)                      # outside (16, 8)
    assert tile_io.spill(R, str(tmp_path)) == 2
    assert not R.get_block(0, 0).any()        # gone: parent_fn zeros
    assert tile_io.restore(R, str(tmp_path)) == 2
    assert np.array_equal(R.get_block(0, 0), a) and np.array_equal(R.get_block(3, 0), b)
    assert not R.get_block(1, 0).any()


def test_header_dtype_pickle_is_restricted():
    evil = base64.b64encode(pickle.dumps(os.getcwd)).decode()
    with pytest.raises(pickle.UnpicklingError):
        tile_io.decode_dtype(evil)
    assert tile_io.decode_dtype(tile_io.encode_dtype(np.float32)) is np.float32
    assert tile_io.decode_dtype(tile_io.encode_dtype(np.dtype("float64"))) == np.dtype("float64")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_import_into_hbm(name, tmp_path, hbm_store):
    from numpywren_amd.device import DeviceTile
    m, meta, dense = _check_import(name, tmp_path)
    assert isinstance(m.get_tile(*([0] * len(m.shape))), DeviceTile)


@pytest.mark.gpu
def test_spill_restore_hbm_bit_exact(tmp_path, hbm_store):
    rng = np.random.default_rng(8)
    X = rng.standard_normal((96, 64))
    M = BigMatrix("spill_hbm", shape=X.shape, shard_sizes=(32, 32))
    shard_matrix(M, X)
    assert tile_io.spill(M, str(tmp_path)) == 6
    with pytest.raises(Exception):
        M.get_block(0, 0)                     # no parent_fn: the tile is really gone
    assert tile_io.restore(M, str(tmp_path)) == 6
    assert np.array_equal(M.numpy(), X)
