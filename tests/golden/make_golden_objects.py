#!/usr/bin/env python
"""Reference-written tile objects, made by RUNNING THE REFERENCE (authoring container only):

    python tests/golden/make_golden_objects.py      # writes tests/golden/objects.npz

For two small matrices the reference's own put_block path (numpywren/matrix.py:312-361 ->
__save_matrix_to_s3__ 519-533: np.save bytes under the key of 457-464) fills the in-memory object store of
make_golden.py; the header object is the JSON of __write_header__ (535-546) with the reference's
__encode_dtype__ (548-551).  objects.npz maps "<matrix>|<object key>" -> the object's raw bytes (uint8) and keeps
the dense arrays they came from.  Data only: no reference source travels.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from numpywren.matrix import BigMatrix  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(20260930)
    cases = [("objA", (5, 7), (2, 3), np.float64), ("objB", (4, 4, 2), (2, 2, 1), np.float32)]
    for name, shape, shards, dtype in cases:
        mg.STORE.clear()
        X = rng.standard_normal(shape).astype(dtype)
        M = BigMatrix(name, shape=shape, shard_sizes=shards, write_header=False, dtype=dtype)
        mg.shard(M, X)
        header = {"shape": M.shape, "shard_sizes": M.shard_sizes, "dtype": M.__encode_dtype__(M.dtype)}
        out[f"{name}|dense"] = X
        out[f"{name}|meta"] = np.frombuffer(json.dumps({"bucket": M.bucket, "key_base": M.key_base, "prefix": M.prefix,
                                                        "key": M.key}).encode(), dtype=np.uint8)
        out[f"{name}|{os.path.join(M.key_base, 'header')}"] = np.frombuffer(json.dumps(header).encode(), dtype=np.uint8)
        for (bucket, key), raw in mg.STORE.items():
            out[f"{name}|{key}"] = np.frombuffer(raw, dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "objects.npz"), **out)
    print("objects.npz:", len(out), "entries")


if __name__ == "__main__":
    main()
