#!/usr/bin/env python
"""Golden vectors for the blocked QR path, made by RUNNING THE REFERENCE
(authoring container only):

    python tests/golden/make_golden_qr.py      # writes tests/golden/qr.npz

  qr.npz  -- known-answer vectors of kernels.qr_factor_triangular (reference kernels.py:107-124, LAPACK
             DTPQRT through the scipy shim of _ref_import.py) and whole runs of alg_wrappers.qr
             (reference alg_wrappers.py:67-89 + algs.py:182-234) with the reference's own instruction objects
             over the in-memory object store of make_golden.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference, installs the object store)

from numpywren import algs, compiler, kernels  # noqa: E402
from numpywren.matrix import BigMatrix  # noqa: E402
from numpywren.matrix_utils import constant_zeros  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(20260929)

    # --- kernel KATs: R factors of two tiles stacked, as the QR tree produces them -------------------------
    for n in (4, 7, 8, 32, 40, 64):
        x0 = np.triu(rng.standard_normal((n, n)))
        x1 = np.triu(rng.standard_normal((n, n)))
        v, t, r = kernels.qr_factor_triangular(x0.copy(), x1.copy())
        out[f"tri_{n}/x0"], out[f"tri_{n}/x1"] = x0, x1
        out[f"tri_{n}/v"], out[f"tri_{n}/t"], out[f"tri_{n}/r"] = np.asarray(v), np.asarray(t), np.asarray(r)
    # inputs that are not triangular: what the strictly-lower parts do
    for n in (8, 40):
        x0 = rng.standard_normal((n, n))
        x1 = rng.standard_normal((n, n))
        v, t, r = kernels.qr_factor_triangular(x0.copy(), x1.copy())
        out[f"tri_full_{n}/x0"], out[f"tri_full_{n}/x1"] = x0, x1
        out[f"tri_full_{n}/v"], out[f"tri_full_{n}/t"], out[f"tri_full_{n}/r"] = np.asarray(v), np.asarray(t), np.asarray(r)

    # --- whole algorithm ------------------------------------------------------------------------------------------
    def run_qr(tag, n, b):
        mg.STORE.clear()
        X = rng.standard_normal((n, n))
        Xb = BigMatrix(f"QR_input_{tag}", shape=X.shape, shard_sizes=(b, b), write_header=False)
        mg.shard(Xb, X)
        nbk = Xb.num_blocks(0)
        levels = max(int(np.ceil(np.log2(nbk) / np.log2(2))), 1) + 1
        mk = lambda name, shape, shards: BigMatrix(name + "_" + tag, shape=shape, shard_sizes=shards, write_header=False,
                                                   parent_fn=constant_zeros, safe=False)
        Vs = mk("Vs", (2 * n, 2 * n, levels), (b, b, 1))
        Ts = mk("Ts", (2 * n, 2 * n, levels), (b, b, 1))
        Rs = mk("Rs", (2 * n, 2 * n, levels), (b, b, 1))
        Ss = mk("Ss", (2 * n, 2 * n, 2 * n, levels * b), (b, b, 1, 1))
        prog = compiler.lpcompile_for_execution(algs.QR, inputs=["I"], outputs=["Rs"])(Xb, Vs, Ts, Rs, Ss, nbk, 0)
        order = mg.run_program(prog)
        out[f"qr_{tag}/X"] = X
        out[f"qr_{tag}/meta"] = np.asarray([n, b, nbk, levels, len(order)], dtype=np.float64)
        for i in range(nbk):
            for k in range(i, nbk):
                out[f"qr_{tag}/R_{i}_{k}"] = Rs.get_block(i, k, 0)

    run_qr("28_7", 28, 7)      # the reference's own test case (tests/test_alg_correctness.py:177-206)
    run_qr("16_8", 16, 8)
    run_qr("24_8", 24, 8)      # 3 block rows: ragged tree
    run_qr("80_40", 80, 40)    # tile wider than DTPQRT's nb = 32
    np.savez_compressed(os.path.join(HERE, "qr.npz"), **out)
    print("qr.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
