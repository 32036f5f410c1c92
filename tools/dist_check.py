#!/usr/bin/env python
"""Distributed Cholesky / GEMM / TSQR on the ranks of a torchrun launch, checked against NumPy on rank 0.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dist_check.py

With NUMPYWREN_AMD_DIST_BACKEND=gloo the payloads go through the host, so several ranks can share one GPU
(tests/test_dist_gpu.py does that on the 1-GPU test box); without it the exchange is RCCL point-to-point."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("NUMPYWREN_AMD_STORE", None)

from numpywren_amd import alg_wrappers, dist  # noqa: E402
from numpywren_amd import lambdapack as lp  # noqa: E402
from numpywren_amd.matrix import BigMatrix  # noqa: E402


def main():
    comm = dist.init_process_group()
    rank, world = comm.rank, comm.world
    n, b = int(os.environ.get("DIST_CHECK_N", "1024")), int(os.environ.get("DIST_CHECK_B", "256"))
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, n))
    A = G @ G.T + n * np.eye(n)
    X = BigMatrix("dist_check_A", shape=A.shape, shard_sizes=(b, b))
    nb = n // b
    for i in range(nb):          # every rank holds only the tiles it owns
        for j in range(nb):
            if comm.owner("dist_check_A", (i, j)) == rank:
                X.put_block(A[i * b:(i + 1) * b, j * b:(j + 1) * b], i, j)
    program, meta = alg_wrappers.cholesky(X)
    program.start()
    res = dist.lambdapack_run_distributed(program, comm)
    ok = program.program_status() == lp.PS.SUCCESS
    L = dist.gather_matrix(meta["outputs"][0], comm)
    err = None
    if rank == 0:
        Lr = np.linalg.cholesky(A)
        err = float(np.abs(np.tril(L) - Lr).max() / np.abs(Lr).max())
    mine = len(res["executed_messages"])
    total = comm.max_over_ranks(0)  # sync point
    counts = [None] * world
    comm.dist.all_gather_object(counts, (mine, res["bytes_sent"], res["transfers"]))
    if rank == 0:
        ntasks = nb * (nb + 1) * (nb + 2) // 6
        print(f"dist_check: world {world} backend {comm.backend} n {n} b {b}: status {'SUCCESS' if ok else 'FAIL'} "
              f"rel err {err:.2e} tasks per rank {[c[0] for c in counts]} (sum {sum(c[0] for c in counts)} of {ntasks}) "
              f"bytes sent {[c[1] for c in counts]}")
        good = ok and err < 1e-12 and sum(c[0] for c in counts) == ntasks and all(c[0] > 0 for c in counts)
        print("dist_check: PASSED" if good else "dist_check: FAILED")
    comm.shutdown()
    del total


if __name__ == "__main__":
    main()
