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


def _owned_scatter(M, A, comm, rank):
    """every rank holds only the tiles it owns"""
    b0, b1 = M.shard_sizes
    for i in range(M.num_blocks(0)):
        for j in range(M.num_blocks(1)):
            if comm.owner(M.key, (i, j)) == rank:
                M.put_block(A[i * b0:(i + 1) * b0, j * b1:(j + 1) * b1], i, j)


def _run(program, comm):
    program.start()
    res = dist.lambdapack_run_distributed(program, comm, pipeline_width=int(os.environ.get("DIST_CHECK_STREAMS", "3")))
    return program.program_status() == lp.PS.SUCCESS, res


def main():
    comm = dist.init_process_group()
    rank, world = comm.rank, comm.world
    n, b = int(os.environ.get("DIST_CHECK_N", "1024")), int(os.environ.get("DIST_CHECK_B", "256"))
    rng = np.random.default_rng(5)
    nb = n // b
    report = []

    # --- Cholesky ------------------------------------------------------------------------------------------------
    G = rng.standard_normal((n, n))
    A = G @ G.T + n * np.eye(n)
    X = BigMatrix("dist_check_A", shape=A.shape, shard_sizes=(b, b))
    _owned_scatter(X, A, comm, rank)
    program, meta = alg_wrappers.cholesky(X)
    ok, res = _run(program, comm)
    L = dist.gather_matrix(meta["outputs"][0], comm)
    counts = [None] * world
    comm.dist.all_gather_object(counts, (len(res["executed_messages"]), res["bytes_sent"]))
    if rank == 0:
        Lr = np.linalg.cholesky(A)
        err = float(np.abs(np.tril(L) - Lr).max() / np.abs(Lr).max())
        ntasks = nb * (nb + 1) * (nb + 2) // 6
        good = ok and err < 1e-12 and sum(c[0] for c in counts) == ntasks and all(c[0] > 0 for c in counts)
        report.append(("cholesky", good, f"rel err {err:.2e} tasks per rank {[c[0] for c in counts]} of {ntasks} "
                                         f"bytes sent {[c[1] for c in counts]}"))

    # --- GEMM (fp64) ---------------------------------------------------------------------------------------------
    Ah, Bh = rng.standard_normal((n, n)), rng.standard_normal((n, n))
    Am = BigMatrix("dist_check_GA", shape=Ah.shape, shard_sizes=(b, b))
    Bm = BigMatrix("dist_check_GB", shape=Bh.shape, shard_sizes=(b, b))
    _owned_scatter(Am, Ah, comm, rank)
    _owned_scatter(Bm, Bh, comm, rank)
    program, meta = alg_wrappers.gemm(Am, Bm)
    ok, res = _run(program, comm)
    C = dist.gather_matrix(meta["outputs"][0], comm)
    if rank == 0:
        ref = Ah @ Bh
        err = float(np.abs(C - ref).max() / np.abs(ref).max())
        report.append(("gemm", ok and err < 1e-12, f"rel err {err:.2e}"))

    # --- TSQR ----------------------------------------------------------------------------------------------------
    # leaves in contiguous chunks per rank (local sub-trees run as batches, only log2(world) R factors travel)
    leaves = 4 * world
    Th = rng.standard_normal((leaves * b, b))
    Tm = BigMatrix("dist_check_T", shape=Th.shape, shard_sizes=(b, b))
    comm.ownership = dist.tsqr_ownership(world, leaves)
    for j in range(leaves):
        if comm.owner("A", (j, 0)) == rank:
            Tm.put_block(np.ascontiguousarray(Th[j * b:(j + 1) * b]), j, 0)
    program, meta = alg_wrappers.tsqr(Tm)
    ok, res = _run(program, comm)
    comm.ownership = None
    levels = int(np.ceil(np.log2(leaves)))
    Rm = meta["outputs"][0]
    have = Rm.get_block(levels, 0) if Rm.tile_exists(levels, 0) else None
    parts = [None] * world
    comm.dist.all_gather_object(parts, have)
    if rank == 0:
        R = next(p for p in parts if p is not None)
        Rr = np.linalg.qr(Th)[1]
        err = float(np.abs(np.abs(R) - np.abs(Rr)).max() / np.abs(Rr).max())
        report.append(("tsqr", ok and err < 1e-11, f"|R| rel err {err:.2e}"))

    if rank == 0:
        for name, good, text in report:
            print(f"dist_check[{name}]: world {world} backend {comm.backend} n {n} b {b}: {'ok' if good else 'BAD'} {text}")
        print("dist_check: PASSED" if all(g for _, g, _ in report) else "dist_check: FAILED")
    comm.shutdown()


if __name__ == "__main__":
    main()
