#!/usr/bin/env python
"""One Cholesky whose tiles do not fit the GPU's memory: N = tiles x 4096, the stored tiles capped by $NUMPYWREN_AMD_HBM_BUDGET
(the tier works ahead of the allocator), the rest in pinned host DRAM.  Prints one JSON line: wall time of the timed run, TFLOP/s,
bytes that left / came back, pinned bytes held, and || A - L L^T ||_F / || A ||_F over a sample of tiles.
    NUMPYWREN_AMD_HBM_BUDGET=200G python tools/beyond_hbm_chol.py --tiles 48"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from numpywren_amd import alg_wrappers, job_runner, matrix  # noqa: E402
from numpywren_amd import lambdapack as lp  # noqa: E402
from numpywren_amd.device import get_backend  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", type=int, default=48)
ap.add_argument("--tile", type=int, default=4096)
ap.add_argument("--runs", type=int, default=2)
a = ap.parse_args()
be = get_backend()
nb, b = a.tiles, a.tile
n = nb * b
X = bench.build_input(be, nb, b, f"beyond_{n}")
times, meta = [], None
for rep in range(a.runs):
    if meta is not None:
        for m in meta["outputs"] + meta["intermediates"]:
            m.free()
    program, meta = alg_wrappers.cholesky(X)
    program.config["executor"]["reclaim_intermediates"] = True
    program.program.tasks
    s0, r0 = be.spilled_bytes_total, be.restored_bytes_total
    be.synchronize()
    t0 = time.time()
    program.start()
    job_runner.lambdapack_run(program, timeout=3600)
    be.synchronize()
    times.append(time.time() - t0)
    assert program.program_status() == lp.PS.SUCCESS, program.exceptions
    out_gb, back_gb = (be.spilled_bytes_total - s0) / 1e9, (be.restored_bytes_total - r0) / 1e9
    program.free()
O = meta["outputs"][0]
# residual on a sample: the last block row's first, middle and diagonal tile, and three early ones
num = den = 0.0
for i, j in [(nb - 1, 0), (nb - 1, nb // 2), (nb - 1, nb - 1), (1, 1), (nb // 2, 1), (nb // 2, nb // 2)]:
    r = aij = X.get_tile(i, j)
    for k in range(j + 1):
        r = be.gemm(O.get_tile(i, k), O.get_tile(j, k), False, True, alpha=-1.0, beta=1.0, C=r)
    num += be.sumsq(r)
    den += be.sumsq(aij)
stored = sum(1 for m in (X, O) for _ in (m._tiles(False) or {}))
print(json.dumps({"what": f"{n}x{n} fp64 Cholesky, {b}^2 tiles, {nb}x{nb} grid ({len(program.program.tasks)} tasks), stored tiles capped at "
                          f"{os.environ.get('NUMPYWREN_AMD_HBM_BUDGET')}", "tile_GiB_total_in_and_out": round(stored * b * b * 8 / 2 ** 30, 1),
                  "device_mem_GiB": round(be.total_mem / 2 ** 30, 1), "seconds": [round(t, 2) for t in times],
                  "tflops": round(n ** 3 / 3 / times[-1] / 1e12, 2), "GB_out": round(out_gb, 1), "GB_back": round(back_gb, 1),
                  "host_link_GBps": round((out_gb + back_gb) / times[-1], 1), "pinned_GiB": round(be.pinned_bytes / 2 ** 30, 1),
                  "residual_sample": float(np.sqrt(num / den)), **matrix.RESIDENCY.stats()}))
