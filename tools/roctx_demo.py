#!/usr/bin/env python
"""Developer aid: a 4 x 4-tile Cholesky with executor.roctx_ranges on -- run it under `rocprofv3 --marker-trace --kernel-trace`
to see the tasks' ranges beside the kernels."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from numpywren_amd import alg_wrappers, job_runner
from numpywren_amd import lambdapack as lp
from numpywren_amd.matrix import BigMatrix
from numpywren_amd.matrix_init import shard_matrix

rng = np.random.default_rng(1)
n, b = 4096, 1024
G = rng.standard_normal((n, 64))
A = G @ G.T + n * np.eye(n)
X = BigMatrix("roctx_demo", shape=A.shape, shard_sizes=(b, b), write_header=True)
shard_matrix(X, A)
program, meta = alg_wrappers.cholesky(X)
program.config["executor"]["roctx_ranges"] = True
program.start()
job_runner.lambdapack_run(program, timeout=120)
program.wait()
assert program.program_status() == lp.PS.SUCCESS, program.exceptions
print("ok")
