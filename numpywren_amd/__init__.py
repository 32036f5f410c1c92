"""numpywren_amd -- an MI355X-native LambdaPACK tile executor with numpywren's operator surface.

    from numpywren_amd.matrix import BigMatrix
    from numpywren_amd.matrix_init import shard_matrix
    from numpywren_amd import alg_wrappers, job_runner, kernels

Tiles live in HBM, tile kernels are hand-written HIP (MFMA fp64/fp32) behind the C-ABI of
libnpw_hip.so, the LambdaPACK DAG is expanded once and driven on HIP streams.
"""
import logging
import os

from .version import __version__  # noqa: F401

logger = logging.getLogger('numpywren')
TMP_DIR = os.environ.get("NUMPYWREN_AMD_TMP", "/tmp/")

from . import config, exceptions, utils  # noqa: E402,F401
from . import _ffi  # noqa: E402,F401
from . import device  # noqa: E402,F401
from . import matrix, matrix_init, matrix_utils  # noqa: E402,F401
from . import kernels  # noqa: E402,F401
from . import frontend, compiler, lambdapack, job_runner, algs, alg_wrappers  # noqa: E402,F401
from .matrix import BigMatrix  # noqa: E402,F401
