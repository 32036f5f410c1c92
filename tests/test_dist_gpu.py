"""The distributed executor with real GPU kernels: several ranks share the one GPU of the test box, payloads staged
through the host (NUMPYWREN_AMD_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).  What this covers that
tests/test_dist_gloo.py (CPU, checker backend) cannot: HIP kernels, device tiles, the allocator and stream/event
hand-off of received tiles under the owner-computes schedule."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world,port", [(2, 29641), (4, 29642)])
def test_distributed_cholesky_on_one_gpu(world, port):
    env = dict(os.environ, NUMPYWREN_AMD_DIST_BACKEND="gloo", DIST_CHECK_N="1024", DIST_CHECK_B="256",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NUMPYWREN_AMD_STORE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "dist_check.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "dist_check: PASSED" in text, text[-3000:]
