#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
for B in 32 64; do
  NUMPYWREN_AMD_QR_BATCH_MAX=64 timeout 600 python tools/bench_aux.py tsqr --leaves 256 --batch $B --keep-vt --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-300 | tee -a $O/tsqr.txt
  NUMPYWREN_AMD_QR_BATCH_MAX=64 timeout 600 python tools/bench_aux.py tsqr --leaves 256 --batch $B --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-300 | tee -a $O/tsqr.txt
done
NUMPYWREN_AMD_QR_BATCH_MAX=64 python - <<'PY' 2>&1 | tee -a $O/tsqr.txt
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from numpywren_amd.device import get_backend
be = get_backend()
n = 4096
for cnt in (32, 64):
    pairs = []
    for z in range(cnt):
        a = be.tri(be.fill_random((n, n), 100 + z), "U"); c = be.tri(be.fill_random((n, n), 300 + z), "U")
        pairs.append((a, c))
    for want_t in (True, False):
        ts = []
        for rep in range(4):
            be.synchronize(); t0 = time.time()
            out = be.tpqrt_batched(pairs, want_t=want_t, want_v=want_t)
            be.synchronize(); ts.append(1e3 * (time.time() - t0)); del out
        print("tpqrt x%d want_t=%s: min %.1f ms (%.2f per node)" % (cnt, want_t, min(ts), min(ts) / cnt))
    del pairs
print("handoff timeouts", be.qr_handoff_timeouts())
PY
