#!/bin/bash
# one run whose tiles do not fit HBM: 196608^2 fp64 Cholesky, 48 x 48 tiles of 4096^2 -- 1176 input tiles (147 GiB) and as many
# factor tiles: 294 GiB on a 288 GB device -- with the stored tiles capped at 200 GiB (the tier works ahead of the allocator
# instead of behind its failures).  The box's cgroup allows 300 GiB of host memory; the tier refuses beyond 225.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05hbm; mkdir -p $O
NUMPYWREN_AMD_HBM_BUDGET=200G timeout 1500 python tools/beyond_hbm_chol.py --tiles 48 > $O/chol48.json 2> $O/chol48.err
echo "rc=$?"; tail -1 $O/chol48.json; tail -3 $O/chol48.err
