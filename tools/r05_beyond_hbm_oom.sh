#!/bin/bash
# the same beyond-HBM problem with NO budget: the tier only acts when a device allocation fails (the allocator's out-of-memory handler)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05hbm; mkdir -p $O
timeout 1500 python tools/beyond_hbm_chol.py --tiles 48 --runs 1 > $O/chol48_oom.json 2> $O/chol48_oom.err
echo "rc=$?"; tail -1 $O/chol48_oom.json; tail -3 $O/chol48_oom.err
