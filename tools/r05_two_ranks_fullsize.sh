#!/bin/bash
# bench.py --gpus 2 at the real sizes (65536^2, 4096^2 tiles) with no launcher: two ranks sharing this box's one GPU
# (host-staged payloads, allowed knowingly) -- the whole N > 1 flow incl. rank 0's anchor, except RCCL itself
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
NUMPYWREN_AMD_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 1 --warmup 1 > $O/two_ranks.json 2> $O/two_ranks.err
echo "rc=$?"; tail -c 2500 $O/two_ranks.json; tail -5 $O/two_ranks.err
