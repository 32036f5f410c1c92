#!/bin/bash
# bench.py --gpus 2 at the real sizes with no launcher: two ranks sharing this box's one GPU (host-staged payloads, allowed
# knowingly) -- the whole N > 1 flow incl. rank 0's anchor, except RCCL itself.  $1 = workload (chol | gemm32)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=${1:-chol}; O=gpurun_out/r05p; mkdir -p $O
NUMPYWREN_AMD_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --workload $W --steps 1 --warmup 1 > $O/two_ranks_$W.json 2> $O/two_ranks_$W.err
echo "rc=$?"; python -c "
import json; l=json.loads([x for x in open('$O/two_ranks_$W.json').read().splitlines() if x.startswith('{')][-1]); c=l['config']
print(l['value'], l['ms_per_step'], l['n_gpus'], c['transport'], c['ranks_joined'], c['one_gpu_anchor'])"
