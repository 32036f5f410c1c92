#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05x; mkdir -p $O
NUMPYWREN_AMD_HBM_BUDGET=96G NPW_BENCH_DEBUG=1 timeout 900 python bench.py --workload tsqr --leaves 256 --steps 3 --warmup 2 > $O/tsqr256_keepvt_96G.json 2> $O/tsqr256_keepvt_96G.err
python -c "
import json; d=json.loads(open('$O/tsqr256_keepvt_96G.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'])"
grep bench $O/tsqr256_keepvt_96G.err | tail -4
