#!/bin/bash
# configs[3]'s 256-leaf TSQR keeping R, V, T (256 GiB of stored tiles) with the stored tiles capped at 96 GiB: 160 GiB of
# factors leave for pinned host DRAM (the box's cgroup allows 300 GiB of host memory: the 512-leaf problem -- 311 GiB to the
# host -- does not fit it; the tier refuses beyond 3/4 of the limit instead of getting the box killed)
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05t; mkdir -p $O
timeout 600 python -m pytest tests/test_residency.py -m gpu -x -q 2>&1 | tail -2
NUMPYWREN_AMD_HBM_BUDGET=96G NPW_BENCH_DEBUG=1 timeout 900 python bench.py --workload tsqr --leaves 256 --steps 3 --warmup 2 > $O/tsqr256_keepvt_96G.json 2> $O/tsqr256_keepvt_96G.err
python -c "
import json; d=json.loads(open('$O/tsqr256_keepvt_96G.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'])"
grep bench $O/tsqr256_keepvt_96G.err | tail -3
for bt in 24 12; do timeout 600 python tools/bench_aux.py spill --tiles 8 --budget-tiles $bt --steps 3 --warmup 4 | python -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['budget']; print($bt, l['resident']['ms'], b['ms'], b['GB_out_per_run'], b['GB_back_per_run'], b['written_through'], l['last_block_row_bitwise_equal'])"; done
