#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05g; mkdir -p $O
for bt in 12 24; do for B in 4 8 32; do for P in 0 2; do
timeout 600 python tools/bench_aux.py spill --tiles 8 --budget-tiles $bt --steps 4 --warmup 5 --batch $B --prefetch $P > $O/spill_${bt}_b${B}_p$P.json 2> $O/spill_${bt}_b${B}_p$P.err
tail -1 $O/spill_${bt}_b${B}_p$P.json | python -c "import json,sys; l=json.loads(sys.stdin.read()); print($bt, $B, $P, l['resident']['ms'], l['budget']['ms'], l['budget']['ms_all'][4:], l['budget']['GB_out_per_run'], l['budget']['GB_back_per_run'])"
done; done; done
