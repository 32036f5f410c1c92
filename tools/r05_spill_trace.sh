#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05d; mkdir -p $O
cd /tmp
for bt in 12 24; do
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_plan_$bt -- python /root/repo/tools/bench_aux.py spill --tiles 8 --budget-tiles $bt --steps 4 --warmup 3 > $O/spill_plan_$bt.json 2> $O/spill_plan_$bt.err
python /root/repo/tools/spill_timeline.py $O/trace_plan_$bt 600 > $O/timeline_plan_$bt.txt 2>&1
find $O/trace_plan_$bt -name "*.csv" -size +1M -delete
done
for bt in 12 24; do cat $O/timeline_plan_$bt.txt; tail -1 $O/spill_plan_$bt.json | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(l[\"budget\"])"; done
