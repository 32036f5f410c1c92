#!/bin/bash
# round 5: the host-DRAM tier at HEAD (plan + write-through + batches of 8) against round 1's policy, and one run beyond HBM
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05h; mkdir -p $O
for bt in 24 12 6; do
  timeout 600 python tools/bench_aux.py spill --tiles 8 --budget-tiles $bt --steps 4 --warmup 5 > $O/spill_plan_$bt.json 2> $O/spill_plan_$bt.err
  timeout 600 python tools/bench_aux.py spill --tiles 8 --budget-tiles $bt --steps 4 --warmup 5 --lru > $O/spill_lru_$bt.json 2> $O/spill_lru_$bt.err
  for pol in plan lru; do tail -1 $O/spill_${pol}_$bt.json | python -c "import json,sys; l=json.loads(sys.stdin.read()); b=l['budget']; print($bt, '$pol', l['resident']['ms'], b['ms'], b['ms_all'][5:], b['GB_out_per_run'], b['GB_back_per_run'], b['policy'], l['last_block_row_bitwise_equal'])"; done
done
cat /sys/fs/cgroup/memory.max > $O/cgroup.txt 2>&1; cat /sys/fs/cgroup/memory/memory.limit_in_bytes >> $O/cgroup.txt 2>&1; ulimit -l >> $O/cgroup.txt; cat $O/cgroup.txt
