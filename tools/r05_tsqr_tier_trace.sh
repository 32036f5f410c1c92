#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O
NUMPYWREN_AMD_HBM_BUDGET=96G timeout 1200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python $R/tools/bench_aux.py tsqr --leaves 256 --keep-vt --steps 2 --warmup 2 > $O/line.json 2> $O/line.err
python $R/tools/spill_timeline.py $O/trace 4400 > $O/timeline.txt 2>&1
find $O/trace -name "*.csv" -size +1M -delete
cat $O/timeline.txt | cut -c1-500; tail -1 $O/line.json | cut -c1-300
