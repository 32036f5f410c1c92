#!/bin/bash
# kernel trace of one run of a tools/bench_aux.py program (after one warm-up run): per-kernel totals and GPU busy time of the
# last <span> ms.     tools/r04_prog_trace.sh <tag> <span_ms> <bench_aux args...>
tag=$1; span=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o prog -- python $GRAFT_REPO_ROOT/tools/bench_aux.py "$@" --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $out/run.log | cut -c1-300
csv=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_agg.py $csv $span | tee $out/agg.txt
rm -rf $out/prof
