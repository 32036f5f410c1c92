#!/bin/bash
# kernel trace of a batch of 32 QR factorisations (2 calls); keeps the raw kernel-trace CSV for analysis off the box.
# Usage: tools/qr_trace_keep.sh <tag> [ENV=VALUE ...]     (QR_SOAK_NO_T=1 for the R-only form)
tag=${1:-r06t}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o qr32 -- python $GRAFT_REPO_ROOT/tools/qr_soak.py 32 2 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
csv=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/qr_chain_trace.py $csv 128 | tee $out/chain.txt
cp $csv $out/kernel_trace.csv
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/prof
