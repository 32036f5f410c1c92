#!/bin/bash
# kernel trace of a batch of 32 QR factorisations (2 calls), analysed by tools/qr_chain_trace.py.  Usage: tools/r04_qr_trace.sh <tag> [env...]
tag=${1:-r04b}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o qr32 -- python $GRAFT_REPO_ROOT/tools/qr_soak.py 32 2 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
csv=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/qr_chain_trace.py $csv 128 | tee $out/chain.txt
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/prof
