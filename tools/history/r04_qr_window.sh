#!/bin/bash
# what runs beside the panel chain in a window of a traced batch of 32.  Usage: tools/r04_qr_window.sh <tag> <from_ms> <to_ms> [ENV=VALUE ...]
tag=$1; lo=$2; hi=$3; shift 3
for v in "$@"; do export "$v"; done
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o qr32 -- python $GRAFT_REPO_ROOT/tools/qr_soak.py 32 2 > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/qr_window_trace.py $(find $out/prof -name "*kernel_trace.csv" | head -1) $lo $hi | tee $out/window.txt
rm -rf $out/prof
