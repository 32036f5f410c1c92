#!/bin/bash
# kernel trace of a batch of 32 stacked-triangle factorisations (TSQR tree nodes), R only, analysed by tools/qr_chain_trace.py;
# once as it runs and once with every launch on one stream.   Usage: tools/r04_tri_trace.sh <tag>
tag=${1:-r04tri}
out=gpurun_out/$tag
mkdir -p $out
export QR_SOAK_NO_T=${QR_SOAK_NO_T:-1}
for serial in 0 1; do
  cd /tmp && export TMPDIR=/tmp
  NPW_QR_SERIAL=$serial rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof$serial -o tri -- python $GRAFT_REPO_ROOT/tools/tpqrt_time.py > $GRAFT_REPO_ROOT/$out/run$serial.log 2>&1
  cd $GRAFT_REPO_ROOT
  csv=$(find $out/prof$serial -name "*kernel_trace.csv" | head -1)
  python tools/qr_chain_trace.py $csv 128 > $out/chain_serial$serial.txt
  rm -rf $out/prof$serial
done
tail -4 $out/run0.log
head -40 $out/chain_serial0.txt
