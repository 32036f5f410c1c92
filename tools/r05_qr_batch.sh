#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
for c in 32 48 64; do
  NUMPYWREN_AMD_QR_BATCH_MAX=64 timeout 300 python tools/qr_soak.py $c 6 2>&1 | grep -v "^all" | tee -a $O/soak.txt
  QR_SOAK_NO_T=1 NUMPYWREN_AMD_QR_BATCH_MAX=64 timeout 300 python tools/qr_soak.py $c 6 2>&1 | grep -v "^all" | tee -a $O/soak.txt
done
