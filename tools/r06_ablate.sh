#!/bin/bash
out=gpurun_out/r06h; mkdir -p $out
for a in 0 1 2 4 8 16 3 11 27 31; do
  echo "== ablate $a" | tee -a $out/ablate.log
  NPW_QR_ABLATE=$a QR_SOAK_NO_T=1 timeout 300 python tools/qr_soak.py 32 5 2>&1 | head -1 | tee -a $out/ablate.log
done
for a in 0 4 8 12; do
  echo "== with T, ablate $a" | tee -a $out/ablate.log
  NPW_QR_ABLATE=$a timeout 300 python tools/qr_soak.py 32 5 2>&1 | head -1 | tee -a $out/ablate.log
done
