#!/bin/bash
# Same-box A/B of the batched QR under two environments (alternating), R only / with T / every launch on one stream.
# Usage: tools/r06_ab.sh <tag> "<ENV=VAL ...>" "<ENV=VAL ...>" [reps]
tag=${1:-r06ab}; A="$2"; B="$3"; reps=${4:-6}
out=gpurun_out/$tag
mkdir -p $out
for round in 1 2; do
  for cfg in "$A" "$B"; do
    echo "== [$cfg]" | tee -a $out/ab.log
    env $cfg QR_SOAK_NO_T=1 timeout 300 python tools/qr_soak.py 32 $reps 2>&1 | head -1 | tee -a $out/ab.log
    env $cfg timeout 300 python tools/qr_soak.py 32 $reps 2>&1 | head -1 | tee -a $out/ab.log
    if [ $round = 1 ]; then
      env $cfg QR_SOAK_NO_T=1 NPW_QR_SERIAL=1 timeout 300 python tools/qr_soak.py 32 3 2>&1 | head -1 | tee -a $out/ab.log
    fi
  done
done
