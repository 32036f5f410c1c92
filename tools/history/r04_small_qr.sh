#!/bin/bash
# Round-4 check of the latency-bound (single / small-batch) QR forms: parity tests of the QR family, then timings with the
# one-launch near update on and off.   Usage: tools/r04_small_qr.sh <tag> [ENV=VALUE ...]
tag=${1:-r04s}; shift
for v in "$@"; do export "$v"; done
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "qr or tsqr or bdfac or lq or tpqrt" > $out/pytest_qr.log 2>&1
tail -3 $out/pytest_qr.log
for f in 1 0; do
  echo "== NPW_QR_NEAR_FUSED=$f" | tee -a $out/soak.log
  export NPW_QR_NEAR_FUSED=$f
  timeout 300 python tools/qr_soak.py 1 6 2>&1 | head -1 | tee -a $out/soak.log
  timeout 300 python tools/qr_soak.py 1 4 8192 4096 2>&1 | head -1 | tee -a $out/soak.log
  timeout 300 python tools/qr_soak.py 4 4 2>&1 | head -1 | tee -a $out/soak.log
  timeout 300 python tools/tpqrt_time.py 2>&1 | head -3 | tee -a $out/soak.log
  timeout 600 python tools/bench_aux.py bdfac --tiles 4 2>&1 | tail -1 | tee -a $out/soak.log
  timeout 600 python tools/bench_aux.py qr --tiles 4 2>&1 | tail -1 | tee -a $out/soak.log
done
