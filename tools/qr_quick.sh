#!/bin/bash
# Round-4 quick check of the batched QR: parity tests of the QR family, then timings (x32 with T, x32 R only, x1, tree nodes, TSQR).
# Usage: tools/r04_qr_quick.sh <tag> [ENV=VALUE ...]
tag=${1:-r04q}; shift
for v in "$@"; do export "$v"; done
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "qr or tsqr or bdfac or lq or tpqrt" > $out/pytest_qr.log 2>&1
tail -3 $out/pytest_qr.log
timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
QR_SOAK_NO_T=1 timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
timeout 300 python tools/qr_soak.py 1 4 2>&1 | head -1 | tee -a $out/soak.log
timeout 300 python tools/qr_soak.py 8 4 2>&1 | head -1 | tee -a $out/soak.log
timeout 300 python tools/tpqrt_time.py 2>&1 | tee -a $out/soak.log
QR_SOAK_NO_T=1 timeout 300 python tools/tpqrt_time.py 2>&1 | tail -2 | tee -a $out/soak.log
timeout 600 python tools/bench_aux.py tsqr --leaves 256 --steps 2 2>&1 | tail -1 | tee -a $out/soak.log
