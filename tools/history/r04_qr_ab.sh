#!/bin/bash
# Round-4 A/B of the batched QR (superblock width, far stream's reserved CUs, R-only): run on the GPU box through gpurun,
# logs under gpurun_out/$1.  Usage: tools/r04_qr_ab.sh <tag> [quick]
tag=${1:-r04a}
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -k "qr or tsqr or bdfac or lq or tpqrt" > $out/pytest_qr.log 2>&1
tail -3 $out/pytest_qr.log
for sb in 128 256 512 1024; do
  echo "== NPW_QR_SB=$sb" | tee -a $out/soak.log
  NPW_QR_SB=$sb timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
  NPW_QR_SB=$sb QR_SOAK_NO_T=1 timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
  NPW_QR_SB=$sb timeout 300 python tools/qr_soak.py 1 4 2>&1 | head -1 | tee -a $out/soak.log
done
for res in 32 64; do
  echo "== NPW_QR_SB=512 NPW_QR_FAR_RESERVE_CUS=$res" | tee -a $out/soak.log
  NPW_QR_FAR_RESERVE_CUS=$res timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
  NPW_QR_FAR_RESERVE_CUS=$res QR_SOAK_NO_T=1 timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
done
for sb in 128 512; do
  echo "== tpqrt NPW_QR_SB=$sb" | tee -a $out/soak.log
  NPW_QR_SB=$sb timeout 300 python tools/tpqrt_time.py 2>&1 | tee -a $out/soak.log
  NPW_QR_SB=$sb QR_SOAK_NO_T=1 timeout 300 python tools/tpqrt_time.py 2>&1 | tail -2 | tee -a $out/soak.log
done
echo "== tsqr 256 leaves R only, SB=512" | tee -a $out/soak.log
timeout 600 python tools/bench_aux.py tsqr --leaves 256 --steps 2 2>&1 | tail -1 | tee -a $out/soak.log
