#!/bin/bash
# A/B of the software-pipelined GEMM epilogue: batched QR (R only / with T / serial) and the headline Cholesky line.
out=gpurun_out/r06c
mkdir -p $out
bash tools/r06_ab.sh r06c "NPW_GEMM_EPI_PIPE=0" "NPW_GEMM_EPI_PIPE=1" 6
for round in 1 2 3; do
  for v in 0 1; do
    echo "== chol EPI_PIPE=$v" | tee -a $out/chol.log
    NPW_GEMM_EPI_PIPE=$v timeout 600 python bench.py --no-cpu-baseline --steps 25 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline']['avg_ms'], l['kernel_ms'], l.get('north_star',{}).get('tflops'))" | tee -a $out/chol.log
  done
done
