#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05r; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_tile4096_gpu.py tests/test_algorithms_gpu.py -m gpu -x -q -k "gemm or Gemm or sgemm or fp32 or config4" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for v in 0 1 0 1; do echo "NPW_SGEMM_BK32=$v"; NPW_SGEMM_BK32=$v python tools/sgemm_time.py 2>&1 | tail -4; done | tee $O/sgemm.txt
for v in 0 1; do NPW_SGEMM_BK32=$v timeout 300 python bench.py --workload gemm32 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bk32=$v', d['value'], d['ms_per_step'], d['config'].get('fused_tflops'))"; done | tee -a $O/sgemm.txt
