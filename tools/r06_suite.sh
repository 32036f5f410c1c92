#!/bin/bash
# Round 6: the new / changed GPU tests, then the configs[3] / configs[4] lines with their new roofline + cpu_baseline objects.
out=gpurun_out/r06f
mkdir -p $out
timeout 1500 python -m pytest tests/test_bench_contract.py tests/test_kernel_surface.py tests/test_checkpoint.py tests/test_dist_gpu.py -m gpu -x -q > $out/pytest_a.log 2>&1; tail -3 $out/pytest_a.log
timeout 900 python -m pytest tests/test_tile4096_gpu.py -m gpu -x -q -k "config2" > $out/pytest_b.log 2>&1; tail -3 $out/pytest_b.log
timeout 900 python bench.py --workload tsqr --steps 2 --warmup 1 2> $out/tsqr.err | tail -1 > $out/tsqr_line.json; head -c 1500 $out/tsqr_line.json; echo
timeout 900 python bench.py --workload gemm32 --steps 3 --warmup 1 2> $out/gemm32.err | tail -1 > $out/gemm32_line.json; head -c 1500 $out/gemm32_line.json; echo
