#!/bin/bash
# Round 6, final state at HEAD: the whole GPU suite, then the three bench lines of this box.
out=gpurun_out/r06final; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
python bench.py > $out/bench_line.json 2> $out/bench.err; cut -c1-200 $out/bench_line.json
python bench.py --workload tsqr --steps 3 --warmup 1 > $out/tsqr_line.json 2> $out/tsqr.err; cut -c1-200 $out/tsqr_line.json
python bench.py --workload gemm32 --steps 3 --warmup 1 > $out/gemm32_line.json 2> $out/gemm32.err; cut -c1-200 $out/gemm32_line.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
