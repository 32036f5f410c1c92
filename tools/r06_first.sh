#!/bin/bash
# Round 6, first contact: the new surface tests on the GPU, then the batched-QR baselines of this box (x32 with T / R only, tree nodes).
out=gpurun_out/r06a
mkdir -p $out
timeout 900 python -m pytest tests/test_kernel_surface.py tests/test_checkpoint.py tests/test_cabi.py -m gpu -x -q > $out/pytest_new.log 2>&1
tail -3 $out/pytest_new.log
timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
QR_SOAK_NO_T=1 timeout 300 python tools/qr_soak.py 32 4 2>&1 | head -1 | tee -a $out/soak.log
QR_SOAK_NO_T=1 NPW_QR_SERIAL=1 timeout 300 python tools/qr_soak.py 32 3 2>&1 | head -1 | tee -a $out/soak.log
timeout 300 python tools/tpqrt_time.py 2>&1 | tee -a $out/soak.log
QR_SOAK_NO_T=1 timeout 300 python tools/tpqrt_time.py 2>&1 | tail -2 | tee -a $out/soak.log
