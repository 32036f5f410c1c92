#!/bin/bash
# round-6 soak at HEAD: the latency-bound kernels repeated with every result checked, then 300 consecutive default steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06soak; mkdir -p $O
timeout 900 python tools/soak.py > $O/soak.log 2>&1; tail -3 $O/soak.log
timeout 600 python tools/chol_stress.py 2>&1 | tail -1 | tee $O/chol_stress.log
timeout 900 python tools/qr_soak.py 32 60 2>&1 | grep -v "^all" | tee $O/qr32_60.log
QR_SOAK_NO_T=1 timeout 900 python tools/qr_soak.py 32 60 2>&1 | grep -v "^all" | tee $O/qr32r_60.log
timeout 900 python tools/qr_soak.py 1 40 2>&1 | grep -v "^all" | tee $O/qr1_40.log
timeout 900 python bench.py --steps 300 --warmup 2 --no-cpu-baseline --no-north-star > $O/bench300.json 2> $O/bench300.err
python -c "
import json, numpy as np; d=json.load(open('$O/bench300.json')); s=np.array(d['step_ms']); print('300 steps:', d['value'], d['ms_per_step'], 'median', d['ms_per_step_median'], 'min', s.min(), 'max', s.max(), 'outliers', d['outliers'], 'residual', d['config']['residual_all_tiles'])" | tee $O/bench300.txt
