#!/bin/bash
# MFMA-busy share by kernel (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)) for a batch of 32 factorisations
# (with T) and for the three bench workloads' dominant kernels: one PMC pass each (--kernel-trace --pmc only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O
QR_RUN_REPS=2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/q -o pmc -- python $R/tools/qr_run.py 32 > /dev/null 2>&1
python $R/tools/pmc_table.py $(find $O/q -name "*counter_collection.csv") > $O/mfma_busy_qr32.txt; rm -rf $O/q
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/t -o pmc -- python $R/bench.py --workload tsqr --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/pmc_table.py $(find $O/t -name "*counter_collection.csv") > $O/mfma_busy_tsqr.txt; rm -rf $O/t
cat $O/mfma_busy_qr32.txt | head -30
