#!/bin/bash
# MFMA-busy and HBM-side bytes per kernel for the default line and for configs[4] (separate --pmc passes, kernel trace only)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pmc; mkdir -p $O
for W in chol gemm32; do
  for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do   # (FETCH_SIZE and WRITE_SIZE in ONE pass took 20 minutes and crashed the profiler: one counter per pass, as tools/r05_profiles.sh does)
    n=$(echo $C | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${W}_$n -o pmc -- python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > /dev/null 2> $O/${W}_$n.err
  done
  python $R/tools/pmc_table.py $(find $O/${W}_GRBM_GUI_ACTIVE $O/${W}_FETCH_SIZE -name "*counter_collection.csv") > $O/${W}_pmc_table.txt 2>&1
  rm -rf $O/${W}_GRBM_GUI_ACTIVE $O/${W}_FETCH_SIZE
  cat $O/${W}_pmc_table.txt | head -14
done
