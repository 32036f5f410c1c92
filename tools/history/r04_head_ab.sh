#!/bin/bash
# Same-box A/B of the default bench line: the library as built before today's QR / runtime commits (gpurun_tmp/libnpw_old.so, built
# from 80be521..fab30c4's qr.hip) against HEAD's, alternating, three runs each; then the PMC passes of the trailing-update kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ab; mkdir -p $O
for i in 1 2 3; do
  for lib in old head; do
    if [ $lib = old ]; then export NUMPYWREN_AMD_LIB=$R/gpurun_tmp/libnpw_old.so; else unset NUMPYWREN_AMD_LIB; fi
    timeout 200 python $R/bench.py --no-cpu-baseline --no-north-star > $O/${lib}_$i.json 2> $O/${lib}_$i.err
    python - <<PY
import json
d=json.load(open("$O/${lib}_$i.json"))
print("$lib $i", d["value"], d["ms_per_step"], "syrk avg_ms", d["roofline"]["avg_ms"], "frac", d["roofline"]["frac"], d["kernel_ms"]["trsm"], d["kernel_ms"]["chol"])
PY
  done
done
unset NUMPYWREN_AMD_LIB
for C in FETCH_SIZE WRITE_SIZE; do
  NUMPYWREN_AMD_CHAIN_CUS=0 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > /dev/null 2>&1
  python $R/tools/pmc_syrk.py $(find $O/pmc_$C -name "*counter_collection.csv") | tee $O/pmc_$C.txt
  rm -rf $O/pmc_$C
done
