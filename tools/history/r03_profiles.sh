#!/bin/bash
# round-3 evidence: kernel stats of the TSQR bench line, the batched QR, the default bench line (+ PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tsqr -o tsqr -- python $R/bench.py --workload tsqr --no-cpu-baseline --steps 2 --warmup 1 > $O/tsqr_line.json 2> $O/tsqr.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/qr32 -o qr32 -- python $R/tools/qr_run.py 32 > $O/qr32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/qr1 -o qr1 -- python $R/tools/qr_run.py 1 > $O/qr1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench.err
python $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $C | cut -d' ' -f1)
  NUMPYWREN_AMD_CHAIN_CUS=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$n -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > /dev/null 2>&1
  python $R/tools/pmc_syrk.py $(find $O/pmc_$n -name "*counter_collection.csv") > $O/pmc_$n.txt
  python - <<PY >> $O/pmc_$n.txt
import csv,glob,collections
f=glob.glob("$O/pmc_$n/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "tril_copy" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
  rm -rf $O/pmc_$n
done
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O $O/*/ | head -60
cat $O/tsqr_line.json $O/bench_line.json | cut -c1-300
cat $O/pmc_*.txt
