#!/bin/bash
# round-4 evidence (run through gpurun; summaries are copied from gpurun_out/r04prof into profiles/ by hand):
#   kernel stats of the default bench line, of the TSQR bench line and of a batch of 32 factorisations (with T / R only),
#   the un-profiled lines of the same box, and the PMC passes of the trailing-update kernel (separate passes per counter group).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04prof; mkdir -p $O
python $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
python $R/bench.py --workload tsqr --steps 3 --warmup 1 > $O/tsqr_line.json 2> $O/tsqr_line.err
python $R/bench.py --workload gemm32 --steps 3 --warmup 1 > $O/gemm32_line.json 2> $O/gemm32_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tsqr -o tsqr -- python $R/bench.py --workload tsqr --steps 2 --warmup 1 > $O/tsqr_under_rocprof.json 2> $O/tsqr.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/qr32 -o qr32 -- python $R/tools/qr_soak.py 32 3 > $O/qr32.log 2>&1
QR_SOAK_NO_T=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/qr32r -o qr32r -- python $R/tools/qr_soak.py 32 3 > $O/qr32r.log 2>&1
for d in bench tsqr qr32 qr32r; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/${d}_kernel_stats.csv; done
python $R/tools/qr_chain_trace.py $(find $O/qr32 -name "*kernel_trace.csv" | head -1) 128 > $O/qr32_chain.txt
python $R/tools/qr_chain_trace.py $(find $O/qr32r -name "*kernel_trace.csv" | head -1) 128 > $O/qr32r_chain.txt
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
rm -rf $O/bench $O/tsqr $O/qr32 $O/qr32r
ls -la $O | head -40
cut -c1-400 $O/bench_line.json $O/tsqr_line.json $O/gemm32_line.json
cat $O/pmc_*.txt
