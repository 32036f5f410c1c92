#!/bin/bash
# round-6 evidence (run through gpurun; summaries are copied from gpurun_out/r06prof into profiles/ by hand):
#   the three bench lines of one box, kernel stats of the default line / the TSQR line / a batch of 32 factorisations, and the PMC
#   passes (separate passes per counter group, --kernel-trace only) behind every `roofline.traffic`: the trailing-update kernel of
#   the default line, the fp32 product of the gemm32 line, the whole batch of 32 factorisations (with T and R only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06prof; mkdir -p $O
python $R/bench.py > $O/bench_line.json 2> $O/bench_line.err
python $R/bench.py --workload tsqr --steps 3 --warmup 1 > $O/tsqr_line.json 2> $O/tsqr_line.err
python $R/bench.py --workload gemm32 --steps 3 --warmup 1 > $O/gemm32_line.json 2> $O/gemm32_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tsqr -o tsqr -- python $R/bench.py --workload tsqr --steps 2 --warmup 1 --no-cpu-baseline > $O/tsqr_under_rocprof.json 2> $O/tsqr.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemm32 -o gemm32 -- python $R/bench.py --workload gemm32 --steps 2 --warmup 1 --no-cpu-baseline > $O/gemm32_under_rocprof.json 2> $O/gemm32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/qr32 -o qr32 -- python $R/tools/qr_soak.py 32 3 > $O/qr32.log 2>&1
for d in bench tsqr gemm32 qr32; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/${d}_kernel_stats.csv; done
python $R/tools/qr_chain_trace.py $(find $O/qr32 -name "*kernel_trace.csv" | head -1) 128 > $O/qr32_chain.txt
rm -rf $O/bench $O/tsqr $O/gemm32 $O/qr32
# ---- PMC: the default line's trailing update (as round 5) ----
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $C | cut -d' ' -f1)
  NUMPYWREN_AMD_CHAIN_CUS=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$n -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-north-star > /dev/null 2>&1
  python $R/tools/pmc_syrk.py $(find $O/pmc_$n -name "*counter_collection.csv") > $O/pmc_$n.txt
  python $R/tools/pmc_table.py $(find $O/pmc_$n -name "*counter_collection.csv") > $O/pmc_table_bench_$n.txt
  rm -rf $O/pmc_$n
done
# ---- PMC: the fp32 product of the gemm32 line ----
for C in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmcg_$n -o pmc -- python $R/bench.py --workload gemm32 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_table.py $(find $O/pmcg_$n -name "*counter_collection.csv") > $O/pmc_table_gemm32_$n.txt
  rm -rf $O/pmcg_$n
done
# ---- PMC: a batch of 32 factorisations, nothing else in the process but the input generator (2 calls each form) ----
for form in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    QR_RUN_NO_T=$form QR_RUN_REPS=2 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmcq_${form}_$C -o pmc -- python $R/tools/qr_run.py 32 > /dev/null 2>&1
    python $R/tools/pmc_table.py $(find $O/pmcq_${form}_$C -name "*counter_collection.csv") > $O/pmc_table_qr32_not${form}_$C.txt
    python - $O/pmcq_${form}_$C $C $form <<'PY' >> $O/qr32_bytes.txt
import csv, glob, sys
d, C, form = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "fill_random" in r["Kernel_Name"]:
        continue
    tot += float(r["Counter_Value"])
print("form_no_t=%s %s sum_KB_over_2_calls=%.1f" % (form, C, tot))
PY
    rm -rf $O/pmcq_${form}_$C
  done
done
ls -la $O | head -60
cut -c1-300 $O/bench_line.json $O/tsqr_line.json $O/gemm32_line.json
cat $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/qr32_bytes.txt; head -5 $O/pmc_table_gemm32_FETCH_SIZE.txt $O/pmc_table_gemm32_WRITE_SIZE.txt
