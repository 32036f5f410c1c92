#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
for f in 0 1; do
NPW_TRSM_FUSED=$f timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$f -- python $R/tools/trsm_run.py > $O/run$f.log 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$O/t$f/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-60:]
t0=int(tail[0]["Start_Timestamp"])
print("NPW_TRSM_FUSED=$f")
for r in tail:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("npw::(anonymous namespace)::","")[:70]
    print("%9.1f %8.1f us  grid %6d wg %4d  %s"%((s-t0)/1e3,(e-s)/1e3,int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])*int(r["Grid_Size_Z"]),int(r["Workgroup_Size_X"]),n))
PY
done > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
tail -130 $O/summary.txt
