#!/bin/bash
# kernel trace of single triangular solves (4096^2, the factor's cached inverses): per-launch timeline of the last one.
out=gpurun_out/${1:-r04t}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o trsm -- python $GRAFT_REPO_ROOT/tools/trsm_run.py > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
csv=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python - $csv <<'PY' | tee $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last solve: the kernels after the last long gap
last = len(rows) - 1
start = last
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 200_000:
    start -= 1
t0 = int(rows[start]["Start_Timestamp"]); prev = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("npw::(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{n:60s} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d} x{r.get('Grid_Size_Y','1'):>2s} x{r.get('Grid_Size_Z','1'):>2s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:5.1f}")
    prev = e
print("span %.1f us" % ((prev - t0) / 1e3))
PY
tail -4 $out/run.log
rm -rf $out/prof
