#!/bin/bash
# HBM-side bytes of a batch of 32 factorisations by kernel (separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE; gfx950
# reports wide coalesced reads at half their size: FETCH_SIZE is doubled in the summary).  Usage: tools/r04_qr_pmc.sh <tag> [ENV=VALUE ...]
tag=${1:-r04x}; shift
for v in "$@"; do export "$v"; done
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/qr_soak.py 32 1 > $out/run_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - $out <<'PY' | tee $out/bytes.txt
import csv, glob, sys, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for C, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
    f = glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    # the factorisation = everything between the input generation and the residual check: take kernels of the QR call by name
    for r in rows:
        n = r["Kernel_Name"].replace("npw::(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*", "", n)[:64]
        if any(k in n for k in ("fill_random", "sumsq")):
            continue
        v = float(r["Counter_Value"]) * 1024.0 * (2.0 if C == "FETCH_SIZE" else 1.0)
        tot[n][col] += v
        if C == "FETCH_SIZE":
            tot[n][0] += 1
gb = lambda x: x / 1e9
print("%-66s %7s %10s %10s" % ("kernel (whole process: 1 batched call + its residual checks)", "calls", "read GB", "write GB"))
for n, (c, rd, wr) in sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-66s %7d %10.2f %10.2f" % (n, c, gb(rd), gb(wr)))
print("%-66s %7s %10.2f %10.2f" % ("sum", "", gb(sum(v[1] for v in tot.values())), gb(sum(v[2] for v in tot.values()))))
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
