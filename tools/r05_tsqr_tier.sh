#!/bin/bash
# configs[3]'s 256-leaf TSQR keeping R, V, T (256 GiB of stored tiles) with the stored tiles capped at 96 GiB: 160 GiB of
# factors leave for pinned host DRAM (the box's cgroup allows 300 GiB of host memory: the 512-leaf problem -- 311 GiB to the
# host -- does not fit it; the tier refuses beyond 3/4 of the limit instead of getting the box killed)
cd /root/repo; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r05i; mkdir -p $O
NUMPYWREN_AMD_HBM_BUDGET=96G NPW_BENCH_DEBUG=1 timeout 900 python bench.py --workload tsqr --leaves 256 --steps 2 --warmup 1 > $O/tsqr256_keepvt_96G.json 2> $O/tsqr256_keepvt_96G.err
tail -c 1200 $O/tsqr256_keepvt_96G.json; grep bench $O/tsqr256_keepvt_96G.err | tail -4
