#!/bin/bash
# round 5, second GPU call: host split after the key memo, the bench's own FORCE_DIST line, spill tier LRU vs plan, RCCL kernel shape
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
free -g > $O/free.txt; nproc >> $O/free.txt
timeout 900 python -m pytest tests/test_comm_gpu.py tests/test_residency.py tests/test_bench_contract.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python tools/dist_host_split.py > $O/host_split.json 2> $O/host_split.err
NUMPYWREN_AMD_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29688 timeout 600 python bench.py --tiles 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_forcedist.json 2> $O/bench_forcedist.err
NUMPYWREN_AMD_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29689 timeout 600 python bench.py --tiles 16 --steps 2 --warmup 1 --streams 3 --no-cpu-baseline > $O/bench_forcedist_s3.json 2> $O/bench_forcedist_s3.err
for bt in 24 12; do
  timeout 600 python tools/bench_aux.py spill --tiles 8 --budget-tiles $bt --lru > $O/spill_lru_$bt.json 2> $O/spill_lru_$bt.err
  timeout 600 python tools/bench_aux.py spill --tiles 8 --budget-tiles $bt > $O/spill_plan_$bt.json 2> $O/spill_plan_$bt.err
done
# the shape of RCCL's point-to-point kernel (grid, workgroup, LDS): kernel trace of the self-exchange test
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/rccl_trace -- python -m pytest /root/repo/tests/test_comm_gpu.py -q -x -k "self_exchange or resident_grid" > /root/repo/$O/rccl_trace.log 2>&1
cd /root/repo
find $O/rccl_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c "head -1 {}; grep -i 'nccl' {} | head -5" > $O/rccl_kernels.txt 2>&1
find $O/rccl_trace -name "*.csv" -size +2M -delete
tail -3 $O/pytest.log; cat $O/free.txt; cat $O/rccl_kernels.txt | cut -c1-600
