#!/bin/bash
# round 5, first GPU call: the -m gpu suite at HEAD, the default bench line, the host split of the distributed walk
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python tools/dist_host_split.py --profile 45 > $O/host_split.json 2> $O/host_split.err
tail -3 $O/pytest.log; head -c 600 $O/bench.json; echo; tail -5 $O/host_split.err
