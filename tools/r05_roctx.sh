#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
cd $R; timeout 300 python -m pytest tests/test_algorithms_gpu.py -m gpu -q -k roctx 2>&1 | tail -2; cd /tmp
timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/trace -- python $R/tools/roctx_demo.py > $O/demo.log 2>&1
ls $O/trace/*/ ; f=$(find $O/trace -name "*marker_api_trace.csv" | head -1); echo "marker file: $f"; head -12 "$f" | cut -c1-220 | tee $O/roctx_ranges_head.txt; wc -l "$f"
find $O/trace -name "*.csv" -size +1M -delete
