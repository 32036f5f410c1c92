cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r04w
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04w/prof -o qr1 -- python $GRAFT_REPO_ROOT/tools/qr_soak.py 1 3 > $GRAFT_REPO_ROOT/gpurun_out/r04w/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/qr_chain_trace.py $(find gpurun_out/r04w/prof -name "*kernel_trace.csv" | head -1) 128 > gpurun_out/r04w/chain.txt 2>&1
python tools/qr_window_trace.py $(find gpurun_out/r04w/prof -name "*kernel_trace.csv" | head -1) 10.0 10.6 > gpurun_out/r04w/window.txt 2>&1
rm -rf gpurun_out/r04w/prof
tail -30 gpurun_out/r04w/chain.txt
