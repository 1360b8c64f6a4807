O=gpurun_out/r03n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o run -- python3 $GRAFT_REPO_ROOT/tools/eig_once.py 20000 20 > $GRAFT_REPO_ROOT/$O/trace_run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name "run_kernel_trace.csv" | head -1)
python tools/qr_start_delay.py $f | tee $O/qr_start_delay.log
python tools/trace_timeline.py $f | tee $O/timeline.log
rm -rf $O/trace
