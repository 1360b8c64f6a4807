# round 5, call 16 (last): the suite, smoke and the C2 / C3 lines at the final commit; the callback-transport stress once more
export TMPDIR=/tmp
EVID=r05x bash tools/quick_run.sh > gpurun_out/r05x_quick.log 2>&1; tail -9 gpurun_out/r05x_quick.log
O=gpurun_out/r05p; mkdir -p $O
rm -rf gpurun_out/trace_stress
timeout 700 python tools/world_trace_stress.py --no-trace --minutes ${M1:-9} > $O/stress_callbacks4.log 2>&1; grep -v "^round .* done" $O/stress_callbacks4.log | cut -c1-330 | tail -24; grep "^round .* done" $O/stress_callbacks4.log | tail -1
