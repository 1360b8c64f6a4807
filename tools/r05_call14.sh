# round 5, call 14: the suite at the current commit, then the callback-transport stress once more (after the agreement
# fix of the verification loop)
export TMPDIR=/tmp
EVID=r05y bash tools/quick_run.sh > gpurun_out/r05y_quick.log 2>&1; tail -12 gpurun_out/r05y_quick.log
O=gpurun_out/r05n; mkdir -p $O
rm -rf gpurun_out/trace_stress
timeout 1100 python tools/world_trace_stress.py --no-trace --minutes ${M1:-14} > $O/stress_callbacks2.log 2>&1; grep -v "^round .* done" $O/stress_callbacks2.log | cut -c1-300 | tail -30; grep "^round .* done" $O/stress_callbacks2.log | tail -1
for d in gpurun_out/trace_stress/*/; do echo "== kept: $d"; tail -5 $d/log.txt | cut -c1-300; done 2>/dev/null | head -60
