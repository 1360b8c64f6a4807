# round 5, call 8: (a) the tests of the verification / redo and of everything touched; (b) the fine trace once more,
# now with every panel's fused small products run twice and compared on the device (a nondeterminism caught in the act)
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python tests/_fault_inject.py > $O/fault_inject.log 2>&1; tail -3 $O/fault_inject.log
rm -rf gpurun_out/oversub_single
BIGKRLS_TRACE_FINE=1 timeout 2000 python tools/oversub_single.py --minutes ${SINGLE_MIN:-24} --procs 36 --reps 8 --small --arms "-|BIGKRLS_NO_SIDE=1" > $O/single_dup.log 2>&1
grep -v "^round .* done" $O/single_dup.log | cut -c1-700 | tail -60
grep "^round .* done" $O/single_dup.log | tail -1
