# round 5, call 13: the single-process control with every context on a stream of its own (not the NULL stream)
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
rm -rf gpurun_out/oversub_single
BIGKRLS_VERIFY=0 timeout 1800 python tools/oversub_single.py --minutes ${SINGLE_MIN:-24} --procs 36 --reps 8 --small --arms "OVERSUB_OWN_STREAM=1" > $O/single_own_stream.log 2>&1
grep -v "^round .* done" $O/single_own_stream.log | cut -c1-500 | tail -40
grep "^round .* done" $O/single_own_stream.log | tail -1
