# round 5, call 3: where does the rare wrong single-GPU fit under oversubscription come from? Same load, same moment,
# two arms: the library as it is / the look-ahead work on the main stream (BIGKRLS_NO_SIDE=1: no cross-stream ordering left)
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
rm -rf gpurun_out/oversub_single
timeout 2400 python tools/oversub_single.py --minutes ${SINGLE_MIN:-30} --procs 32 --reps 8 --small --arms "-|BIGKRLS_NO_SIDE=1" > $O/single_arms.log 2>&1
grep -v "^round .* done" $O/single_arms.log | cut -c1-400 | tail -60
grep "^round .* done" $O/single_arms.log | tail -2
