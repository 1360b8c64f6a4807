# phase ticks (BIGKRLS_VERBOSE) + kernel stats for the eigensolver at N = 5000 and 20000, the mailbox pingpong probe
O=gpurun_out/${EVID:-r04b}; mkdir -p $O
export TMPDIR=/tmp
tools/pingpong > $O/pingpong.log 2>&1
for cfg in "5000 10" "20000 20"; do
  tag=$(echo $cfg | tr ' ' '_')
  BIGKRLS_VERBOSE=1 python tools/eig_once.py $cfg > $O/eig_verbose_$tag.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o run -- python3 tools/eig_once.py $cfg > $O/eig_prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/eig_${tag}_kernel_stats.csv
  t=$(find $O/prof_$tag -name "*kernel_trace.csv" | head -1); python tools/trace_timeline.py "$t" > $O/eig_${tag}_timeline.log 2>&1
  rm -rf $O/prof_$tag
done
cat $O/pingpong.log; tail -60 $O/eig_verbose_5000_10.log
