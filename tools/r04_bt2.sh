O=gpurun_out/${EVID:-r04d}; mkdir -p $O
tools/dpp_probe
for cfg in "5000 10" "20000 20"; do
  tag=$(echo $cfg | tr ' ' '_')
  BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | tail -24 > $O/eig_verbose_$tag.log
done
cat $O/eig_verbose_5000_10.log $O/eig_verbose_20000_20.log | grep "^rep\|stage 2"  | tail -8
