O=gpurun_out/${EVID:-r04c}; mkdir -p $O
python -m pytest tests/test_gpu_level1.py -x -q -k "eigen" 2>&1 | tail -3
python tools/eig_stress.py 2>&1 | tail -12
for cfg in "5000 10" "20000 20"; do
  tag=$(echo $cfg | tr ' ' '_')
  BIGKRLS_VERBOSE=1 python tools/eig_once.py $cfg 2>&1 | tail -24 > $O/eig_verbose_$tag.log
done
cat $O/eig_verbose_5000_10.log $O/eig_verbose_20000_20.log | grep -v "^rep" | grep "d&c\|divide"
