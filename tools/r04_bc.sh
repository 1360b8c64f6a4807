O=gpurun_out/${EVID:-r04e}; mkdir -p $O
for cfg in "5000 10" "20000 20"; do
  tag=$(echo $cfg | tr ' ' '_')
  BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg > $O/eig_verbose_$tag.log 2>&1
  BIGKRLS_BC=lds BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | grep "stage 2 (band" | tail -1
done
cat $O/eig_verbose_5000_10.log $O/eig_verbose_20000_20.log | grep "^rep\|stage 2"  | tail -8
timeout 900 python -m pytest tests/test_gpu_level1.py -x -q -k "eigen" 2>&1 | tail -5
timeout 300 python tools/eig_stress.py 2>&1 | grep -c CHECK
