timeout 600 python -m pytest tests/test_gpu_level1.py -x -q -k "eigen" 2>&1 | tail -2
timeout 300 python tools/eig_stress.py 2>&1 | grep -c CHECK
for cfg in "5000 10" "20000 20"; do BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | grep "^rep\|divide\|depth  [0123]" | tail -6; done
