export TMPDIR=/tmp
O=gpurun_out/dcab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_level1.py -x -q -k "eigen" > $O/tests.log 2>&1; tail -3 $O/tests.log
for n in "20000 20" "5000 10"; do
  echo "== $n"; BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $n 2>&1 | grep -E "d&c|divide|^rep" | tail -16 | cut -c1-230
done
