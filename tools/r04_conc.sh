for knobs in "" ; do
  echo "== knobs: $knobs"
  for rep in 1 2 3; do env $knobs BIGKRLS_VERBOSE=1 timeout 300 python tools/contention_check.py 5000 2>&1 | grep -v "d&c\|eigen n=\|panels left" | tail -5; done
done
python tools/contention_check.py 12000 2>&1 | tail -4
python -m pytest tests/test_gpu_fit.py -x -q 2>&1 | tail -3
for cfg in "5000 10" "20000 20"; do BIGKRLS_VERBOSE=1 timeout 300 python tools/eig_once.py $cfg 2>&1 | grep "^rep\|stage 2" | tail -3; done
python tools/kb_bench.py 20000 20 2>&1 | tail -3; python tools/kb_bench.py 50000 20 2>&1 | tail -3
