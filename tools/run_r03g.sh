set -x
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
BIGKRLS_SKIP_WORLD_RUNS=1 python -m pytest tests/test_gpu_level1.py tests/test_gpu_fit_capi.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
BIGKRLS_KB=tiled BIGKRLS_SKIP_WORLD_RUNS=1 python -m pytest tests/test_gpu_level1.py -m gpu -x -q > $O/tests_tiled.log 2>&1; tail -3 $O/tests_tiled.log
for cfg in "5000 10" "20000 20" "50000 20" "100000 50"; do
  for kb in wave tiled; do
    echo "KB=$kb" >> $O/kb_bench.log
    BIGKRLS_KB=$kb timeout 300 python tools/kb_bench.py $cfg >> $O/kb_bench.log 2>&1
  done
done
grep -v amdgpu $O/kb_bench.log
