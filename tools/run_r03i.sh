set -x
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
BIGKRLS_SKIP_WORLD_RUNS=1 python -m pytest tests/test_gpu_level1.py tests/test_gpu_fit_capi.py tests/test_gpu_golden_and_properties.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
for cfg in "20000 20" "50000 20" "100000 50"; do
  echo "head" >> $O/kb_ab.log; timeout 300 python tools/kb_bench.py $cfg tools/libbigkrls_head.so >> $O/kb_ab.log 2>&1
  echo "new" >> $O/kb_ab.log; timeout 300 python tools/kb_bench.py $cfg >> $O/kb_ab.log 2>&1
done
done
BIGKRLS_VERBOSE=1 timeout 600 python tools/fit_bench.py 50000 20 512 > $O/kry_C4.log 2>&1
BIGKRLS_VERBOSE=1 timeout 600 python tools/fit_bench.py 100000 50 1024 > $O/kry_C5.log 2>&1
grep -v amdgpu $O/kb_ab.log; grep "block Lanczos" $O/kry_C4.log | tail -12; grep "block Lanczos" $O/kry_C5.log | tail -12
