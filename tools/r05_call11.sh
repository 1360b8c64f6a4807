export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
BIGKRLS_POISON=1 BIGKRLS_SKIP_WORLD_RUNS=1 timeout 900 python -m pytest tests/test_gpu_fit.py -q -x -k "two_contexts or crossvalid or folds" > $O/tests_poison2.log 2>&1; tail -4 $O/tests_poison2.log
timeout 400 python tools/knob_ab.py 5000 10 - "BIGKRLS_VERIFY=0" > $O/knob_verify_c2.log 2>&1; grep best $O/knob_verify_c2.log
timeout 400 python tools/knob_ab.py 20000 20 - "BIGKRLS_VERIFY=0" > $O/knob_verify_c3.log 2>&1; grep best $O/knob_verify_c3.log
