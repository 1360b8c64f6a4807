O=gpurun_out/r03m; mkdir -p $O
BIGKRLS_PQ_WIDE=1 python -m pytest tests/test_gpu_level1.py tests/test_gpu_golden_and_properties.py tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests_wide.log
python tools/knob_ab.py 20000 20 - BIGKRLS_PQ_WIDE=1 BIGKRLS_PQ_WIDE=5000 BIGKRLS_PQ_WIDE=9000 BIGKRLS_PQ_WIDE=14000 2>&1 | grep best | tee $O/knob_wide_C3.log
python tools/knob_ab.py 5000 10 - BIGKRLS_PQ_WIDE=1 2>&1 | grep best | tee $O/knob_wide_C2.log
