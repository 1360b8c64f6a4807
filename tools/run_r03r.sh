python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "choleskyqr or ill_conditioned or aggregated" 2>&1 | tail -8
