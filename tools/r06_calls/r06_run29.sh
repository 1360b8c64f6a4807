# round 6, GPU call 33: the Lanczos tests once more after the last one-line change (the norms of replaced blocks add up)
O=gpurun_out/${EVID:-r06x8}; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_level1.py tests/test_gpu_configs.py -m gpu -q -k "lanczos or low_rank or c4 or fault or recovery or watchdog" > $O/gpu_tests_kry.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_kry.log
grep -v amdgpu.ids $O/gpu_tests_kry.log | tail -6
python tools/lowrank_check.py 20000 2 2048 2>&1 | grep -v amdgpu.ids | grep "eigen:\|fit:\|check:\|max |theta" 
